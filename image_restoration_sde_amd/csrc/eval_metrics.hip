// Evaluation tail on the device: the per-image metrics of codes/config/deraining/test.py:131-178
// without leaving the GPU and without cv2.
//   tensor2img  (codes/utils/img_utils.py:136-163): clamp to [0,1], *255 (fp32), round half-to-even, uint8
//   PSNR        (:182-189) on the uint8 RGB images and on the Y channel
//   SSIM        (:192-234) 11x11 Gaussian (sigma 1.5) 'valid' window, float64, averaged over channels
//   bgr2ycbcr   (codes/data/util.py:177-198, only_y) on the float image q/255
// Stage 1 quantises both images and produces the Y planes; stage 2 writes per-tile partial sums (fixed order,
// no atomics => deterministic); the host adds the few partials per image in float64.
#include "common.h"

namespace irsde {
namespace {

// uint8 value of tensor2img: fp32 clamp, fp32 multiply, rint (ties to even), as numpy's float32 `.round()`
__device__ __forceinline__ float quant255(float v) {
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    return rintf(v * 255.0f);
}

// bgr2ycbcr(only_y) of the float image q/255, times 255 again (what calculate_psnr / ssim receive)
__device__ __forceinline__ double y255(float r, float g, float b) {
    const double rb = ((double)b / 255.0) * 255.0, rg = ((double)g / 255.0) * 255.0, rr = ((double)r / 255.0) * 255.0;
    const double dot = rb * 24.966 + rg * 128.553 + rr * 65.481;
    return ((dot / 255.0 + 16.0) / 255.0) * 255.0;
}

// out / gt: NCHW fp32 [B][C][H][W] (C = 1 or 3, RGB).  q: [2][B][C][H][W] uint8-valued floats; yq: [2][B][H][W] doubles.
__global__ void eval_quantize_kernel(const float* __restrict__ out, const float* __restrict__ gt, float* __restrict__ q,
                                     double* __restrict__ yq, int B, int C, int HW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)B * HW;
    if (i >= n) return;
    const int b = (int)(i / HW);
    const int pix = (int)(i - (size_t)b * HW);
    const size_t plane = (size_t)B * C * HW;
    float vo[3], vg[3];
    for (int c = 0; c < C; ++c) {
        const size_t idx = ((size_t)b * C + c) * HW + pix;
        vo[c] = quant255(out[idx]);
        vg[c] = quant255(gt[idx]);
        q[idx] = vo[c];
        q[plane + idx] = vg[c];
    }
    if (C == 3) {
        yq[i] = y255(vo[0], vo[1], vo[2]);
        yq[n + i] = y255(vg[0], vg[1], vg[2]);
    }
}

struct GaussWin {
    double g[11];
};

// One plane pair (a, b) of one image: squared-error sum over the cropped region and SSIM-map sum over its valid
// region.  grid = (tiles, planes); block = 256 threads over a 16x16 output tile.  T = float (uint8-valued) or double.
template <typename T>
__global__ void eval_plane_kernel(const T* __restrict__ a_all, const T* __restrict__ b_all, int H, int W, int crop,
                                  GaussWin gw, double* __restrict__ partial /* [planes][tiles][2] */, int tiles_x) {
    const int plane = blockIdx.y;
    const T* a = a_all + (size_t)plane * H * W;
    const T* b = b_all + (size_t)plane * H * W;
    const int Hc = H - 2 * crop, Wc = W - 2 * crop;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y = ty * 16 + (threadIdx.x >> 4), x = tx * 16 + (threadIdx.x & 15);
    double se = 0.0, ss = 0.0;
    if (y < Hc && x < Wc) {
        const double d = (double)a[(size_t)(y + crop) * W + x + crop] - (double)b[(size_t)(y + crop) * W + x + crop];
        se = d * d;
        if (y < Hc - 10 && x < Wc - 10) {  // window anchored at the valid position (y+5, x+5) of the cropped image
            double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
            for (int i = 0; i < 11; ++i) {
                const T* ra = a + (size_t)(y + crop + i) * W + x + crop;
                const T* rb = b + (size_t)(y + crop + i) * W + x + crop;
                for (int j = 0; j < 11; ++j) {
                    const double w = gw.g[i] * gw.g[j];
                    const double va = (double)ra[j], vb = (double)rb[j];
                    m1 += w * va;
                    m2 += w * vb;
                    s11 += w * (va * va);
                    s22 += w * (vb * vb);
                    s12 += w * (va * vb);
                }
            }
            const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
            const double m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
            ss = ((2 * m12 + C1) * (2 * (s12 - m12) + C2)) / ((m11 + m22 + C1) * ((s11 - m11) + (s22 - m22) + C2));
        }
    }
    // deterministic block reduction (fixed tree)
    __shared__ double red[2][256];
    red[0][threadIdx.x] = se;
    red[1][threadIdx.x] = ss;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* dst = partial + ((size_t)plane * gridDim.x + blockIdx.x) * 2;
        dst[0] = red[0][0];
        dst[1] = red[1][0];
    }
}

}  // namespace

// metrics_host[b] = {sse_rgb, ssim_sum_rgb, sse_y, ssim_sum_y}; the caller divides by the element counts.
void eval_metrics(const float* out, const float* gt, int B, int C, int H, int W, int crop, double* sums_host,
                  hipStream_t s) {
    if (C != 1 && C != 3) throw HipError("eval_metrics: C must be 1 or 3");
    if (crop < 0 || H - 2 * crop < 11 || W - 2 * crop < 11) throw HipError("eval_metrics: image smaller than the 11x11 SSIM window");
    const int HW = H * W;
    const size_t nq = (size_t)B * C * HW;
    float* q = nullptr;
    double *yq = nullptr, *partial = nullptr;
    IRSDE_HIP_CHECK(hipMalloc(&q, 2 * nq * sizeof(float)));
    IRSDE_HIP_CHECK(hipMalloc(&yq, 2 * (size_t)B * HW * sizeof(double)));
    const int Hc = H - 2 * crop, Wc = W - 2 * crop;
    const int tiles_x = (Wc + 15) / 16, tiles = tiles_x * ((Hc + 15) / 16);
    const int planes_rgb = B * C, planes_y = C == 3 ? B : 0;
    IRSDE_HIP_CHECK(hipMalloc(&partial, (size_t)(planes_rgb + planes_y) * tiles * 2 * sizeof(double)));
    GaussWin gw;
    {
        double sum = 0;
        for (int i = 0; i < 11; ++i) {
            gw.g[i] = exp(-((double)(i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
            sum += gw.g[i];
        }
        for (int i = 0; i < 11; ++i) gw.g[i] /= sum;
    }
    try {
        hipLaunchKernelGGL(eval_quantize_kernel, dim3((unsigned)(((size_t)B * HW + 255) / 256)), dim3(256), 0, s, out, gt, q, yq,
                           B, C, HW);
        IRSDE_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(eval_plane_kernel<float>, dim3(tiles, planes_rgb), dim3(256), 0, s, q, q + nq, H, W, crop, gw,
                           partial, tiles_x);
        IRSDE_HIP_CHECK(hipGetLastError());
        if (planes_y) {
            hipLaunchKernelGGL(eval_plane_kernel<double>, dim3(tiles, planes_y), dim3(256), 0, s, yq, yq + (size_t)B * HW, H, W,
                               crop, gw, partial + (size_t)planes_rgb * tiles * 2, tiles_x);
            IRSDE_HIP_CHECK(hipGetLastError());
        }
        std::vector<double> hp((size_t)(planes_rgb + planes_y) * tiles * 2);
        IRSDE_HIP_CHECK(hipMemcpyAsync(hp.data(), partial, hp.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        IRSDE_HIP_CHECK(hipStreamSynchronize(s));
        for (int b = 0; b < B; ++b) {
            double acc[4] = {0, 0, 0, 0};
            for (int c = 0; c < C; ++c)
                for (int t = 0; t < tiles; ++t) {
                    acc[0] += hp[((size_t)(b * C + c) * tiles + t) * 2];
                    acc[1] += hp[((size_t)(b * C + c) * tiles + t) * 2 + 1];
                }
            if (planes_y)
                for (int t = 0; t < tiles; ++t) {
                    acc[2] += hp[((size_t)(planes_rgb + b) * tiles + t) * 2];
                    acc[3] += hp[((size_t)(planes_rgb + b) * tiles + t) * 2 + 1];
                }
            for (int k = 0; k < 4; ++k) sums_host[b * 4 + k] = acc[k];
        }
    } catch (...) {
        (void)hipFree(q); (void)hipFree(yq); (void)hipFree(partial);
        throw;
    }
    (void)hipFree(q); (void)hipFree(yq); (void)hipFree(partial);
}

// tensor2img on the device: NCHW fp32 -> HWC uint8, channel order reversed (RGB -> BGR) as the reference returns it
namespace {
__global__ void tensor2img_kernel(const float* __restrict__ in, unsigned char* __restrict__ out, int B, int C, int HW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const int b = (int)(i / HW);
    const int pix = (int)(i - (size_t)b * HW);
    for (int c = 0; c < C; ++c)
        out[i * C + (C - 1 - c)] = (unsigned char)quant255(in[((size_t)b * C + c) * HW + pix]);
}
}  // namespace

void tensor2img_u8(const float* in, unsigned char* out, int B, int C, int H, int W, hipStream_t s) {
    hipLaunchKernelGGL(tensor2img_kernel, dim3((unsigned)(((size_t)B * H * W + 255) / 256)), dim3(256), 0, s, in, out, B, C, H * W);
    IRSDE_HIP_CHECK(hipGetLastError());
}

}  // namespace irsde
