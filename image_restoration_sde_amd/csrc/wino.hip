// Winograd F(2x2, 3x3) for the wide 3x3 stride-1 convolutions (Cin, Cout >= 256) of the score network.
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A         (Lavin & Gray, "Fast Algorithms for Convolutional Neural
//                                                        Networks", CVPR 2016; 2.25x fewer multiplies, fp32 throughout)
// Three launches per convolution:
//   wino_input_kernel   V[k][tile][c]  = (B^T d B)_k   of every 4x4 input patch (stride 2), both concat sources and
//                                         the fused nearest-x2 upsample are gathered here           (HBM bound)
//   conv_igemm_kernel   M[k][tile][n]  = sum_c V[k][tile][c] * U[k][n][c], 16 independent GEMMs (blockIdx.z = k) on the
//                                         fp32 MFMA pipe — the same kernel as the direct path, run as a 1x1 conv
//   wino_output_kernel  y = A^T M A (2x2 pixels per tile) + bias / FiLM / SiLU / residual            (HBM bound)
// U = G g G^T is computed once at weight-load time (engine.hip).
#include "common.h"

namespace irsde {
namespace {

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float silu1(float v) { return v / (1.0f + expf(-v)); }

__global__ __launch_bounds__(256) void wino_input_kernel(const WinoParams p) {
    const int C4 = (p.C0 + p.C1) >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)p.T * C4;
    if (idx >= total) return;
    const int cg = (int)(idx % C4);
    const int t = (int)(idx / C4);
    const int tx = t % p.TW;
    const int t1 = t / p.TW;
    const int ty = t1 % p.TH;
    const int b = t1 / p.TH;
    const int c = cg * 4;
    const float* src;
    int pix;
    if (c < p.C0) {
        src = p.in0 + c; pix = p.C0;
    } else {
        src = p.in1 + (c - p.C0); pix = p.C1;
    }
    const int Hv = p.Hin << p.in_shift, Wv = p.Win << p.in_shift;
    float4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int iy = 2 * ty - 1 + r;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ix = 2 * tx - 1 + s;
            const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
            const size_t pixel = (size_t)b * p.Hin * p.Win + (size_t)((ok ? iy : 0) >> p.in_shift) * p.Win + ((ok ? ix : 0) >> p.in_shift);
            const float4 v = *reinterpret_cast<const float4*>(src + pixel * pix);
            d[r][s] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // B^T d  (rows), then (.) B (columns);  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    float4 w[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        w[0][s] = f4sub(d[0][s], d[2][s]);
        w[1][s] = f4add(d[1][s], d[2][s]);
        w[2][s] = f4sub(d[2][s], d[1][s]);
        w[3][s] = f4sub(d[1][s], d[3][s]);
    }
    const size_t Ctot = (size_t)(p.C0 + p.C1);
    float* vp = p.V + (size_t)t * Ctot + c;
    const size_t kstride = (size_t)p.T * Ctot;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        *reinterpret_cast<float4*>(vp + (size_t)(r * 4 + 0) * kstride) = f4sub(w[r][0], w[r][2]);
        *reinterpret_cast<float4*>(vp + (size_t)(r * 4 + 1) * kstride) = f4add(w[r][1], w[r][2]);
        *reinterpret_cast<float4*>(vp + (size_t)(r * 4 + 2) * kstride) = f4sub(w[r][2], w[r][1]);
        *reinterpret_cast<float4*>(vp + (size_t)(r * 4 + 3) * kstride) = f4sub(w[r][1], w[r][3]);
    }
}

__global__ __launch_bounds__(256) void wino_output_kernel(const WinoParams p) {
    const int N4 = p.Cout >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)p.T * N4;
    if (idx >= total) return;
    const int ng = (int)(idx % N4);
    const int t = (int)(idx / N4);
    const int tx = t % p.TW;
    const int t1 = t / p.TW;
    const int ty = t1 % p.TH;
    const int b = t1 / p.TH;
    const int n = ng * 4;
    const float* mp = p.M + (size_t)t * p.Cout + n;
    const size_t kstride = (size_t)p.T * p.Cout;
    float4 m[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) m[r][s] = *reinterpret_cast<const float4*>(mp + (size_t)(r * 4 + s) * kstride);
    // A^T m (rows), then (.) A;  A^T = [1 1 1 0; 0 1 -1 -1]
    float4 u[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u[0][s] = f4add(f4add(m[0][s], m[1][s]), m[2][s]);
        u[1][s] = f4sub(f4sub(m[1][s], m[2][s]), m[3][s]);
    }
    float4 y[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        y[i][0] = f4add(f4add(u[i][0], u[i][1]), u[i][2]);
        y[i][1] = f4sub(f4sub(u[i][1], u[i][2]), u[i][3]);
    }
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = bias;
    if (p.bias) bias = *reinterpret_cast<const float4*>(p.bias + n);
    if (p.film) {
        const float* f = p.film + (size_t)(p.film_bstride ? b : 0) * p.film_bstride;
        const float4 s4 = *reinterpret_cast<const float4*>(f + n);
        sc = make_float4(s4.x + 1.0f, s4.y + 1.0f, s4.z + 1.0f, s4.w + 1.0f);
        sh = *reinterpret_cast<const float4*>(f + p.Cout + n);
    }
    const int Ho = 2 * p.TH, Wo = 2 * p.TW;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t pixel = ((size_t)b * Ho + 2 * ty + i) * Wo + 2 * tx + j;
            float v[4] = {y[i][j].x + bias.x, y[i][j].y + bias.y, y[i][j].z + bias.z, y[i][j].w + bias.w};
            if (p.film) {
                v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
            }
            if (p.silu) {
                v[0] = silu1(v[0]); v[1] = silu1(v[1]); v[2] = silu1(v[2]); v[3] = silu1(v[3]);
            }
            if (p.res) {
                const float4 r4 = *reinterpret_cast<const float4*>(p.res + pixel * p.res_stride + n);
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            *reinterpret_cast<float4*>(p.out + pixel * p.out_stride + n) = make_float4(v[0], v[1], v[2], v[3]);
        }
}

}  // namespace

void launch_wino_input(const WinoParams& p, hipStream_t s) {
    const long long total = (long long)p.T * ((p.C0 + p.C1) >> 2);
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_wino_output(const WinoParams& p, hipStream_t s) {
    const long long total = (long long)p.T * (p.Cout >> 2);
    hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
    IRSDE_HIP_CHECK(hipGetLastError());
}

// U[k][n][c] = (G g G^T)_k with G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1];  g given as [Cout][3][3][Cin] (packed layout)
void wino_transform_weights(const float* w_packed, int Cout, int Cin, float* U) {
    static const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const size_t kstride = (size_t)Cout * Cin;
    for (int n = 0; n < Cout; ++n)
        for (int c = 0; c < Cin; ++c) {
            float g[3][3];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w_packed[(((size_t)n * 3 + ky) * 3 + kx) * Cin + c];
            float tmp[4][3];
            for (int r = 0; r < 4; ++r)
                for (int kx = 0; kx < 3; ++kx) tmp[r][kx] = G[r][0] * g[0][kx] + G[r][1] * g[1][kx] + G[r][2] * g[2][kx];
            for (int r = 0; r < 4; ++r)
                for (int s = 0; s < 4; ++s)
                    U[(size_t)(r * 4 + s) * kstride + (size_t)n * Cin + c] =
                        tmp[r][0] * G[s][0] + tmp[r][1] * G[s][1] + tmp[r][2] * G[s][2];
        }
}

}  // namespace irsde
