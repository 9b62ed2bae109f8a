// Winograd F(m x m, 3x3), m = 2 (default) or 4 (opt-in), for the wide 3x3 stride-1 convolutions of the score network.
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A         (Lavin & Gray, "Fast Algorithms for Convolutional Neural
//                                                        Networks", CVPR 2016; fp32 throughout)
//   m = 2: 16 products per 4 outputs  (2.25x fewer multiplies than direct), V/M tensors 4x    the input/output size
//   m = 4: 36 products per 16 outputs (4x    fewer multiplies),             V/M tensors 2.25x the input/output size,
//          ~15-30x the rounding error of the direct kernel (1e-5 instead of 5e-7 relative per layer)
// Three launches per convolution:
//   wino_input_kernel   V[k][tile][c]  = (B^T d B)_k   of every (m+2)x(m+2) input patch (stride m); both concat sources
//                                         and the fused nearest-x2 upsample are gathered here          (HBM bound)
//   conv_igemm_kernel   M[k][tile][n]  = sum_c V[k][tile][c] * U[k][n][c], (m+2)^2 independent GEMMs (blockIdx.z = k) on
//                                         the fp32 MFMA pipe — the same kernel as the direct path, run as a 1x1 conv
//   wino_output_kernel  y = A^T M A (m x m pixels per tile) + bias / FiLM / SiLU / residual            (HBM bound)
// U = G g G^T is computed once at weight-load time (engine_weights.hip).
#include "common.h"

namespace irsde {
namespace {

__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float silu1(float v) { return v / (1.0f + expf(-v)); }
// V = float4 (4 channels per thread) or float (1 channel per thread: ~90 instead of 170-220 VGPRs, so that transform waves
// can share a SIMD with the resident GEMM waves of another stream; 64 lanes x 4 B is still a 256-byte coalesced access)
template <typename V> struct VecOps;
template <> struct VecOps<float4> {
    static constexpr int N = 4;
    static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ float4 one() { return make_float4(1.f, 1.f, 1.f, 1.f); }
    static __device__ __forceinline__ float4 mad(float4 v, float4 a, float4 b) { return make_float4(v.x * a.x + b.x, v.y * a.y + b.y, v.z * a.z + b.z, v.w * a.w + b.w); }
    static __device__ __forceinline__ float4 silu(float4 v) { return make_float4(silu1(v.x), silu1(v.y), silu1(v.z), silu1(v.w)); }
};
template <> struct VecOps<float> {
    static constexpr int N = 1;
    static __device__ __forceinline__ float zero() { return 0.f; }
    static __device__ __forceinline__ float one() { return 1.f; }
    static __device__ __forceinline__ float mad(float v, float a, float b) { return v * a + b; }
    static __device__ __forceinline__ float silu(float v) { return silu1(v); }
};

// one application of B^T (input side) / A^T (output side) along one axis
template <int TILE, typename V>
__device__ __forceinline__ void bt_apply(const V* d, V* t) {
    if constexpr (TILE == 2) {  // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        t[0] = d[0] - d[2];
        t[1] = d[1] + d[2];
        t[2] = d[2] - d[1];
        t[3] = d[1] - d[3];
    } else {  // B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
        t[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
        t[1] = (d[3] + d[4]) - 4.0f * (d[1] + d[2]);
        t[2] = 4.0f * (d[1] - d[2]) + (d[4] - d[3]);
        t[3] = 2.0f * (d[3] - d[1]) + (d[4] - d[2]);
        t[4] = 2.0f * (d[1] - d[3]) + (d[4] - d[2]);
        t[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
    }
}
template <int TILE, typename V>
__device__ __forceinline__ void at_apply(const V* m, V* y) {
    if constexpr (TILE == 2) {  // A^T = [1 1 1 0; 0 1 -1 -1]
        y[0] = m[0] + m[1] + m[2];
        y[1] = m[1] - m[2] - m[3];
    } else {  // A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
        const V s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
        y[0] = m[0] + s12 + s34;
        y[1] = d12 + 2.0f * d34;
        y[2] = s12 + 4.0f * s34;
        y[3] = d12 + 8.0f * d34 + m[5];
    }
}

template <int TILE, typename V>
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoParams p) {
    constexpr int A = TILE + 2;
    constexpr int VN = VecOps<V>::N;
    const int C4 = (p.C0 + p.C1) / VN;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)p.T * C4;
    if (idx >= total) return;
    const int cg = (int)(idx % C4);
    const int t = (int)(idx / C4);
    const int tx = t % p.TW;
    const int t1 = t / p.TW;
    const int ty = t1 % p.TH;
    const int b = t1 / p.TH;
    const int c = cg * VN;
    const float* src;
    int pix;
    if (c < p.C0) {
        src = p.in0 + c; pix = p.C0;
    } else {
        src = p.in1 + (c - p.C0); pix = p.C1;
    }
    const int Hv = p.Hin << p.in_shift, Wv = p.Win << p.in_shift;
    V w[A][A];  // after the row pass: w[r][s] = (B^T d)[r][s]
    {
        V d[A][A];
#pragma unroll
        for (int r = 0; r < A; ++r) {
            const int iy = TILE * ty - 1 + r;
#pragma unroll
            for (int s = 0; s < A; ++s) {
                const int ix = TILE * tx - 1 + s;
                const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                const size_t pixel = (size_t)b * p.Hin * p.Win + (size_t)((ok ? iy : 0) >> p.in_shift) * p.Win +
                                     ((ok ? ix : 0) >> p.in_shift);
                const V v = *reinterpret_cast<const V*>(src + pixel * pix);
                d[r][s] = ok ? v : VecOps<V>::zero();
            }
        }
#pragma unroll
        for (int s = 0; s < A; ++s) {
            V col[A], tc[A];
#pragma unroll
            for (int r = 0; r < A; ++r) col[r] = d[r][s];
            bt_apply<TILE, V>(col, tc);
#pragma unroll
            for (int r = 0; r < A; ++r) w[r][s] = tc[r];
        }
    }
    const size_t Ctot = (size_t)(p.C0 + p.C1);
    float* vp = p.V + (size_t)t * Ctot + c;
    const size_t kstride = (size_t)p.T * Ctot;
#pragma unroll
    for (int r = 0; r < A; ++r) {
        V o[A];
        bt_apply<TILE, V>(w[r], o);
#pragma unroll
        for (int s = 0; s < A; ++s) *reinterpret_cast<V*>(vp + (size_t)(r * A + s) * kstride) = o[s];
    }
}

// Split-operand mode (gemm_split.hip): the same F(4x4,3x3) input transform, 4 channels per thread, but every V element is
// written as NPL bf16 pieces (round to nearest even, residual exact in f32) into NPL planes: 2 NPL bytes per element instead
// of 4, and the component GEMMs become bf16 GEMMs with f32-equivalent (NPL = 3) or 16-bit (NPL = 2) operands.
template <int NPL, bool PAIRS = false, bool F16 = false>
__global__ __launch_bounds__(256) void wino_input_split_kernel(const WinoParams p) {
    constexpr int A = 6;
    const int C4 = (p.C0 + p.C1) / 4;
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
    float4 w[A][A];
    {
        float4 d[A][A];
#pragma unroll
        for (int r = 0; r < A; ++r) {
            const int iy = 4 * ty - 1 + r;
#pragma unroll
            for (int s = 0; s < A; ++s) {
                const int ix = 4 * tx - 1 + s;
                const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                const size_t pixel = (size_t)b * p.Hin * p.Win + (size_t)((ok ? iy : 0) >> p.in_shift) * p.Win +
                                     ((ok ? ix : 0) >> p.in_shift);
                const float4 v = *reinterpret_cast<const float4*>(src + pixel * pix);
                d[r][s] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int s = 0; s < A; ++s) {
            float4 col[A], tc[A];
#pragma unroll
            for (int r = 0; r < A; ++r) col[r] = d[r][s];
            bt_apply<4, float4>(col, tc);
#pragma unroll
            for (int r = 0; r < A; ++r) w[r][s] = tc[r];
        }
    }
    const size_t Ctot = (size_t)(p.C0 + p.C1);
    // PAIRS: [component][tile][c / 32][plane][c % 32] (both pieces of a 32-channel block in one 128-byte line)
    unsigned short* vp = PAIRS ? p.Vs + ((size_t)t * (Ctot / 32) + c / 32) * 64 + (c & 31) : p.Vs + (size_t)t * Ctot + c;
    const size_t kstride = PAIRS ? (size_t)p.T * Ctot * 2 : (size_t)p.T * Ctot;
    const size_t plstride = PAIRS ? 32 : (size_t)p.v_plane;
#pragma unroll
    for (int r = 0; r < A; ++r) {
        float4 o[A];
        bt_apply<4, float4>(w[r], o);
#pragma unroll
        for (int s = 0; s < A; ++s) {
            float rem[4] = {o[s].x, o[s].y, o[s].z, o[s].w};
            if constexpr (F16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) rem[e] *= p.v_scale;
            }
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                unsigned short q[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (F16) {
                        const _Float16 hb = (_Float16)rem[e];
                        q[e] = __builtin_bit_cast(unsigned short, hb);
                        rem[e] -= (float)hb;
                    } else {
                        const __bf16 hb = (__bf16)rem[e];
                        q[e] = __builtin_bit_cast(unsigned short, hb);
                        rem[e] -= (float)hb;
                    }
                }
                uint2 pk;
                pk.x = (unsigned)q[0] | ((unsigned)q[1] << 16);
                pk.y = (unsigned)q[2] | ((unsigned)q[3] << 16);
                *reinterpret_cast<uint2*>(vp + (size_t)pl * plstride + (size_t)(r * A + s) * kstride) = pk;
            }
        }
    }
}

template <int TILE, typename V>
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoParams p) {
    constexpr int A = TILE + 2;
    constexpr int VN = VecOps<V>::N;
    const int N4 = p.Cout / VN;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)p.T * N4;
    if (idx >= total) return;
    const int ng = (int)(idx % N4);
    const int t = (int)(idx / N4);
    const int tx = t % p.TW;
    const int t1 = t / p.TW;
    const int ty = t1 % p.TH;
    const int b = t1 / p.TH;
    const int n = ng * VN;
    const float* mp = p.M + (size_t)t * p.Cout + n;
    const size_t kstride = (size_t)p.T * p.Cout;
    V u[TILE][A];  // u = A^T m (row pass)
    {
        V m[A][A];
#pragma unroll
        for (int r = 0; r < A; ++r)
#pragma unroll
            for (int s = 0; s < A; ++s) m[r][s] = *reinterpret_cast<const V*>(mp + (size_t)(r * A + s) * kstride);
#pragma unroll
        for (int s = 0; s < A; ++s) {
            V col[A], yc[TILE];
#pragma unroll
            for (int r = 0; r < A; ++r) col[r] = m[r][s];
            at_apply<TILE, V>(col, yc);
#pragma unroll
            for (int i = 0; i < TILE; ++i) u[i][s] = yc[i];
        }
    }
    V bias = VecOps<V>::zero(), sc = VecOps<V>::one(), sh = VecOps<V>::zero();
    if (p.bias) bias = *reinterpret_cast<const V*>(p.bias + n);
    if (p.film) {
        const float* f = p.film + (size_t)(p.film_bstride ? b : 0) * p.film_bstride;
        sc = *reinterpret_cast<const V*>(f + n) + VecOps<V>::one();
        sh = *reinterpret_cast<const V*>(f + p.Cout + n);
    }
    const int Ho = TILE * p.TH, Wo = TILE * p.TW;
#pragma unroll
    for (int i = 0; i < TILE; ++i) {
        V y[TILE];
        at_apply<TILE, V>(u[i], y);
#pragma unroll
        for (int j = 0; j < TILE; ++j) {
            const size_t pixel = ((size_t)b * Ho + TILE * ty + i) * Wo + TILE * tx + j;
            V v = y[j] + bias;
            if (p.film) v = VecOps<V>::mad(v, sc, sh);
            if (p.silu) v = VecOps<V>::silu(v);
            if (p.res) v = v + *reinterpret_cast<const V*>(p.res + pixel * p.res_stride + n);
            *reinterpret_cast<V*>(p.out + pixel * p.out_stride + n) = v;
        }
    }
}

}  // namespace

// IRSDE_WINO_VEC=1 (experiment): one channel per thread instead of four
static int wino_vec() {
    static const int v = tuning_env_int("IRSDE_WINO_VEC", 4);
    return v;
}

void launch_wino_input(const WinoParams& p, hipStream_t s) {
    if (p.Vs) {  // split-operand mode: bf16 planes
        if (p.tile != 4 || (p.nplanes != 2 && p.nplanes != 3) || (p.C0 % 4) || (p.C1 % 4)) throw HipError("wino_input (split): F(4x4,3x3), 2 or 3 planes");
        const long long tot = (long long)p.T * ((p.C0 + p.C1) / 4);
        const dim3 gr((unsigned)((tot + 255) / 256));
        if (p.v_pairs) {
            if (p.nplanes != 2 || (p.C0 + p.C1) % 32) throw HipError("wino_input (pairs): two planes, channels a multiple of 32");
            if (p.v_f16) hipLaunchKernelGGL((wino_input_split_kernel<2, true, true>), gr, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((wino_input_split_kernel<2, true, false>), gr, dim3(256), 0, s, p);
        } else if (p.nplanes == 3) hipLaunchKernelGGL((wino_input_split_kernel<3, false>), gr, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((wino_input_split_kernel<2, false>), gr, dim3(256), 0, s, p);
        IRSDE_HIP_CHECK(hipGetLastError());
        return;
    }
    const int vn = wino_vec() == 1 ? 1 : 4;
    const long long total = (long long)p.T * ((p.C0 + p.C1) / vn);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (p.tile == 4) {
        if (vn == 1) hipLaunchKernelGGL((wino_input_kernel<4, float>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((wino_input_kernel<4, float4>), grid, dim3(256), 0, s, p);
    } else {
        if (vn == 1) hipLaunchKernelGGL((wino_input_kernel<2, float>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((wino_input_kernel<2, float4>), grid, dim3(256), 0, s, p);
    }
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_wino_output(const WinoParams& p, hipStream_t s) {
    const int vn = wino_vec() == 1 ? 1 : 4;
    const long long total = (long long)p.T * (p.Cout / vn);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (p.tile == 4) {
        if (vn == 1) hipLaunchKernelGGL((wino_output_kernel<4, float>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((wino_output_kernel<4, float4>), grid, dim3(256), 0, s, p);
    } else {
        if (vn == 1) hipLaunchKernelGGL((wino_output_kernel<2, float>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((wino_output_kernel<2, float4>), grid, dim3(256), 0, s, p);
    }
    IRSDE_HIP_CHECK(hipGetLastError());
}

// U[k][n][c] = (G g G^T)_k;  g given as [Cout][3][3][Cin] (packed layout); tile = 2 or 4 (host, at weight load)
void wino_transform_weights(const float* w_packed, int Cout, int Cin, float* U, int tile) {
    static const float G2[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    static const float G4[6][3] = {{1.f / 4, 0.f, 0.f},          {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                                   {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    const int A = tile + 2;
    const float(*G)[3] = tile == 4 ? G4 : G2;
    const size_t kstride = (size_t)Cout * Cin;
    for (int n = 0; n < Cout; ++n)
        for (int c = 0; c < Cin; ++c) {
            double g[3][3];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w_packed[(((size_t)n * 3 + ky) * 3 + kx) * Cin + c];
            double tmp[6][3];
            for (int r = 0; r < A; ++r)
                for (int kx = 0; kx < 3; ++kx)
                    tmp[r][kx] = (double)G[r][0] * g[0][kx] + (double)G[r][1] * g[1][kx] + (double)G[r][2] * g[2][kx];
            for (int r = 0; r < A; ++r)
                for (int s = 0; s < A; ++s)
                    U[(size_t)(r * A + s) * kstride + (size_t)n * Cin + c] =
                        (float)(tmp[r][0] * G[s][0] + tmp[r][1] * G[s][1] + tmp[r][2] * G[s][2]);
        }
}

}  // namespace irsde
