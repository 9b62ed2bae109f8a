// GPU check of the vector-pipe group sums of csrc/kernels_misc.hip (group_sum<W>, half_sum, half_max: DPP quad permutes / row mirrors +
// v_permlane16_swap / v_permlane32_swap) against the __shfl_xor butterflies they replace.   build: hipcc --offload-arch=gfx950 -O3 group_sum_probe.hip -o group_sum_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
struct SwapPair { float a, b; };
__device__ __forceinline__ SwapPair swap16(const float v) { SwapPair r{v, v}; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(r.a), "+v"(r.b)); return r; }
__device__ __forceinline__ SwapPair swap32(const float v) { SwapPair r{v, v}; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r.a), "+v"(r.b)); return r; }
template <int W>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (W >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    if constexpr (W >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    if constexpr (W >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    if constexpr (W >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    if constexpr (W >= 32) { const SwapPair r = swap16(v); v = r.a + r.b; }
    if constexpr (W >= 64) { const SwapPair r = swap32(v); v = r.a + r.b; }
    return v;
}
template <int W>
__global__ void probe(const float* in, float* out_new, float* out_ref, float* out_max) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const float v = in[i];
    out_new[i] = group_sum<W>(v);
    float s = v;
    for (int o = W >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    out_ref[i] = s;
    const SwapPair r = swap32(v);
    out_max[i] = fmaxf(r.a, r.b) - fmaxf(v, __shfl_xor(v, 32, 64));
}
template <int W>
static int run(const float* din, float* d1, float* d2, float* d3, const std::vector<float>& h, int n) {
    probe<W><<<n / 64, 64>>>(din, d1, d2, d3);
    std::vector<float> a(n), b(n), c(n);
    hipMemcpy(a.data(), d1, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d2, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d3, n * 4, hipMemcpyDeviceToHost);
    double worst = 0, wmax = 0;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        // exact group total in double as the judge of both orders
        double t = 0;
        const int g0 = i / W * W;
        for (int k = 0; k < W; ++k) t += h[g0 + k];
        const double e = std::fabs(a[i] - t) / (std::fabs(t) + 1.0), er = std::fabs(b[i] - t) / (std::fabs(t) + 1.0);
        worst = std::max(worst, e);
        if (e > 2e-5 || er > 2e-5) ++bad;
        if (a[i] != a[g0]) ++bad;   // every lane of the group holds the same bits
        wmax = std::max(wmax, (double)std::fabs(c[i]));
    }
    printf("group_sum<%2d>: worst relative error vs the exact sum %.2e, half_max mismatch %.1e, bad %d\n", W, worst, wmax, bad);
    return bad + (wmax != 0.0);
}
int main() {
    const int n = 64 * 64;
    std::vector<float> h(n);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 8) % 2001 - 1000) / 37.0f; }
    float *din, *d1, *d2, *d3;
    hipMalloc(&din, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4); hipMalloc(&d3, n * 4);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    int bad = 0;
    bad += run<2>(din, d1, d2, d3, h, n); bad += run<4>(din, d1, d2, d3, h, n); bad += run<8>(din, d1, d2, d3, h, n);
    bad += run<16>(din, d1, d2, d3, h, n); bad += run<32>(din, d1, d2, d3, h, n); bad += run<64>(din, d1, d2, d3, h, n);
    printf(bad ? "FAILED\n" : "OK\n");
    return bad != 0;
}
