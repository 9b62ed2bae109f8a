// Probe (r06): do vector-ALU instructions take matrix-pipe time away from v_mfma_f32_16x16x4_f32 on one MI355X SIMD?
// Every CU runs one 512-thread block = 2 waves per SIMD (the fused Winograd kernels' occupancy).  A wave's loop body = 8 independent MFMAs (256 matrix cycles)
// followed by NV vector instructions on private registers (inline assembly: nothing the compiler can fold or move) —
//   MODE 0  every wave issues both streams (the two-tile-group kernel wino4_fused64t: the transform inside the matrix waves)
//   MODE 1  waves 0-3 (one per SIMD) issue MFMAs only, waves 4-7 the vector stream only, 2 NV per loop (wino4_fused64p: producer waves beside matrix waves)
// KIND: 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_permlane32_swap, 3 ds_write_b64 (LDS, 8 B / lane), 4 v_mov_b32, 5 v_exp_f32, 6 v_rcp_f32.
// Prints the MFMA rate against the no-vector-work case.  If the two pipes issued independently the rate would not move until NV x (cycles per op) reached 256.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_coissue tools/probe/mfma_valu_coissue.hip && ./mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// part 2 (r06): the same question for the 16-bit matrix instruction of the bf16 / fp16 modes (v_mfma_f32_32x32x16_bf16: 16 passes = 64 cycles? measured):
// 4 independent MFMAs + NV vector instructions per loop, every wave both streams
template <int KIND, int NV>
__global__ __launch_bounds__(512, 2) void kb(const float* __restrict__ src, float* __restrict__ out, const int iters) {
    const floatx4 u0 = reinterpret_cast<const floatx4*>(src)[threadIdx.x * 2], u1 = reinterpret_cast<const floatx4*>(src)[threadIdx.x * 2 + 1];
    const bf16x8 a = __builtin_bit_cast(bf16x8, u0), b = __builtin_bit_cast(bf16x8, u1);
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    floatx2 r[8];
    for (int i = 0; i < 8; ++i) r[i] = floatx2{u0[i & 3] * 1e-3f, u1[i & 3] * 1e-3f};
    for (int it = 0; it < iters; ++it) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, acc[3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i & 7]) : "v"(r[(i + 3) & 7]), "v"(r[(i + 5) & 7]));
            else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i & 7].x) : "v"(r[(i + 3) & 7].y), "v"(r[(i + 5) & 7].x));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND, int NV>
__device__ __forceinline__ void vec_ops(floatx2 (&r)[8], float* lds) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i & 7]) : "v"(r[(i + 3) & 7]), "v"(r[(i + 5) & 7]));
        else if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i & 7].x) : "v"(r[(i + 3) & 7].y), "v"(r[(i + 5) & 7].x));
        else if constexpr (KIND == 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i & 7].x), "+v"(r[i & 7].y));
        else if constexpr (KIND == 3) asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(size_t)lds), "v"(r[i & 7]) : "memory");
        else if constexpr (KIND == 5) asm volatile("v_exp_f32 %0, %1" : "=v"(r[i & 7].x) : "v"(r[(i + 3) & 7].y));
        else if constexpr (KIND == 6) asm volatile("v_rcp_f32 %0, %1" : "=v"(r[i & 7].x) : "v"(r[(i + 3) & 7].y));
        else asm volatile("v_mov_b32 %0, %1" : "=v"(r[i & 7].x) : "v"(r[(i + 3) & 7].y));
    }
    if constexpr (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int MODE, int KIND, int NV>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* __restrict__ out, const int iters) {
    __shared__ float lds[8 * 64 * 2 + 64];
    const floatx4 u0 = reinterpret_cast<const floatx4*>(src)[threadIdx.x * 2], u1 = reinterpret_cast<const floatx4*>(src)[threadIdx.x * 2 + 1];
    const int wave = threadIdx.x >> 6;
    floatx4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx2 r[8];
    for (int i = 0; i < 8; ++i) r[i] = floatx2{u0[i & 3] * 1e-3f, u1[i & 3] * 1e-3f};
    float* my = lds + threadIdx.x * 2;
    const bool do_mfma = MODE == 0 || wave < 4, do_vec = MODE == 0 || wave >= 4;
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.x, u1.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.y, u1.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.z, u1.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.w, u1.w, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.x, u0.x, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.y, u0.y, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.z, u0.z, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.w, u0.w, acc[7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (do_vec) {
            if constexpr (MODE == 0) vec_ops<KIND, NV>(r, my); else { vec_ops<KIND, NV>(r, my); vec_ops<KIND, NV>(r, my); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w + r[i].x + r[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static float* g_out;
static const float* g_src;

template <int MODE, int KIND, int NV>
double run() {
    const int nb = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 20000;
    hipLaunchKernelGGL((k<MODE, KIND, NV>), dim3(nb), dim3(512), 0, 0, g_src, g_out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, KIND, NV>), dim3(nb), dim3(512), 0, 0, g_src, g_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * 120.0 / ms);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, KIND, NV>), dim3(nb), dim3(512), 0, 0, g_src, g_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = MODE == 0 ? 8.0 : 4.0;
    const double flop = (double)nb * waves * iters * 8 * 2048.0;
    // cycles per loop iteration per SIMD at 2.4 GHz nominal
    const double cyc = ms * 1e-3 * 2.4e9 / iters;
    const double tf = flop / ms / 1e9;
    printf("mode %d kind %d NV %3d : %7.1f TFLOP/s MFMA (%.3f of 157.3)   %.0f nominal cycles per loop and SIMD (MFMA floor %d)\n", MODE, KIND, NV, tf, tf / 157.3, cyc, MODE == 0 ? 512 : 256);
    return tf;
}

template <int KIND, int NV>
void runb() {
    const int nb = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int iters = 20000;
    hipLaunchKernelGGL((kb<KIND, NV>), dim3(nb), dim3(512), 0, 0, g_src, g_out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kb<KIND, NV>), dim3(nb), dim3(512), 0, 0, g_src, g_out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * 120.0 / ms);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kb<KIND, NV>), dim3(nb), dim3(512), 0, 0, g_src, g_out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)nb * 8.0 * iters * 4 * 32768.0;
    const double tf = flop / ms / 1e9;
    printf("v_mfma_f32_32x32x16_bf16 x 4 + %3d x %s per wave and loop : %8.1f TFLOP/s (%.3f of 2516.6)   %.0f nominal cycles per loop and SIMD\n", NV,
           KIND == 0 ? "v_pk_fma_f32" : "v_fma_f32", tf, tf / 2516.6, ms * 1e-3 * 2.4e9 / iters);
}

template <int MODE, int KIND>
void sweep(const char* name) {
    printf("---- %s, %s\n", name, MODE == 0 ? "every wave: 8 MFMAs + NV vector ops per loop (2 waves per SIMD)" : "matrix waves (1 per SIMD) beside vector waves (1 per SIMD, 2 NV ops per loop)");
    run<MODE, KIND, 0>();
    run<MODE, KIND, 4>();
    run<MODE, KIND, 8>();
    run<MODE, KIND, 16>();
    run<MODE, KIND, 32>();
    run<MODE, KIND, 64>();
}

int main() {
    const size_t n = 512 * 8;
    std::vector<float> hr(n);
    srand(7);
    for (auto& v : hr) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float* dr;
    hipMalloc(&dr, n * 4);
    hipMemcpy(dr, hr.data(), n * 4, hipMemcpyHostToDevice);
    g_src = dr;
    hipMalloc(&g_out, (size_t)256 * 512 * 4);
    sweep<0, 0>("v_pk_fma_f32");
    sweep<0, 1>("v_fma_f32");
    sweep<0, 2>("v_permlane32_swap_b32");
    sweep<0, 3>("ds_write_b64");
    sweep<0, 4>("v_mov_b32");
    sweep<0, 5>("v_exp_f32 (transcendental)");
    sweep<0, 6>("v_rcp_f32 (transcendental)");
    sweep<1, 0>("v_pk_fma_f32");
    sweep<1, 1>("v_fma_f32");
    sweep<1, 4>("v_mov_b32");
    printf("---- 16-bit matrix instruction (random operands: the chip clocks down under it, profiles/r05_mfma_sustained.txt), every wave 4 MFMAs + NV vector ops per loop\n");
    runb<0, 0>(); runb<0, 4>(); runb<0, 8>(); runb<0, 16>(); runb<0, 32>(); runb<0, 64>();
    runb<1, 0>(); runb<1, 8>(); runb<1, 16>(); runb<1, 32>(); runb<1, 64>();
    return 0;
}
