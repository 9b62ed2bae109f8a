// Probe (r05): what the matrix pipe of one MI355X SUSTAINS on a pure MFMA stream — every CU, 2 waves per SIMD, independent accumulators, no memory traffic in the
// loop — for the three instructions the library's hot kernels issue, on zero operands and on random operands (the chip clocks to its power budget: dense MFMA streams
// on toggling data run at a lower clock than the nominal 2.4 GHz the peak figures assume; guides/MI355X_MICROARCH.md, DVFS).  ~0.3 s per case so the clock settles.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_sustained tools/probe/mfma_sustained.hip && ./mfma_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* __restrict__ out, const int iters) {
    // operands: 8 registers per lane from memory (zeros or random), fixed for the whole loop
    const floatx4 u0 = reinterpret_cast<const floatx4*>(src)[threadIdx.x * 2], u1 = reinterpret_cast<const floatx4*>(src)[threadIdx.x * 2 + 1];
    float s = 0.f;
    if constexpr (KIND == 0) {   // v_mfma_f32_32x32x2_f32: 64 cycles, 4096 FLOP... (32*32*2*2)
        floatx16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.x, u1.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.y, u1.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.z, u1.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.w, u1.w, acc[3], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else if constexpr (KIND == 1) {   // v_mfma_f32_16x16x4_f32: 32 cycles
        floatx4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.x, u1.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.y, u1.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.z, u1.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.w, u1.w, acc[3], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.x, u0.x, acc[4], 0, 0, 0);
                acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.y, u0.y, acc[5], 0, 0, 0);
                acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.z, u0.z, acc[6], 0, 0, 0);
                acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.w, u0.w, acc[7], 0, 0, 0);
            }
        }
        for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    } else {   // v_mfma_f32_32x32x16_bf16: 32 cycles
        floatx16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const bf16x8 a = __builtin_bit_cast(bf16x8, u0), b = __builtin_bit_cast(bf16x8, u1);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, acc[3], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, const float* dsrc, const char* data, double flop_per_mfma, int mfma_per_iter, int cycles, double nominal) {
    const int nb = 256 * 2;   // 2 blocks of 8 waves per CU would not fit 2 waves per SIMD: launch_bounds(512, 2) -> one 8-wave block per CU = 2 waves per SIMD
    float* d; hipMalloc(&d, (size_t)nb * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    hipLaunchKernelGGL((k<KIND>), dim3(nb), dim3(512), 0, 0, dsrc, d, iters);   // warm-up
    hipDeviceSynchronize();
    // size the run to ~0.3 s
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(nb), dim3(512), 0, 0, dsrc, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * 300.0 / ms);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(nb), dim3(512), 0, 0, dsrc, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)nb * 8 * iters * mfma_per_iter;   // wave-instructions
    const double tf = n_mfma * flop_per_mfma / ms / 1e9;
    // implied clock if the pipe issues back to back: per SIMD (1024 of them) n_mfma / 1024 * cycles cycles in ms
    const double ghz = n_mfma / 1024.0 * cycles / (ms * 1e-3) / 1e9;
    printf("%-28s %-7s %8.1f ms  %8.1f TFLOP/s  = %.3f of the nominal %.1f; implied clock at back-to-back issue %.2f GHz\n", name, data, ms, tf, tf / nominal, nominal, ghz);
    hipFree(d);
}

// ---- part 2: the two f32 instructions against occupancy (1 / 2 / 4 waves per SIMD), accumulator reuse distance (4 / 8) and issue density (an s_sleep gap every 8 MFMAs) ----
template <int KIND, int NACC, int GAP>
__global__ __launch_bounds__(256) void k2(const float* __restrict__ src, float* __restrict__ out, const int iters) {
    extern __shared__ float smem2[];
    const floatx4 u0 = reinterpret_cast<const floatx4*>(src)[threadIdx.x * 2], u1 = reinterpret_cast<const floatx4*>(src)[threadIdx.x * 2 + 1];
    float s = 0.f;
    if constexpr (KIND == 0) {
        floatx16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 32; ++g) {
                acc[g % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0[g & 3], u1[(g >> 2) & 3], acc[g % NACC], 0, 0, 0);
                if (GAP && (g & 7) == 7) __builtin_amdgcn_s_sleep(GAP);
            }
        }
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        floatx4 acc[NACC * 4];
        for (int i = 0; i < NACC * 4; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 64; ++g) {
                acc[g % (NACC * 4)] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0[g & 3], u1[(g >> 2) & 3], acc[g % (NACC * 4)], 0, 0, 0);
                if (GAP && (g & 15) == 15) __builtin_amdgcn_s_sleep(GAP);
            }
        }
        for (int i = 0; i < NACC * 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    }
    if (threadIdx.x == 9999) smem2[0] = s;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NACC, int GAP>
void run2(const float* dsrc, int waves_per_simd) {
    const int nb = 256 * waves_per_simd;
    const int lds = waves_per_simd == 1 ? 150 * 1024 : waves_per_simd == 2 ? 72 * 1024 : 36 * 1024;
    float* d; hipMalloc(&d, (size_t)nb * 256 * 4);
    hipFuncSetAttribute((const void*)k2<KIND, NACC, GAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    hipLaunchKernelGGL((k2<KIND, NACC, GAP>), dim3(nb), dim3(256), lds, 0, dsrc, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k2<KIND, NACC, GAP>), dim3(nb), dim3(256), lds, 0, dsrc, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * 250.0 / ms);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k2<KIND, NACC, GAP>), dim3(nb), dim3(256), lds, 0, dsrc, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)nb * 4 * iters * (KIND == 0 ? 32 * 4096.0 : 64 * 2048.0);
    printf("%-24s waves/SIMD %d  accumulators %2d  s_sleep %d per 512 MFMA cycles : %8.1f TFLOP/s (%.3f of 157.3)\n", KIND == 0 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32",
           waves_per_simd, KIND == 0 ? NACC : NACC * 4, GAP, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
    hipFree(d);
}

int main() {
    const size_t n = 512 * 8;
    std::vector<float> hz(n, 0.f), hr(n);
    srand(7);
    for (auto& v : hr) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *dz, *dr;
    hipMalloc(&dz, n * 4); hipMalloc(&dr, n * 4);
    hipMemcpy(dz, hz.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dr, hr.data(), n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("v_mfma_f32_32x32x2_f32", dz, "zeros", 4096.0, 32, 64, 157.3);
        run<0>("v_mfma_f32_32x32x2_f32", dr, "random", 4096.0, 32, 64, 157.3);
        run<1>("v_mfma_f32_16x16x4_f32", dz, "zeros", 2048.0, 64, 32, 157.3);
        run<1>("v_mfma_f32_16x16x4_f32", dr, "random", 2048.0, 64, 32, 157.3);
        run<2>("v_mfma_f32_32x32x16_bf16", dz, "zeros", 32768.0, 32, 32, 2516.6);
        run<2>("v_mfma_f32_32x32x16_bf16", dr, "random", 32768.0, 32, 32, 2516.6);
    }
    for (int w : {1, 2, 4}) { run2<0, 4, 0>(dr, w); run2<1, 4, 0>(dr, w); }
    run2<0, 8, 0>(dr, 2); run2<1, 2, 0>(dr, 2);
    run2<0, 4, 1>(dr, 2); run2<1, 4, 1>(dr, 2);
    run2<0, 4, 2>(dr, 2); run2<1, 4, 2>(dr, 2);
    run2<0, 4, 0>(dz, 2); run2<1, 4, 0>(dz, 2);
    return 0;
}
