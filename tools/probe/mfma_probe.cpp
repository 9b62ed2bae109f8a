// Probe: how well do two co-resident waves per SIMD overlap an MFMA phase with a non-MFMA gap on gfx950?
// Each wave: repeat { 64 x v_mfma_f32_32x32x2_f32 (16 groups of 4 chained) ; gap } ; blocks of 4 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int GAP, int PRIO>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, int gaplen) {
    extern __shared__ float smem[];
    if (PRIO) {
        unsigned hwid = __builtin_amdgcn_s_getreg((3 << 11) | 4);
        if (hwid & 1) __builtin_amdgcn_s_setprio(PRIO);
    }
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    float v = a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g & 3], 0, 0, 0);
            acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[g & 3], 0, 0, 0);
            acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[g & 3], 0, 0, 0);
            acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[g & 3], 0, 0, 0);
        }
        if (GAP == 1) {  // pure wait
            for (int s = 0; s < gaplen; ++s) __builtin_amdgcn_s_sleep(1);
        } else if (GAP == 2) {  // dependent VALU chain
            for (int s = 0; s < gaplen; ++s) v = v * 1.0001f + 0.5f;
            a += v * 1e-30f;
        } else if (GAP == 3) {  // LDS write + barrier + LDS read (a K-step boundary)
            smem[threadIdx.x + (it & 1) * 256] = v;
            __syncthreads();
            v = smem[(threadIdx.x ^ 1) + (it & 1) * 256];
            a += v * 1e-30f;
            for (int s = 0; s < gaplen; ++s) __builtin_amdgcn_s_sleep(1);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + v;
}

template <int GAP, int PRIO>
void run(const char* name, int lds, int gaplen) {
    const int nb = 2048, iters = 200;
    float* d; hipMalloc(&d, nb * 256 * 4);
    hipFuncSetAttribute((const void*)k<GAP, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<GAP, PRIO>), dim3(nb), dim3(256), lds, 0, d, iters, gaplen);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<GAP, PRIO>), dim3(nb), dim3(256), lds, 0, d, iters, gaplen);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)nb * 4 * iters * 64 * 4096.0;
    printf("%-44s lds %6d gap %4d : %7.3f ms  %6.1f TF/s\n", name, lds, gaplen, ms, fl / ms / 1e9);
    hipFree(d);
}
int main() {
    for (int lds : {120 * 1024, 70 * 1024}) {   // 1 block/CU vs 2 blocks/CU
        run<0, 0>("no gap", lds, 0);
        run<1, 0>("sleep gap", lds, 12);
        run<1, 0>("sleep gap", lds, 24);
        run<1, 2>("sleep gap, prio by slot", lds, 24);
        run<2, 0>("VALU-chain gap", lds, 200);
        run<2, 0>("VALU-chain gap", lds, 400);
        run<2, 2>("VALU-chain gap, prio by slot", lds, 400);
        run<3, 0>("lds+barrier gap", lds, 0);
        run<3, 0>("lds+barrier + sleep gap", lds, 12);
        run<3, 2>("lds+barrier + sleep gap, prio", lds, 12);
    }
    return 0;
}
