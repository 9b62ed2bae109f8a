// Micro-benchmark (gfx950): do the f32 MFMAs of one wave overlap with the ordinary vector instructions of ANOTHER wave on
// the same SIMD?  256 threads per block, one block per CU, 8 waves = 2 per SIMD (waves w and w + 4 share a SIMD).  Waves 0-3
// run NM dependent-free MFMAs, waves 4-7 run NV packed / scalar FMAs; each role can be switched off.  Prints cycles per block.
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap tools/probe/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>  // 0: f32 MFMA 32x32x2, 1: bf16 MFMA 32x32x16
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int nm, int nv, int do_m, int do_v, int packed) {
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (do_m) {
            floatx16 acc[4];
            for (int i = 0; i < 4; ++i)
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
            bf16x8 ab, bb;
            for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
            for (int it = 0; it < nm; it += 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
                }
            }
            float s = 0.f;
            for (int i = 0; i < 4; ++i)
                for (int r = 0; r < 16; ++r) s += acc[i][r];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        }
    } else if (do_v) {
        if (packed) {
            floatx2 x[8];
            for (int i = 0; i < 8; ++i) x[i] = floatx2{threadIdx.x * 1e-3f + i, 1.0f};
            const floatx2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
            for (int it = 0; it < nv; it += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], m, c);
            }
            floatx2 s = x[0];
            for (int i = 1; i < 8; ++i) s += x[i];
            out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y;
        } else {
            float x[8];
            for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
            for (int it = 0; it < nv; it += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 1e-3f);
            }
            float s = 0.f;
            for (int i = 0; i < 8; ++i) s += x[i];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static double run(int nm, int nv, int do_m, int do_v, int packed) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, cyc, nm, nv, do_m, do_v, packed);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    hipFree(out); hipFree(cyc);
    return s / 256;
}

int main() {
    const int NM = 4096, NV = 16384;
    for (int packed = 0; packed < 2; ++packed) {
        printf("== vector stream: %s FMA, %d instructions per wave ==\n", packed ? "v_pk_fma_f32" : "v_fma_f32", NV);
        printf("f32  MFMA x%d alone: %.0f cycles | vector alone: %.0f | both: %.0f\n", NM, run<0>(NM, NV, 1, 0, packed),
               run<0>(NM, NV, 0, 1, packed), run<0>(NM, NV, 1, 1, packed));
        printf("bf16 MFMA x%d alone: %.0f cycles | vector alone: %.0f | both: %.0f\n", 2 * NM, run<1>(2 * NM, NV, 1, 0, packed),
               run<1>(2 * NM, NV, 0, 1, packed), run<1>(2 * NM, NV, 1, 1, packed));
    }
    return 0;
}
