// Micro-benchmark (gfx950): how many ordinary vector instructions fit BETWEEN the MFMAs of the SAME wave for free?
// One wave per SIMD (256 threads per block, one block per CU).  Each wave runs NM MFMAs (4 independent accumulators), with
// K independent v_pk_fma_f32 / v_fma_f32 after every MFMA.  Prints cycles per MFMA for K = 0..16.
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_samewave tools/probe/mfma_valu_samewave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int K, int PACKED>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int nm) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
    floatx2 x[16];
    for (int i = 0; i < 16; ++i) x[i] = floatx2{threadIdx.x * 1e-3f + i, 1.0f};
    const floatx2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    for (int it = 0; it < nm; it += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (PACKED) x[j] = __builtin_elementwise_fma(x[j], m, c);
                else x[j].x = __builtin_fmaf(x[j].x, 1.0001f, 1e-3f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int K, int PACKED>
static double run(int nm) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, K, PACKED>), dim3(256), dim3(256), 0, 0, out, cyc, nm);
    (void)hipDeviceSynchronize();
    unsigned long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    (void)hipFree(out); (void)hipFree(cyc);
    return s / 256 / nm;
}

template <int MODE, int PACKED>
static void row(const char* name) {
    const int NM = 4096;
    printf("%-34s K=0 %.1f | 1 %.1f | 2 %.1f | 4 %.1f | 6 %.1f | 8 %.1f | 12 %.1f | 16 %.1f   cycles per MFMA\n", name,
           run<MODE, 0, PACKED>(NM), run<MODE, 1, PACKED>(NM), run<MODE, 2, PACKED>(NM), run<MODE, 4, PACKED>(NM),
           run<MODE, 6, PACKED>(NM), run<MODE, 8, PACKED>(NM), run<MODE, 12, PACKED>(NM), run<MODE, 16, PACKED>(NM));
}

int main() {
    row<0, 0>("f32 32x32x2  + K x v_fma_f32");
    row<0, 1>("f32 32x32x2  + K x v_pk_fma_f32");
    row<1, 0>("bf16 32x32x16 + K x v_fma_f32");
    row<1, 1>("bf16 32x32x16 + K x v_pk_fma_f32");
    return 0;
}
