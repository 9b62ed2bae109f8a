// Probe for the limiter of the fused Winograd kernels (profiles/r04_wino_fused64_notes.md): what does a producer-style burst of vector-memory
// instructions cost to ISSUE on a CU whose four other waves run back-to-back f32 MFMAs and stream one-kilobyte weight fragments?
// Block = 512 threads on one CU (grid = 256), as in wino4_fused64p_kernel:
//   waves 0-3 ("MFMA waves"): per unit 4 x v_mfma_f32_16x16x4_f32 (128 cycles) + one 1 KB buffer load into a ring of 12 (from a 2 MB buffer: L2 hits);
//              reports cycles per 72 units (= one 32-channel chunk; MFMA floor 9216)
//   waves 4-7 ("producers"): once per ~chunk a burst of loads, reports cycles to ISSUE the burst (until the last instruction has left the wave) and to COMPLETE it
// Switches: mfma (0: the MFMA waves sleep instead of issuing MFMAs), ring (0: no weight loads), kind of the producer loads:
//   0 none; 1 36 x buffer_load_dwordx2 gathers (4 lines of 128 B per instruction: the register-patch kernel); 2 18 x buffer_load_dwordx4 (1 KB contiguous);
//   3 18 x buffer_load_dwordx4 ... lds (LDS-DMA, the halo kernel); src: 0 = an L2-resident window, 1 = a 6 GB buffer (every line a first touch)
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/vmem_issue.hip -o tools/probe/vmem_issue ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void probe(const float* small, unsigned small_bytes, const float* big, unsigned long long big_bytes, int mfma, int ring_on,
                                                int kind, int src, int chunks, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(small), 0, small_bytes, 0x00020000);
    float accs = 0.f;
    if (wave < 4) {
        floatx4 acc[8] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        floatx4 ring[12];
        unsigned off = ((blockIdx.x * 4 + wave) * 7919u * 1024u) % (small_bytes - 4096u);
        for (int i = 0; i < 12; ++i) {
            ring[i] = floatx4{1.f, 1.f, 1.f, 1.f};
            if (ring_on) ring[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (int)off, 0));
            off = (off + 1024u) % (small_bytes - 4096u);
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int c = 0; c < chunks; ++c) {
#pragma unroll 1
            for (int u6 = 0; u6 < 6; ++u6) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    if (mfma) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[4 * (i & 1) + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[i][j], 1.0f, acc[4 * (i & 1) + j], 0, 0, 0);
                    } else {
                        accs += ring[i][0];
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (ring_on) ring[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (int)off, 0));
                    off = (off + 1024u) % (small_bytes - 4096u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        accs += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][0] + acc[5][1] + acc[6][2] + acc[7][3];
        if (lane == 0) out[(blockIdx.x * 8 + wave) * 4 + 0] = (t1 - t0) / chunks;
    } else {
        unsigned long long pos = ((unsigned long long)(blockIdx.x * 8 + wave) * 1000003ull * 4096ull) % (big_bytes - (1ull << 24));
        unsigned spos = ((blockIdx.x * 8 + wave) * 104729u * 512u) % (256u << 10);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src ? big : small), 0, 0x7fffffffu, 0x00020000);
        unsigned long long t_issue = 0, t_done = 0;
        for (int c = 0; c < chunks && kind; ++c) {
            const unsigned long long base = src ? pos : spos;
            const unsigned long long a = __builtin_amdgcn_s_memtime();
            if (kind == 1) {
                floatx2 r[36];
#pragma unroll
                for (int e = 0; e < 36; ++e)
                    r[e] = *reinterpret_cast<const floatx2*>(reinterpret_cast<const char*>(src ? big : small) + base + (unsigned long long)e * 1024ull + (lane >> 4) * (src ? 262144 : 16384) + (lane & 15) * 8);
                const unsigned long long b = __builtin_amdgcn_s_memtime();
#pragma unroll
                for (int e = 0; e < 36; ++e) accs += r[e][0];
                const unsigned long long d = __builtin_amdgcn_s_memtime();
                t_issue += b - a; t_done += d - a;
            } else if (kind == 2) {
                floatx4 r[18];
#pragma unroll
                for (int e = 0; e < 18; ++e)
                    r[e] = *reinterpret_cast<const floatx4*>(reinterpret_cast<const char*>(src ? big : small) + base + (unsigned long long)e * 4096ull + lane * 16);
                const unsigned long long b = __builtin_amdgcn_s_memtime();
#pragma unroll
                for (int e = 0; e < 18; ++e) accs += r[e][0];
                const unsigned long long d = __builtin_amdgcn_s_memtime();
                t_issue += b - a; t_done += d - a;
            } else {
#pragma unroll
                for (int e = 0; e < 18; ++e)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(lds + (wave - 4) * 4608 + e * 256), 16, (int)(lane * 16), (int)((base + e * 4096ull) & 0x3fffffffull), 0, 0);
                const unsigned long long b = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long d = __builtin_amdgcn_s_memtime();
                t_issue += b - a; t_done += d - a;
            }
            pos = (pos + 36ull * 262144ull + 4096ull * 131ull) % (big_bytes - (1ull << 24));
            spos = (spos + 73728u) % (256u << 10);
            __builtin_amdgcn_s_sleep(100);   // ~6.4k cycles: one burst per chunk period or so
        }
        if (lane == 0) {
            out[(blockIdx.x * 8 + wave) * 4 + 1] = kind ? t_issue / chunks : 0;
            out[(blockIdx.x * 8 + wave) * 4 + 2] = kind ? t_done / chunks : 0;
        }
    }
    if (accs == 123.456f) sink[0] = accs;
}

int main() {
    const unsigned small_bytes = 2u << 20;   // fits one XCD's 4 MB L2 (8 MB did not: every weight fragment came from the MALL, 5 B/clk per CU)
    const unsigned long long big_bytes = 6ull << 30;
    float *small, *big, *sink; unsigned long long* out;
    hipMalloc(&small, small_bytes); hipMalloc(&big, big_bytes); hipMalloc(&sink, 64); hipMalloc(&out, 256 * 8 * 4 * 8);
    hipMemset(small, 0, small_bytes); hipMemset(big, 0, big_bytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int chunks = 40;
    static const char* kinds[4] = {"none", "36 x dwordx2 gather (4 lines each)", "18 x dwordx4 (1 KB each)", "18 x dwordx4 LDS-DMA (1 KB each)"};
    printf("%-5s %-5s %-38s %-8s | MFMA waves: cycles per 72 units (floor 9216) | producer burst: cycles to issue / to complete (per instruction)\n", "mfma", "ring", "producer loads", "source");
    for (int mfma = 1; mfma >= 0; --mfma)
        for (int ring = 1; ring >= 0; --ring)
            for (int kind = 0; kind < 4; ++kind)
                for (int src = 0; src < (kind ? 2 : 1); ++src) {
                    for (int rep = 0; rep < 2; ++rep) {
                        hipMemset(out, 0, 256 * 8 * 4 * 8);
                        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 100 * 1024, 0, small, small_bytes, big, big_bytes, mfma, ring, kind, src, chunks, out, sink);
                        hipDeviceSynchronize();
                    }
                    std::vector<unsigned long long> h(256 * 8 * 4);
                    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
                    double tm = 0, ti = 0, td = 0;
                    for (int b = 0; b < 256; ++b) {
                        for (int w = 0; w < 4; ++w) tm += h[(b * 8 + w) * 4];
                        for (int w = 4; w < 8; ++w) { ti += h[(b * 8 + w) * 4 + 1]; td += h[(b * 8 + w) * 4 + 2]; }
                    }
                    const int n = kind == 1 ? 36 : 18;
                    printf("%-5d %-5d %-38s %-8s | %10.0f | %8.0f / %8.0f  (%.0f / %.0f)\n", mfma, ring, kinds[kind], kind ? (src ? "HBM" : "L2") : "-", tm / 1024, ti / 1024, td / 1024,
                           ti / 1024 / n, td / 1024 / n);
                }
    return 0;
}
