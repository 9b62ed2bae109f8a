// Probe: does an HBM-missing load stream of one wave delay the L2-hitting loads of ANOTHER wave on the same CU (in-order vector L1 / TA path)?
// Block = 2 waves on one CU.  Wave 0 ("hit" stream): 12 x 1 KB buffer loads in flight from a small L2-resident buffer (the weight-fragment
// ring of wino4_fused64p_kernel), timing each wait.  Wave 1 ("side" stream), per mode:
//   0 idle;  1 bursts of 36 x 512 B loads from a 4 GB buffer (every line an HBM miss: the patch loads);  2 the same bursts from the small buffer;
//   3 bursts of 9 x 1 KB... (halo-style: fewer, wider loads, same misses per byte)
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/tcp_order.hip -o tools/probe/tcp_order ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(320) void probe(const float* small, unsigned small_bytes, const float* big, unsigned long long big_bytes, int mode, int iters, int pace,
                                             unsigned long long* out, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(small), 0, small_bytes, 0x00020000);
    float acc = 0.f;
    if (wave == 0) {
        floatx4 ring[12];
        unsigned off = (blockIdx.x * 7919u * 1024u) % small_bytes;
        for (int i = 0; i < 12; ++i) {
            ring[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (int)off, 0));
            off = (off + 1024u) % small_bytes;
        }
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), worst = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const unsigned long long a = __builtin_amdgcn_s_memtime();
                acc += ring[i][0] + ring[i][3];          // waits for the oldest load
                const unsigned long long b = __builtin_amdgcn_s_memtime();
                worst = b - a > worst ? b - a : worst;
                ring[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (int)off, 0));
                off = (off + 1024u) % small_bytes;
                if (pace) __builtin_amdgcn_s_sleep(1);   // ~64+ cycles of "MFMA work" per unit
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[blockIdx.x * 4 + 0] = t1 - t0; out[blockIdx.x * 4 + 1] = worst; }
    } else {
        if (mode == 0) { if (lane == 0) { out[blockIdx.x * 4 + 2] = 0; } }
        else {
            unsigned long long pos = ((unsigned long long)(blockIdx.x * 4 + wave) * 1000003ull * 4096ull) % (big_bytes - (1ull << 22));
            unsigned spos = ((blockIdx.x * 4 + wave) * 104729u * 512u) % (small_bytes - (1u << 16));
            unsigned long long t0 = __builtin_amdgcn_s_memtime();
            const int bursts = iters * 12 / (pace ? 72 : 200) + 1;     // one burst per ~72 units of wave 0 (one 32-channel chunk)
            for (int bi = 0; bi < bursts; ++bi) {
                if (mode == 1) {
                    floatx2 r[36];
#pragma unroll
                    for (int e = 0; e < 36; ++e) r[e] = *reinterpret_cast<const floatx2*>(reinterpret_cast<const char*>(big) + pos + (unsigned long long)e * 65536ull + (lane >> 4) * 16384 + (lane & 15) * 8);
#pragma unroll
                    for (int e = 0; e < 36; ++e) acc += r[e][0];
                    pos = (pos + 36ull * 65536ull + 4096ull * 131ull) % (big_bytes - (1ull << 22));
                } else if (mode == 2) {
                    floatx2 r[36];
#pragma unroll
                    for (int e = 0; e < 36; ++e) r[e] = *reinterpret_cast<const floatx2*>(reinterpret_cast<const char*>(small) + spos + e * 1024 + (lane >> 4) * 256 + (lane & 15) * 8);
#pragma unroll
                    for (int e = 0; e < 36; ++e) acc += r[e][0];
                    spos = (spos + 36 * 1024) % (small_bytes - (1u << 16));
                } else {
                    floatx4 r[10];
#pragma unroll
                    for (int e = 0; e < 10; ++e) r[e] = *reinterpret_cast<const floatx4*>(reinterpret_cast<const char*>(big) + pos + (unsigned long long)e * 65536ull + (lane >> 3) * 4096 + (lane & 7) * 16);
#pragma unroll
                    for (int e = 0; e < 10; ++e) acc += r[e][0];
                    pos = (pos + 10ull * 65536ull + 4096ull * 131ull) % (big_bytes - (1ull << 22));
                }
                __builtin_amdgcn_s_sleep(100);           // ~6.4k cycles between bursts
            }
            if (lane == 0 && wave == 1) out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime() - t0;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const unsigned small_bytes = 8u << 20;              // 8 MB: L2 / MALL resident
    const unsigned long long big_bytes = 6ull << 30;
    float *small, *big, *sink; unsigned long long* out;
    hipMalloc(&small, small_bytes); hipMalloc(&big, big_bytes); hipMalloc(&sink, 64); hipMalloc(&out, 256 * 4 * 8);
    hipMemset(small, 0, small_bytes); hipMemset(big, 0, big_bytes);
    const int iters = 2000;
    for (int pace = 1; pace >= 0; --pace)
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(out, 0, 256 * 4 * 8);
            hipLaunchKernelGGL(probe, dim3(256), dim3(320), 0, 0, small, small_bytes, big, big_bytes, mode, iters, pace, out, sink);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(256 * 4);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double t = 0, w = 0, s = 0;
        for (int b = 0; b < 256; ++b) { t += h[b * 4]; w += h[b * 4 + 1]; s += h[b * 4 + 2]; }
        printf("pace %d mode %d (%s): hit stream %.0f cycles per 1 KB unit (worst single wait %.0f); side wave total %.0f cycles\n", pace, mode,
               mode == 0 ? "side idle" : mode == 1 ? "side: 36 x 512 B HBM-miss bursts" : mode == 2 ? "side: 36 x 512 B L2-hit bursts" : "side: 10 x 1 KB HBM-miss bursts",
               t / 256 / (iters * 12.0), w / 256, s / 256);
    }
    return 0;
}
