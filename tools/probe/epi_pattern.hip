// r06 probe: what does the CU's vector-memory path charge for the EPILOGUE access shapes of the fused Winograd kernels (16-byte per lane stores / loads of an
// NHWC tile), all 8 waves of a 512-thread block bursting at once as they do at the end of a work item?
//   pattern 0: 1 KB contiguous per wave instruction (the ideal)
//   pattern 1: 4 pixels x 256 contiguous bytes (64 output channels of one pixel per 16 lanes)          <- an LDS-transposed epilogue
//   pattern 2: 16 pixels x 64 bytes (16 output channels per 4 lanes; the four waves of a group fill the lines) <- the lane-local epilogue of wino4_fused64p / 64t
//   pattern 3: as 2 with two 8-byte instructions per lane
// kind: 0 buffer_store_dwordx4 (nt), 1 buffer_store_dwordx4, 2 buffer_load_dwordx4 into registers (nt), 3 buffer_load_dwordx4 ... lds (nt)
// Each wave issues N = 16 instructions per "item" (its share of a 32-tile x 64-channel output tile, pixel stride = C * 4 bytes), `items` times, over a tensor of
// B x 256 x 256 x C floats; prints shader cycles per item: to ISSUE the 16 instructions (until the last one has left the wave) and until they have completed.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/epi_pattern.hip -o tools/probe/epi_pattern ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

template <int KIND, int PATTERN>
__global__ __launch_bounds__(512) void probe(float* buf, unsigned bytes, int C, int W, int items, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, bytes, 0x00020000);
    const int cb = wave & 3, h = wave >> 2, l15 = lane & 15, g = lane >> 4;
    unsigned long long t_issue = 0, t_done = 0;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    const floatx4 val = {1.f + lane, 2.f, 3.f, 4.f};
    const int TW = W / 4, groups_x = TW / 8;
    for (int it = 0; it < items; ++it) {
        // work item: 4 x 8 tiles; (gy, gx, image) from a virtual id, as the kernels walk them
        const int v = blockIdx.x + it * gridDim.x;
        const int gx = v % groups_x, gy = (v / groups_x) % (W / 16), b = v / (groups_x * (W / 16));
        unsigned off;
        if (PATTERN == 0) {
            off = (unsigned)(((v * 8 + wave) * 16) * 1024 + lane * 16);   // + k * 1024 per instruction
        } else if (PATTERN == 1) {
            // wave (cb, h): tile group h, tile column cb, rows by lane: lane = (tile row l15 >> 2 ... ) 4 pixels x 16 lanes x 16 B = 4 x 256 B
            const int ty = gy * 4 + (l15 >> 2), tx = gx * 8 + h * 4 + cb;
            const unsigned pix = (unsigned)((b * W + 4 * ty) * W + 4 * tx);
            off = (pix * (unsigned)C + (unsigned)(((l15 & 3) * 4 + g) * 4)) * 4u;
        } else {
            const int ty = gy * 4 + (l15 >> 2), tx = gx * 8 + h * 4 + (l15 & 3);
            const unsigned pix = (unsigned)((b * W + 4 * ty) * W + 4 * tx);
            off = (pix * (unsigned)C + (unsigned)(cb * 16 + 4 * g)) * 4u;
        }
        const int rowb = W * C * 4, pixb = C * 4;
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int so = PATTERN == 0 ? k * 1024 : (k >> 2) * rowb + (k & 3) * pixb;
            if (KIND <= 1) {
                if (PATTERN == 3) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, floatx2{val.x, val.y}), rs, (int)off, so, KIND == 0 ? 2 : 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, floatx2{val.z, val.w}), rs, (int)off + 8, so, KIND == 0 ? 2 : 0);
                } else
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, val), rs, (int)off, so, KIND == 0 ? 2 : 0);
            } else if (KIND == 2) {
                acc += __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, so, 2));
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(lds) + wave * 16384 + k * 1024), 16, (int)off, so, 0, 2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (KIND == 2) asm volatile("" : "+v"(acc));
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t_issue += t1 - t0;
        t_done += t2 - t0;
    }
    if (lane == 0) {
        out[(blockIdx.x * 8 + wave) * 2] = t_issue;
        out[(blockIdx.x * 8 + wave) * 2 + 1] = t_done;
    }
    if (acc.x == 12345.f) sink[0] = acc.x + acc.y + acc.z + acc.w + lds[lane];
}

template <int KIND, int PATTERN>
void run(float* buf, size_t bytes, int C, int W, int B, unsigned long long* dout, float* sink, const char* kname, const char* pname) {
    const int total = B * (W / 16) * (W / 32);
    const int items = total / 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<KIND, PATTERN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<KIND, PATTERN>), dim3(256), dim3(512), 8 * 16384, 0, buf, (unsigned)bytes, C, W, items, dout, sink);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(256 * 8 * 2);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double si = 0, sd = 0;
    for (int i = 0; i < 256 * 8; ++i) { si += (double)h[2 * i]; sd += (double)h[2 * i + 1]; }
    printf("%-28s %-34s C=%3d: %7.0f cycles to issue, %7.0f until complete, per item (16 instructions per wave, 8 waves = 128 KB per CU)\n", kname, pname, C,
           si / (256 * 8) / items, sd / (256 * 8) / items);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, W = 256;
    for (int C : {64, 128}) {
        const size_t bytes = (size_t)B * W * W * C * 4;
        float* buf = nullptr;
        unsigned long long* dout = nullptr;
        float* sink = nullptr;
        hipMalloc(&buf, bytes);
        hipMemset(buf, 0, bytes);
        hipMalloc(&dout, 256 * 8 * 2 * 8);
        hipMalloc(&sink, 64);
        const char* pn[4] = {"1 KB contiguous", "4 pixels x 256 B", "16 pixels x 64 B", "16 pixels x 64 B, 2 x 8-byte"};
#define RUN(K, KN) run<K, 0>(buf, bytes, C, W, B, dout, sink, KN, pn[0]); run<K, 1>(buf, bytes, C, W, B, dout, sink, KN, pn[1]); run<K, 2>(buf, bytes, C, W, B, dout, sink, KN, pn[2]);
        RUN(0, "buffer_store_dwordx4 nt")
        run<0, 3>(buf, bytes, C, W, B, dout, sink, "buffer_store_dwordx2 nt x 2", pn[3]);
        RUN(1, "buffer_store_dwordx4")
        RUN(2, "buffer_load_dwordx4 nt")
        RUN(3, "buffer_load_dwordx4 lds nt")
#undef RUN
        hipFree(buf); hipFree(dout); hipFree(sink);
    }
    return 0;
}
