// Probe: which HW_ID wave slots do two co-resident 256-thread blocks (73 KB LDS each) get on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
    extern __shared__ float smem[];
    smem[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    long long t0 = clock64();
    float acc = smem[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
    if ((threadIdx.x & 63) == 0) {
        unsigned* o = out + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        o[0] = hwid; o[1] = xcc; o[2] = (unsigned)(t0 & 0xffffffff); o[3] = (unsigned)acc;
    }
}
int main() {
    const int nb = 1024;
    unsigned* d; hipMalloc(&d, nb * 16 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 73728, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 16);
    hipMemcpy(h.data(), d, nb * 16 * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> slots, simd;
    std::map<unsigned long long, std::vector<int>> percu;
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 4; ++w) {
            unsigned hw = h[(b * 4 + w) * 4], xcc = h[(b * 4 + w) * 4 + 1];
            slots[hw & 0xF]++; simd[(hw >> 4) & 3]++;
            unsigned long long key = ((unsigned long long)xcc << 32) | ((hw >> 8) & 0xFFFF0 ? 0 : 0) | ((hw >> 8) & 0xFF) ;
            key = ((unsigned long long)xcc << 32) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 7) | ((hw >> 8) & 0xF);
            if (w == 0) percu[key].push_back(b);
        }
    printf("wave slot histogram:"); for (auto& kv : slots) printf(" [%u]=%d", kv.first, kv.second); printf("\n");
    printf("simd histogram:"); for (auto& kv : simd) printf(" [%u]=%d", kv.first, kv.second); printf("\n");
    printf("distinct (xcc,se,sh,cu): %zu\n", percu.size());
    int shown = 0;
    for (auto& kv : percu) { if (shown++ > 6) break; printf("cu key %llx blocks:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    for (int b = 0; b < 4; ++b) for (int w = 0; w < 4; ++w) printf("block %d wave %d hwid %08x xcc %u t0 %u\n", b, w, h[(b*4+w)*4], h[(b*4+w)*4+1], h[(b*4+w)*4+2]);
    return 0;
}
