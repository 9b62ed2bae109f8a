// Probe (r06): does a hipMemsetAsync captured into a hipGraph replay the value it was called with?  The first form of the split NAFBlock chain zeroed its barrier
// counters with hipMemsetAsync(ctr, 0, (B + 1) * 4, stream) in the launch function; under the step graph's replay the counters and the error word came up as
// 0x01010101.  Reproduce in isolation: capture { memset(p, 0, n) ; kernel: out[i] = p[i]; p[i] = 0xdeadbeef } and replay it.
//   hipcc --offload-arch=gfx950 -O2 -o graph_memset_probe tools/probe/graph_memset_probe.hip && ./graph_memset_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void snap(unsigned* p, unsigned* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[i] = p[i]; p[i] = 0xdeadbeefu; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int run(size_t bytes, int value) {
    const int n = (int)(bytes / 4);
    unsigned *p, *out;
    CK(hipMalloc(&p, bytes + 64)); CK(hipMalloc(&out, bytes + 64));
    CK(hipMemset(p, 0x55, bytes + 64));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(p, value, bytes, s));
    hipLaunchKernelGGL(snap, dim3((n + 63) / 64), dim3(64), 0, s, p, out, n);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    std::vector<unsigned> h(n);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), out, bytes, hipMemcpyDeviceToHost));
        int bad = 0; unsigned first = 0;
        const unsigned want = (unsigned)(value & 0xff) * 0x01010101u;
        for (int i = 0; i < n; ++i) if (h[i] != want) { if (!bad) first = h[i]; ++bad; }
        printf("memset(%d, %zu bytes) in a graph, replay %d: %d of %d words differ from 0x%08x%s", value, bytes, rep, bad, n, want, bad ? "" : "\n");
        if (bad) printf(" (first 0x%08x)\n", first);
    }
    // eager for comparison
    CK(hipMemsetAsync(p, value, bytes, s));
    hipLaunchKernelGGL(snap, dim3((n + 63) / 64), dim3(64), 0, s, p, out, n);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), out, bytes, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += h[i] != (unsigned)(value & 0xff) * 0x01010101u;
    printf("  eager: %d of %d words differ\n", bad, n);
    return 0;
}
// the engine's shape: TWO memset + kernel pairs on different buffers in one captured graph (the encoder-level and decoder-level chain launches of a step),
// a first kernel in front, 100 replays back to back without a host sync in between
int run2() {
    unsigned *p1, *p2, *o1, *o2, *acc;
    const int n1 = 33, n2 = 9;
    CK(hipMalloc(&p1, 4096)); CK(hipMalloc(&p2, 4096)); CK(hipMalloc(&o1, 4096)); CK(hipMalloc(&o2, 4096)); CK(hipMalloc(&acc, 4096));
    CK(hipMemset(p1, 0x55, 4096)); CK(hipMemset(p2, 0x55, 4096)); CK(hipMemset(acc, 0, 4096));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(snap, dim3(1), dim3(64), 0, s, acc, acc + 512, 8);
    CK(hipMemsetAsync(p1, 0, n1 * 4, s));
    hipLaunchKernelGGL(snap, dim3(1), dim3(64), 0, s, p1, o1, n1);
    hipLaunchKernelGGL(snap, dim3(1), dim3(64), 0, s, acc, acc + 512, 8);
    CK(hipMemsetAsync(p2, 0, n2 * 4, s));
    hipLaunchKernelGGL(snap, dim3(1), dim3(64), 0, s, p2, o2, n2);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    int badtot = 0;
    for (int rep = 0; rep < 100; ++rep) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    unsigned h1[33], h2[9];
    CK(hipMemcpy(h1, o1, sizeof h1, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2, o2, sizeof h2, hipMemcpyDeviceToHost));
    for (unsigned v : h1) badtot += v != 0u;
    for (unsigned v : h2) badtot += v != 0u;
    printf("two memset nodes in one graph, 100 replays: %d words differ from 0 after the last replay (first words 0x%08x 0x%08x)\n", badtot, h1[0], h2[0]);
    return 0;
}
int main() {
    run(132, 0); run(36, 0); run(4096, 0); run(132, 7); run(1 << 20, 0);
    run2();
    return 0;
}
