// Helpers shared by the fused Winograd F(4x4,3x3) kernels (wino_fused.hip: the 32- and 64-cout kernels; wino_fused_t.hip: the two-tile-group kernel):
// vector types, the B^T input transform, the persistent kernels' work-item map.  Internal linkage: every translation unit gets its own copy.
#pragma once
#include "common.h"

namespace irsde {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

namespace {

constexpr unsigned WF_OOB = 0x80000000u;   // voffset of a tap that must read 0 (tensors are < 2 GiB, checked at launch)

__device__ __forceinline__ float silu_w(float v) { return silu_hw(v); }

// B^T of F(4x4,3x3) along one axis (same matrix as wino.hip), 14 packed operations with explicit FMAs: the producer's vector
// instructions compete with the f32 MFMAs of the wave that shares its SIMD (the f32 matrix rate IS the vector FMA rate), so
// every operation saved here is matrix-pipe time (profiles/r02_wino_fused_notes.md).
__device__ __forceinline__ floatx2 fma2(const float a, const floatx2 b, const floatx2 c) {
    return __builtin_elementwise_fma(floatx2{a, a}, b, c);
}
__device__ __forceinline__ void bt6(const floatx2* d, floatx2* t) {
    t[0] = fma2(-5.0f, d[2], fma2(4.0f, d[0], d[4]));          // 4 d0 - 5 d2 + d4
    t[1] = fma2(-4.0f, d[1] + d[2], d[3] + d[4]);              // (d3 + d4) - 4 (d1 + d2)
    t[2] = fma2(4.0f, d[1] - d[2], d[4] - d[3]);               // 4 (d1 - d2) + (d4 - d3)
    const floatx2 a = d[3] - d[1], b = d[4] - d[2];
    t[3] = fma2(2.0f, a, b);                                   // 2 (d3 - d1) + (d4 - d2)
    t[4] = fma2(-2.0f, a, b);                                  // 2 (d1 - d3) + (d4 - d2)
    t[5] = fma2(-5.0f, d[3], fma2(4.0f, d[1], d[5]));          // 4 d1 - 5 d3 + d5
}
// Work item of the persistent kernels = (cout block, tile group): virtual block id v (v % 8 = the XCD of the block that runs it) -> item.
// Default: each XCD walks a contiguous range of (tile group, cout block), cout block fastest (the NB blocks of a tile group share its patches in
// one L2); xcd_nb: cout block = xcd % NB (an XCD reads one slice of the weights only; wino_fused64_xcd_nb()).
struct W6Item { int nblk, gx, gy, b; };

__device__ __forceinline__ W6Item w6_item(const int v, const int total, const int NB, const int GX, const int GY, const int xcd_nb) {
    int nblk, g_;
    if (xcd_nb) {
        const int xcd = v & 7;
        nblk = xcd % NB;
        g_ = (xcd / NB) * (total / 8) + (v >> 3);
    } else {
        const int xcd = v & 7, q = total >> 3, r = total & 7;
        const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        nblk = wgid % NB;
        g_ = wgid / NB;
    }
    W6Item it;
    it.nblk = nblk;
    it.gx = g_ % GX; g_ /= GX;
    it.gy = g_ % GY;
    it.b = g_ / GY;
    return it;
}

}  // namespace
}  // namespace irsde
