// Fused Winograd F(4x4,3x3) convolution for the big feature maps: input transform, the 36 component GEMMs and the output
// transform + epilogue in ONE kernel, so the transformed tensors V (2.25x the input) and M (2.25x the output) never reach
// HBM.  Replaces, for the layers where the transforms cost as much as the GEMM they serve (64..256 channels on the
// 256^2 / 128^2 levels: Block.proj, module_util.py:108-122, and the fused-upsample default_conv, DenoisingUNet_arch.py:67),
// the three-launch path of wino.hip (wino_input -> gemm_zloop -> wino_output: 24 % of the r01 step, 69 GB per evaluation).
//
// Block = 32 tiles (8 wide x 4 tall = 32 x 16 output pixels of one image) x 32 output channels x all 36 components;
// K loop over the input channels in chunks of 16.  512 threads, specialised by wave (1 MFMA wave + 1 producer wave per SIMD):
//   waves 0-3  MFMA: wave z-group zg owns components 9zg .. 9zg+8: 9 accumulator tiles M_z[32 tiles][32 couts] of
//              v_mfma_f32_32x32x2_f32 (144 registers).  A = V_z[tile][k] from LDS (one ds_read_b128 = 4 k per lane),
//              B = U_z[cout][k] straight from L2 into registers (the weight slice of a (component, 32 couts) pair is
//              private to one wave, so LDS staging would buy nothing): U is stored pre-swizzled so that a wave's
//              fragment of one K sub-step is 1 KB contiguous (buffer_load_dwordx4, refilled in place right after use).
//   waves 4-7  producers: lane = (tile, channel pair of the 16-channel chunk): 36 buffer_load_dwordx2 of the 6x6 input patch
//              (out-of-image taps, tiles past the edge: out-of-range offset -> hardware returns 0;
//              concat sources and the fused nearest x2 upsample are resolved in the row / column offsets), B^T d B in
//              registers, ds_write_b64 into the V buffer of the NEXT chunk (double buffer, one barrier per chunk).
// The MFMA wave of each SIMD has no vector instruction in its K loop; the producer wave of the same SIMD overlaps its loads,
// LDS writes and waits with the MFMAs, but NOT its transform arithmetic: on gfx950 an f32 MFMA and the vector instructions of
// the other wave on the SIMD serialise (tools/probe/mfma_valu_overlap.hip), which is what holds the kernel at ~0.5 MFMA-busy.
// Epilogue (all 8 waves): accumulators -> LDS (one pass), thread = (tile, 4 couts, row pair): A^T M A,
// bias -> FiLM -> SiLU -> +residual, 16-byte stores.
//
// Arithmetic is exact fp32 (f32 MFMA = fmaf chain); the result differs from wino.hip's only in summation order.
#include "common.h"
#include "wino_fused_shared.h"
#include <mutex>

namespace irsde {

namespace {

constexpr int WF_NT = 512;
constexpr int WF_KC = 16;                  // input channels per chunk (two K sub-steps of 8)
constexpr int WF_HS = 136;                 // LDS floats between the two k-halves (32 tiles x 4 k + 8 pad)
constexpr int WF_SS = 2 * WF_HS;           // ... between the two sub-steps of a chunk (272 = 16 mod 32: conflict-free b64 writes)
constexpr int WF_ZS = 2 * WF_SS;           // ... between components (544)
constexpr int WF_VBUF = 36 * WF_ZS;        // floats per V buffer (78 336 B)
constexpr int WF_LDS_BYTES = 2 * WF_VBUF * 4;
// r04: the same transform in 12 operations (t1 / t2 share p = d4 - 4 d2, q = d3 - 4 d1); same number of roundings per output
__device__ __forceinline__ void bt6_12(const floatx2* d, floatx2* t) {
    t[0] = fma2(-5.0f, d[2], fma2(4.0f, d[0], d[4]));
    const floatx2 p = fma2(-4.0f, d[2], d[4]), q = fma2(-4.0f, d[1], d[3]);
    t[1] = p + q;                                              // (d3 + d4) - 4 (d1 + d2)
    t[2] = p - q;                                              // 4 (d1 - d2) + (d4 - d3)
    const floatx2 a = d[3] - d[1], b = d[4] - d[2];
    t[3] = fma2(2.0f, a, b);
    t[4] = fma2(-2.0f, a, b);
    t[5] = fma2(-5.0f, d[3], fma2(4.0f, d[1], d[5]));
}
// A^T of F(4x4,3x3) along one axis
__device__ __forceinline__ void at6(const float* m, float* y) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.0f * d34;
    y[2] = s12 + 4.0f * s34;
    y[3] = d12 + 8.0f * d34 + m[5];
}

// Output transform + epilogue (every wave runs it): thread = (tile, 4 consecutive couts, row pair), 512 threads = 32 tiles x
// 8 cout quads x 2 row pairs.  16-byte LDS reads, residual loads and output stores: a wave-level store covers 8 x 128
// contiguous bytes (4-byte-per-lane stores made the epilogue cost more than a K=64 main loop: the store tail is issue-bound).
__device__ __forceinline__ void wf_epilogue(const ConvParams& p, const float* Ms, const int tid, const int b, const int gy,
                                            const int gx, const int TH, const int TW, const int n0) {
    const int quad = tid & 7, half = (tid >> 3) & 1, t = tid >> 4;
    const int tyy = gy * 4 + (t >> 3), txx = gx * 8 + (t & 7);
    const int n = n0 + 4 * quad;
    if (tyy >= TH || txx >= TW) return;
    floatx4 bias = {0.f, 0.f, 0.f, 0.f}, fsc = {1.f, 1.f, 1.f, 1.f}, fsh = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias = *reinterpret_cast<const floatx4*>(p.bias + n);
    if (p.film) {
        const float* f = p.film + (size_t)b * p.film_bstride;
        fsc = *reinterpret_cast<const floatx4*>(f + n) + 1.0f;
        fsh = *reinterpret_cast<const floatx4*>(f + p.Cout + n);
    }
    // this thread's two output rows: 2 half, 2 half + 1
    const size_t pix0 = ((size_t)b * p.Ho + 4 * tyy + 2 * half) * p.Wo + 4 * txx;
    floatx4 rv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) rv[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (p.res) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) rv[i][j] = *reinterpret_cast<const floatx4*>(p.res + (pix0 + (size_t)i * p.Wo + j) * p.res_stride + n);
    }
    // A^T rows (2 half, 2 half + 1): half 0: [1 1 1 1 1 0], [0 1 -1 2 -2 0];  half 1: [0 1 1 4 4 0], [0 1 -1 8 -8 1]
    // (multiplying by the constants 0 / 1 is exact)
    const float c0 = half ? 0.f : 1.f, ka = half ? 4.f : 1.f, kb = half ? 8.f : 2.f, c5 = half ? 1.f : 0.f;
    floatx4 u[2][6];
    const float* mp = Ms + t * 32 + 4 * quad;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        floatx4 m[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) m[r] = *reinterpret_cast<const floatx4*>(mp + (r * 6 + s) * 1024);
        const floatx4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
        u[0][s] = c0 * m[0] + s12 + ka * s34;
        u[1][s] = d12 + kb * d34 + c5 * m[5];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const floatx4 s12 = u[i][1] + u[i][2], d12 = u[i][1] - u[i][2], s34 = u[i][3] + u[i][4], d34 = u[i][3] - u[i][4];
        floatx4 y[4];
        y[0] = u[i][0] + s12 + s34;
        y[1] = d12 + 2.0f * d34;
        y[2] = s12 + 4.0f * s34;
        y[3] = d12 + 8.0f * d34 + u[i][5];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            floatx4 v = (y[j] + bias) * fsc + fsh;
            if (p.silu) {
                v.x = silu_w(v.x); v.y = silu_w(v.y); v.z = silu_w(v.z); v.w = silu_w(v.w);
            }
            *reinterpret_cast<floatx4*>(p.out + (pix0 + (size_t)i * p.Wo + j) * p.out_stride + n) = v + rv[i][j];
        }
    }
}

// The two wave roles run separate code paths (their register sets do not add up: accumulators on one side, the input
// patch / offset table / transform temporaries on the other); both execute the same sequence of s_barrier instructions:
// nch + 1 in the K loop, 1 in the epilogue.
template <bool STAMPS>
__global__ __launch_bounds__(WF_NT, 2) void wino4_fused_kernel(const ConvParams p, const float* __restrict__ Uf, const int GX,
                                                                const int GY, const int NB, const unsigned in0_bytes,
                                                                const unsigned in1_bytes, const unsigned uf_bytes,
                                                                unsigned long long* __restrict__ dbg, const int dflags) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform for the role branch
    // tuning aid (irsde_bench_conv variant 82): per-wave shader-clock stamps at the phase boundaries; dbg == nullptr otherwise
#define WF_STAMP(K)                                                                                                  \
    if (STAMPS && dbg) {                                                                                                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                  \
        if (lane == 0) dbg[((size_t)blockIdx.x * 8 + wave) * 16 + (K)] = t_;                                          \
    }
    WF_STAMP(0)
    if (STAMPS && dbg && lane == 0) dbg[((size_t)blockIdx.x * 8 + wave) * 16 + 7] = __builtin_amdgcn_s_memrealtime();
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // XCD-aware bijective block remap: an XCD walks a contiguous range of (tile group, cout group) pairs, cout groups
    // fastest, so the NB blocks that read the same input patch share one L2
    int wgid;
    {
        const int orig = blockIdx.x, nwg = gridDim.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int nblk = wgid % NB;
    int g = wgid / NB;
    const int gx = g % GX; g /= GX;
    const int gy = g % GY;
    const int b = g / GY;
    const int TH = p.Ho >> 2, TW = p.Wo >> 2;
    const int Ctot = p.C0 + p.C1;
    const int nch = Ctot / WF_KC;
    const int nsub = Ctot / 8;

    float* Ms = smem;  // [36][32 tiles][32 couts] (147 456 B), aliases the V buffers after the K loop
    const int n0 = nblk * 32;

    if (wave < 4) {
        // =============================== MFMA waves ===============================
        const int zg = wave;
        if (dflags & 8) __builtin_amdgcn_s_setprio(2);
        const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Uf), 0, uf_bytes, 0x00020000);
        floatx16 acc[9];
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        floatx4 breg[9];  // B fragments: breg[zi] is refilled in place for the next K sub-step right after its 4 MFMAs
        int uvoff[9];
#pragma unroll
        for (int zi = 0; zi < 9; ++zi) {
            uvoff[zi] = ((((zg * 9 + zi) * NB + nblk) * nsub) * 64 + h * 32 + l31) * 16;
            if (dflags & 2) uvoff[zi] = (int)WF_OOB;  // tuning aid: weight fragments read as zeros without touching memory
            breg[zi] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, uvoff[zi], 0, 0));
        }
        __syncthreads();  // iteration 0: the producers fill V[0]
        WF_STAMP(1)
        for (int c = 0; c < nch; ++c) {
            const float* vb = smem + (c & 1) * WF_VBUF + zg * 9 * WF_ZS + h * WF_HS + l31 * 4;
            floatx4 a_cur = *reinterpret_cast<const floatx4*>(vb);
            // 18 groups (2 K sub-steps x 9 components) of { A fragment of the next group, 4 MFMAs, refill of this group's
            // B fragment for the next sub-step }.  The scheduling barrier pins that order: left alone, the compiler sinks
            // the refills to a few MFMAs before their use (it runs out of registers for 18 A fragments held up front),
            // which exposes the L2 latency; here every refill has 36 MFMAs (2304 cycles) of cover.
#pragma unroll
            for (int gi = 0; gi < 18; ++gi) {
                const int sub = gi / 9, zi = gi % 9;
                const int ss = 2 * c + sub;
                const int nxt = (ss + 1 < nsub ? ss + 1 : ss) * 1024;  // byte offset of the next sub-step's fragments
                floatx4 a_next = a_cur;
                if (gi + 1 < 18) a_next = *reinterpret_cast<const floatx4*>(vb + ((gi + 1) % 9) * WF_ZS + ((gi + 1) / 9) * WF_SS);
                const floatx4 bb = breg[zi];
                if (!(dflags & 128)) {  // (tuning aid: no MFMAs)
                acc[zi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, bb.x, acc[zi], 0, 0, 0);
                acc[zi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, bb.y, acc[zi], 0, 0, 0);
                acc[zi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, bb.z, acc[zi], 0, 0, 0);
                acc[zi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, bb.w, acc[zi], 0, 0, 0);
                }
                breg[zi] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, uvoff[zi], nxt, 0));
                a_cur = a_next;
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
        WF_STAMP(2)
        // epilogue: accumulators -> LDS (accumulator register r of lane (l31, h) holds tile (r&3) + 8 (r>>2) + 4h, cout l31)
#pragma unroll
        for (int zi = 0; zi < 9; ++zi)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ms[((zg * 9 + zi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[zi][r];
        __syncthreads();
        WF_STAMP(3)
        wf_epilogue(p, Ms, tid, b, gy, gx, TH, TW, n0);
        WF_STAMP(4)
    } else {
        // =============================== producer waves ===============================
        // lane = (tile column, channel pair of the 16-channel chunk).  Per chunk: column pass of the 6x6 patch (consumes the
        // patch registers), the 36 loads of the NEXT chunk into the same registers, row pass + 36 ds_write_b64.
        // (Measured alternatives, profiles/r02_wino_fused_notes.md: two chunks of loads in flight, full-line / quad-channel
        // dwordx4 loads, 16 loads for the fused-upsample layers, scalar per-channel transforms, wave priorities, a second
        // barrier separating the transform from the MFMA phase — none faster.  What bounds the period: this wave's ~200 vector
        // instructions per chunk and the 72 f32 MFMAs of the wave it shares the SIMD with do not overlap
        // (tools/probe/mfma_valu_overlap.hip), so every instruction below is matrix-pipe time.)
        if (dflags & 4) __builtin_amdgcn_s_setprio(2);  // tuning aid
        const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0), 0, in0_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc1 =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in1 ? p.in1 : p.in0), 0, p.in1 ? in1_bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0), 0, 0u, 0x00020000);
        const int cp = lane & 7;                           // channel pair of the chunk: channels 2cp, 2cp+1
        const int trow = wave - 4, tcol = lane >> 3;       // this lane's tile inside the group
        // LDS float offset of (tile, channel pair): [sub = c>>3][hh = (c>>2)&1][tile][kk = c&3]
        const int vw_base = (cp >> 2) * WF_SS + ((cp >> 1) & 1) * WF_HS + (trow * 8 + tcol) * 4 + 2 * (cp & 1);
        const int tyy = gy * 4 + trow, txx = gx * 8 + tcol;
        const bool tile_ok = tyy < TH && txx < TW && !(dflags & 1);  // (dflags & 1, tuning aid: every patch load reads zeros, no memory traffic)
        const int Hv = p.Hin << p.in_shift, Wv = p.Win << p.in_shift;
        int rowpix[6], colpix[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int y = 4 * tyy - 1 + r, x = 4 * txx - 1 + r;
            rowpix[r] = (tile_ok && (unsigned)y < (unsigned)Hv) ? (b * p.Hin + (y >> p.in_shift)) * p.Win : -1;
            colpix[r] = (tile_ok && (unsigned)x < (unsigned)Wv) ? (x >> p.in_shift) : -1;
        }
        unsigned voff[36];
        floatx2 rawA[36], rawB[36];
#define WF_BUILD_VOFF(PIXF)                                                                                                  \
    _Pragma("unroll") for (int r = 0; r < 6; ++r) _Pragma("unroll") for (int s = 0; s < 6; ++s) voff[r * 6 + s] =            \
        (rowpix[r] >= 0 && colpix[s] >= 0) ? (unsigned)(rowpix[r] + colpix[s]) * (unsigned)((PIXF)*4) + (unsigned)(cp * 8) : WF_OOB;
        // ONE unconditional load site per register set (a load under `if (ci < nch)` turns the patch registers into a phi
        // and costs 36 register-pair copies per chunk); past the last chunk the descriptor has zero bytes: zeros, no traffic.
#define WF_LOAD_RAW(RAW, CI)                                                                                                 \
    {                                                                                                                        \
        const int cc_ = (CI)*WF_KC;                                                                                          \
        const bool second_ = cc_ >= p.C0;                                                                                    \
        const int soff_ = (second_ ? cc_ - p.C0 : cc_) * 4;                                                                  \
        const __amdgpu_buffer_rsrc_t rs_ = (CI) >= nch ? rsrc_none : second_ ? rsrc1 : rsrc0;                                \
        _Pragma("unroll") for (int e = 0; e < 36; ++e) RAW[e] =                                                              \
            __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rs_, (int)voff[e], soff_, 0));                 \
    }
        // one chunk: the NEXT chunk's 36 loads into the other register set first (they have the whole transform of this
        // chunk to land), column pass, row pass + 36 ds_write_b64.  The chunk loop is unrolled by two so that the two
        // register sets swap roles without copies (C0 and C1 are multiples of 32: the chunk count is even).
#define WF_CHUNK(CUR, NXT, IT)                                                                                               \
    {                                                                                                                        \
        if (((IT) + 1) * WF_KC == p.C0) { WF_BUILD_VOFF(p.pix1) }                                                            \
        WF_LOAD_RAW(NXT, (IT) + 1)                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        floatx2 w[6][6];                                                                                                     \
        _Pragma("unroll") for (int s = 0; s < 6; ++s) {                                                                      \
            floatx2 col[6], tc[6];                                                                                           \
            _Pragma("unroll") for (int r = 0; r < 6; ++r) col[r] = CUR[r * 6 + s];                                           \
            bt6(col, tc);                                                                                                    \
            _Pragma("unroll") for (int r = 0; r < 6; ++r) w[r][s] = tc[r];                                                   \
        }                                                                                                                    \
        float* vw = smem + ((IT)&1) * WF_VBUF + vw_base;                                                                     \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                                      \
            floatx2 o[6];                                                                                                    \
            bt6(w[r], o);                                                                                                    \
            _Pragma("unroll") for (int s = 0; s < 6; ++s) *reinterpret_cast<floatx2*>(vw + (r * 6 + s) * WF_ZS) = o[s];      \
        }                                                                                                                    \
        __syncthreads();                                                                                                     \
    }
        WF_BUILD_VOFF(p.pix0)
        WF_LOAD_RAW(rawA, 0)
        for (int it = 0; it < nch; it += 2) {
            if (it == 2) { WF_STAMP(1) }
            WF_CHUNK(rawA, rawB, it)
            WF_CHUNK(rawB, rawA, it + 1)
        }
#undef WF_CHUNK
#undef WF_BUILD_VOFF
#undef WF_LOAD_RAW
        WF_STAMP(2)
        __syncthreads();  // the MFMA waves' last chunk
        __syncthreads();  // the accumulators are in LDS
        WF_STAMP(3)
        wf_epilogue(p, Ms, tid, b, gy, gx, TH, TW, n0);
        WF_STAMP(4)
    }
}

#undef WF_STAMP

// ================================================================================================================
// r03: the 64-cout variant.  r02's kernel recomputes B^T d B of a patch once per 32-cout block and runs 4.5 vector
// instructions per MFMA on SIMDs where f32 MFMAs and the other wave's vector instructions serialise
// (tools/probe/mfma_valu_overlap.hip): 0.49 MFMA-busy.  Here a block owns 16 tiles (4 x 4 tiles = 16 x 16 output pixels)
// x 64 output channels x all 36 components, K chunks of 32 input channels, on v_mfma_f32_16x16x4_f32 (same FLOP per cycle as
// 32x32x2, same 144 accumulator registers per MFMA wave): per chunk the producers do the SAME number of loads, transform
// operations and LDS writes as before while the MFMA waves have twice the work (288 MFMA-equivalents of 32 cycles... i.e.
// 9216 matrix-pipe cycles per wave), so the vector work per MFMA halves; the patch is transformed Cout/64 instead of
// Cout/32 times (once for the 64-channel level-0 layers).  Cost: the weight fragments (private to a wave) are 8 KB per
// (component, chunk) instead of 4 KB per two blocks — the same bytes per FLOP from L2, twice the load instructions per MFMA.
//   MFMA waves 0-3: wave zg owns components 9zg .. 9zg+8.  A operand = U (16 couts x 4 k), B operand = V (4 k x 16 tiles):
//       D[cout][tile], lane (tile = l & 15, g = l >> 4) holds couts 4g .. 4g+3 of one 16-cout block -> one ds_write_b128
//       per accumulator into the LDS staging of the output transform.  One ds_read_b128 of V per 16 MFMAs; U fragments
//       come from L2 through a ring of W6_RING (component, 16-cout block) units refilled W6_RING - 1 units (x 128 cycles)
//       ahead.
//   producer waves 4-7: lane = (tile, channel pair of the 32-channel chunk): 16 lanes x 8 B = one full 128-byte line per
//       patch pixel; same transform code as above.
// V layout in LDS (floats): [component z][r = c >> 4][g = (c >> 2) & 3][tile ^ g][j = c & 3], r stride 272, z stride 544: the
// MFMA waves' b128 reads are 1 KB contiguous per (z, r); the XOR spreads the producers' b64 writes of one tile over all banks.
// ================================================================================================================
constexpr int W6_KC = 32;
constexpr int W6_RING = 18, W6_RING_ALT = 12;  // U units in flight per MFMA wave (x 4 registers); ALT: irsde_bench_conv variant 93
constexpr int W6_RS = 272, W6_ZS = 544;      // floats between r halves / components (same footprint as the 32-cout kernel)
constexpr int W6_VBUF = 36 * W6_ZS;
constexpr int W6_MS = 68;                    // floats per (component, tile) row of the output staging: 64 couts + 4 pad
constexpr int W6_LDS_BYTES = 2 * W6_VBUF * 4;  // 156 672 B = 36 * 16 * 68 * 4 (the staging aliases the V buffers)
static_assert(36 * 16 * W6_MS * 4 <= W6_LDS_BYTES, "output staging must fit in the V buffers");

typedef _Float16 wf_f16x8 __attribute__((ext_vector_type(8)));

// (x0, x1) -> the 8 bytes [hi x0, hi x1, lo x0, lo x1] of fp16 pieces of x / 16 (round to nearest even; the residual x - hi is exact in f32)
__device__ __forceinline__ floatx2 wf_split_pair(const floatx2 v) {
    const float a0 = v.x * kWinoFused64PairVScale, a1 = v.y * kWinoFused64PairVScale;
    const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
    const _Float16 l0 = (_Float16)(a0 - (float)h0), l1 = (_Float16)(a1 - (float)h1);
    const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
    floatx2 o;
    o.x = __builtin_bit_cast(float, hi);
    o.y = __builtin_bit_cast(float, lo);
    return o;
}

// Uf (wino_fused64_pack_weights order, f32) -> the PAIR kernel's weights: every group of 4 consecutive floats (one lane's 4
// channels of a fragment) becomes the 8 halves [hi c0, hi c1, lo c0, lo c1, hi c2, hi c3, lo c2, lo c3] of scale * U
__global__ __launch_bounds__(256) void wf64_split_weights_kernel(const float* __restrict__ Uf, uint4* __restrict__ out, const size_t nquads,
                                                                 const float scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nquads) return;
    const float4 u = reinterpret_cast<const float4*>(Uf)[i];
    const float a[4] = {u.x * scale, u.y * scale, u.z * scale, u.w * scale};
    unsigned short h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 hb = (_Float16)a[e];
        const _Float16 lb = (_Float16)(a[e] - (float)hb);
        h[e] = __builtin_bit_cast(unsigned short, hb);
        l[e] = __builtin_bit_cast(unsigned short, lb);
    }
    uint4 o;
    o.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
    o.y = (unsigned)l[0] | ((unsigned)l[1] << 16);
    o.z = (unsigned)h[2] | ((unsigned)h[3] << 16);
    o.w = (unsigned)l[2] | ((unsigned)l[3] << 16);
    out[i] = o;
}

// NT: residual loads and output stores carry the non-temporal hint (streamed once: they should not evict the weight fragments from L2)
template <bool NT>
__device__ __forceinline__ void wf64_epilogue(const ConvParams& p, const float* Ms, const int tid, const int b, const int gy,
                                              const int gx, const int TH, const int TW, const int n0) {
    // thread = (tile, 4 consecutive couts, row pair): 512 = 16 tiles x 16 quads x 2 row pairs
    const int quad = tid & 15, half = (tid >> 4) & 1, t = tid >> 5;
    const int tyy = gy * 4 + (t >> 2), txx = gx * 4 + (t & 3);
    const int n = n0 + 4 * quad;
    if (tyy >= TH || txx >= TW) return;
    floatx4 bias = {0.f, 0.f, 0.f, 0.f}, fsc = {1.f, 1.f, 1.f, 1.f}, fsh = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias = *reinterpret_cast<const floatx4*>(p.bias + n);
    if (p.film) {
        const float* f = p.film + (size_t)b * p.film_bstride;
        fsc = *reinterpret_cast<const floatx4*>(f + n) + 1.0f;
        fsh = *reinterpret_cast<const floatx4*>(f + p.Cout + n);
    }
    const size_t pix0 = ((size_t)b * p.Ho + 4 * tyy + 2 * half) * p.Wo + 4 * txx;
    floatx4 rv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) rv[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (p.res) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const floatx4* rp = reinterpret_cast<const floatx4*>(p.res + (pix0 + (size_t)i * p.Wo + j) * p.res_stride + n);
                rv[i][j] = NT ? __builtin_nontemporal_load(rp) : *rp;
            }
    }
    const float c0 = half ? 0.f : 1.f, ka = half ? 4.f : 1.f, kb = half ? 8.f : 2.f, c5 = half ? 1.f : 0.f;
    floatx4 u[2][6];
    const float* mp = Ms + t * W6_MS + 4 * quad;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        floatx4 m[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) m[r] = *reinterpret_cast<const floatx4*>(mp + (r * 6 + s) * (16 * W6_MS));
        const floatx4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
        u[0][s] = c0 * m[0] + s12 + ka * s34;
        u[1][s] = d12 + kb * d34 + c5 * m[5];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const floatx4 s12 = u[i][1] + u[i][2], d12 = u[i][1] - u[i][2], s34 = u[i][3] + u[i][4], d34 = u[i][3] - u[i][4];
        floatx4 y[4];
        y[0] = u[i][0] + s12 + s34;
        y[1] = d12 + 2.0f * d34;
        y[2] = s12 + 4.0f * s34;
        y[3] = d12 + 8.0f * d34 + u[i][5];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            floatx4 v = (y[j] + bias) * fsc + fsh;
            if (p.silu) {
                v.x = silu_w(v.x); v.y = silu_w(v.y); v.z = silu_w(v.z); v.w = silu_w(v.w);
            }
            floatx4* op = reinterpret_cast<floatx4*>(p.out + (pix0 + (size_t)i * p.Wo + j) * p.out_stride + n);
            if (NT) __builtin_nontemporal_store(v + rv[i][j], op);
            else *op = v + rv[i][j];
        }
    }
}

// NOWT / NOPATCH: measurement twins (irsde_bench_conv variants 91 / 92: weight fragments resp. patch loads read zeros without
// memory traffic) — template parameters, so the production instance <false, false> carries no run-time tuning branch.
//
// PAIR (IRSDE_FLAG_SPLIT_F16X2): the same kernel with the component GEMMs on v_mfma_f32_16x16x32_f16.  Every f32 operand value x
// is the exact sum of two fp16 pieces hi = RNE(x), lo = RNE(x - hi) (V scaled by 1/16, U by a per-layer power of two: exact,
// undone on the accumulators).  The 16 bytes a lane holds of a (16 rows x 16 channels) fragment — 4 floats in the f32 kernel —
// are the 8 halves [hi c0, hi c1, lo c0, lo c1, hi c2, hi c3, lo c2, lo c3] of its 4 channels: same LDS layout, same weight
// fragment order, same producer write (8 bytes per channel pair).  With B1 = the V fragment as stored and B2 = the same
// registers with the (hi, lo) dwords swapped,  A.B1 = sum hi.hi + lo.lo  and  A.B2 = sum hi.lo + lo.hi : two MFMAs of
// ~17 cycles per unit give all FOUR cross products of 16 channels where the f32 kernel issues four MFMAs of 32 cycles.
// NTMODE (irsde_bench_conv 406 / 407): 1 = non-temporal residual loads / output stores, 2 = also the patch loads
template <int RING, bool NOWT, bool NOPATCH, bool PAIR = false, int NTMODE = 0>
__global__ __launch_bounds__(WF_NT, 2) void wino4_fused64_kernel(const ConvParams p, const float* __restrict__ Uf, const int GX,
                                                                  const int GY, const int NB, const unsigned in0_bytes,
                                                                  const unsigned in1_bytes, const unsigned uf_bytes, const int xcd_nb) {
    static_assert(72 % RING == 0, "the ring must divide the 72 (component, r, cout block) units of a chunk");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Block -> (cout block, tile group).  Default: each XCD (block id % 8) walks a contiguous range of (tile group, cout block) with the
    // cout block fastest: the NB blocks of a tile group share its patches in one L2, but every round of 32 blocks touches ALL of U, and
    // U (36 Cout Cin floats: 2.4 .. 38 MB) does not survive in a 4 MB L2 next to the streamed patches — it is re-fetched every round
    // (r03 PMC: 22 of the kernel's 35 GB of fabric reads per evaluation).  xcd_nb (the launcher's choice, layers whose input is small next
    // to U x rounds): the cout block is a function of the XCD (NB in {1, 2, 4, 8}: cout block = xcd % NB, 8 / NB XCDs share the tile
    // groups of one cout block): an XCD then reads one U slice only, the patches are fetched by NB XCDs instead of one.
    int nblk, g_;
    if (xcd_nb) {
        const int orig = blockIdx.x, xcd = orig & 7;
        nblk = xcd % NB;
        g_ = (xcd / NB) * (int)(gridDim.x / 8) + (orig >> 3);   // gridDim.x = G NB, G % (8 / NB) == 0: each XCD owns G NB / 8 tile groups
    } else {
        const int orig = blockIdx.x, nwg = gridDim.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        nblk = wgid % NB;
        g_ = wgid / NB;
    }
    const int gx = g_ % GX; g_ /= GX;
    const int gy = g_ % GY;
    const int b = g_ / GY;
    const int TH = p.Ho >> 2, TW = p.Wo >> 2;
    const int Ctot = p.C0 + p.C1;
    const int nch = Ctot / W6_KC;
    const int nsub = Ctot / 16;   // 16-channel k groups (one U unit row each)
    float* Ms = smem;
    const int n0 = nblk * 64;

    if (wave < 4) {
        // =============================== MFMA waves ===============================
        const int zg = wave;
        const int l15 = lane & 15, g = lane >> 4;
        const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Uf), 0, NOWT ? 0u : uf_bytes, 0x00020000);
        floatx4 acc[9][4];
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[i][cb] = floatx4{0.f, 0.f, 0.f, 0.f};
        // unit u (global over the K loop) = ((s * 9) + zi) * 4 + cb with s = 2 chunk + r: fragment of (component zg*9+zi, 16-cout
        // block cb, k group s) = 1 KB at Uf + ((((zg*9+zi) * NB + nblk) * nsub + s) * 4 + cb) * 1 KB; lane reads 16 B
        const int uv_lane = lane * 16;
        const int zstride = NB * nsub * 4096;                       // bytes between components
        const int ubase = (zg * 9 * NB + nblk) * nsub * 4096;       // component zg*9, this cout block, s = 0
        // unit K (counted from the start of chunk c; K may run past 71 into the following chunks): cout block K & 3, component
        // (K >> 2) % 9 and k group 2c + (K >> 2) / 9 are compile-time functions of K except for c: a handful of scalar
        // instructions per load.  Units past the end of the K loop re-read the last k group (never used).
        auto unit_soff = [&](const int c, const int K) {
            const int cb = K & 3, zi = (K >> 2) % 9;
            int sidx = 2 * c + (K >> 2) / 9;
            sidx = sidx < nsub ? sidx : nsub - 1;
            return ubase + zi * zstride + sidx * 4096 + cb * 1024;
        };
        floatx4 ring[RING];
#pragma unroll
        for (int i = 0; i < RING; ++i)
            ring[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, uv_lane, unit_soff(0, i), 0));
        const int v_lane = g * 64 + ((l15 ^ g) * 4);
        __syncthreads();  // iteration 0: the producers fill V[0]
        for (int c = 0; c < nch; ++c) {
            const float* vb = smem + (c & 1) * W6_VBUF + zg * 9 * W6_ZS + v_lane;
            floatx4 v_cur = *reinterpret_cast<const floatx4*>(vb);
            // 18 groups (r major, component minor) of { V fragment of the next group, 4 units of { 4 MFMAs, refill of the unit's
            // ring slot for RING units ahead } }.  The scheduling barriers pin that order (see the 32-cout kernel).
#pragma unroll
            for (int gi = 0; gi < 18; ++gi) {
                const int r = gi / 9, zi = gi % 9;
                floatx4 v_next = v_cur;
                if (gi + 1 < 18) v_next = *reinterpret_cast<const floatx4*>(vb + ((gi + 1) % 9) * W6_ZS + ((gi + 1) / 9) * W6_RS);
                // k step outer, cout block inner: consecutive MFMAs hit different accumulators (a dependent v_mfma_f32_16x16x4_f32
                // issues after 40 cycles instead of 32, MI355X_MICROARCH.md)
                if constexpr (PAIR) {
                    const floatx4 v_sw = {v_cur[1], v_cur[0], v_cur[3], v_cur[2]};
                    const wf_f16x8 b1 = __builtin_bit_cast(wf_f16x8, v_cur), b2 = __builtin_bit_cast(wf_f16x8, v_sw);
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[zi][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wf_f16x8, ring[(gi * 4 + cb) % RING]), b1, acc[zi][cb], 0, 0, 0);
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[zi][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wf_f16x8, ring[(gi * 4 + cb) % RING]), b2, acc[zi][cb], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            acc[zi][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[(gi * 4 + cb) % RING][j], v_cur[j], acc[zi][cb], 0, 0, 0);
                }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const int ul = gi * 4 + cb;            // unit index inside the chunk (72 % RING == 0: the slot is static)
                    ring[ul % RING] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, uv_lane, unit_soff(c, ul + RING), 0));
                }
                __builtin_amdgcn_sched_barrier(0);
                (void)r;
                v_cur = v_next;
            }
            __syncthreads();
        }
        // accumulators -> LDS: lane (tile = l15, g) holds couts 16 cb + 4 g .. + 3 of component zi: one 16-byte write each
#pragma unroll
        for (int zi = 0; zi < 9; ++zi)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                *reinterpret_cast<floatx4*>(Ms + ((zg * 9 + zi) * 16 + l15) * W6_MS + cb * 16 + 4 * g) = PAIR ? acc[zi][cb] * p.pair_scale : acc[zi][cb];
        __syncthreads();
        wf64_epilogue<(NTMODE >= 1)>(p, Ms, tid, b, gy, gx, TH, TW, n0);
    } else {
        // =============================== producer waves ===============================
        const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0), 0, in0_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc1 =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in1 ? p.in1 : p.in0), 0, p.in1 ? in1_bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0), 0, 0u, 0x00020000);
        const int cp = lane & 15;                          // channel pair of the chunk: channels 2cp, 2cp+1
        const int tile = (wave - 4) * 4 + (lane >> 4);     // tile inside the 4 x 4 group
        const int trow = tile >> 2, tcol = tile & 3;
        const int kg = (cp >> 1) & 3;
        // LDS float offset of (tile, channel pair): [r = cp >> 3][g = (cp >> 1) & 3][tile ^ g][j = 2 (cp & 1)]
        const int vw_base = (cp >> 3) * W6_RS + kg * 64 + ((tile ^ kg) * 4) + 2 * (cp & 1);
        const int tyy = gy * 4 + trow, txx = gx * 4 + tcol;
        const bool tile_ok = tyy < TH && txx < TW && !NOPATCH;
        const int Hv = p.Hin << p.in_shift, Wv = p.Win << p.in_shift;
        int rowpix[6], colpix[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int y = 4 * tyy - 1 + r, x = 4 * txx - 1 + r;
            rowpix[r] = (tile_ok && (unsigned)y < (unsigned)Hv) ? (b * p.Hin + (y >> p.in_shift)) * p.Win : -1;
            colpix[r] = (tile_ok && (unsigned)x < (unsigned)Wv) ? (x >> p.in_shift) : -1;
        }
        unsigned voff[36];
        floatx2 rawA[36], rawB[36];
#define W6_BUILD_VOFF(PIXF)                                                                                                  \
    _Pragma("unroll") for (int r = 0; r < 6; ++r) _Pragma("unroll") for (int s = 0; s < 6; ++s) voff[r * 6 + s] =            \
        (rowpix[r] >= 0 && colpix[s] >= 0) ? (unsigned)(rowpix[r] + colpix[s]) * (unsigned)((PIXF)*4) + (unsigned)(cp * 8) : WF_OOB;
#define W6_LOAD_RAW(RAW, CI)                                                                                                 \
    {                                                                                                                        \
        const int cc_ = (CI)*W6_KC;                                                                                          \
        const bool second_ = cc_ >= p.C0;                                                                                    \
        const int soff_ = (second_ ? cc_ - p.C0 : cc_) * 4;                                                                  \
        const __amdgpu_buffer_rsrc_t rs_ = (CI) >= nch ? rsrc_none : second_ ? rsrc1 : rsrc0;                                \
        _Pragma("unroll") for (int e = 0; e < 36; ++e) RAW[e] =                                                              \
            __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rs_, (int)voff[e], soff_, NTMODE >= 2 ? 2 : 0)); \
    }
#define W6_CHUNK(CUR, NXT, IT)                                                                                               \
    {                                                                                                                        \
        if (((IT) + 1) * W6_KC == p.C0) { W6_BUILD_VOFF(p.pix1) }                                                            \
        W6_LOAD_RAW(NXT, (IT) + 1)                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        floatx2 w[6][6];                                                                                                     \
        _Pragma("unroll") for (int s = 0; s < 6; ++s) {                                                                      \
            floatx2 col[6], tc[6];                                                                                           \
            _Pragma("unroll") for (int r = 0; r < 6; ++r) col[r] = CUR[r * 6 + s];                                           \
            bt6(col, tc);                                                                                                    \
            _Pragma("unroll") for (int r = 0; r < 6; ++r) w[r][s] = tc[r];                                                   \
        }                                                                                                                    \
        float* vw = smem + ((IT)&1) * W6_VBUF + vw_base;                                                                     \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                                      \
            floatx2 o[6];                                                                                                    \
            bt6(w[r], o);                                                                                                    \
            _Pragma("unroll") for (int s = 0; s < 6; ++s)                                                                    \
                *reinterpret_cast<floatx2*>(vw + (r * 6 + s) * W6_ZS) = PAIR ? wf_split_pair(o[s]) : o[s];                  \
        }                                                                                                                    \
        __syncthreads();                                                                                                     \
    }
        W6_BUILD_VOFF(p.pix0)
        W6_LOAD_RAW(rawA, 0)
        for (int it = 0; it < nch; it += 2) {   // Ctot is a multiple of 64: the chunk count is even
            W6_CHUNK(rawA, rawB, it)
            W6_CHUNK(rawB, rawA, it + 1)
        }
#undef W6_CHUNK
#undef W6_BUILD_VOFF
#undef W6_LOAD_RAW
        __syncthreads();  // the MFMA waves' last chunk
        __syncthreads();  // the accumulators are in LDS
        wf64_epilogue<(NTMODE >= 1)>(p, Ms, tid, b, gy, gx, TH, TW, n0);
    }
}


// ================================================================================================================
// r04: wino4_fused64p_kernel — the 64-cout kernel as a PERSISTENT block with the output transform in registers.
//
// r03's kernel spent ~11 us per block outside its K loop (first patch fetch + transform before the first MFMA, then
// accumulators -> LDS -> output transform -> stores) with ONE block per CU, so nothing overlapped it: 0.35 of the MFMA roof on
// the 2-chunk 64 -> 64 layers, 0.54 on 4-chunk layers (profiles/r03_wino_fused64_notes.md).  Two changes remove it:
//  * wave = 16-cout block (not 9 components): MFMA wave w owns couts 16 w .. 16 w + 15 of the block's 64 and ALL 36 components
//    (36 accumulators of 4 registers = the same 144).  With A = U (16 couts x 4 k) and B = V (4 k x 16 tiles) a lane (tile =
//    l & 15, g = l >> 4) ends up holding M_z[tile][4 g .. 4 g + 3] for every component z — everything A^T M A needs for its
//    (tile, 4 couts).  The output transform is lane-local: no LDS staging, no barrier, and the V buffers are never aliased.
//    Cost: every wave reads all of V from LDS (4 x the ds_read_b128 traffic: 288 KB per chunk and CU, ~30 B/clk of the 256 the
//    LDS delivers) and a store instruction covers 16 segments of 64 B (the four waves fill the other quarters of the same lines).
//  * one block per CU walks its tile groups (virtual block id = blockIdx.x + k gridDim.x, the same XCD-aware item map as
//    before): the producer waves run straight on into the next tile group — its chunk 0 is transformed while the MFMA waves
//    finish the last chunk, chunk 1 during their epilogue — and the weight-fragment ring prefetches across the boundary.
// V double buffer, one barrier per 32-channel chunk, exactly as in wino4_fused64_kernel; producer code unchanged.
// ================================================================================================================
// lane-local output transform + epilogue of the persistent kernel: acc[z] = M_z[tile][n .. n + 3].
// Branch-free and address-arithmetic-free: residual / output go through buffer descriptors (lane offset in one VGPR, the pixel (i, j) of the
// 4 x 4 tile in the scalar offset; tiles past the edge of a ragged group carry an out-of-range offset: loads return 0, stores are dropped).
// The residual rows 0 / 1 are requested before the first transform stage, rows 2 / 3 behind it (the accumulators are dead by then), so
// their latency hides under the ~400 vector instructions of A^T M A instead of being paid once per output row.
// Between the two stages (the accumulators are dead, 144 registers free) the weight-fragment ring is primed with the next tile group's first units:
// the ring is NOT live across the first stage (accumulators + ring + residual rows would not fit in 256 registers).
template <bool NT, bool PAIR, bool RES, bool SILU, int RING>
__device__ __forceinline__ void wf64p_epilogue(const ConvParams& p, floatx4 (&acc)[36], const unsigned lane_off_out, const unsigned lane_off_res,
                                               const __amdgpu_buffer_rsrc_t rs_out, const __amdgpu_buffer_rsrc_t rs_res, const floatx4 bias,
                                               const floatx4 fsc, const floatx4 fsh, floatx4 (&ring)[RING], const __amdgpu_buffer_rsrc_t rsrc_u,
                                               const int uv_lane, const int nubase, const int zstride) {
    constexpr int AUX = NT ? 2 : 0;
    const int orow = p.Wo * p.out_stride * 4, opix = p.out_stride * 4;
    const int rrow = p.Wo * p.res_stride * 4, rpix = p.res_stride * 4;
    floatx4 rv[4][4];
    if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                rv[i][j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)lane_off_res, i * rrow + j * rpix, AUX));
        __builtin_amdgcn_sched_barrier(0);
    }
    // rows: u[i][s] = sum_r A^T[i][r] M[r][s]
    floatx4 u[4][6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const floatx4 m0 = acc[s], m1 = acc[6 + s], m2 = acc[12 + s], m3 = acc[18 + s], m4 = acc[24 + s], m5 = acc[30 + s];
        const floatx4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        u[0][s] = m0 + s12 + s34;
        u[1][s] = d12 + 2.0f * d34;
        u[2][s] = s12 + 4.0f * s34;
        u[3][s] = d12 + 8.0f * d34 + m5;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RING; ++i)   // unit i of the next tile group's first chunk (r = 0, component i)
        ring[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, uv_lane, nubase + i * zstride, 0));
    if constexpr (RES) {
#pragma unroll
        for (int i = 2; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                rv[i][j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)lane_off_res, i * rrow + j * rpix, AUX));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const floatx4 s12 = u[i][1] + u[i][2], d12 = u[i][1] - u[i][2], s34 = u[i][3] + u[i][4], d34 = u[i][3] - u[i][4];
        floatx4 y[4];
        y[0] = u[i][0] + s12 + s34;
        y[1] = d12 + 2.0f * d34;
        y[2] = s12 + 4.0f * s34;
        y[3] = d12 + 8.0f * d34 + u[i][5];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            floatx4 v = ((PAIR ? y[j] * p.pair_scale : y[j]) + bias) * fsc + fsh;   // (pair_scale: the operand scales, powers of two, undone exactly)
            if constexpr (SILU) {
                v.x = silu_w(v.x); v.y = silu_w(v.y); v.z = silu_w(v.z); v.w = silu_w(v.w);
            }
            if constexpr (RES) v = v + rv[i][j];
            // The pixel offset goes into the VECTOR offset (one v_add), not the scalar one: with an SGPR soffset hipcc (ROCm 7.2) pads no wait state
            // between a buffer_store_dwordx4 and the next VALU write of its data registers, and on gfx950 the store's last lanes (12-15 of every
            // row of 16) then read the NEW value of the later dwords — measured: component .y of tile row 3 came out shifted by one pixel.
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), rs_out,
                                                   (int)(lane_off_out + (unsigned)(i * orow + j * opix)), 0, AUX);
        }
    }
}

// STAMP (irsde_bench_conv 435): per-wave cycle totals into dbg[(block * 8 + wave) * 8 ..]: MFMA waves { K-loop compute, barrier wait, epilogue, whole
// kernel, items }, producer waves { load issue, wait + transform + LDS writes, barrier wait, whole kernel, chunks }
// EPI: the epilogue this instance is compiled for — bit 0 SiLU, bit 1 residual (0 .. 3: the production instances; four epilogue bodies behind
// run-time branches in ONE kernel cost 250 spilled registers); -1: all four behind run-time branches (the measurement twins only)
// OPT (r04 tuning bits, measurement twins only — irsde_bench_conv 1000 + OPT; production is OPT = 0: none of them paid, profiles/r04_wino_fused64_notes.md):
//   1  the ring refills of a tile group's LAST chunk that would fetch units past it read out of range (zeros, no traffic): the ring is not live across
//      the epilogue (it is primed between the two transform stages), so those 12 KB per wave and tile group were fetched twice, and the epilogue began
//      by draining them (s_waitcnt vmcnt(0) before their registers could be reused)
//   2  bt6_12 instead of bt6 in the producers
//   4  the producers issue the next chunk's 36 patch loads in six groups between the column passes instead of up front (the issue of a load blocks
//      for ~280 cycles on the shared, saturated vector-memory queue: up front that is ~10k cycles during which the wave cannot transform)
//   8  residual tile warmed into L2 during the last chunk (two LDS-DMA dword loads per MFMA lane into a scratch corner of LDS: no registers)
//  16  measurement twin: every patch load reads from the first 512 KB of the input (L2-resident patches)
//  32  bias / FiLM rows of the epilogue requested before the last chunk's barrier
//  64  producer waves at s_setprio 3 (the MFMA waves stay at 0)
template <int RING, bool NOWT, bool NOPATCH, bool PAIR, bool NT, int EPI, bool STAMP = false, int OPT = 0>
__global__ __launch_bounds__(WF_NT, 2) void wino4_fused64p_kernel(const ConvParams p, const float* __restrict__ Uf, const int GX, const int GY,
                                                                   const int NB, const unsigned in0_bytes, const unsigned in1_bytes,
                                                                   const unsigned uf_bytes, const unsigned out_bytes, const unsigned res_bytes,
                                                                   const int xcd_nb, const int total, unsigned long long* __restrict__ dbg) {
    unsigned long long st_a = 0, st_b = 0, st_c = 0, st_n = 0, st_t0 = 0, st_t = 0;
    if constexpr (STAMP) st_t0 = st_t = __builtin_amdgcn_s_memtime();
#define W6P_STAMP(ACC)                                                    \
    if constexpr (STAMP) {                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        ACC += now_ - st_t;                                               \
        st_t = now_;                                                      \
    }
    static_assert(72 % RING == 0 && RING % 4 == 0, "the ring must divide the 72 (component, k group) units of a chunk, in whole groups of 4");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TH = p.Ho >> 2, TW = p.Wo >> 2;
    const int Ctot = p.C0 + p.C1;
    const int nch = Ctot / W6_KC;
    const int nsub = Ctot / 16;   // 16-channel k groups (one U unit each)
    const int nblocks = gridDim.x;

    if (wave < 4) {
        // =============================== MFMA waves: wave = 16-cout block ===============================
        const int l15 = lane & 15, g = lane >> 4;
        const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Uf), 0, NOWT ? 0u : uf_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.out), 0, p.res ? res_bytes : 0u, 0x00020000);
        floatx4 acc[36];
#pragma unroll
        for (int z = 0; z < 36; ++z) acc[z] = floatx4{0.f, 0.f, 0.f, 0.f};
        // unit (component z, k group s) of this wave's 16-cout block: 1 KB at Uf + (((z NB + nblk) nsub + s) 4 + wave) KB; lane reads 16 B
        const int uv_lane = lane * 16;
        const int zstride = NB * nsub * 4096;                       // bytes between components
        int v = blockIdx.x;
        W6Item it = w6_item(v, total, NB, GX, GY, xcd_nb);
        int ubase = it.nblk * nsub * 4096 + wave * 1024;
        // unit K of a chunk (0 .. 71; r = K / 36 major, component K % 36 minor) relative to the chunk's first k group
        auto unit_rel = [&](const int K) { return (K % 36) * zstride + (K / 36) * 4096; };
        floatx4 ring[RING];
#pragma unroll
        for (int i = 0; i < RING; ++i)
            ring[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, uv_lane, ubase + unit_rel(i), 0));
        const int v_lane = g * 64 + ((l15 ^ g) * 4);
        __syncthreads();  // the producers have filled V[0] of the first tile group
        W6P_STAMP(st_b)
        while (v < total) {
            const int nv = v + nblocks;
            const W6Item nit = w6_item(nv < total ? nv : v, total, NB, GX, GY, xcd_nb);
            const int nubase = nit.nblk * nsub * 4096 + wave * 1024;
            floatx4 e_bias = {0.f, 0.f, 0.f, 0.f}, e_fsc = {1.f, 1.f, 1.f, 1.f}, e_fsh = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < nch; ++c) {
                const float* vb = smem + (c & 1) * W6_VBUF + v_lane;
                const int cur_off = ubase + c * 8192;
                // units past this chunk: the next chunk / the next tile group's first (OPT & 1: the latter are re-fetched by the epilogue anyway: read out of range)
                const int nxt_off = ((OPT & 1) || c + 1 < nch) ? cur_off + 8192 : nubase;
                const int uv_nx = ((OPT & 1) && c + 1 == nch) ? (int)WF_OOB : uv_lane;
                floatx4 vq[2][4];   // V fragments of the current / next group (ping-pong by group parity: no register copies)
#pragma unroll
                for (int i = 0; i < 4; ++i) vq[0][i] = *reinterpret_cast<const floatx4*>(vb + i * W6_ZS);
                // 18 groups (r major) of 4 components: { V fragments of the next group, 16 MFMAs (k step outer, component inner: consecutive
                // MFMAs hit different accumulators), refill of the 4 ring slots RING units ahead }.  The scheduling barriers pin that order.
#pragma unroll
                for (int gi = 0; gi < 18; ++gi) {
                    const int zq = gi % 9, cu = gi & 1, nx = cu ^ 1;
                    if (gi + 1 < 18) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            vq[nx][i] = *reinterpret_cast<const floatx4*>(vb + (4 * ((gi + 1) % 9) + i) * W6_ZS + ((gi + 1) / 9) * W6_RS);
                    }
                    if constexpr (PAIR) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[4 * zq + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wf_f16x8, ring[(gi * 4 + i) % RING]),
                                                                                     __builtin_bit_cast(wf_f16x8, vq[cu][i]), acc[4 * zq + i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const floatx4 v_sw = {vq[cu][i][1], vq[cu][i][0], vq[cu][i][3], vq[cu][i][2]};
                            acc[4 * zq + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wf_f16x8, ring[(gi * 4 + i) % RING]),
                                                                                     __builtin_bit_cast(wf_f16x8, v_sw), acc[4 * zq + i], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                acc[4 * zq + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[(gi * 4 + i) % RING][j], vq[cu][i][j], acc[4 * zq + i], 0, 0, 0);
                            // without this the scheduler regroups a group's MFMAs by accumulator (4 dependent MFMAs in a row: 40 instead of 32 cycles each)
                            if (j < 3) __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ul = gi * 4 + i, K = ul + RING;   // 72 % RING == 0: the slot is static
                        const int off = K < 72 ? cur_off + unit_rel(K) : nxt_off + unit_rel(K - 72);
                        ring[ul % RING] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, K < 72 ? uv_lane : uv_nx, off, 0));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                W6P_STAMP(st_a)
                if ((OPT & 32) && c + 1 == nch) {   // the epilogue's bias / FiLM rows: their latency hides under the barrier wait
                    const int n = it.nblk * 64 + wave * 16 + 4 * g;
                    if (p.bias) e_bias = *reinterpret_cast<const floatx4*>(p.bias + n);
                    if (p.film) {
                        const float* f = p.film + (size_t)it.b * p.film_bstride;
                        e_fsc = *reinterpret_cast<const floatx4*>(f + n);
                        e_fsh = *reinterpret_cast<const floatx4*>(f + p.Cout + n);
                    }
                }
                __syncthreads();
                W6P_STAMP(st_b)
            }
            // the producers are already transforming the next tile group; this wave's accumulators hold everything its output needs
            {
                const int n = it.nblk * 64 + wave * 16 + 4 * g;
                const int tyy = it.gy * 4 + (l15 >> 2), txx = it.gx * 4 + (l15 & 3);
                const bool ok = tyy < TH && txx < TW;
                const unsigned pix = (unsigned)((it.b * p.Ho + 4 * tyy) * p.Wo + 4 * txx);
                const unsigned off_out = ok ? (pix * (unsigned)p.out_stride + (unsigned)n) * 4u : WF_OOB;
                const unsigned off_res = ok ? (pix * (unsigned)p.res_stride + (unsigned)n) * 4u : WF_OOB;
                floatx4 bias = e_bias, fsc = e_fsc, fsh = e_fsh;
                if constexpr (OPT & 32) {
                    if (p.film) fsc = fsc + 1.0f;
                } else {
                    if (p.bias) bias = *reinterpret_cast<const floatx4*>(p.bias + n);
                    if (p.film) {
                        const float* f = p.film + (size_t)it.b * p.film_bstride;
                        fsc = *reinterpret_cast<const floatx4*>(f + n) + 1.0f;
                        fsh = *reinterpret_cast<const floatx4*>(f + p.Cout + n);
                    }
                }
#define W6P_EPI(RES_, SILU_) wf64p_epilogue<NT, PAIR, RES_, SILU_, RING>(p, acc, off_out, off_res, rs_out, rs_res, bias, fsc, fsh, ring, rsrc_u, uv_lane, nubase, zstride)
                if constexpr (EPI >= 0) {
                    W6P_EPI((EPI & 2) != 0, (EPI & 1) != 0);
                } else if (p.res) {
                    if (p.silu) W6P_EPI(true, true); else W6P_EPI(true, false);
                } else {
                    if (p.silu) W6P_EPI(false, true); else W6P_EPI(false, false);
                }
#undef W6P_EPI
            }
#pragma unroll
            for (int z = 0; z < 36; ++z) acc[z] = floatx4{0.f, 0.f, 0.f, 0.f};
            v = nv; it = nit; ubase = nubase;
            if constexpr (STAMP) st_n += 1;
            W6P_STAMP(st_c)
        }
    } else {
        // =============================== producer waves (as in wino4_fused64_kernel, running on across tile groups) ===============================
        if constexpr ((OPT & 64) != 0) __builtin_amdgcn_s_setprio(3);   // OPT 64: the producers' vector-memory / vector instructions win the SIMD's issue arbitration
        const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0), 0, in0_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc1 =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in1 ? p.in1 : p.in0), 0, p.in1 ? in1_bytes : 0u, 0x00020000);
        const int cp = lane & 15;                          // channel pair of the chunk: channels 2cp, 2cp+1
        const int tile = (wave - 4) * 4 + (lane >> 4);     // tile inside the 4 x 4 group
        const int trow = tile >> 2, tcol = tile & 3;
        const int kg = (cp >> 1) & 3;
        // LDS float offset of (tile, channel pair): [r = cp >> 3][g = (cp >> 1) & 3][tile ^ g][j = 2 (cp & 1)]
        const int vw_base = (cp >> 3) * W6_RS + kg * 64 + ((tile ^ kg) * 4) + 2 * (cp & 1);
        const int Hv = p.Hin << p.in_shift, Wv = p.Win << p.in_shift;
        int rowpix[6], colpix[6];
        unsigned voff[36];
        floatx2 rawA[36], rawB[36];
        int v = blockIdx.x;
#define W6P_SET_ITEM(VID)                                                                                                    \
    {                                                                                                                        \
        const bool live_ = (VID) < total && !NOPATCH;                                                                        \
        const W6Item pi_ = w6_item(live_ ? (VID) : 0, total, NB, GX, GY, xcd_nb);                                            \
        const int tyy_ = pi_.gy * 4 + trow, txx_ = pi_.gx * 4 + tcol;                                                        \
        const bool tile_ok_ = live_ && tyy_ < TH && txx_ < TW;                                                               \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                                      \
            const int y = 4 * tyy_ - 1 + r, x = 4 * txx_ - 1 + r;                                                            \
            rowpix[r] = (tile_ok_ && (unsigned)y < (unsigned)Hv) ? (pi_.b * p.Hin + (y >> p.in_shift)) * p.Win : -1;         \
            colpix[r] = (tile_ok_ && (unsigned)x < (unsigned)Wv) ? (x >> p.in_shift) : -1;                                   \
        }                                                                                                                    \
    }
#define W6P_BUILD_VOFF(PIXF)                                                                                                 \
    _Pragma("unroll") for (int r = 0; r < 6; ++r) _Pragma("unroll") for (int s = 0; s < 6; ++s) voff[r * 6 + s] =            \
        (rowpix[r] >= 0 && colpix[s] >= 0)                                                                                   \
            ? (((unsigned)(rowpix[r] + colpix[s]) * (unsigned)((PIXF)*4) + (unsigned)(cp * 8)) & ((OPT & 16) ? 0x7ffffu : 0xffffffffu)) : WF_OOB;
// OPT & 8: the residual tile of tile group VID (the one the MFMA waves are finishing) into L2: lane = (pixel row wave - 4, pixel column lane >> 4) of
// tile lane & 15, the two 128-byte lines of its 64 output channels; the data lands in a scratch corner of LDS and is never read
#define W6P_TOUCH(VID)                                                                                                       \
    if ((VID) >= (int)blockIdx.x) {                                                                                          \
        const W6Item ti_ = w6_item((VID), total, NB, GX, GY, xcd_nb);                                                        \
        const int ty_ = ti_.gy * 4 + ((lane & 15) >> 2), tx_ = ti_.gx * 4 + (lane & 3);                                      \
        if (ty_ < TH && tx_ < TW) {                                                                                          \
            const size_t pix_ = ((size_t)ti_.b * p.Ho + 4 * ty_ + (wave - 4)) * p.Wo + 4 * tx_ + (lane >> 4);                \
            const float* rp_ = p.res + pix_ * p.res_stride + ti_.nblk * 64;                                                  \
            __attribute__((address_space(3))) void* sc_ = (__attribute__((address_space(3))) void*)(smem + 2 * W6_VBUF + (wave - 4) * 64); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)rp_, sc_, 4, 0, 0);              \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rp_ + 32), sc_, 4, 0, 0);       \
        }                                                                                                                    \
    }
#define W6P_BT6(D, T) { if constexpr (OPT & 2) bt6_12(D, T); else bt6(D, T); }
// one chunk: loads of the NEXT chunk (the next tile group's chunk 0 behind the last one), transform of the current one into V[IT & 1]
#define W6P_CHUNK(CUR, NXT, IT)                                                                                              \
    {                                                                                                                        \
        if constexpr ((OPT & 8) != 0 && EPI != 0 && EPI != 1) {                                                              \
            if ((IT) == 0 && p.res) { W6P_TOUCH(v - nblocks) }                                                               \
        }                                                                                                                    \
        int ci_ = (IT) + 1;                                                                                                  \
        if ((IT) + 1 == nch) {                                                                                               \
            W6P_SET_ITEM(v + nblocks)                                                                                        \
            W6P_BUILD_VOFF(p.pix0)                                                                                           \
            ci_ = 0;                                                                                                         \
        } else if (((IT) + 1) * W6_KC == p.C0) {                                                                             \
            W6P_BUILD_VOFF(p.pix1)                                                                                           \
        }                                                                                                                    \
        const int cc_ = ci_ * W6_KC;                                                                                         \
        const bool second_ = cc_ >= p.C0;                                                                                    \
        const int soff_ = (second_ ? cc_ - p.C0 : cc_) * 4;                                                                  \
        const __amdgpu_buffer_rsrc_t rs_ = second_ ? rsrc1 : rsrc0;                                                          \
        if constexpr (!(OPT & 4)) {                                                                                          \
            _Pragma("unroll") for (int e = 0; e < 36; ++e) NXT[e] =                                                          \
                __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rs_, (int)voff[e], soff_, 0));              \
        }                                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        W6P_STAMP(st_a)                                                                                                      \
        floatx2 w[6][6];                                                                                                     \
        _Pragma("unroll") for (int s = 0; s < 6; ++s) {                                                                      \
            if constexpr ((OPT & 4) != 0) {                                                                                  \
                _Pragma("unroll") for (int e = 6 * s; e < 6 * s + 6; ++e) NXT[e] =                                           \
                    __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rs_, (int)voff[e], soff_, 0));          \
                __builtin_amdgcn_sched_barrier(0);                                                                           \
            }                                                                                                                \
            floatx2 col[6], tc[6];                                                                                           \
            _Pragma("unroll") for (int r = 0; r < 6; ++r) col[r] = CUR[r * 6 + s];                                           \
            W6P_BT6(col, tc)                                                                                                 \
            _Pragma("unroll") for (int r = 0; r < 6; ++r) w[r][s] = tc[r];                                                   \
            if constexpr ((OPT & 4) != 0) __builtin_amdgcn_sched_barrier(0);                                                 \
        }                                                                                                                    \
        float* vw = smem + ((IT)&1) * W6_VBUF + vw_base;                                                                     \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                                      \
            floatx2 o[6];                                                                                                    \
            W6P_BT6(w[r], o)                                                                                                 \
            _Pragma("unroll") for (int s = 0; s < 6; ++s)                                                                    \
                *reinterpret_cast<floatx2*>(vw + (r * 6 + s) * W6_ZS) = PAIR ? wf_split_pair(o[s]) : o[s];                  \
        }                                                                                                                    \
        if constexpr (STAMP) st_n += 1;                                                                                      \
        W6P_STAMP(st_b)                                                                                                      \
        __syncthreads();                                                                                                     \
        W6P_STAMP(st_c)                                                                                                      \
    }
        W6P_SET_ITEM(v)
        W6P_BUILD_VOFF(p.pix0)
        _Pragma("unroll") for (int e = 0; e < 36; ++e) rawA[e] =
            __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rsrc0, (int)voff[e], 0, 0));
        while (v < total) {
            for (int it = 0; it < nch; it += 2) {   // Ctot is a multiple of 64: the chunk count is even
                W6P_CHUNK(rawA, rawB, it)
                W6P_CHUNK(rawB, rawA, it + 1)
            }
            v += nblocks;
        }
        if constexpr ((OPT & 8) != 0 && EPI != 0 && EPI != 1) {
            if (p.res) { W6P_TOUCH(v - nblocks) }
        }
#undef W6P_CHUNK
#undef W6P_BT6
#undef W6P_TOUCH
#undef W6P_BUILD_VOFF
#undef W6P_SET_ITEM
        __syncthreads();  // the MFMA waves' last chunk
    }
    if constexpr (STAMP) {
        if (lane == 0 && dbg) {
            unsigned long long* d = dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
            d[0] = st_a; d[1] = st_b; d[2] = st_c; d[3] = __builtin_amdgcn_s_memtime() - st_t0; d[4] = st_n;
        }
    }
#undef W6P_STAMP
}


// ================================================================================================================
// r04: wino4_fused64h_kernel — the persistent 64-cout kernel with the input patches staged through LDS ("halo" kernel).
//
// What the cycle stamps of wino4_fused64p_kernel said (profiles/r04_wino_fused64p_stamps_b.txt): its MFMA waves run a 32-channel chunk in
// ~10.1k cycles (floor 9216) and then wait ~5.5k cycles at the barrier for the producers, whose 36 buffer_load_dwordx2 per chunk take 15 - 20k
// cycles to ISSUE (~430 cycles per instruction with four producer waves in the queue).  Patches served from an L2-resident window: no change;
// no weight traffic: no change; patch loads out of range (no data): kernel -25 ... -30 %.  The cost is the number of small gather
// instructions through the CU's vector-memory path, not bytes, L2 or HBM.  So:
//  * the 18 x 18 pixel halo of a 4 x 4 tile group (the 16 overlapping 6 x 6 patches: 324 instead of 576 pixels) is fetched once per 16-channel
//    chunk by LDS-DMA (buffer_load_dwordx4 ... lds: 16 pixels x 64 B per instruction, 21 instead of 144 / 2 instructions, no registers) into a
//    ring of four halo buffers, three chunks ahead of its use;
//  * the producers read their 6 x 6 patches from LDS (36 ds_read_b64), transform, and write V as before;
//  * K chunk = 16 channels (V double buffer 2 x 39 KB + halos 4 x 20 KB = 158 KB); one barrier per chunk ("step"); the four producer waves work as
//    two pairs on alternate chunks (pair = chunk parity = V buffer): in its own step a pair only transforms (at raised wave priority: its ~250 vector
//    instructions otherwise wait for gaps between the MFMAs of the wave that shares the SIMD and the transform takes a whole step,
//    profiles/r04_wino_fused64h_stamps_a.txt), in the other step it waits for the halo of its next chunk and requests the one after.
// Halo layout (bytes): [row 18][slot 18][64 B = 16 channels]; pixel column col < 16 sits in slot (col & 3) * 4 + (col >> 2), columns 16 / 17 in
// slots 16 / 17: the four tiles of a tile row read four slots that are distinct mod 4 (one 256-byte bank window; the one exception, the last
// patch column, is a 2-way conflict), so the ds_read_b64 of 32 lanes (4 tiles x 8 channel pairs) is conflict-free; lanes are ordered so that every
// 16-lane group of the V writes holds two tiles from different tile rows (their tile ^ g columns fall into different halves of the 128-byte write window).
// Pixels outside the image carry an out-of-range buffer offset: the DMA writes zeros.  The fused nearest x2 upsample reads input pixel
// (y >> 1, x >> 1) for halo pixel (y, x).
// MFMA waves, item walk, weight ring and the lane-local output transform are wino4_fused64p_kernel's (OPT bit 1: no double-fetched units).
// ================================================================================================================
constexpr int W7_KC = 16;
constexpr int W7_ZS = 272;                        // floats per component plane of a V buffer: [g 4][tile ^ g 16][j 4] + 16 pad
constexpr int W7_VBUF = 36 * W7_ZS;               // 39 168 B
constexpr int W7_HROW = 18;                       // pixel slots per halo row
constexpr int W7_NDMA = 21;                       // 1 KB LDS-DMA instructions per halo: 324 slots (the last instruction: 4 slots = 16 lanes)
constexpr int W7_HBUF_BYTES = 18 * W7_HROW * 64;  // 20 736 B
constexpr int W7_NH = 4;
constexpr int W7_LDS_BYTES = 2 * W7_VBUF * 4 + W7_NH * W7_HBUF_BYTES;   // 161 280 B

// STAMP: per-wave cycle totals (irsde_bench_conv 2010 ..): MFMA waves { K loop, barrier wait, epilogue, kernel, items }, producer waves { halo DMA issue,
// transform, barrier wait, wait for the halo in the off step, kernel }
template <int RING, bool NOWT, bool NOPATCH, bool PAIR, bool NT, int EPI, bool STAMP = false>
__global__ __launch_bounds__(WF_NT, 2) void wino4_fused64h_kernel(const ConvParams p, const float* __restrict__ Uf, const int GX, const int GY,
                                                                   const int NB, const unsigned in0_bytes, const unsigned in1_bytes,
                                                                   const unsigned uf_bytes, const unsigned out_bytes, const unsigned res_bytes,
                                                                   const int xcd_nb, const int total, unsigned long long* __restrict__ dbg) {
    unsigned long long st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_n = 0, st_t0 = 0, st_t = 0;
    if constexpr (STAMP) st_t0 = st_t = __builtin_amdgcn_s_memtime();
#define W7_STAMP(ACC)                                                     \
    if constexpr (STAMP) {                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        ACC += now_ - st_t;                                               \
        st_t = now_;                                                      \
    }
    static_assert(36 % RING == 0 && RING % 4 == 0, "the ring must divide the 36 units of a chunk, in whole groups of 4");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TH = p.Ho >> 2, TW = p.Wo >> 2;
    const int Ctot = p.C0 + p.C1;
    const int nch = Ctot / W7_KC;      // a multiple of 4
    const int nsub = nch;              // 16-channel k groups = chunks
    const int nblocks = gridDim.x;

    if (wave < 4) {
        // =============================== MFMA waves: wave = 16-cout block ===============================
        const int l15 = lane & 15, g = lane >> 4;
        const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Uf), 0, NOWT ? 0u : uf_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.out), 0, p.res ? res_bytes : 0u, 0x00020000);
        floatx4 acc[36];
#pragma unroll
        for (int z = 0; z < 36; ++z) acc[z] = floatx4{0.f, 0.f, 0.f, 0.f};
        // unit (component z, k group s) of this wave's 16-cout block: 1 KB at Uf + (((z NB + nblk) nsub + s) 4 + wave) KB; lane reads 16 B
        const int uv_lane = lane * 16;
        const int zstride = NB * nsub * 4096;                       // bytes between components
        int v = blockIdx.x;
        W6Item it = w6_item(v, total, NB, GX, GY, xcd_nb);
        int ubase = it.nblk * nsub * 4096 + wave * 1024;
        floatx4 ring[RING];
#pragma unroll
        for (int i = 0; i < RING; ++i)
            ring[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, uv_lane, ubase + i * zstride, 0));
        const int v_lane = g * 64 + ((l15 ^ g) * 4);
        __syncthreads();  // P: the first two halos have landed
        __syncthreads();  // B_0: V[0] of the first tile group is ready
        W7_STAMP(st_b)
        while (v < total) {
            const int nv = v + nblocks;
            const W6Item nit = w6_item(nv < total ? nv : v, total, NB, GX, GY, xcd_nb);
            const int nubase = nit.nblk * nsub * 4096 + wave * 1024;
            for (int c = 0; c < nch; ++c) {
                const float* vb = smem + (c & 1) * W7_VBUF + v_lane;
                const int cur_off = ubase + c * 4096;
                // units past this chunk belong to the next chunk; past the tile group's last chunk nothing is fetched (out-of-range lane offset: zeros, no traffic):
                // the ring is primed for the next tile group between the two stages of the output transform
                const int uv_nx = (c + 1 == nch) ? (int)WF_OOB : uv_lane;
                floatx4 vq[2][4];   // V fragments of the current / next group of 4 components
#pragma unroll
                for (int i = 0; i < 4; ++i) vq[0][i] = *reinterpret_cast<const floatx4*>(vb + i * W7_ZS);
#pragma unroll
                for (int gi = 0; gi < 9; ++gi) {
                    const int cu = gi & 1, nx = cu ^ 1;
                    if (gi + 1 < 9) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) vq[nx][i] = *reinterpret_cast<const floatx4*>(vb + (4 * (gi + 1) + i) * W7_ZS);
                    }
                    if constexpr (PAIR) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[4 * gi + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wf_f16x8, ring[(gi * 4 + i) % RING]),
                                                                                     __builtin_bit_cast(wf_f16x8, vq[cu][i]), acc[4 * gi + i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const floatx4 v_sw = {vq[cu][i][1], vq[cu][i][0], vq[cu][i][3], vq[cu][i][2]};
                            acc[4 * gi + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wf_f16x8, ring[(gi * 4 + i) % RING]),
                                                                                     __builtin_bit_cast(wf_f16x8, v_sw), acc[4 * gi + i], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                acc[4 * gi + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[(gi * 4 + i) % RING][j], vq[cu][i][j], acc[4 * gi + i], 0, 0, 0);
                            if (j < 3) __builtin_amdgcn_sched_barrier(0);   // (keeps consecutive MFMAs on different accumulators)
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ul = gi * 4 + i, K = ul + RING;   // 36 % RING == 0: the slot is static
                        const int off = K < 36 ? cur_off + K * zstride : cur_off + 4096 + (K - 36) * zstride;
                        ring[ul % RING] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, K < 36 ? uv_lane : uv_nx, off, 0));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                W7_STAMP(st_a)
                __syncthreads();
                W7_STAMP(st_b)
            }
            {
                const int n = it.nblk * 64 + wave * 16 + 4 * g;
                const int tyy = it.gy * 4 + (l15 >> 2), txx = it.gx * 4 + (l15 & 3);
                const bool ok = tyy < TH && txx < TW;
                const unsigned pix = (unsigned)((it.b * p.Ho + 4 * tyy) * p.Wo + 4 * txx);
                const unsigned off_out = ok ? (pix * (unsigned)p.out_stride + (unsigned)n) * 4u : WF_OOB;
                const unsigned off_res = ok ? (pix * (unsigned)p.res_stride + (unsigned)n) * 4u : WF_OOB;
                floatx4 bias = {0.f, 0.f, 0.f, 0.f}, fsc = {1.f, 1.f, 1.f, 1.f}, fsh = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) bias = *reinterpret_cast<const floatx4*>(p.bias + n);
                if (p.film) {
                    const float* f = p.film + (size_t)it.b * p.film_bstride;
                    fsc = *reinterpret_cast<const floatx4*>(f + n) + 1.0f;
                    fsh = *reinterpret_cast<const floatx4*>(f + p.Cout + n);
                }
                wf64p_epilogue<NT, PAIR, (EPI & 2) != 0, (EPI & 1) != 0, RING>(p, acc, off_out, off_res, rs_out, rs_res, bias, fsc, fsh, ring, rsrc_u, uv_lane, nubase, zstride);
            }
#pragma unroll
            for (int z = 0; z < 36; ++z) acc[z] = floatx4{0.f, 0.f, 0.f, 0.f};
            v = nv; it = nit; ubase = nubase;
            if constexpr (STAMP) st_n += 1;
            W7_STAMP(st_c)
        }
    } else {
        // =============================== producer waves: two pairs on alternate chunks ===============================
        const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0), 0, NOPATCH ? 0u : in0_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc1 =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in1 ? p.in1 : p.in0), 0, (p.in1 && !NOPATCH) ? in1_bytes : 0u, 0x00020000);
        __builtin_amdgcn_s_setprio(3);   // the producers' few instructions go first: their halo requests queue behind the weight loads of the MFMA waves otherwise
        const int pw = wave - 4, pair = pw >> 1, pp = pw & 1;
        const int q = lane >> 3, cp = lane & 7;
        const int trl = 2 * pp + (q & 1);                       // tile row / column inside the 4 x 4 group (see the lane order above)
        const int tc = q < 4 ? q : ((q - 4) ^ 1);
        const int tile = 4 * trl + tc, kg = cp >> 1;
        // V float offset of (tile, channel pair) inside a component plane: [g = cp >> 1][tile ^ g][j = 2 (cp & 1)]
        const int vw_base = kg * 64 + ((tile ^ kg) * 4) + 2 * (cp & 1);
        // byte offsets of this lane's patch origin inside a halo buffer: patch columns 0 .. 3 (slot 4 s + tc), column 4 and column 5
        const int hrd0 = ((4 * trl) * W7_HROW + tc) * 64 + cp * 8;
        const int hrd4 = ((4 * trl) * W7_HROW + (tc < 3 ? tc + 1 : 16)) * 64 + cp * 8;
        const int hrd5 = ((4 * trl) * W7_HROW + (tc < 3 ? tc + 5 : 17)) * 64 + cp * 8;
        char* const hbase = reinterpret_cast<char*>(smem) + 2 * W7_VBUF * 4;
        const int Hv = p.Hin << p.in_shift, Wv = p.Win << p.in_shift;
        unsigned voff0[11], voff1[11];   // this lane's 16 bytes of DMA instruction d = 2 dd + pp: byte offset inside source 0 / 1
        int vi = blockIdx.x;             // tile group and chunk of this pair's next halo fetch
        int cd = pair;
#define W7_SET_ITEM(VID)                                                                                                     \
    {                                                                                                                        \
        const bool live_ = (VID) < total;                                                                                    \
        const W6Item pi_ = w6_item(live_ ? (VID) : 0, total, NB, GX, GY, xcd_nb);                                            \
        _Pragma("unroll") for (int dd = 0; dd < 11; ++dd) {                                                                  \
            const int sig_ = 16 * (2 * dd + pp) + (lane >> 2);                                                               \
            const int hr_ = sig_ / W7_HROW, sc_ = sig_ - hr_ * W7_HROW;                                                      \
            const int hc_ = sc_ < 16 ? 4 * (sc_ & 3) + (sc_ >> 2) : sc_;                                                     \
            const int y_ = 16 * pi_.gy - 1 + hr_, x_ = 16 * pi_.gx - 1 + hc_;                                                \
            const bool ok_ = live_ && sig_ < 18 * W7_HROW && (unsigned)y_ < (unsigned)Hv && (unsigned)x_ < (unsigned)Wv;     \
            const unsigned pidx_ = (unsigned)((pi_.b * p.Hin + (y_ >> p.in_shift)) * p.Win + (x_ >> p.in_shift));           \
            voff0[dd] = ok_ ? pidx_ * (unsigned)(p.pix0 * 4) + (unsigned)((lane & 3) * 16) : WF_OOB;                         \
            voff1[dd] = ok_ ? pidx_ * (unsigned)(p.pix1 * 4) + (unsigned)((lane & 3) * 16) : WF_OOB;                         \
        }                                                                                                                    \
    }
// this wave's half of the halo of chunk cd of tile group vi into halo buffer HB, then advance (vi, cd) to the pair's next chunk
// (the 21st instruction covers 4 slots: lanes 0 .. 15 only, the others would write past the buffer)
#define W7_DMA_ONE(RS, VOFF, DD)                                                                                             \
    if (2 * (DD) + pp < W7_NDMA - 1 || (2 * (DD) + pp == W7_NDMA - 1 && lane < 16))                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(hb_ + (DD) * 2048), 16, (int)VOFF[DD], soff_, 0, 0);
#define W7_DMA(HB)                                                                                                           \
    {                                                                                                                        \
        const int cc_ = cd * W7_KC;                                                                                          \
        const bool second_ = cc_ >= p.C0;                                                                                    \
        const int soff_ = (second_ ? cc_ - p.C0 : cc_) * 4;                                                                  \
        char* const hb_ = hbase + (HB) * W7_HBUF_BYTES + pp * 1024;                                                          \
        if (second_) {                                                                                                       \
            _Pragma("unroll") for (int dd = 0; dd < 11; ++dd) { W7_DMA_ONE(rsrc1, voff1, dd) }                               \
        } else {                                                                                                             \
            _Pragma("unroll") for (int dd = 0; dd < 11; ++dd) { W7_DMA_ONE(rsrc0, voff0, dd) }                               \
        }                                                                                                                    \
        cd += 2;                                                                                                             \
        if (cd >= nch) {                                                                                                     \
            cd -= nch;                                                                                                       \
            vi += nblocks;                                                                                                   \
            W7_SET_ITEM(vi)                                                                                                  \
        }                                                                                                                    \
    }
        // prologue: the pair's chunks before its first off step (off step G requests chunk G + 3): pair 0 chunks 0 and 2, pair 1 chunk 1
        W7_SET_ITEM(vi)
        W7_DMA(pair)
        if (pair == 0) { W7_DMA(2) }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // P
        W7_STAMP(st_c)
        const int nitems = (total - (int)blockIdx.x + nblocks - 1) / nblocks;
        const int Gtot = nitems * nch;
        for (int G = 0; G < Gtot; ++G) {
            if ((G & 1) == pair) {
                // own step: B^T d B of this lane's (tile, channel pair) from halo G % 4 into V[pair]
                const char* hq = hbase + (G & 3) * W7_HBUF_BYTES;
                // all 36 patch reads first (one LDS latency instead of one per column: the LDS queue is busy with the V reads of four MFMA waves)
                floatx2 raw[36];
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    const char* hp = hq + (s < 4 ? hrd0 + s * 256 : s == 4 ? hrd4 : hrd5);
#pragma unroll
                    for (int r = 0; r < 6; ++r) raw[r * 6 + s] = *reinterpret_cast<const floatx2*>(hp + r * (W7_HROW * 64));
                }
                __builtin_amdgcn_sched_barrier(0);
                floatx2 w[6][6];
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    floatx2 col[6], tcv[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) col[r] = raw[r * 6 + s];
                    bt6(col, tcv);
#pragma unroll
                    for (int r = 0; r < 6; ++r) w[r][s] = tcv[r];
                }
                float* vw = smem + pair * W7_VBUF + vw_base;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    floatx2 o[6];
                    bt6(w[r], o);
#pragma unroll
                    for (int s = 0; s < 6; ++s) *reinterpret_cast<floatx2*>(vw + (r * 6 + s) * W7_ZS) = PAIR ? wf_split_pair(o[s]) : o[s];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                W7_STAMP(st_b)
            } else {
                // off step: the halo of the next own chunk (requested two steps ago, or in the prologue) has landed; request chunk G + 3
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                W7_STAMP(st_d)
                if (G + 3 < Gtot) { W7_DMA((G + 3) & 3) }
                W7_STAMP(st_a)
            }
            // B_G as a raw barrier: __syncthreads() would drain vmcnt(0) here (the LDS-DMA counts as a pending LDS store)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            W7_STAMP(st_c)
        }
#undef W7_DMA
#undef W7_DMA_ONE
#undef W7_SET_ITEM
        __syncthreads();  // the MFMA waves' last chunk
    }
    if constexpr (STAMP) {
        if (lane == 0 && dbg) {
            unsigned long long* d = dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
            d[0] = st_a; d[1] = st_b; d[2] = st_c; d[3] = __builtin_amdgcn_s_memtime() - st_t0; d[4] = st_n; d[5] = st_d;
        }
    }
#undef W7_STAMP
}

// ================================================================================================================
// r04 (late): wino4_fused64s_kernel — the persistent 64-cout kernel with ONE role per wave ("single-stream" kernel).
//
// What the two experiments after the halo kernel said (profiles/r04_wino_fused64_notes.md): an instruction of a producer wave —
// vector, LDS or vector-memory — waits for a gap between the back-to-back f32 MFMAs of the wave that shares its SIMD (~27 cycles per vector
// instruction, ~430 per gather), while the MFMA waves' own weight-fragment loads issue without any such wait (their K loop runs at 10.0-10.4k
// cycles per chunk against a floor of 9.2k).  So the producer's work moves INTO the MFMA waves' instruction stream:
//  * block = 4 waves, one per SIMD (the whole 512-register file of the SIMD: accumulators in the AccVGPR half);
//    wave w = couts 16 w .. 16 w + 15 of the block's 64, all 36 components (as in wino4_fused64p_kernel) AND the input transform of tiles
//    4 w .. 4 w + 3 (lane = (tile, channel pair), as the producer wave w + 4 did);
//  * one register set of 36 patch pairs.  During the K loop of chunk q the wave transforms chunk q + 1 into the other V buffer between its
//    MFMA groups: groups 6-11 the column pass (in place), groups 12-17 the row pass of row r, its six ds_write_b64, and — into the registers
//    that row just freed — the six gathers of row r of chunk q + 2 (>= 7 groups, ~3.5k cycles, before the column pass needs them; packing the
//    passes into groups 12-17 to give them 13 was slower: twelve gathers in a row from all four waves queue up in the CU's one address unit).
//  * item walk, weight ring, V layout, lane-local output transform: wino4_fused64p_kernel's.  Same arithmetic in the same order: bit-identical.
// ================================================================================================================
template <int RING, bool NOPATCH, bool PAIR, bool NT, int EPI, bool STAMP = false, int DBG = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void wino4_fused64s_kernel(const ConvParams p, const float* __restrict__ Uf, const int GX, const int GY, const int NB, const unsigned in0_bytes,
                           const unsigned in1_bytes, const unsigned uf_bytes, const unsigned out_bytes, const unsigned res_bytes, const int xcd_nb,
                           const int total, unsigned long long* __restrict__ dbg) {
    unsigned long long st_a = 0, st_b = 0, st_c = 0, st_n = 0, st_t0 = 0, st_t = 0;
    if constexpr (STAMP) st_t0 = st_t = __builtin_amdgcn_s_memtime();
#define W8_STAMP(ACC)                                                     \
    if constexpr (STAMP) {                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        ACC += now_ - st_t;                                               \
        st_t = now_;                                                      \
    }
    static_assert(72 % RING == 0 && RING % 4 == 0, "the ring must divide the 72 (component, k group) units of a chunk, in whole groups of 4");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TH = p.Ho >> 2, TW = p.Wo >> 2;
    const int Ctot = p.C0 + p.C1;
    const int nch = Ctot / W6_KC;
    const int nsub = Ctot / 16;
    const int nblocks = gridDim.x;
    // ---- matrix role
    const int l15 = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Uf), 0, uf_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.out), 0, p.res ? res_bytes : 0u, 0x00020000);
    floatx4 acc[36];
#pragma unroll
    for (int z = 0; z < 36; ++z) acc[z] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int uv_lane = lane * 16;
    const int zstride = NB * nsub * 4096;
    int v = blockIdx.x;
    W6Item it = w6_item(v, total, NB, GX, GY, xcd_nb);
    int ubase = it.nblk * nsub * 4096 + wave * 1024;
    auto unit_rel = [&](const int K) { return (K % 36) * zstride + (K / 36) * 4096; };
    floatx4 ring[RING];
    const int v_lane = g * 64 + ((l15 ^ g) * 4);
    // ---- transform role: lane = (tile 4 wave + (lane >> 4), channel pair lane & 15)
    const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0), 0, in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in1 ? p.in1 : p.in0), 0, p.in1 ? in1_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_null = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0), 0, 0u, 0x00020000);
    const int cp = lane & 15;
    const int tile = wave * 4 + (lane >> 4);
    const int trow = tile >> 2, tcol = tile & 3;
    const int kg = (cp >> 1) & 3;
    const int vw_base = (cp >> 3) * W6_RS + kg * 64 + ((tile ^ kg) * 4) + 2 * (cp & 1);
    const int Hv = p.Hin << p.in_shift, Wv = p.Win << p.in_shift;
    const int N = (((total - 1 - (int)blockIdx.x) / nblocks) + 1) * nch;   // chunks of this block over all its tile groups (even)
    unsigned voff[36];
#pragma unroll
    for (int e = 0; e < 36; ++e) voff[e] = WF_OOB;
    floatx2 raw[36];
    int gq = 0, gv = blockIdx.x, gc = 0;        // the next chunk to gather: global index, tile group, chunk of the group
    int go_v = -1, go_second = -1;              // what rowoff / coloff were built for
    int g_soff = 0, g_second = 0;
    unsigned g_dead = 0;
// offsets / descriptor choice of chunk gq (the 36 gather offsets are rebuilt when the tile group or the concat source changes)
#define W8_SETUP()                                                                                                           \
    {                                                                                                                        \
        const bool live_ = gq < N;                                                                                           \
        const int cc_ = gc * W6_KC;                                                                                          \
        g_second = cc_ >= p.C0 ? 1 : 0;                                                                                      \
        if (live_ && (gv != go_v || g_second != go_second)) {                                                                \
            const W6Item pi_ = w6_item(gv, total, NB, GX, GY, xcd_nb);                                                       \
            const int tyy_ = pi_.gy * 4 + trow, txx_ = pi_.gx * 4 + tcol;                                                    \
            const bool tile_ok_ = !NOPATCH && tyy_ < TH && txx_ < TW;                                                        \
            const unsigned pixb_ = (unsigned)((g_second ? p.pix1 : p.pix0) * 4);                                             \
            unsigned rowoff_[6], coloff_[6];                                                                                 \
            _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                                  \
                const int y = 4 * tyy_ - 1 + r, x = 4 * txx_ - 1 + r;                                                        \
                rowoff_[r] = (tile_ok_ && (unsigned)y < (unsigned)Hv) ? (unsigned)((pi_.b * p.Hin + (y >> p.in_shift)) * p.Win) * pixb_ + (unsigned)(cp * 8) : WF_OOB; \
                coloff_[r] = (tile_ok_ && (unsigned)x < (unsigned)Wv) ? (unsigned)(x >> p.in_shift) * pixb_ : WF_OOB;        \
            }                                                                                                                \
            _Pragma("unroll") for (int r = 0; r < 6; ++r) _Pragma("unroll") for (int s = 0; s < 6; ++s)                      \
                voff[r * 6 + s] = ((rowoff_[r] | coloff_[s]) & WF_OOB) ? WF_OOB : rowoff_[r] + coloff_[s];                   \
            go_v = gv; go_second = g_second;                                                                                 \
        }                                                                                                                    \
        g_dead = live_ ? 0u : 1u;                                                                                            \
        g_soff = (g_second ? cc_ - p.C0 : cc_) * 4;                                                                          \
    }
#define W8_GATHER_ROW(R)                                                                                                     \
    {                                                                                                                        \
        const __amdgpu_buffer_rsrc_t rs_ = g_dead ? rsrc_null : (g_second ? rsrc1 : rsrc0);   /* past the block's last chunk: out of range */ \
        _Pragma("unroll") for (int s = 0; s < 6; ++s) raw[(R)*6 + s] =                                                       \
            __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rs_, (int)voff[(R)*6 + s], g_soff, 0));         \
    }
#define W8_ADVANCE() { gq += 1; gc += 1; if (gc == nch) { gc = 0; gv += nblocks; } }
#define W8_COLPASS(S)                                                                                                        \
    {                                                                                                                        \
        floatx2 col_[6], tc_[6];                                                                                             \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) col_[r] = raw[r * 6 + (S)];                                            \
        bt6(col_, tc_);                                                                                                      \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) raw[r * 6 + (S)] = tc_[r];                                             \
    }
#define W8_ROWPASS_WRITE(R, BUF)                                                                                             \
    {                                                                                                                        \
        floatx2 o_[6];                                                                                                       \
        bt6(&raw[(R)*6], o_);                                                                                                \
        float* vw_ = smem + (BUF)*W6_VBUF + vw_base;                                                                         \
        _Pragma("unroll") for (int s = 0; s < 6; ++s)                                                                        \
            *reinterpret_cast<floatx2*>(vw_ + ((R)*6 + s) * W6_ZS) = PAIR ? wf_split_pair(o_[s]) : o_[s];                    \
    }
    // prologue: chunk 0 of the first tile group into V[0], chunk 1 requested, the ring primed
    W8_SETUP()
#pragma unroll
    for (int r = 0; r < 6; ++r) W8_GATHER_ROW(r)
    W8_ADVANCE()
#pragma unroll
    for (int i = 0; i < RING; ++i)
        ring[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, uv_lane, ubase + unit_rel(i), 0));
#pragma unroll
    for (int s = 0; s < 6; ++s) W8_COLPASS(s)
#pragma unroll
    for (int r = 0; r < 6; ++r) W8_ROWPASS_WRITE(r, 0)
    W8_SETUP()
#pragma unroll
    for (int r = 0; r < 6; ++r) W8_GATHER_ROW(r)
    W8_ADVANCE()
    __syncthreads();
    W8_STAMP(st_b)
    while (v < total) {
        const int nv = v + nblocks;
        const W6Item nit = w6_item(nv < total ? nv : v, total, NB, GX, GY, xcd_nb);
        const int nubase = nit.nblk * nsub * 4096 + wave * 1024;
        for (int c = 0; c < nch; ++c) {
            // (the chunk count of a tile group is even: the parity of the global chunk index is the parity of c)
            const float* vb = smem + (c & 1) * W6_VBUF + v_lane;
            const int wbuf = (c & 1) ^ 1;
            const int cur_off = ubase + c * 8192;
            const int nxt_off = c + 1 < nch ? cur_off + 8192 : nubase;
            W8_SETUP()   // chunk (this + 2): its rows are gathered in groups 12 .. 17
            floatx4 vq[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) vq[0][i] = *reinterpret_cast<const floatx4*>(vb + i * W6_ZS);
#pragma unroll
            for (int gi = 0; gi < 18; ++gi) {
                const int zq = gi % 9, cu = gi & 1, nx = cu ^ 1;
                if (gi + 1 < 18) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        vq[nx][i] = *reinterpret_cast<const floatx4*>(vb + (4 * ((gi + 1) % 9) + i) * W6_ZS + ((gi + 1) / 9) * W6_RS);
                }
                if constexpr (PAIR) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[4 * zq + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wf_f16x8, ring[(gi * 4 + i) % RING]),
                                                                                 __builtin_bit_cast(wf_f16x8, vq[cu][i]), acc[4 * zq + i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const floatx4 v_sw = {vq[cu][i][1], vq[cu][i][0], vq[cu][i][3], vq[cu][i][2]};
                        acc[4 * zq + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wf_f16x8, ring[(gi * 4 + i) % RING]),
                                                                                 __builtin_bit_cast(wf_f16x8, v_sw), acc[4 * zq + i], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[4 * zq + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[(gi * 4 + i) % RING][j], vq[cu][i][j], acc[4 * zq + i], 0, 0, 0);
                        if (j < 3) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // the transform slice of this group (chunk + 1 -> V[wbuf]; rows of chunk + 2 into the registers the row pass freed)
                // (DBG, measurement twins: 1 no column pass, 2 no row pass / V writes, 8 no gathers inside the K loop)
                if constexpr (!(DBG & 1)) { if (gi >= 6 && gi < 12) W8_COLPASS(gi - 6) }
                if (gi >= 12) {
                    if constexpr (!(DBG & 2)) W8_ROWPASS_WRITE(gi - 12, wbuf)
                    if constexpr (!(DBG & 8)) W8_GATHER_ROW(gi - 12)
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ul = gi * 4 + i, K = ul + RING;
                    const int off = K < 72 ? cur_off + unit_rel(K) : nxt_off + unit_rel(K - 72);
                    ring[ul % RING] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, uv_lane, off, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            W8_ADVANCE()
            W8_STAMP(st_a)
            __syncthreads();
            W8_STAMP(st_b)
        }
        {
            const int n = it.nblk * 64 + wave * 16 + 4 * g;
            const int tyy = it.gy * 4 + (l15 >> 2), txx = it.gx * 4 + (l15 & 3);
            const bool ok = tyy < TH && txx < TW;
            const unsigned pix = (unsigned)((it.b * p.Ho + 4 * tyy) * p.Wo + 4 * txx);
            const unsigned off_out = ok ? (pix * (unsigned)p.out_stride + (unsigned)n) * 4u : WF_OOB;
            const unsigned off_res = ok ? (pix * (unsigned)p.res_stride + (unsigned)n) * 4u : WF_OOB;
            floatx4 bias = {0.f, 0.f, 0.f, 0.f}, fsc = {1.f, 1.f, 1.f, 1.f}, fsh = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias = *reinterpret_cast<const floatx4*>(p.bias + n);
            if (p.film) {
                const float* f = p.film + (size_t)it.b * p.film_bstride;
                fsc = *reinterpret_cast<const floatx4*>(f + n) + 1.0f;
                fsh = *reinterpret_cast<const floatx4*>(f + p.Cout + n);
            }
            wf64p_epilogue<NT, PAIR, (EPI & 2) != 0, (EPI & 1) != 0, RING>(p, acc, off_out, off_res, rs_out, rs_res, bias, fsc, fsh, ring, rsrc_u, uv_lane, nubase, zstride);
        }
#pragma unroll
        for (int z = 0; z < 36; ++z) acc[z] = floatx4{0.f, 0.f, 0.f, 0.f};
        v = nv; it = nit; ubase = nubase;
        if constexpr (STAMP) st_n += 1;
        W8_STAMP(st_c)
    }
#undef W8_ROWPASS_WRITE
#undef W8_COLPASS
#undef W8_ADVANCE
#undef W8_GATHER_ROW
#undef W8_SETUP
    if constexpr (STAMP) {
        if (lane == 0 && dbg) {
            unsigned long long* d = dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
            d[0] = st_a; d[1] = st_b; d[2] = st_c; d[3] = __builtin_amdgcn_s_memtime() - st_t0; d[4] = st_n;
        }
    }
#undef W8_STAMP
}

}  // namespace

// Blocks the launch of launch_wino_fused(p, ...) creates (the size of the variant-82 stamp buffer: 64 stamps per block)
int wino_fused_num_blocks(const ConvParams& p) {
    return p.B * ((p.Ho / 4 + 3) / 4) * ((p.Wo / 4 + 7) / 8) * (p.Cout / 32);
}

void wino_fused_global_init() {
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        160 * 1024));
#ifdef IRSDE_PROBES
#define W6_ATTR(...) IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_fused64_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    W6_ATTR(W6_RING, false, false);
    W6_ATTR(W6_RING, true, false);
    W6_ATTR(W6_RING, false, true);
    W6_ATTR(W6_RING_ALT, false, false);
    W6_ATTR(W6_RING, false, false, true);
    W6_ATTR(W6_RING_ALT, false, false, true);
    W6_ATTR(W6_RING, false, false, false, 1);
    W6_ATTR(W6_RING, false, false, false, 2);
    W6_ATTR(W6_RING_ALT, false, false, true, 1);
    W6_ATTR(W6_RING_ALT, false, false, false, 1);
#undef W6_ATTR
#endif
#define W6P_ATTR(...) IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_fused64p_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
#define W6P_ATTR4(...) W6P_ATTR(__VA_ARGS__, 0); W6P_ATTR(__VA_ARGS__, 1); W6P_ATTR(__VA_ARGS__, 2); W6P_ATTR(__VA_ARGS__, 3)
    W6P_ATTR4(W6_RING_ALT, false, false, false, true);
    W6P_ATTR4(W6_RING_ALT, false, false, true, true);
#ifdef IRSDE_PROBES
    W6P_ATTR4(W6_RING_ALT, true, false, false, true);
    W6P_ATTR4(W6_RING_ALT, false, true, false, true);
    W6P_ATTR4(W6_RING_ALT, false, false, false, false);
    W6P_ATTR(W6_RING_ALT, false, false, false, true, 0, true); W6P_ATTR(W6_RING_ALT, false, false, false, true, 1, true);
    W6P_ATTR(W6_RING_ALT, false, false, false, true, 2, true); W6P_ATTR(W6_RING_ALT, false, false, false, true, 3, true);
    // r04 tuning twins (irsde_bench_conv 436 .. 440): timed instances per OPT value, stamp instances for OPT = all / no weights / no patches / hot patches
#define W6P_OPT_TIMED(O) W6P_ATTR(W6_RING_ALT, false, false, false, true, 0, false, O); W6P_ATTR(W6_RING_ALT, false, false, false, true, 1, false, O); W6P_ATTR(W6_RING_ALT, false, false, false, true, 3, false, O)
    W6P_OPT_TIMED(1); W6P_OPT_TIMED(2); W6P_OPT_TIMED(4); W6P_OPT_TIMED(8); W6P_OPT_TIMED(15); W6P_OPT_TIMED(64); W6P_OPT_TIMED(65);
#undef W6P_OPT_TIMED
    W6P_ATTR(W6_RING_ALT, false, false, false, true, 1, true, 15); W6P_ATTR(W6_RING_ALT, false, false, false, true, 3, true, 15);
    W6P_ATTR(W6_RING_ALT, true, false, false, true, 1, true, 0); W6P_ATTR(W6_RING_ALT, true, false, false, true, 3, true, 0);
    W6P_ATTR(W6_RING_ALT, false, true, false, true, 1, true, 0); W6P_ATTR(W6_RING_ALT, false, true, false, true, 3, true, 0);
    W6P_ATTR(W6_RING_ALT, false, false, false, true, 1, true, 16); W6P_ATTR(W6_RING_ALT, false, false, false, true, 3, true, 16);
#endif
#undef W6P_ATTR4
#undef W6P_ATTR
#ifdef IRSDE_PROBES
#define W8_ATTR(...) IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_fused64s_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
#define W8_ATTR4(...) W8_ATTR(__VA_ARGS__, 0); W8_ATTR(__VA_ARGS__, 1); W8_ATTR(__VA_ARGS__, 2); W8_ATTR(__VA_ARGS__, 3)
    W8_ATTR4(W6_RING_ALT, false, false, true);
    W8_ATTR4(W6_RING_ALT, true, false, true);
    W8_ATTR4(W6_RING_ALT, false, true, true);
    W8_ATTR(W6_RING_ALT, false, false, true, 1, true); W8_ATTR(W6_RING_ALT, false, false, true, 3, true);
    W8_ATTR(W6_RING_ALT, false, false, true, 1, false, 11); W8_ATTR(W6_RING_ALT, false, false, true, 3, false, 11);
    W8_ATTR(W6_RING_ALT, false, false, true, 1, false, 3); W8_ATTR(W6_RING_ALT, false, false, true, 3, false, 3);
    W8_ATTR(W6_RING_ALT, false, false, true, 1, false, 8); W8_ATTR(W6_RING_ALT, false, false, true, 3, false, 8);
#undef W8_ATTR4
#undef W8_ATTR
#define W7_ATTR(...) IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_fused64h_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
#define W7_ATTR4(...) W7_ATTR(__VA_ARGS__, 0); W7_ATTR(__VA_ARGS__, 1); W7_ATTR(__VA_ARGS__, 2); W7_ATTR(__VA_ARGS__, 3)
    W7_ATTR4(W6_RING_ALT, false, false, false, true);
    W7_ATTR4(W6_RING_ALT, false, false, true, true);
    W7_ATTR4(W6_RING_ALT, true, false, false, true);
    W7_ATTR4(W6_RING_ALT, false, true, false, true);
    W7_ATTR(W6_RING_ALT, false, false, false, true, 1, true); W7_ATTR(W6_RING_ALT, false, false, false, true, 3, true);
    W7_ATTR(W6_RING_ALT, true, false, false, true, 1, true); W7_ATTR(W6_RING_ALT, true, false, false, true, 3, true);
    W7_ATTR(W6_RING_ALT, false, true, false, true, 1, true); W7_ATTR(W6_RING_ALT, false, true, false, true, 3, true);
#undef W7_ATTR4
#undef W7_ATTR
#endif   // IRSDE_PROBES
}

// Geometry / feature check only (the plan decides where the fused kernel pays)
bool wino_fused_eligible(const ConvParams& p) {
    if (p.w_bf || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_y != 1 || p.pad_x != 1 || p.splits != 1 || p.nz != 1) return false;
    if (p.gate || p.shuffle || p.ch_scale || p.in_scale || p.ln_g || p.in_bf16 || p.out_bf16) return false;
    if (p.in_shift != 0 && p.in_shift != 1) return false;
    if (p.Ho != (p.Hin << p.in_shift) || p.Wo != (p.Win << p.in_shift) || (p.Ho & 3) || (p.Wo & 3)) return false;
    if (p.C0 % 32 || p.C1 % 32 || p.C0 + p.C1 == 0 || p.Cout % 32) return false;  // chunk pairs (2 x 16 channels) never straddle the sources
    if (p.C1 && !p.in1) return false;
    if ((p.out_stride & 3) || (p.res && (p.res_stride & 3))) return false;  // 16-byte epilogue accesses
    const double lim = 2147483648.0 - 65536.0;  // buffer offsets are 32-bit; 0x80000000 is the "reads zero" offset
    if (4.0 * p.B * p.Hin * p.Win * (double)p.pix0 >= lim || (p.C1 && 4.0 * p.B * p.Hin * p.Win * (double)p.pix1 >= lim)) return false;
    if (36.0 * 4.0 * p.Cout * (double)(p.C0 + p.C1) >= lim) return false;
    return true;
}

// U[z][n][c] (wino_transform_weights, tile 4) -> the fused kernel's fragment order
//   Uf[z][n >> 5][c >> 3][(c >> 2) & 1][n & 31][c & 3]
// so that the B fragments of one (component, 32 couts, K sub-step of 8) are 1 KB contiguous: lane (cout = l & 31, h = l >> 5)
// reads the 16 bytes k = 4h .. 4h+3, the same k assignment as the A fragments in LDS.
void wino_fused_pack_weights(const float* U, int Cout, int Cin, float* Uf) {
    const int NB = Cout / 32, nsub = Cin / 8;
    for (int z = 0; z < 36; ++z)
        for (int n = 0; n < Cout; ++n)
            for (int c = 0; c < Cin; ++c) {
                const size_t dst = ((((size_t)(z * NB + (n >> 5)) * nsub + (c >> 3)) * 2 + ((c >> 2) & 1)) * 32 + (n & 31)) * 4 + (c & 3);
                Uf[dst] = U[((size_t)z * Cout + n) * Cin + c];
            }
}

void launch_wino_fused(const ConvParams& p, const float* Uf, hipStream_t s, unsigned long long* dbg, int dflags) {
    if (!wino_fused_eligible(p)) throw HipError("launch_wino_fused: layer not eligible");
    if (!Uf) throw HipError("launch_wino_fused: fused weights missing");
    const int TH = p.Ho / 4, TW = p.Wo / 4;
    const int GX = (TW + 7) / 8, GY = (TH + 3) / 4, NB = p.Cout / 32;
    const unsigned in0_bytes = (unsigned)((size_t)p.B * p.Hin * p.Win * p.pix0 * 4);
    const unsigned in1_bytes = p.C1 ? (unsigned)((size_t)p.B * p.Hin * p.Win * p.pix1 * 4) : 0u;
    const unsigned uf_bytes = (unsigned)((size_t)36 * p.Cout * (p.C0 + p.C1) * 4);
    if (dbg)  // tuning aid: the instrumented twin (the stamps cost ~10 % even when they are branched over)
        hipLaunchKernelGGL(wino4_fused_kernel<true>, dim3((unsigned)(p.B * GY * GX * NB)), dim3(WF_NT), WF_LDS_BYTES, s, p, Uf, GX, GY, NB,
                           in0_bytes, in1_bytes, uf_bytes, dbg, dflags);
    else
        hipLaunchKernelGGL(wino4_fused_kernel<false>, dim3((unsigned)(p.B * GY * GX * NB)), dim3(WF_NT), WF_LDS_BYTES, s, p, Uf, GX, GY, NB,
                           in0_bytes, in1_bytes, uf_bytes, dbg, dflags);
    IRSDE_HIP_CHECK(hipGetLastError());
}


// ---- the 64-cout variant (r03) ----
bool wino_fused64_eligible(const ConvParams& p) {
    if (!wino_fused_eligible(p) || p.Cout % 64 || (p.C0 + p.C1) % 64) return false;   // 32-channel chunks, processed in pairs
    // r04: the persistent kernel's epilogue addresses the output / residual through 32-bit buffer offsets too
    const double lim = 2147483648.0 - 65536.0, npix = (double)p.B * p.Ho * p.Wo;
    return 4.0 * npix * p.out_stride < lim && (!p.res || 4.0 * npix * p.res_stride < lim);
}

// U[z][n][c] -> Uf[z][n >> 6][c >> 4][(n >> 4) & 3][(c >> 2) & 3][n & 15][c & 3]: the fragment of one (component, 16-cout
// block, 16-channel k group) is 1 KB contiguous, lane (cout = l & 15, g = l >> 4) reads the 16 bytes k = 4g .. 4g+3 of it.
void wino_fused64_pack_weights(const float* U, int Cout, int Cin, float* Uf) {
    const int NB = Cout / 64, nsub = Cin / 16;
    for (int z = 0; z < 36; ++z)
        for (int n = 0; n < Cout; ++n)
            for (int c = 0; c < Cin; ++c) {
                const size_t dst = ((((((size_t)(z * NB + (n >> 6)) * nsub + (c >> 4)) * 4 + ((n >> 4) & 3)) * 4 + ((c >> 2) & 3)) * 16 + (n & 15)) * 4) + (c & 3);
                Uf[dst] = U[((size_t)z * Cout + n) * Cin + c];
            }
}

// Block mapping of the 64-cout kernel (see the kernel): true = cout block by XCD.  Modelled fabric reads per launch:
//   default  : patches x 1.27 (halo) + U x (blocks / 32)            (every round of 32 blocks on an XCD streams all of U)
//   xcd_nb   : patches x 1.27 x NB   + U slice per XCD, once if it fits L2 (<= 3 MB) else once per round
// (IRSDE_TUNING=1: IRSDE_WINO_FUSED64_XNB = 0 never / 1 whenever legal / -1 the model)
bool wino_fused64_xcd_nb(const ConvParams& p) {
    static const int mode = tuning_env_int("IRSDE_WINO_FUSED64_XNB", -1);
    if (mode == 0) return false;
    const int NB = p.Cout / 64;
    const long long G = (long long)p.B * ((p.Ho / 4 + 3) / 4) * ((p.Wo / 4 + 3) / 4);
    if (NB < 2 || 8 % NB || G % (8 / NB)) return false;
    if (mode == 1) return true;
    const double Ctot = p.C0 + p.C1;
    const double in_b = 4.0 * p.B * p.Hin * p.Win * Ctot * 1.27, u_b = 36.0 * 4.0 * p.Cout * Ctot;
    const double blocks = (double)G * NB;
    const double a = in_b + u_b * blocks / 32.0;
    const double slice = u_b / NB, rounds = blocks / 256.0;
    const double b = in_b * NB + (slice <= 3.0e6 ? u_b * (8.0 / NB) : 8.0 * slice * rounds);
    return b < 0.75 * a;
}

// tuning aid (irsde_bench_conv 435): the stamp buffer of the STAMP twin, 8 waves x 8 counters per persistent block
static unsigned long long* g_w6p_dbg = nullptr;
void wino_fused64_set_debug(unsigned long long* buf) { g_w6p_dbg = buf; }
// tuning aid: the OPT value of launch variants 26 (timed) / 27 (stamps)
static int g_w6p_opt = 0;
void wino_fused64_set_opt(int opt) { g_w6p_opt = opt; }

int wino_fused64_num_blocks(const ConvParams& p) {
    return p.B * ((p.Ho / 4 + 3) / 4) * ((p.Wo / 4 + 3) / 4) * (p.Cout / 64);
}

void launch_wino_fused64_split_weights(const float* Uf, unsigned short* out, size_t nfloats, float scale, hipStream_t s) {
    const size_t nq = nfloats / 4;
    hipLaunchKernelGGL(wf64_split_weights_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, Uf, reinterpret_cast<uint4*>(out), nq, scale);
    IRSDE_HIP_CHECK(hipGetLastError());
}

// variant: 0 production (= 10 unless IRSDE_WINO_FUSED64_NT=0); 1 weight fragments read zeros (no L2 traffic); 2 patch loads read zeros; 3 the shorter U ring;
// 4 / 5: the fp16-pair kernel (Uf = the wf64_split_weights_kernel output, p.pair_scale = 1 / (kWinoFused64PairVScale * weight scale)), ring 12 (production:
// 3 - 6 % faster than 18 on every layer class, profiles/r03_wino_fused64_pair_sweep.txt — the MFMAs of a unit take 34 instead of 128 cycles, the
// deeper ring only costs registers) / 18; 6 / 7 / 8: the f32 kernel with non-temporal residual loads + output stores / + patch loads / without
// any hint; 10: 12 units in flight + the epilogue hint = production since late r03 (0 .. -4 % against 6 on every layer class,
// profiles/r03_wino_fused64_nt.txt).  Production (0, 4) carries the epilogue hint (IRSDE_WINO_FUSED64_NT=0 under IRSDE_TUNING=1 switches it off): the streamed
// epilogue traffic no longer evicts the weight fragments from the XCD's 4 MB L2 - 128 -> 128 @ 256^2 1.14 -> 1.01 ms, profiles/r03_wino_fused64_nt.txt
void launch_wino_fused64(const ConvParams& p, const float* Uf, hipStream_t s, int variant) {
    if (!wino_fused64_eligible(p)) throw HipError("launch_wino_fused64: layer not eligible");
    if (!Uf) throw HipError("launch_wino_fused64: fused weights missing");
    const int TH = p.Ho / 4, TW = p.Wo / 4;
    const int GX = (TW + 3) / 4, GY = (TH + 3) / 4, NB = p.Cout / 64;
    const unsigned in0_bytes = (unsigned)((size_t)p.B * p.Hin * p.Win * p.pix0 * 4);
    const unsigned in1_bytes = p.C1 ? (unsigned)((size_t)p.B * p.Hin * p.Win * p.pix1 * 4) : 0u;
    const unsigned uf_bytes = (unsigned)((size_t)36 * p.Cout * (p.C0 + p.C1) * 4);
    const dim3 grid((unsigned)(p.B * GY * GX * NB));
    // variant + 64: the cout-block-by-XCD mapping wherever it is legal (test hook: the production choice follows the traffic model)
    const bool force_xnb = (variant & 64) != 0;
    variant &= 63;
    const long long G_ = (long long)p.B * GY * GX;
    const int xcd_nb = (force_xnb ? (NB >= 2 && 8 % NB == 0 && G_ % (8 / NB) == 0) : wino_fused64_xcd_nb(p)) ? 1 : 0;
#define W6_LAUNCH(...) hipLaunchKernelGGL((wino4_fused64_kernel<__VA_ARGS__>), grid, dim3(WF_NT), W6_LDS_BYTES, s, p, Uf, GX, GY, NB, in0_bytes, in1_bytes, uf_bytes, xcd_nb)
    static const bool nt = tuning_env_int("IRSDE_WINO_FUSED64_NT", 1) != 0;
    // r04: the persistent kernel (one block per CU walks its tile groups, output transform in registers) is production;
    // IRSDE_WINO_FUSED64_PERSIST=0 under IRSDE_TUNING=1 selects r03's one-block-per-tile-group kernel
    // (r04, later) 2 selects the halo kernel (patches staged through LDS by LDS-DMA), 3 the single-stream kernel; 1 = the register-patch persistent kernel is production
    static const int persist = tuning_env_int("IRSDE_WINO_FUSED64_PERSIST", 1);
    if (persist == 1 && (variant == 0 || variant == 4)) variant = variant == 0 ? 20 : 24;
    if (persist == 2 && (variant == 0 || variant == 4)) variant = variant == 0 ? 40 : 44;
    if (persist >= 3 && (variant == 0 || variant == 4)) variant = variant == 0 ? 48 : 52;   // 3: the single-stream kernel (one role per wave)
    if (variant >= 20) {   // 20 production f32, 21 weight fragments read zeros, 22 patch loads read zeros, 23 no non-temporal hint, 24 fp16 pairs
        const int ncu = device_cu_count();   // per device (a process may hold parts with different CU counts)
        const int total = (int)grid.x;
        const size_t npix_out = (size_t)p.B * p.Ho * p.Wo;
        const size_t ob = npix_out * p.out_stride * 4, rb = p.res ? npix_out * p.res_stride * 4 : 0;
        if (ob >= 0x7fff0000ull || rb >= 0x7fff0000ull) throw HipError("launch_wino_fused64: output / residual tensor too large for 32-bit buffer offsets");
        const unsigned out_bytes = (unsigned)ob, res_bytes = (unsigned)rb;
        // one block per CU; a multiple of 8 so that virtual block id % 8 stays the XCD of the block that runs it
        const dim3 pgrid((unsigned)std::min(total, std::max(8, ncu & ~7)));
#define W6P_LAUNCH(...) hipLaunchKernelGGL((wino4_fused64p_kernel<__VA_ARGS__>), pgrid, dim3(WF_NT), W6_LDS_BYTES + 1024, s, p, Uf, GX, GY, NB, in0_bytes, in1_bytes, uf_bytes, out_bytes, res_bytes, xcd_nb, total, g_w6p_dbg)
        const int epi = (p.silu ? 1 : 0) | (p.res ? 2 : 0);   // one kernel instance per epilogue (see the EPI template parameter)
#define W6P_LAUNCH_EPI(...)                                          \
    switch (epi) {                                                   \
        case 0: W6P_LAUNCH(W6_RING_ALT, __VA_ARGS__, 0); break;      \
        case 1: W6P_LAUNCH(W6_RING_ALT, __VA_ARGS__, 1); break;      \
        case 2: W6P_LAUNCH(W6_RING_ALT, __VA_ARGS__, 2); break;      \
        default: W6P_LAUNCH(W6_RING_ALT, __VA_ARGS__, 3); break;     \
    }
#define W6P_LAUNCH_EPI_STAMP()                                                          \
    switch (epi) {                                                                      \
        case 0: W6P_LAUNCH(W6_RING_ALT, false, false, false, true, 0, true); break;     \
        case 1: W6P_LAUNCH(W6_RING_ALT, false, false, false, true, 1, true); break;     \
        case 2: W6P_LAUNCH(W6_RING_ALT, false, false, false, true, 2, true); break;     \
        default: W6P_LAUNCH(W6_RING_ALT, false, false, false, true, 3, true); break;    \
    }
#ifdef IRSDE_PROBES   // superseded / measurement kernels (r04 halo and single-stream kernels, their stamp and ablation twins): not in the product library
        if (variant >= 48 && variant <= 54) {   // the single-stream kernel: 48 f32, 50 patch loads read zeros, 52 fp16 pairs, 53 cycle stamps (epilogues 1 / 3)
#define W8_LAUNCH(...) hipLaunchKernelGGL((wino4_fused64s_kernel<W6_RING_ALT, __VA_ARGS__>), pgrid, dim3(256), W6_LDS_BYTES + 1024, s, p, Uf, GX, GY, NB, in0_bytes, in1_bytes, uf_bytes, out_bytes, res_bytes, xcd_nb, total, g_w6p_dbg)
#define W8_LAUNCH_EPI(...)                                  \
    switch (epi) {                                          \
        case 0: W8_LAUNCH(__VA_ARGS__, 0); break;           \
        case 1: W8_LAUNCH(__VA_ARGS__, 1); break;           \
        case 2: W8_LAUNCH(__VA_ARGS__, 2); break;           \
        default: W8_LAUNCH(__VA_ARGS__, 3); break;          \
    }
            switch (variant) {
                case 48: W8_LAUNCH_EPI(false, false, true) break;
                case 50: W8_LAUNCH_EPI(true, false, true) break;
                case 52: W8_LAUNCH_EPI(false, true, true) break;
                case 53:
                    if (epi != 1 && epi != 3) throw HipError("launch_wino_fused64: stamp twins exist for epilogues 1 / 3");
                    if (epi == 1) W8_LAUNCH(false, false, true, 1, true); else W8_LAUNCH(false, false, true, 3, true);
                    break;
                case 49: case 51: case 54:   // measurement twins (results are garbage): 49 MFMAs + ring + V reads only, 51 + gathers, 54 everything but the gathers
                    if (epi != 1 && epi != 3) throw HipError("launch_wino_fused64: measurement twins exist for epilogues 1 / 3");
                    if (variant == 49) { if (epi == 1) W8_LAUNCH(false, false, true, 1, false, 11); else W8_LAUNCH(false, false, true, 3, false, 11); }
                    if (variant == 51) { if (epi == 1) W8_LAUNCH(false, false, true, 1, false, 3); else W8_LAUNCH(false, false, true, 3, false, 3); }
                    if (variant == 54) { if (epi == 1) W8_LAUNCH(false, false, true, 1, false, 8); else W8_LAUNCH(false, false, true, 3, false, 8); }
                    break;
                default: throw HipError("launch_wino_fused64: bad variant");
            }
#undef W8_LAUNCH_EPI
#undef W8_LAUNCH
            IRSDE_HIP_CHECK(hipGetLastError());
            return;
        }
        if (variant >= 40 && variant <= 47) {   // the halo kernel: 40 production f32, 41 weight fragments read zeros, 42 halo fetches read zeros, 44 fp16 pairs
#define W7_LAUNCH(...) hipLaunchKernelGGL((wino4_fused64h_kernel<W6_RING_ALT, __VA_ARGS__>), pgrid, dim3(WF_NT), W7_LDS_BYTES, s, p, Uf, GX, GY, NB, in0_bytes, in1_bytes, uf_bytes, out_bytes, res_bytes, xcd_nb, total, g_w6p_dbg)
#define W7_LAUNCH_EPI(...)                                  \
    switch (epi) {                                          \
        case 0: W7_LAUNCH(__VA_ARGS__, 0); break;           \
        case 1: W7_LAUNCH(__VA_ARGS__, 1); break;           \
        case 2: W7_LAUNCH(__VA_ARGS__, 2); break;           \
        default: W7_LAUNCH(__VA_ARGS__, 3); break;          \
    }
            switch (variant) {
                case 40: W7_LAUNCH_EPI(false, false, false, true) break;
                case 41: W7_LAUNCH_EPI(true, false, false, true) break;
                case 42: W7_LAUNCH_EPI(false, true, false, true) break;
                case 44: W7_LAUNCH_EPI(false, false, true, true) break;
                case 45: case 46: case 47:   // stamp twins (epilogues 1 / 3): production / no weight traffic / no halo traffic
                    if (epi != 1 && epi != 3) throw HipError("launch_wino_fused64: stamp twins exist for epilogues 1 / 3");
                    if (variant == 45) { if (epi == 1) W7_LAUNCH(false, false, false, true, 1, true); else W7_LAUNCH(false, false, false, true, 3, true); }
                    if (variant == 46) { if (epi == 1) W7_LAUNCH(true, false, false, true, 1, true); else W7_LAUNCH(true, false, false, true, 3, true); }
                    if (variant == 47) { if (epi == 1) W7_LAUNCH(false, true, false, true, 1, true); else W7_LAUNCH(false, true, false, true, 3, true); }
                    break;
                default: throw HipError("launch_wino_fused64: bad variant");
            }
#undef W7_LAUNCH_EPI
#undef W7_LAUNCH
            IRSDE_HIP_CHECK(hipGetLastError());
            return;
        }
#else
        if (variant >= 40) throw HipError("launch_wino_fused64: the halo / single-stream kernels are measurement variants: build with make PROBES=1 (libirsde_hip_probes.so)");
#endif
        switch (variant) {
            case 20: W6P_LAUNCH_EPI(false, false, false, true) break;
#ifdef IRSDE_PROBES
            case 21: W6P_LAUNCH_EPI(true, false, false, true) break;    // weight fragments read zeros
            case 22: W6P_LAUNCH_EPI(false, true, false, true) break;    // patch loads read zeros
            case 23: W6P_LAUNCH_EPI(false, false, false, false) break;  // no non-temporal hint
#endif
            case 24: W6P_LAUNCH_EPI(false, false, true, true) break;    // fp16 pairs
#ifdef IRSDE_PROBES
            case 25: W6P_LAUNCH_EPI_STAMP() break;                      // cycle stamps into the buffer of wino_fused64_set_debug()
            case 26: {   // timed tuning twins: OPT = wino_fused64_set_opt() (epilogues 0 / 1 / 3 only)
#define W6P_LAUNCH_OPT(O)                                                                  \
    switch (epi) {                                                                         \
        case 0: W6P_LAUNCH(W6_RING_ALT, false, false, false, true, 0, false, O); break;    \
        case 1: W6P_LAUNCH(W6_RING_ALT, false, false, false, true, 1, false, O); break;    \
        case 3: W6P_LAUNCH(W6_RING_ALT, false, false, false, true, 3, false, O); break;    \
        default: throw HipError("launch_wino_fused64: tuning twins exist for epilogues 0 / 1 / 3");  \
    }
                switch (g_w6p_opt) {
                    case 1: W6P_LAUNCH_OPT(1) break;
                    case 2: W6P_LAUNCH_OPT(2) break;
                    case 4: W6P_LAUNCH_OPT(4) break;
                    case 8: W6P_LAUNCH_OPT(8) break;
                    
                    case 15: W6P_LAUNCH_OPT(15) break;
                    case 64: W6P_LAUNCH_OPT(64) break;
                    case 65: W6P_LAUNCH_OPT(65) break;
                    default: throw HipError("launch_wino_fused64: no timed twin for this OPT");
                }
#undef W6P_LAUNCH_OPT
                break;
            }
            case 27: case 28: case 29: case 30: {   // stamp twins: 27 OPT = all, 28 no weight traffic, 29 no patch traffic, 30 hot patches
                if (epi != 1 && epi != 3) throw HipError("launch_wino_fused64: stamp twins exist for epilogues 1 / 3");
                if (variant == 27) { if (epi == 1) W6P_LAUNCH(W6_RING_ALT, false, false, false, true, 1, true, 15); else W6P_LAUNCH(W6_RING_ALT, false, false, false, true, 3, true, 15); }
                if (variant == 28) { if (epi == 1) W6P_LAUNCH(W6_RING_ALT, true, false, false, true, 1, true, 0); else W6P_LAUNCH(W6_RING_ALT, true, false, false, true, 3, true, 0); }
                if (variant == 29) { if (epi == 1) W6P_LAUNCH(W6_RING_ALT, false, true, false, true, 1, true, 0); else W6P_LAUNCH(W6_RING_ALT, false, true, false, true, 3, true, 0); }
                if (variant == 30) { if (epi == 1) W6P_LAUNCH(W6_RING_ALT, false, false, false, true, 1, true, 16); else W6P_LAUNCH(W6_RING_ALT, false, false, false, true, 3, true, 16); }
                break;
            }
#endif   // IRSDE_PROBES
            default: throw HipError("launch_wino_fused64: bad variant (the ablation / stamp / tuning twins need a make PROBES=1 build)");
        }
#undef W6P_LAUNCH_EPI_STAMP
#undef W6P_LAUNCH_EPI
#undef W6P_LAUNCH
        IRSDE_HIP_CHECK(hipGetLastError());
        return;
    }
#ifndef IRSDE_PROBES
    (void)grid; (void)nt;
    throw HipError("launch_wino_fused64: r03's one-block-per-tile-group kernel (IRSDE_WINO_FUSED64_PERSIST=0, debug_conv 34 / 35 with persist off) is a superseded variant: build with make PROBES=1");
#else
    if (variant == 0 && nt) variant = 10;   // 12 units in flight: with the epilogue hint the weights hit L2 more often, and the shorter ring has no spills (237 VGPRs)
    if (variant == 4 && nt) variant = 9;
    switch (variant) {
        case 0: case 8: W6_LAUNCH(W6_RING, false, false); break;
        case 9: W6_LAUNCH(W6_RING_ALT, false, false, true, 1); break;
        case 10: W6_LAUNCH(W6_RING_ALT, false, false, false, 1); break;   // f32, 12 units in flight, non-temporal epilogue (irsde_bench_conv 410)
        case 1: W6_LAUNCH(W6_RING, true, false); break;
        case 2: W6_LAUNCH(W6_RING, false, true); break;
        case 3: W6_LAUNCH(W6_RING_ALT, false, false); break;
        case 4: W6_LAUNCH(W6_RING_ALT, false, false, true); break;
        case 5: W6_LAUNCH(W6_RING, false, false, true); break;
        case 6: W6_LAUNCH(W6_RING, false, false, false, 1); break;
        case 7: W6_LAUNCH(W6_RING, false, false, false, 2); break;
        default: throw HipError("launch_wino_fused64: bad variant");
    }
    IRSDE_HIP_CHECK(hipGetLastError());
#endif   // IRSDE_PROBES
#undef W6_LAUNCH
}

}  // namespace irsde
