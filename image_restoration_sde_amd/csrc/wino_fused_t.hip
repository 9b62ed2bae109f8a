// r06: wino4_fused64t_kernel — the fused Winograd F(4x4,3x3) convolution with every weight fragment feeding TWO tile groups.
//
// Replaces (for the layers with enough work items: the plan decides) wino4_fused64p_kernel (wino_fused.hip) on the 3x3 s1 Block.proj / fused-upsample
// convolutions of the big feature maps (module_util.py:108-122, DenoisingUNet_arch.py:67).  What bound that kernel (profiles/r04_wino_fused64_notes.md,
// TD_TD_BUSY 0.85 at 0.6 MFMA-busy): each of its four MFMA waves streams a private 1 KB weight fragment from L2 for every four
// v_mfma_f32_16x16x4_f32 — 288 KB per 9216 MFMA cycles and CU through a register-return path that delivers ~30 B/clk — and the patch gathers of
// the producer waves queue behind them.  Here the bytes per MFMA are halved without touching the transform work per MFMA:
//   * work item = 32 tiles (two 4 x 4 tile groups side by side = 16 x 32 output pixels of one image) x 64 output channels x all 36 components;
//     K chunks of 16 input channels (one 1 KB weight unit = 16 couts x 16 channels per (component, cout block) and chunk);
//   * 8 waves, ALL of them matrix waves, two per SIMD: wave (cb = w & 3, h = w >> 2) owns the 16-cout block cb and the 18 components of the component
//     columns s = 3h .. 3h + 2 (z = 6 r + s), for BOTH tile groups: 18 x 2 x 4 = 144 accumulator registers, and every weight unit it loads
//     feeds 8 MFMAs (4 k steps x 2 tile groups): 144 KB of fragments per 9216 MFMA cycles and CU;
//   * no producer waves (with 144 accumulators per wave and two waves per SIMD the register file is spent): the input transform runs INSIDE the matrix
//     waves' instruction streams, and split so that it needs 36 instead of 72 patch registers: lane = (column half hh, tile, channel pair);
//     the lane gathers the 6 x 3 half patch d[0..5][3hh .. 3hh+2] (18 buffer_load_dwordx2), runs the column pass B^T d on its three columns, trades
//     rows with its partner lane (lane ^ 32) by 18 v_permlane32_swap — afterwards the lanes 0-31 hold rows 0-2 and the lanes 32-63 rows 3-5 of
//     B^T d, all six columns — runs the row pass on its three rows and writes 18 components (ds_write_b64).  A wave transforms 4 tiles x 8 channel pairs per
//     chunk: 8 waves = the 32 tiles x 16 channels of the chunk.  Same operations in the same order as the producers of wino4_fused64p_kernel: V is bit-identical;
//   * chunk c + 1 is transformed during the K loop of chunk c (units 6 .. 12 of its 18 weight units), its patch was gathered during chunk c - 1 (units 13 .. 15):
//     one barrier per chunk, V double buffer of 2 x 72 KB;
//   * output transform: wave (cb, h) applies A^T . over the component rows to its three component columns (lane-local, both tile groups); the row of the
//     second stage y[i][j] = sum_s u[i][s] A[s][j] needs the other wave's columns: the partial terms (u[i][0], u1 + u2, u1 - u2) resp. (u3 + u4, u3 - u4, u[i][5])
//     of the tile group the OTHER wave finishes go through LDS (12 x 16 B per lane, one pass, fixed order: deterministic), wave (cb, h) finishes tile group h
//     with exactly the expressions of wf64p_epilogue: the output is bit-identical to wino4_fused64p_kernel's (tests/test_gpu_parity.py).
// Weights: the fragment order of wino_fused64_pack_weights (the same buffer serves both kernels).
#include "common.h"
#include "wino_fused_shared.h"
#include <mutex>

namespace irsde {

namespace {

constexpr int WT_NT = 512;
constexpr int WT_KC = 16;                       // input channels per chunk
constexpr int WT_PS = 256;                      // floats per (component, tile group) plane of V: [g = (c >> 2) & 3][tile ^ 2g][j = c & 3]
constexpr int WT_ZS = 2 * WT_PS;                // ... per component
constexpr int WT_VBUF = 36 * WT_ZS;             // floats per V buffer (73 728 B)
constexpr int WT_XCH_BYTES = 8 * 12 * 64 * 16;   // the epilogue's exchange (8 waves x 12 KB): aliases the V buffers, bytes [0, 96 K)
constexpr int WT_RES_BYTES = 8 * 8 * 1024;       // ... and behind it the residual staging (8 waves x 8 LDS-DMA rows of 1 KB): bytes [96 K, 160 K)
constexpr int WT_LDS_BYTES = 160 * 1024;         // V double buffer 147 456 B + 16 KB
static_assert(WT_XCH_BYTES + WT_RES_BYTES <= WT_LDS_BYTES && 2 * WT_VBUF * 4 <= WT_LDS_BYTES, "LDS budget");
constexpr unsigned WT_ROW_OOB = 0x80000000u;    // row / column part of a gather offset that must read 0: every sum with one of them is >= 0x40000000,
constexpr unsigned WT_COL_OOB = 0x40000000u;    // past the end of any tensor this kernel accepts (<= 1 GiB, wino_fused64t_eligible)

// rows 1 (lanes 32 - 63) of a change places with rows 0 (lanes 0 - 31) of b, six register pairs per statement.  Inline assembly: hipcc (ROCm 7.2)
// miscompiles __builtin_amdgcn_permlane32_swap's result pair (kernels_misc.hip, group_sum); s_nop 1 = the wait states behind a VALU write of the operands.
__device__ __forceinline__ void swap32x6(float& a0, float& b0, float& a1, float& b1, float& a2, float& b2, float& a3, float& b3, float& a4, float& b4,
                                         float& a5, float& b5) {
    asm volatile(
        "s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\t"
        "v_permlane32_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %8, %9\n\tv_permlane32_swap_b32 %10, %11"
        : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3), "+v"(a4), "+v"(b4), "+v"(a5), "+v"(b5));
}

// 16 bytes per lane as one buffer_load_dwordx4 / buffer_store_dwordx4, or (SPLIT, measurement twins) as two 8-byte instructions
template <bool SPLIT, int AUX>
__device__ __forceinline__ floatx4 wt_load16(const __amdgpu_buffer_rsrc_t rs, const int voff, const int soff) {
    if constexpr (SPLIT) {
        const floatx2 a = __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, AUX));
        const floatx2 b = __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 8, soff, AUX));
        return floatx4{a.x, a.y, b.x, b.y};
    } else {
        return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX));
    }
}

// First stage of the output transform of one work item by wave (cb, H), lane-local: u[i][sc] = sum_r A^T[i][r] M[r][3 H + sc] on the wave's three
// component columns for both tile groups, then the partial terms of the second stage —
//   H = 0 (columns 0 1 2): u0, s12 = u1 + u2, d12 = u1 - u2;   H = 1 (columns 3 4 5): s34 = u3 + u4, d34 = u3 - u4, u5   (index 3 i + k, i = output row)
// — of tile group H into keep, those of the tile group the partner wave finishes (wave ^ 4: same cout block, same lane = same (tile, couts)) into the
// exchange area (12 KB per wave, 1 KB rows).
template <int H>
__device__ __forceinline__ void wt_stage1(const floatx4 (&acc)[18][2], floatx4 (&keep)[12], float* xs) {
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
        floatx4 u[4][3];
#pragma unroll
        for (int sc = 0; sc < 3; ++sc) {
            const floatx4 m0 = acc[sc][tg], m1 = acc[3 + sc][tg], m2 = acc[6 + sc][tg], m3 = acc[9 + sc][tg], m4 = acc[12 + sc][tg], m5 = acc[15 + sc][tg];
            const floatx4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            u[0][sc] = m0 + s12 + s34;
            u[1][sc] = d12 + 2.0f * d34;
            u[2][sc] = s12 + 4.0f * s34;
            u[3][sc] = d12 + 8.0f * d34 + m5;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const floatx4 t0 = H == 0 ? u[i][0] : u[i][0] + u[i][1];
            const floatx4 t1 = H == 0 ? u[i][1] + u[i][2] : u[i][0] - u[i][1];
            const floatx4 t2 = H == 0 ? u[i][1] - u[i][2] : u[i][2];
            if (tg == H) {
                keep[3 * i] = t0; keep[3 * i + 1] = t1; keep[3 * i + 2] = t2;
            } else {
                *reinterpret_cast<floatx4*>(xs + (3 * i) * 256) = t0;
                *reinterpret_cast<floatx4*>(xs + (3 * i + 1) * 256) = t1;
                *reinterpret_cast<floatx4*>(xs + (3 * i + 2) * 256) = t2;
            }
        }
    }
}

// Second stage + epilogue of the output rows 2 HALF, 2 HALF + 1 of tile group H by wave (cb, H): keep = this wave's partial terms, recv = the partner wave's.
// Same expressions as wf64p_epilogue (wino_fused.hip): bit-identical outputs.
template <int H, int HALF, bool NT, bool RES, bool SILU, bool SPLIT_ST = false>
__device__ __forceinline__ void wt_finish_rows(const ConvParams& p, const floatx4 (&keep)[12], const floatx4 (&recv)[12], const floatx4 (&rv)[2][4],
                                               const unsigned lane_off_out, const __amdgpu_buffer_rsrc_t rs_out, const floatx4 bias, const floatx4 fsc,
                                               const floatx4 fsh) {
    constexpr int AUX = NT ? 2 : 0;
    const int orow = p.Wo * p.out_stride * 4, opix = p.out_stride * 4;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * HALF + ii;
        const floatx4 u0 = H == 0 ? keep[3 * i] : recv[3 * i], s12 = H == 0 ? keep[3 * i + 1] : recv[3 * i + 1], d12 = H == 0 ? keep[3 * i + 2] : recv[3 * i + 2];
        const floatx4 s34 = H == 0 ? recv[3 * i] : keep[3 * i], d34 = H == 0 ? recv[3 * i + 1] : keep[3 * i + 1], u5 = H == 0 ? recv[3 * i + 2] : keep[3 * i + 2];
        floatx4 y[4];
        y[0] = u0 + s12 + s34;
        y[1] = d12 + 2.0f * d34;
        y[2] = s12 + 4.0f * s34;
        y[3] = d12 + 8.0f * d34 + u5;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            floatx4 v = (y[j] + bias) * fsc + fsh;
            if constexpr (SILU) {
                v.x = silu_w(v.x); v.y = silu_w(v.y); v.z = silu_w(v.z); v.w = silu_w(v.w);
            }
            if constexpr (RES) v = v + rv[ii][j];
            // (pixel offset in the VECTOR offset: see wf64p_epilogue — with an SGPR soffset hipcc pads no wait state behind a buffer_store_dwordx4)
            if constexpr (SPLIT_ST) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, floatx2{v.x, v.y}), rs_out,
                                                      (int)(lane_off_out + (unsigned)(i * orow + j * opix)), 0, AUX);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, floatx2{v.z, v.w}), rs_out,
                                                      (int)(lane_off_out + (unsigned)(i * orow + j * opix) + 8u), 0, AUX);
            } else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), rs_out,
                                                   (int)(lane_off_out + (unsigned)(i * orow + j * opix)), 0, AUX);
        }
    }
}

// NR: weight units in flight per wave (x 4 registers; 18 % NR == 0).  NOWT / NOPATCH: measurement twins (weight fragments / patch gathers read zeros without
// memory traffic).  EPI: bit 0 SiLU, bit 1 residual.  STAMP: per-wave cycle totals into dbg[(block * 8 + wave) * 8 ..]: { K loops, barrier waits, epilogue +
// (DBG, measurement twins: 1 no column pass, 2 no row pass / V writes, 4 no swaps, 8 no gathers inside the K loop — results are garbage;
//  16 / 32 / 64: residual loads / output stores / weight units as two 8-byte instructions per lane instead of one 16-byte one; 256: the residual tile
//  gathered into registers instead of through LDS — results unchanged)
// first-chunk transform, whole kernel, items, first stage up to the exchange barrier, exchange (two barriers), second stage + stores }.
template <int NR, bool NOWT, bool NOPATCH, bool NT, int EPI, bool STAMP = false, int DBG = 0>
__global__ __launch_bounds__(WT_NT, 2) void wino4_fused64t_kernel(const ConvParams p, const float* __restrict__ Uf, const int GX, const int GY, const int NB,
                                                                   const unsigned in0_bytes, const unsigned in1_bytes, const unsigned uf_bytes,
                                                                   const unsigned out_bytes, const unsigned res_bytes, const int xcd_nb, const int total,
                                                                   unsigned long long* __restrict__ dbg, const int skew) {
    static_assert(18 % NR == 0, "the ring must divide the 18 weight units of a chunk");
    constexpr bool RES = (EPI & 2) != 0, SILU = (EPI & 1) != 0;
    unsigned long long st_a = 0, st_b = 0, st_c = 0, st_n = 0, st_t0 = 0, st_t = 0, st_e1 = 0, st_e2 = 0, st_e3 = 0;
    if constexpr (STAMP) st_t0 = st_t = __builtin_amdgcn_s_memtime();
#define WT_STAMP(ACC)                                                     \
    if constexpr (STAMP) {                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        ACC += now_ - st_t;                                               \
        st_t = now_;                                                      \
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave & 3, h = wave >> 2;
    if (skew) {   // start skew: the blocks of an XCD enter the kernel in 8 phase classes, `skew` cycles apart (see launch_wino_fused64t)
        const unsigned long long until = __builtin_amdgcn_s_memtime() + (unsigned long long)skew * ((blockIdx.x >> 3) & 7);
        while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(32);
    }
    const int TH = p.Ho >> 2, TW = p.Wo >> 2;
    const int Ctot = p.C0 + p.C1;
    const int nch = Ctot / WT_KC;   // chunks = 16-channel k groups (one weight unit each); a multiple of 4
    const int nblocks = gridDim.x;
    // ---- matrix role: lane (tile = l & 15, g = l >> 4) of wave (cb, h)
    const int l15 = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Uf), 0, NOWT ? 0u : uf_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.out), 0, p.res ? res_bytes : 0u, 0x00020000);
    floatx4 acc[18][2];   // [unit zl = 3 r + sc: component z = 6 r + 3 h + sc][tile group]
    const floatx4 kZero4 = {0.f, 0.f, 0.f, 0.f};
    // unit (component z, k group s) of cout block (nblk, cb): 1 KB at Uf + (((z NB + nblk) nch + s) 4 + cb) KB; lane reads 16 B
    const int uv_lane = lane * 16;
    const int zstride = NB * nch * 4096;   // bytes between components
    int v = blockIdx.x;
    W6Item it = w6_item(v, total, NB, GX, GY, xcd_nb);
    int ubase = it.nblk * nch * 4096 + cb * 1024 + 3 * h * zstride;
    auto unit_rel = [&](const int zl) { return ((zl / 3) * 6 + zl % 3) * zstride; };
    floatx4 ring[NR];
    const int v_lane = g * 64 + (((l15 ^ (2 * g)) & 15) * 4);
    // ---- transform role: lane = (column half hh, tile 4 (wave & 3) + tl of tile group wave >> 2, channel pair cp of the chunk)
    const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0), 0, NOPATCH ? 0u : in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in1 ? p.in1 : p.in0), 0, (p.in1 && !NOPATCH) ? in1_bytes : 0u, 0x00020000);
    const int hh = lane >> 5, tl = (lane >> 3) & 3, cp = lane & 7;
    const int ptile = 4 * (wave & 3) + tl, ptg = wave >> 2;
    const int kg = cp >> 1;
    // LDS float offset of the lane's first component (row 3 hh, column 0): [z][tile group][g][tile ^ 2g][j]
    const int vw_base = hh * (18 * WT_ZS) + ptg * WT_PS + kg * 64 + (((ptile ^ (2 * kg)) & 15) * 4) + 2 * (cp & 1);
    const int Hv = p.Hin << p.in_shift, Wv = p.Win << p.in_shift;
    floatx2 raw[18];   // [i = patch row][jl = local column]; after the swaps [rl (+3: columns 3 .. 5)][jl]
    unsigned rowoff[6], coloff[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) rowoff[i] = WT_ROW_OOB;
#pragma unroll
    for (int j = 0; j < 3; ++j) coloff[j] = WT_COL_OOB;
    int gv = blockIdx.x, gc = 0;           // the next chunk to gather: item, chunk of the item
    W6Item git = it, nit = it;             // ... its coordinates; the item after the one in the K loop
    int go_v = -1, go_second = -1;         // what rowoff / coloff were built for
    int g_soff = 0, g_second = 0;
// offsets / descriptor choice of the next chunk to gather (rebuilt when the item or the concat source changes; past the last item: every offset out of range)
#define WT_SETUP()                                                                                                           \
    {                                                                                                                        \
        const bool live_ = gv < total;                                                                                       \
        const int cc_ = gc * WT_KC;                                                                                          \
        g_second = cc_ >= p.C0 ? 1 : 0;                                                                                      \
        if (gv != go_v || g_second != go_second) {                                                                           \
            const W6Item pi_ = git;                                                                                          \
            const int tyy_ = pi_.gy * 4 + (wave & 3), txx_ = pi_.gx * 8 + ptg * 4 + tl;                                      \
            const bool tile_ok_ = live_ && tyy_ < TH && txx_ < TW;                                                           \
            const unsigned pixb_ = (unsigned)((g_second ? p.pix1 : p.pix0) * 4);                                             \
            _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                                  \
                const int y = 4 * tyy_ - 1 + i;                                                                              \
                rowoff[i] = (tile_ok_ && (unsigned)y < (unsigned)Hv) ? (unsigned)((pi_.b * p.Hin + (y >> p.in_shift)) * p.Win) * pixb_ + (unsigned)(cp * 8) : WT_ROW_OOB; \
            }                                                                                                                \
            _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                  \
                const int x = 4 * txx_ - 1 + 3 * hh + j;                                                                     \
                coloff[j] = ((unsigned)x < (unsigned)Wv) ? (unsigned)(x >> p.in_shift) * pixb_ : WT_COL_OOB;                 \
            }                                                                                                                \
            go_v = gv; go_second = g_second;                                                                                 \
        }                                                                                                                    \
        g_soff = (g_second ? cc_ - p.C0 : cc_) * 4;                                                                          \
    }
#define WT_GATHER_ROW(I)                                                                                                     \
    {                                                                                                                        \
        const __amdgpu_buffer_rsrc_t rs_ = g_second ? rsrc1 : rsrc0;                                                        \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) raw[(I)*3 + j] =                                                       \
            __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rs_, (int)(rowoff[(I)] + coloff[j]), g_soff, 0)); \
    }
#define WT_ADVANCE() { gc += 1; if (gc == nch) { gc = 0; gv += nblocks; git = nit; } }   /* (the stream reaches the next item while `nit` is that item) */
// column pass on local column J (in place): raw[i][J] <- (B^T d)[i][J]
#define WT_COLPASS(J)                                                                                                        \
    {                                                                                                                        \
        floatx2 col_[6], tc_[6];                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) col_[i] = raw[i * 3 + (J)];                                            \
        bt6(col_, tc_);                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) raw[i * 3 + (J)] = tc_[i];                                             \
    }
// rows 3 .. 5 of the lanes 0 - 31 change places with rows 0 .. 2 of the lanes 32 - 63: afterwards raw[rl][j] / raw[rl + 3][j] = columns j / 3 + j of row rl + 3 hh
#define WT_SWAPS()                                                                                                           \
    {                                                                                                                        \
        _Pragma("unroll") for (int rt = 0; rt < 3; ++rt) {                                                                   \
            float a_[6], b_[6];                                                                                              \
            _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                  \
                a_[2 * j] = raw[rt * 3 + j].x; a_[2 * j + 1] = raw[rt * 3 + j].y;                                            \
                b_[2 * j] = raw[(rt + 3) * 3 + j].x; b_[2 * j + 1] = raw[(rt + 3) * 3 + j].y;                                \
            }                                                                                                                \
            swap32x6(a_[0], b_[0], a_[1], b_[1], a_[2], b_[2], a_[3], b_[3], a_[4], b_[4], a_[5], b_[5]);                    \
            _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                  \
                raw[rt * 3 + j] = floatx2{a_[2 * j], a_[2 * j + 1]};                                                         \
                raw[(rt + 3) * 3 + j] = floatx2{b_[2 * j], b_[2 * j + 1]};                                                   \
            }                                                                                                                \
        }                                                                                                                    \
    }
// row pass on the lane's row RL + 3 hh, its six components into V[BUF]
#define WT_ROWPASS_WRITE(RL, BUF)                                                                                            \
    {                                                                                                                        \
        floatx2 w_[6], o_[6];                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) { w_[j] = raw[(RL)*3 + j]; w_[3 + j] = raw[((RL) + 3) * 3 + j]; }      \
        bt6(w_, o_);                                                                                                         \
        float* vw_ = smem + (BUF)*WT_VBUF + vw_base;                                                                         \
        _Pragma("unroll") for (int s = 0; s < 6; ++s) *reinterpret_cast<floatx2*>(vw_ + ((RL)*6 + s) * WT_ZS) = o_[s];       \
    }
// one chunk of the K loop: 18 weight units of { V fragments of the next unit, 8 MFMAs (k step outer, tile group inner: consecutive MFMAs hit different
// accumulators), a slice of the transform of the next chunk (TF), refill of the unit's ring slot NR units ahead }.  The scheduling barriers pin that order.
#define WT_CHUNK(C, TF, GA, LAST, FIRST)                                                                                                      \
    {                                                                                                                        \
        const float* vb = smem + ((C)&1) * WT_VBUF + 3 * h * WT_PS * 2 + v_lane;                                             \
        const int wbuf = ((C)&1) ^ 1;                                                                                        \
        const int cur_off = ubase + (C)*4096;                                                                                \
        /* transform slices: column passes at units TC .. TC + 2, swaps TC + 3, row passes TC + 4 .. TC + 6, then the gathers of the chunk after next.       \
           FIRST (chunk 0 of an item): chunk 1 is gathered HERE, at units 0 .. 2 — not inside the epilogue, where these gathers queued behind the      \
           item's 128 output stores in the CU's address unit and the whole block waited for them — and everything else moves three units back */      \
        constexpr int TC = (FIRST) ? 9 : 6;                                                                                  \
        if constexpr (GA) WT_SETUP()                                                                                         \
        floatx4 vq[2][2];                                                                                                    \
        vq[0][0] = *reinterpret_cast<const floatx4*>(vb);                                                                    \
        vq[0][1] = *reinterpret_cast<const floatx4*>(vb + WT_PS);                                                            \
        _Pragma("unroll") for (int u = 0; u < 18; ++u) {                                                                     \
            const int cu = u & 1, nx = cu ^ 1;                                                                               \
            if (u + 1 < 18) {                                                                                                \
                const int zo = ((((u + 1) / 3) * 6) + (u + 1) % 3) * WT_ZS;                                                  \
                vq[nx][0] = *reinterpret_cast<const floatx4*>(vb + zo);                                                      \
                vq[nx][1] = *reinterpret_cast<const floatx4*>(vb + zo + WT_PS);                                              \
            }                                                                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                  \
                acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[u % NR][j], vq[cu][0][j], ((FIRST) && j == 0) ? kZero4 : acc[u][0], 0, 0, 0); \
                acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[u % NR][j], vq[cu][1][j], ((FIRST) && j == 0) ? kZero4 : acc[u][1], 0, 0, 0); \
                if (j < 3) __builtin_amdgcn_sched_barrier(0);                                                                \
            }                                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
            if constexpr (TF) {                                                                                              \
                if constexpr ((FIRST) && !(DBG & 8)) {                                                                       \
                    if (u < 3) { WT_GATHER_ROW(2 * u) WT_GATHER_ROW(2 * u + 1) }                                             \
                    if (u == 2) { WT_ADVANCE() WT_SETUP() }                                                                  \
                }                                                                                                            \
                if constexpr (!(DBG & 1)) { if (u >= TC && u < TC + 3) WT_COLPASS(u - TC) }                                  \
                if constexpr (!(DBG & 4)) { if (u == TC + 3) WT_SWAPS() }                                                    \
                if constexpr (!(DBG & 2)) { if (u >= TC + 4 && u < TC + 7) WT_ROWPASS_WRITE(u - TC - 4, wbuf) }              \
                if constexpr (GA && !(DBG & 8)) {                                                                            \
                    if constexpr (FIRST) {                                                                                   \
                        if (u >= 16) { WT_GATHER_ROW(3 * (u - 16)) WT_GATHER_ROW(3 * (u - 16) + 1) WT_GATHER_ROW(3 * (u - 16) + 2) } \
                    } else {                                                                                                 \
                        if (u >= 13 && u < 16) { WT_GATHER_ROW(2 * (u - 13)) WT_GATHER_ROW(2 * (u - 13) + 1) }               \
                    }                                                                                                        \
                }                                                                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                                           \
            }                                                                                                                \
            if (!(LAST) || u + NR < 18) {   /* LAST: the ring is not live across the output transform (primed again behind its first stage) */ \
                const int K = u + NR;                                                                                        \
                const int off = K < 18 ? cur_off + unit_rel(K) : cur_off + 4096 + unit_rel(K - 18);                          \
                ring[u % NR] = wt_load16<(DBG & 64) != 0, 0>(rsrc_u, uv_lane, off);                                        \
            }                                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
        }                                                                                                                    \
        if constexpr (GA) WT_ADVANCE()                                                                                       \
        WT_STAMP(st_a)                                                                                                       \
        __syncthreads();                                                                                                     \
        WT_STAMP(st_b)                                                                                                       \
    }

    // prologue: chunk 0 of the first item into V[0], the ring primed
    WT_SETUP()
#pragma unroll
    for (int i = 0; i < 6; ++i) WT_GATHER_ROW(i)
    WT_ADVANCE()
#pragma unroll
    for (int i = 0; i < NR; ++i)
        ring[i] = wt_load16<(DBG & 64) != 0, 0>(rsrc_u, uv_lane, ubase + unit_rel(i));
#pragma unroll
    for (int j = 0; j < 3; ++j) WT_COLPASS(j)
    WT_SWAPS()
#pragma unroll
    for (int r = 0; r < 3; ++r) WT_ROWPASS_WRITE(r, 0)
    WT_SETUP()   // (chunk 1: gathered at the head of chunk 0's K loop)
    __syncthreads();
    WT_STAMP(st_c)
    while (v < total) {
        const int nv = v + nblocks;
        nit = w6_item(nv < total ? nv : v, total, NB, GX, GY, xcd_nb);
        const int nubase = nit.nblk * nch * 4096 + cb * 1024 + 3 * h * zstride;
        WT_CHUNK(0, true, true, false, true)     // (the first MFMA of every accumulator takes C = 0: no zeroing pass between the items)
        for (int c = 1; c < nch - 2; ++c) WT_CHUNK(c, true, true, false, false)
        WT_CHUNK(nch - 2, true, false, false, false)    // transforms the item's last chunk; chunk 0 of the next item is gathered inside the epilogue (no patch registers live across its first stage)
        WT_CHUNK(nch - 1, false, false, true, false)
        // ---------------- output transform + epilogue (this wave finishes tile group h) ----------------
        {
            const int n = it.nblk * 64 + cb * 16 + 4 * g;
            const int tyy = it.gy * 4 + (l15 >> 2), txx = it.gx * 8 + h * 4 + (l15 & 3);
            const bool ok = tyy < TH && txx < TW;
            const unsigned pix = (unsigned)((it.b * p.Ho + 4 * tyy) * p.Wo + 4 * txx);
            unsigned off_out = ok ? (pix * (unsigned)p.out_stride + (unsigned)n) * 4u : WF_OOB;
            unsigned off_res = ok ? (pix * (unsigned)p.res_stride + (unsigned)n) * 4u : WF_OOB;
            if constexpr ((DBG & 128) != 0) {   // measurement twin (garbage results): the same bytes per instruction as 4 pixels x 256 contiguous bytes instead of 16 x 64
                const int tx4 = it.gx * 8 + h * 4 + cb;   // (tile column cb of the group: the four waves of a group cover four different pixels)
                const unsigned pixc = (unsigned)((it.b * p.Ho + 4 * tyy) * p.Wo + 4 * tx4);
                const unsigned nn = (unsigned)(it.nblk * 64 + ((l15 & 3) * 4 + g) * 4);
                off_out = (tyy < TH && tx4 < TW) ? (pixc * (unsigned)p.out_stride + nn) * 4u : WF_OOB;
                off_res = (tyy < TH && tx4 < TW) ? (pixc * (unsigned)p.res_stride + nn) * 4u : WF_OOB;
            }
            constexpr int AUX = NT ? 2 : 0;
            const int rrow = p.Wo * p.res_stride * 4, rpix = p.res_stride * 4;
            floatx4 rv0[2][4], rv1[2][4];
            // The residual tile comes through LDS (buffer_load_dwordx4 ... lds: lane l's 16 bytes land at slot + 16 l; ~45 cycles of the CU's address unit per
            // 1 KB row where the same gather into registers takes 150 - 500, profiles/r04_vmem_issue_probe.txt, r06_notes.md): rows 0 / 1 now, rows 2 / 3 into the
            // same slots once rows 0 / 1 have been read.  (DBG & 256: the register gathers, measurement twin.)
            constexpr bool RDMA = RES && !(DBG & 256);
            char* const rslot = reinterpret_cast<char*>(smem) + WT_XCH_BYTES + wave * 8192;
            if constexpr (RES) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (RDMA)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_res, (__attribute__((address_space(3))) void*)(rslot + (i * 4 + j) * 1024), 16, (int)off_res, i * rrow + j * rpix, 0, AUX);
                        else
                            rv0[i][j] = wt_load16<(DBG & 16) != 0, AUX>(rs_res, (int)off_res, i * rrow + j * rpix);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            floatx4 keep[12];
            float* xs = smem + (wave * 12 * 64 + lane) * 4;
            if (h == 0) wt_stage1<0>(acc, keep, xs); else wt_stage1<1>(acc, keep, xs);
            __builtin_amdgcn_sched_barrier(0);
            // the accumulators are dead: the weight ring of the next item
#pragma unroll
            for (int i = 0; i < NR; ++i)
                ring[i] = wt_load16<(DBG & 64) != 0, 0>(rsrc_u, uv_lane, nubase + unit_rel(i));
            floatx4 bias = {0.f, 0.f, 0.f, 0.f}, fsc = {1.f, 1.f, 1.f, 1.f}, fsh = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias = *reinterpret_cast<const floatx4*>(p.bias + n);
            if (p.film) {
                const float* f = p.film + (size_t)it.b * p.film_bstride;
                fsc = *reinterpret_cast<const floatx4*>(f + n) + 1.0f;
                fsh = *reinterpret_cast<const floatx4*>(f + p.Cout + n);
            }
            WT_STAMP(st_e1)
            __syncthreads();
            floatx4 recv[12];
            const float* xr = smem + ((wave ^ 4) * 12 * 64 + lane) * 4;
#pragma unroll
            for (int k = 0; k < 12; ++k) recv[k] = *reinterpret_cast<const floatx4*>(xr + k * 256);
            if constexpr (RES && !RDMA) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        rv1[i][j] = wt_load16<(DBG & 16) != 0, AUX>(rs_res, (int)off_res, (2 + i) * rrow + j * rpix);
            }
            __syncthreads();   // every wave has read its terms: V[0] may be overwritten (chunk 0 of the next item, below)
            WT_STAMP(st_e2)
            if constexpr (RDMA) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // rows 0 / 1 have landed (two barriers ago; also the ring and the bias / FiLM rows)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) rv0[i][j] = *reinterpret_cast<const floatx4*>(rslot + (i * 4 + j) * 1024 + lane * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and have been read: rows 2 / 3 into the same slots
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_res, (__attribute__((address_space(3))) void*)(rslot + (i * 4 + j) * 1024), 16, (int)off_res, (2 + i) * rrow + j * rpix, 0, AUX);
                __builtin_amdgcn_sched_barrier(0);
            }
            // chunk 0 of the next item (past the last item: every offset out of range) — requested in FRONT of the output stores: the CU's address unit works
            // its queue in order, and behind the 128 stores of an item these gathers (and the transform that waits for them) came 4 - 6k cycles later
            WT_SETUP()
#pragma unroll
            for (int i = 0; i < 6; ++i) WT_GATHER_ROW(i)
            WT_ADVANCE()
            __builtin_amdgcn_sched_barrier(0);
            if (h == 0) wt_finish_rows<0, 0, NT, RES, SILU, (DBG & 32) != 0>(p, keep, recv, rv0, off_out, rs_out, bias, fsc, fsh);
            else wt_finish_rows<1, 0, NT, RES, SILU, (DBG & 32) != 0>(p, keep, recv, rv0, off_out, rs_out, bias, fsc, fsh);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (RDMA) {
                // behind the 8 residual rows 2 / 3: the 18 gathers and the 8 (16: 8-byte twin) stores of the first half
                if constexpr ((DBG & 32) != 0) asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) rv1[i][j] = *reinterpret_cast<const floatx4*>(rslot + (i * 4 + j) * 1024 + lane * 16);
            }
            if (h == 0) wt_finish_rows<0, 1, NT, RES, SILU, (DBG & 32) != 0>(p, keep, recv, rv1, off_out, rs_out, bias, fsc, fsh);
            else wt_finish_rows<1, 1, NT, RES, SILU, (DBG & 32) != 0>(p, keep, recv, rv1, off_out, rs_out, bias, fsc, fsh);
            WT_STAMP(st_e3)
        }
        v = nv; it = nit; ubase = nubase;
        if (v < total) {   // its transform into V[0]
#pragma unroll
            for (int j = 0; j < 3; ++j) WT_COLPASS(j)
            WT_SWAPS()
#pragma unroll
            for (int r = 0; r < 3; ++r) WT_ROWPASS_WRITE(r, 0)
            WT_SETUP()   // (chunk 1: gathered at the head of chunk 0's K loop)
        }
        __syncthreads();
        if constexpr (STAMP) st_n += 1;
        WT_STAMP(st_c)
    }
#undef WT_CHUNK
#undef WT_ROWPASS_WRITE
#undef WT_SWAPS
#undef WT_COLPASS
#undef WT_ADVANCE
#undef WT_GATHER_ROW
#undef WT_SETUP
    if constexpr (STAMP) {
        if (lane == 0 && dbg) {
            unsigned long long* d = dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
            d[0] = st_a; d[1] = st_b; d[2] = st_c; d[3] = __builtin_amdgcn_s_memtime() - st_t0; d[4] = st_n; d[5] = st_e1; d[6] = st_e2; d[7] = st_e3;
        }
    }
#undef WT_STAMP
}

constexpr int WT_NR = 6;

}  // namespace

// Work items of a launch: (image, 4 x 8 tiles, 64-cout block)
long long wino_fused64t_num_items(const ConvParams& p) {
    return (long long)p.B * ((p.Ho / 4 + 3) / 4) * ((p.Wo / 4 + 7) / 8) * (p.Cout / 64);
}

bool wino_fused64t_eligible(const ConvParams& p) {
    if (!wino_fused64_eligible(p)) return false;
    // invalid gather offsets are built from a row part 0x80000000 and a column part 0x40000000: every valid byte offset stays below 0x40000000
    const double lim = 1073741824.0 - 65536.0;
    if (4.0 * p.B * p.Hin * p.Win * (double)p.pix0 >= lim || (p.C1 && 4.0 * p.B * p.Hin * p.Win * (double)p.pix1 >= lim)) return false;
    return true;
}

void wino_fused_t_global_init() {
#define WT_ATTR(...) IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_fused64t_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
#define WT_ATTR4(...) WT_ATTR(__VA_ARGS__, 0); WT_ATTR(__VA_ARGS__, 1); WT_ATTR(__VA_ARGS__, 2); WT_ATTR(__VA_ARGS__, 3)
    WT_ATTR4(WT_NR, false, false, true);
#ifdef IRSDE_PROBES
    WT_ATTR4(WT_NR, true, false, true);
    WT_ATTR4(WT_NR, false, true, true);
    WT_ATTR4(3, false, false, true);
    WT_ATTR(WT_NR, false, false, true, 0, true); WT_ATTR(WT_NR, false, false, true, 1, true);
    WT_ATTR(WT_NR, false, false, true, 2, true); WT_ATTR(WT_NR, false, false, true, 3, true);
    WT_ATTR(WT_NR, false, true, true, 1, true); WT_ATTR(WT_NR, false, true, true, 3, true);
    WT_ATTR(WT_NR, false, false, true, 1, false, 7); WT_ATTR(WT_NR, false, false, true, 3, false, 7);
    WT_ATTR(WT_NR, false, false, true, 1, false, 15); WT_ATTR(WT_NR, false, false, true, 3, false, 15);
    WT_ATTR(WT_NR, false, false, true, 1, false, 8); WT_ATTR(WT_NR, false, false, true, 3, false, 8);
    WT_ATTR4(WT_NR, false, false, false);
    WT_ATTR(WT_NR, false, false, true, 1, false, 16); WT_ATTR(WT_NR, false, false, true, 3, false, 16);
    WT_ATTR(WT_NR, false, false, true, 1, false, 32); WT_ATTR(WT_NR, false, false, true, 3, false, 32);
    WT_ATTR(WT_NR, false, false, true, 1, false, 64); WT_ATTR(WT_NR, false, false, true, 3, false, 64);
    WT_ATTR(WT_NR, false, false, true, 1, true, 128); WT_ATTR(WT_NR, false, false, true, 3, true, 128);
    WT_ATTR(WT_NR, false, false, true, 3, false, 256); WT_ATTR(WT_NR, false, false, true, 2, false, 256);
#endif
#undef WT_ATTR4
#undef WT_ATTR
}

static unsigned long long* g_wt_dbg = nullptr;
static int g_wt_skew = 0;   // tuning: start skew in cycles per phase class (irsde_bench_conv 4700 + k: k x 1000 cycles)
void wino_fused64t_set_skew(int cycles) { g_wt_skew = cycles; }
void wino_fused64t_set_debug(unsigned long long* buf) { g_wt_dbg = buf; }

// variant: 0 production; (PROBES build) 1 weight fragments read zeros, 2 patch gathers read zeros, 3 three instead of six weight units in flight, 5 cycle stamps
// into the buffer of wino_fused64t_set_debug(); + 64: the cout-block-by-XCD item map wherever it is legal (test hook)
void launch_wino_fused64t(const ConvParams& p, const float* Uf, hipStream_t s, int variant) {
    if (!wino_fused64t_eligible(p)) throw HipError("launch_wino_fused64t: layer not eligible");
    if (!Uf) throw HipError("launch_wino_fused64t: fused weights missing");
    const int TH = p.Ho / 4, TW = p.Wo / 4;
    const int GX = (TW + 7) / 8, GY = (TH + 3) / 4, NB = p.Cout / 64;
    const unsigned in0_bytes = (unsigned)((size_t)p.B * p.Hin * p.Win * p.pix0 * 4);
    const unsigned in1_bytes = p.C1 ? (unsigned)((size_t)p.B * p.Hin * p.Win * p.pix1 * 4) : 0u;
    const unsigned uf_bytes = (unsigned)((size_t)36 * p.Cout * (p.C0 + p.C1) * 4);
    const long long G_ = (long long)p.B * GY * GX;
    const int total = (int)(G_ * NB);
    const bool force_xnb = (variant & 64) != 0;
    variant &= 63;
    const bool xnb_legal = NB >= 2 && 8 % NB == 0 && G_ % (8 / NB) == 0;
    // (the same traffic model as the 16-tile kernel's: the weights per round against the patches per cout block)
    const int xcd_nb = (force_xnb ? xnb_legal : (xnb_legal && wino_fused64_xcd_nb(p))) ? 1 : 0;
    const size_t npix_out = (size_t)p.B * p.Ho * p.Wo;
    const size_t ob = npix_out * p.out_stride * 4, rb = p.res ? npix_out * p.res_stride * 4 : 0;
    if (ob >= 0x7fff0000ull || rb >= 0x7fff0000ull) throw HipError("launch_wino_fused64t: output / residual tensor too large for 32-bit buffer offsets");
    // (measurement twins 13 / 14: output stores / residual loads out of range — dropped by the address unit, no traffic; 12: stamps of the NOPATCH twin)
    const unsigned out_bytes = variant == 13 ? 0u : (unsigned)ob, res_bytes = variant == 14 ? 0u : (unsigned)rb;
    const int ncu = device_cu_count();
    // one block per CU; a multiple of 8 so that virtual item id % 8 stays the XCD of the block that runs it
    const dim3 pgrid((unsigned)std::min(total, std::max(8, ncu & ~7)));
    const int epi = (p.silu ? 1 : 0) | (p.res ? 2 : 0);
#define WT_LAUNCH(...) hipLaunchKernelGGL((wino4_fused64t_kernel<__VA_ARGS__>), pgrid, dim3(WT_NT), WT_LDS_BYTES, s, p, Uf, GX, GY, NB, in0_bytes, in1_bytes, uf_bytes, out_bytes, res_bytes, xcd_nb, total, g_wt_dbg, g_wt_skew)
#define WT_LAUNCH_EPI(...)                                  \
    switch (epi) {                                          \
        case 0: WT_LAUNCH(__VA_ARGS__, 0); break;           \
        case 1: WT_LAUNCH(__VA_ARGS__, 1); break;           \
        case 2: WT_LAUNCH(__VA_ARGS__, 2); break;           \
        default: WT_LAUNCH(__VA_ARGS__, 3); break;          \
    }
    switch (variant) {
        case 0: WT_LAUNCH_EPI(WT_NR, false, false, true) break;
#ifdef IRSDE_PROBES
        case 1: WT_LAUNCH_EPI(WT_NR, true, false, true) break;
        case 2: WT_LAUNCH_EPI(WT_NR, false, true, true) break;
        case 3: WT_LAUNCH_EPI(3, false, false, true) break;
        case 12:
            if (epi != 1 && epi != 3) throw HipError("launch_wino_fused64t: measurement twins exist for epilogues 1 / 3");
            if (epi == 1) WT_LAUNCH(WT_NR, false, true, true, 1, true); else WT_LAUNCH(WT_NR, false, true, true, 3, true);
            break;
        case 16:   // the residual tile gathered into registers (r06's first form; epilogues 2 / 3)
            if (epi == 3) WT_LAUNCH(WT_NR, false, false, true, 3, false, 256); else if (epi == 2) WT_LAUNCH(WT_NR, false, false, true, 2, false, 256);
            else throw HipError("launch_wino_fused64t: variant 16 is a residual-layer twin");
            break;
        case 15:   // stamps of the coalesced-epilogue twin (garbage results)
            if (epi != 1 && epi != 3) throw HipError("launch_wino_fused64t: measurement twins exist for epilogues 1 / 3");
            if (epi == 1) WT_LAUNCH(WT_NR, false, false, true, 1, true, 128); else WT_LAUNCH(WT_NR, false, false, true, 3, true, 128);
            break;
        case 5: case 13: case 14:
            switch (epi) {
                case 0: WT_LAUNCH(WT_NR, false, false, true, 0, true); break;
                case 1: WT_LAUNCH(WT_NR, false, false, true, 1, true); break;
                case 2: WT_LAUNCH(WT_NR, false, false, true, 2, true); break;
                default: WT_LAUNCH(WT_NR, false, false, true, 3, true); break;
            }
            break;
        case 4: WT_LAUNCH_EPI(WT_NR, false, false, false) break;   // no non-temporal hint on the residual loads / output stores
        case 6: case 7: case 8:   // measurement twins (garbage results; epilogues 1 / 3): 6 no transform arithmetic / V writes, 7 + no gathers, 8 no gathers only
            if (epi != 1 && epi != 3) throw HipError("launch_wino_fused64t: measurement twins exist for epilogues 1 / 3");
            if (variant == 6) { if (epi == 1) WT_LAUNCH(WT_NR, false, false, true, 1, false, 7); else WT_LAUNCH(WT_NR, false, false, true, 3, false, 7); }
            if (variant == 7) { if (epi == 1) WT_LAUNCH(WT_NR, false, false, true, 1, false, 15); else WT_LAUNCH(WT_NR, false, false, true, 3, false, 15); }
            if (variant == 8) { if (epi == 1) WT_LAUNCH(WT_NR, false, false, true, 1, false, 8); else WT_LAUNCH(WT_NR, false, false, true, 3, false, 8); }
            break;
        case 9: case 10: case 11:   // 8-byte twins (epilogues 1 / 3): 9 residual loads, 10 output stores, 11 weight units
            if (epi != 1 && epi != 3) throw HipError("launch_wino_fused64t: measurement twins exist for epilogues 1 / 3");
            if (variant == 9) { if (epi == 1) WT_LAUNCH(WT_NR, false, false, true, 1, false, 16); else WT_LAUNCH(WT_NR, false, false, true, 3, false, 16); }
            if (variant == 10) { if (epi == 1) WT_LAUNCH(WT_NR, false, false, true, 1, false, 32); else WT_LAUNCH(WT_NR, false, false, true, 3, false, 32); }
            if (variant == 11) { if (epi == 1) WT_LAUNCH(WT_NR, false, false, true, 1, false, 64); else WT_LAUNCH(WT_NR, false, false, true, 3, false, 64); }
            break;
#endif
        default: throw HipError("launch_wino_fused64t: bad variant (the measurement twins need a make PROBES=1 build)");
    }
#undef WT_LAUNCH_EPI
#undef WT_LAUNCH
    IRSDE_HIP_CHECK(hipGetLastError());
}

}  // namespace irsde
