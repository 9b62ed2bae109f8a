// fp32-equivalent GEMM on the bf16 matrix pipe (r03 prototype; SURVEY.md 7 lists the route, VERDICT r02 #3 asks for it).
//
// gfx950 has no tf32 / xf32 MFMA: an exact-f32 product costs v_mfma_f32_32x32x2_f32 = 1/16 of the bf16 rate.  Split every
// f32 operand into NPL bf16 pieces (round-to-nearest-even, the residual is exact in f32):
//     a = a0 + a1 + a2 + e,   |a1| <= 2^-9 |a|,  |a2| <= 2^-18 |a|,  |e| <= 2^-27 |a|          (8 + 8 + 8 significand bits)
// and a product of two pieces (8 x 8 bits) is exact in the MFMA's f32 accumulator, so
//     NPL = 3:  a b ~= a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)      6 MFMAs, dropped terms <= 3 * 2^-27 |a b|
//     NPL = 2:  a b ~= a0 b0 + (a0 b1 + a1 b0)                                3 MFMAs, dropped terms <= 3 * 2^-18 |a b|
// i.e. NPL = 3 carries the operands' full 24 bits (the dropped part is below f32's own 2^-24 rounding of each
// accumulation) at 6/16 of the f32-MFMA cycles; NPL = 2 is a 16-bit-significand mode at 3/16.  Accumulation is f32 in both.
// This is NOT the bit-exact fmaf chain of the native kernels (different rounding points), so it is an opt-in engine mode
// (IRSDE_FLAG_SPLIT_BF16 / IRSDE_FLAG_SPLIT_BF16X2) with its own measured error table (profiles/r03_split_gemm_*.txt).
//
// Where: the component GEMMs of the three-launch Winograd layers, M_z[t][n] = sum_c V_z[t][c] U_z[n][c] (reference call site:
// Block.proj, module_util.py:108-122).  wino.hip writes V already split (NPL planes of bf16: 2 NPL bytes per element instead
// of 4), U is split once at weight load, so the GEMM kernel below is a pure bf16 GEMM over NPL + NPL operand planes:
//   block 128 x 128 outputs, 4 waves of 64 x 64 (2 x 2 tiles of v_mfma_f32_32x32x16_bf16), K-step 32;
//   per K-step and plane one 128 x 32 bf16 slice of A and of B in LDS (80-byte rows: conflict-free ds_read_b128, the layout of
//   the bf16 kernels in conv_igemm.hip), double-buffered, global -> register -> LDS staging one K-step ahead;
//   per 16-k sub-step a wave reads NPL A + NPL B fragments per tile row / column and issues NPROD MFMAs per output tile:
//   one LDS fragment read per MFMA (x3) instead of two for the plain bf16 kernel, which is LDS-read-bound.
//   Like gemm_zloop_kernel a block walks n_inner components of its (row tile, column tile) as one pipelined K loop and
//   writes finished accumulators straight from registers (buffer stores, rows past M dropped by the descriptor).
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace irsde {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SG_BK = 32;            // k per K-step and plane
constexpr int SG_ROWB = 80;          // LDS bytes per row: 32 bf16 + 16 B pad
constexpr int SG_PLANE = 128 * SG_ROWB;

template <int NPL>
__global__ __launch_bounds__(256, 1) void gemm_split_kernel(const SplitGemmArgs g) {
    constexpr int NPROD = NPL == 3 ? 6 : 3;
    // smallest terms first; consecutive MFMAs of one product hit the four different accumulators of the wave
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0};
    constexpr int PB[6] = {2, 1, 0, 1, 0, 0};
    constexpr int P0 = 6 - NPROD;  // NPL = 2 uses the last three pairs: (0,1), (1,0), (0,0)
    constexpr int A_BYTES = NPL * SG_PLANE;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* As = lds;
    char* Bs = lds + 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    int wgid;
    {
        const int orig = blockIdx.x, nwg = gridDim.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int mblk = wgid / g.nblk_n, nblk = wgid - mblk * g.nblk_n;
    const int m0 = mblk * 128, n0 = nblk * 128;
    const int plane0 = blockIdx.y * g.n_inner;  // first component of this block
    const int nk = g.K / SG_BK;
    const int steps = g.n_inner * nk;

    const int piece = tid & 3, row0 = tid >> 2;  // 16-byte piece of a 64-byte row slice; rows row0 and row0 + 64
    const char* arow[2];
    const char* brow[2];
    int st_i = 0;
    auto set_tile_ptrs = [&]() {
        const long long z = plane0 + st_i;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            int m = m0 + row0 + ps * 64;
            m = m < g.M ? m : g.M - 1;   // clamped rows feed accumulator rows that are never stored
            arow[ps] = reinterpret_cast<const char*>(g.a + z * g.pA + (long long)m * g.lda) + piece * 16;
            int n = n0 + row0 + ps * 64;
            n = n < g.N ? n : g.N - 1;
            brow[ps] = reinterpret_cast<const char*>(g.b + z * g.pB + (long long)n * g.K) + piece * 16;
        }
    };
    set_tile_ptrs();
    uintx4 rs[4 * NPL];
    int kk = 0;
    const long long plA2 = g.plA * 2, plB2 = g.plB * 2;
    auto load_all = [&]() {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                rs[ps * NPL + p] = *reinterpret_cast<const uintx4*>(arow[ps] + p * plA2 + kk * 2);
                rs[2 * NPL + ps * NPL + p] = *reinterpret_cast<const uintx4*>(brow[ps] + p * plB2 + kk * 2);
            }
    };
    auto store_all = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                *reinterpret_cast<uintx4*>(As + buf * A_BYTES + p * SG_PLANE + (row0 + ps * 64) * SG_ROWB + piece * 16) = rs[ps * NPL + p];
                *reinterpret_cast<uintx4*>(Bs + buf * A_BYTES + p * SG_PLANE + (row0 + ps * 64) * SG_ROWB + piece * 16) = rs[2 * NPL + ps * NPL + p];
            }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_all();
    store_all(0);
    __syncthreads();

    const int wm_s = __builtin_amdgcn_readfirstlane(wm), wn_s = __builtin_amdgcn_readfirstlane(wn);
    unsigned o_voff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) o_voff[r] = (unsigned)(((r & 3) + 8 * (r >> 2) + 4 * h) * g.ldc + l31) * 4u;
    auto flush = [&](int fi) {
        const int rowb = m0 + wm_s * 64, colb = n0 + wn_s * 64;
        float* ob = g.out + (long long)(plane0 + fi) * g.pO + (long long)rowb * g.ldc;
        const int rows = g.M - rowb;
        const unsigned nrec = rows <= 0 ? 0u : (unsigned)(rows < 64 ? rows : 64) * (unsigned)g.ldc * 4u;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ob, 0, nrec, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int colu = colb + j * 32;
                if (colu + l31 < g.N) {
                    const int soff = (i * 32 * g.ldc + colu) * 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r];  // (bit_cast straight on the vector element stored element 0 sixteen times: hipcc 7.2)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, (int)o_voff[r], soff, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
    };
    int kdone = 0, cu_i = 0, fl_i = 0;
    bool pending = false;
    for (int st = 0; st < steps; ++st) {
        const int buf = st & 1;
        if (st + 1 < steps) {  // advance the staging position (the last step re-stages its own operands into the dead buffer)
            kk += SG_BK;
            if (kk == g.K) {
                kk = 0;
                ++st_i;
                set_tile_ptrs();
            }
        }
        if (pending) {
            flush(fl_i);
            pending = false;
        }
        load_all();
        const char* a = As + buf * A_BYTES + (wm * 64 + l31) * SG_ROWB + h * 16;
        const char* b = Bs + buf * A_BYTES + (wn * 64 + l31) * SG_ROWB + h * 16;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            bf16x8 fa[NPL][2], fb[NPL][2];
#pragma unroll
            for (int p = 0; p < NPL; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[p][i] = *reinterpret_cast<const bf16x8*>(a + p * SG_PLANE + i * 32 * SG_ROWB + sb * 32);
                    fb[p][i] = *reinterpret_cast<const bf16x8*>(b + p * SG_PLANE + i * 32 * SG_ROWB + sb * 32);
                }
#pragma unroll
            for (int pr = P0; pr < 6; ++pr)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pr]][i], fb[PB[pr]][j], acc[i][j], 0, 0, 0);
        }
        store_all(buf ^ 1);
        if (++kdone == nk) {
            kdone = 0;
            fl_i = cu_i++;
            pending = true;
        }
        __syncthreads();
    }
    flush(fl_i);
}

// ---------------------------------------------------------------------------------------------------------------
// The engine's kernel: two planes, pair-interleaved operands, LDS-DMA, 256 x 256 tiles.  Its predecessor (v2, removed: plane-major
// operands, register-staged ds_write_b128, same tiles; numbers in profiles/r03_split_gemm_notes.md) showed with ablation twins that
// without its global loads the K loop takes 69 % of the time and without MFMAs 72 %: the 64 wave-loads of a K-step (1 KB each, but
// 16 separate 64-byte row segments) cost as much as the 96 MFMAs they feed, and the 64 ds_write_b128 (13 cycles each) and their
// VGPR staging come on top.  Here:
//   * operand layout [row][k / 32][plane][32 k]: the hi and lo pieces of a row's 32-k block are ONE 128-byte line, so a wave-load
//     of 1 KB covers 8 rows x 128 B = 8 full lines instead of 16 half lines (wino_input / split_planes write this layout);
//   * global_load_lds_dwordx4: global -> LDS without VGPRs or ds_write; the LDS image of a stage is lane-linear (1 KB per
//     wave-load = 8 rows), the XOR swizzle of the 16-byte pieces (by (row >> 1) & 7: conflict-free ds_read_b128 lane groups on
//     128-byte rows) is applied to the per-lane SOURCE address and to the fragment reads (cdna_hip_programming.md rule 21);
//   * the loads of K-step t+1 are issued right after the barrier that ends step t-1 and have the whole MFMA phase of step t to
//     land; __syncthreads() (vmcnt(0) + s_barrier) ends the step.
// Block tile 256 x 256, 8 waves of 128 x 64, K-step 32, two LDS stages of 64 KB.
// ---------------------------------------------------------------------------------------------------------------
#define IRSDE_GLDS16(GPTR, LPTR) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(GPTR), (__attribute__((address_space(3))) void*)(LPTR), 16, 0, 0)

// ABL: measurement twins (irsde_bench_conv 473 / 475 / 476): 1 = no global loads in the K loop, 3 = no MFMAs, 4 = no output stores;
// 5 = output stores with the non-temporal hint (IRSDE_SPLIT_NT=1 under IRSDE_TUNING=1 makes it the fp16 instance).  Measured, not used:
// the GEMMs +-1 %, but wino_output, which reads M right after, 5 - 10 % slower (M no longer waits in the L2 / Infinity Cache): 28.66 vs 28.72 ms
// F16: the two pieces are IEEE binary16 (11 significand bits each: hi + lo carry 22+ of f32's 24 bits, products exact in f32) on
// v_mfma_f32_32x32x16_f16 — fp32-equivalent per product at the same three MFMAs; the writers scale the operands by powers of two
// into fp16's range and g.out_scale undoes it here.
template <int ABL = 0, bool F16 = false>
__global__ __launch_bounds__(512, 2) void gemm_split2i_kernel(const SplitGemmArgs g) {
    using frag_t = typename std::conditional<F16, f16x8, bf16x8>::type;
    constexpr int TM = 4, TN = 2, WN = 4, BM = 256, BN = 256;
    constexpr int A_STAGE = BM * 128, STAGE = (BM + BN) * 128;   // bytes
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    // XCD-aware work assignment (workgroup id w runs on XCD w % 8): a unit = (component, row tile) with all its column tiles;
    // XCD x owns the contiguous unit range [x U / 8, (x + 1) U / 8).  An A row tile is then read through ONE L2 (its column
    // tiles run side by side on that XCD and walk K in step), a component's B through at most two; with the plain
    // (tile, component) grid every A tile crossed two L2s and every B tile four, and the kernel sat on the HBM floor of
    // that traffic (v3 without MFMAs = 0.19 of its 0.25 ms, profiles/r03_split_gemm_notes.md).
    const int mtiles = (g.M + BM - 1) / BM;
    const int units = mtiles * g.n_inner;          // (n_inner carries the component count here)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int ulo = (int)((long long)xcd * units / 8), uhi = (int)((long long)(xcd + 1) * units / 8);
    const int unit = ulo + jx / g.nblk_n;
    if (unit >= uhi) return;
    const int nblk = jx % g.nblk_n;
    const int plane0 = unit / mtiles, mblk = unit - plane0 * mtiles;
    const int m0 = mblk * BM, n0 = nblk * BN;
    const int nk = g.K / SG_BK;
    const int steps = nk;
    const unsigned rowb = (unsigned)nk * 128u;   // bytes per operand row: K / 32 blocks of (hi 64 B | lo 64 B)

    // staging: wave-load (pass ps) = LDS chunk c = ps * 8 + wave = rows 8c .. 8c+7 of the A (B) tile; lane = (row 8c + lane / 8, slot lane % 8)
    // fetches the row's piece slot ^ ((row >> 1) & 7)
    unsigned a_voff[4], b_voff[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int row = (ps * 8 + wave) * 8 + (lane >> 3);
        const unsigned pc = (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
        int m = m0 + row;
        m = m < g.M ? m : g.M - 1;
        a_voff[ps] = (unsigned)m * rowb + pc;
        int n = n0 + row;
        n = n < g.N ? n : g.N - 1;
        b_voff[ps] = (unsigned)n * rowb + pc;
    }
    const char* abase = reinterpret_cast<const char*>(g.a);
    const char* bbase = reinterpret_cast<const char*>(g.b);
    const long long pA2 = g.pA * 4, pB2 = g.pB * 4;   // bytes between components: 2 planes x 2 bytes per element
    int kb = 0;
    const char* acomp = abase + (long long)plane0 * pA2;
    const char* bcomp = bbase + (long long)plane0 * pB2;
    auto issue_loads = [&](int buf) {
        char* la = lds + buf * STAGE + wave * 1024;
        const char* ga = acomp + kb * 128;
        const char* gb = bcomp + kb * 128;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) IRSDE_GLDS16(ga + a_voff[ps], la + ps * 8192);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) IRSDE_GLDS16(gb + b_voff[ps], la + A_STAGE + ps * 8192);
    };

    auto issue_loads_part = [&](int buf, int part) {   // the 8 loads of a stage in three groups: A 0-2 | A 3, B 0-1 | B 2-3
        char* la = lds + buf * STAGE + wave * 1024;
        const char* ga = acomp + kb * 128;
        const char* gb = bcomp + kb * 128;
        if (part == 0) {
            IRSDE_GLDS16(ga + a_voff[0], la);
            IRSDE_GLDS16(ga + a_voff[1], la + 8192);
            IRSDE_GLDS16(ga + a_voff[2], la + 2 * 8192);
        } else if (part == 1) {
            IRSDE_GLDS16(ga + a_voff[3], la + 3 * 8192);
            IRSDE_GLDS16(gb + b_voff[0], la + A_STAGE);
            IRSDE_GLDS16(gb + b_voff[1], la + A_STAGE + 8192);
        } else {
            IRSDE_GLDS16(gb + b_voff[2], la + A_STAGE + 2 * 8192);
            IRSDE_GLDS16(gb + b_voff[3], la + A_STAGE + 3 * 8192);
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue_loads(0);
    __syncthreads();

    const int wm_s = wm, wn_s = wn;
    unsigned o_voff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) o_voff[r] = (unsigned)(((r & 3) + 8 * (r >> 2) + 4 * h) * g.ldc + l31) * 4u;
    auto flush = [&](int fi) {
        const int rowb_ = m0 + wm_s * TM * 32, colb = n0 + wn_s * TN * 32;
        float* ob = g.out + (long long)(plane0 + fi) * g.pO + (long long)rowb_ * g.ldc;
        const int rows = g.M - rowb_;
        const unsigned nrec = rows <= 0 ? 0u : (unsigned)(rows < TM * 32 ? rows : TM * 32) * (unsigned)g.ldc * 4u;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ob, 0, nrec, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int colu = colb + j * 32;
                if (colu + l31 < g.N && (ABL != 4 || acc[i][j][0] == 1.2345e30f)) {
                    const int soff = (i * 32 * g.ldc + colu) * 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = F16 ? acc[i][j][r] * g.out_scale : acc[i][j][r];   // (exact: a power of two)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, (int)o_voff[r], soff, ABL == 5 ? 2 : 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
    };
    // fragment offsets inside a 128-byte row: piece = plane * 4 + sb * 2 + h, swizzled by (l31 >> 1) & 7 (tile bases are multiples of 32 rows)
    const int swz = (l31 >> 1) & 7;
    int fr_off[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) fr_off[p][sb] = l31 * 128 + (((p * 4 + sb * 2 + h) ^ swz) * 16);
    for (int st = 0; st < steps; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < steps;
        if (more) ++kb;   // (stage buf^1 was last read in step st-1: every wave is past that step's barrier when the loads below are issued)
        const char* a = lds + buf * STAGE + wm * TM * 32 * 128;
        const char* b = lds + buf * STAGE + A_STAGE + wn * TN * 32 * 128;
        // Pinned order (r03, A/B in profiles/r03_split_gemm_notes.md): fragments of both 16-k sub-steps up front (the second set lands
        // while the first multiplies), the next stage's LDS-DMA loads spread behind the first MFMA groups instead of one burst
        frag_t fa[2][2][TM], fb[2][2][TN];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[sb][p][i] = *reinterpret_cast<const frag_t*>(a + i * 32 * 128 + fr_off[p][sb]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[sb][p][j] = *reinterpret_cast<const frag_t*>(b + j * 32 * 128 + fr_off[p][sb]);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (ABL != 3) {
                            if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sb][pr == 1 ? 1 : 0][i], fb[sb][pr == 0 ? 1 : 0][j], acc[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sb][pr == 1 ? 1 : 0][i], fb[sb][pr == 0 ? 1 : 0][j], acc[i][j], 0, 0, 0);
                        }
                        else acc[i][j][0] += (float)fa[sb][pr == 1 ? 1 : 0][i][0] * (float)fb[sb][pr == 0 ? 1 : 0][j][0];
                if (sb == 0 && ABL != 1 && more) issue_loads_part(buf ^ 1, pr);   // 3 + 3 + 2 loads behind the first three MFMA groups
                __builtin_amdgcn_sched_barrier(0);
            }
        __syncthreads();
    }
    flush(0);
}
#undef IRSDE_GLDS16

// f32 -> pair-interleaved hi / lo pieces: element (row, k) of a [rows][K] matrix -> out[(row * K / 32 + k / 32) * 64 + plane * 32 + k % 32].
// F16: the pieces are IEEE binary16 of in * scale (scale = a power of two that brings the tensor into fp16's range).
template <bool F16>
__global__ __launch_bounds__(256) void split_pairs_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, const size_t n, const int K,
                                                          const float scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t row = i / K;
    const int k = (int)(i - row * K);
    float r = in[i];
    const size_t base = (row * (size_t)(K / 32) + k / 32) * 64 + (k & 31);
    if constexpr (F16) {
        r *= scale;
        const _Float16 hb = (_Float16)r;
        out[base] = __builtin_bit_cast(unsigned short, hb);
        r -= (float)hb;
        const _Float16 lb = (_Float16)r;
        out[base + 32] = __builtin_bit_cast(unsigned short, lb);
    } else {
        const __bf16 hb = (__bf16)r;
        out[base] = __builtin_bit_cast(unsigned short, hb);
        r -= (float)hb;
        const __bf16 lb = (__bf16)r;
        out[base + 32] = __builtin_bit_cast(unsigned short, lb);
    }
}

// f32 -> NPL bf16 planes (round to nearest even; every residual is exact in f32); F16: IEEE fp16 pieces of in * scale
template <int NPL, bool F16 = false>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, const size_t n,
                                                           const size_t plane, const float scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = F16 ? in[i] * scale : in[i];
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        if constexpr (F16) {
            const _Float16 hb = (_Float16)r;
            out[p * plane + i] = __builtin_bit_cast(unsigned short, hb);
            r -= (float)hb;
        } else {
            const __bf16 hb = (__bf16)r;
            out[p * plane + i] = __builtin_bit_cast(unsigned short, hb);
            r -= (float)hb;
        }
    }
}

}  // namespace

void gemm_split_global_init() {
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split2i_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split2i_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split2i_kernel<5, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split2i_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split2i_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split2i_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}

// components per block: the largest divisor of `ncomp` that still leaves >= 512 blocks (2 per CU)
int gemm_split_inner(int M, int N, int ncomp) {
    const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
    for (int d = ncomp; d >= 1; --d)
        if (ncomp % d == 0 && tiles * (ncomp / d) >= 512) return d;
    return 1;
}

// pair-interleaved operands (g.a / g.b in the [row][k/32][plane][32] layout, g.pA / g.pB = elements per component and plane)
void launch_gemm_split_pairs(const SplitGemmArgs& a, int ncomp, hipStream_t s, int abl, bool f16) {
    if (a.K % SG_BK) throw HipError("gemm_split_pairs: K must be a multiple of 32");
    if ((unsigned long long)a.M * a.K * 4ull >= 0xffffffffull || (unsigned long long)a.N * a.K * 4ull >= 0xffffffffull ||
        (size_t)a.M * a.ldc * 4 >= 0xffffffffull)
        throw HipError("gemm_split_pairs: a component exceeds the 32-bit offset range");
    SplitGemmArgs g = a;
    g.nblk_n = (a.N + 255) / 256;
    g.n_inner = ncomp;   // this kernel: one (component, row tile, column tile) per block; n_inner carries the component count
    const int units = ((a.M + 255) / 256) * ncomp;
    int per_xcd = 0;
    for (int x = 0; x < 8; ++x) per_xcd = std::max(per_xcd, (int)((long long)(x + 1) * units / 8) - (int)((long long)x * units / 8));
    const dim3 grid((unsigned)(8 * per_xcd * g.nblk_n));
    const size_t lds = (size_t)2 * 512 * 128;
    if (f16) {
        if (abl != 0) throw HipError("gemm_split_pairs: the ablation twins exist for the bf16 kernel only");
        static const bool nt = tuning_env_int("IRSDE_SPLIT_NT", 0) != 0;
        if (nt) hipLaunchKernelGGL((gemm_split2i_kernel<5, true>), grid, dim3(512), lds, s, g);
        else hipLaunchKernelGGL((gemm_split2i_kernel<0, true>), grid, dim3(512), lds, s, g);
        IRSDE_HIP_CHECK(hipGetLastError());
        return;
    }
    switch (abl) {
        case 0: hipLaunchKernelGGL(gemm_split2i_kernel<0>, grid, dim3(512), lds, s, g); break;
        case 1: hipLaunchKernelGGL(gemm_split2i_kernel<1>, grid, dim3(512), lds, s, g); break;
        case 3: hipLaunchKernelGGL(gemm_split2i_kernel<3>, grid, dim3(512), lds, s, g); break;
        case 4: hipLaunchKernelGGL(gemm_split2i_kernel<4>, grid, dim3(512), lds, s, g); break;
        default: throw HipError("gemm_split_pairs: bad ablation variant");
    }
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_split_pairs(const float* in, unsigned short* out, size_t rows, int K, hipStream_t s, bool f16, float scale) {
    if (K % 32) throw HipError("split_pairs: K must be a multiple of 32");
    const size_t n = rows * (size_t)K;
    if (f16) hipLaunchKernelGGL(split_pairs_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n, K, scale);
    else hipLaunchKernelGGL(split_pairs_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n, K, 1.0f);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_gemm_split(const SplitGemmArgs& a, int nplanes, int ncomp, hipStream_t s) {
    if (nplanes != 2 && nplanes != 3) throw HipError("gemm_split: 2 or 3 planes");
    if (a.K % SG_BK || a.n_inner < 1 || ncomp % a.n_inner) throw HipError("gemm_split: K must be a multiple of 32, n_inner a divisor of the component count");
    if ((size_t)a.M * a.ldc * 4 >= 0xffffffffull) throw HipError("gemm_split: output plane exceeds the 32-bit buffer range");
    SplitGemmArgs g = a;
    g.nblk_n = (a.N + 127) / 128;
    const dim3 grid((unsigned)(((a.M + 127) / 128) * g.nblk_n), (unsigned)(ncomp / a.n_inner));
    const size_t lds = (size_t)4 * nplanes * SG_PLANE;
    if (nplanes == 3) hipLaunchKernelGGL(gemm_split_kernel<3>, grid, dim3(256), lds, s, g);
    else hipLaunchKernelGGL(gemm_split_kernel<2>, grid, dim3(256), lds, s, g);
    IRSDE_HIP_CHECK(hipGetLastError());
}

void launch_split_planes(const float* in, unsigned short* out, size_t n, size_t plane, int nplanes, hipStream_t s, bool f16, float scale) {
    const dim3 grid((unsigned)((n + 255) / 256));
    if (f16 && nplanes != 2) throw HipError("split_planes: fp16 pieces come in pairs");
    if (f16) hipLaunchKernelGGL((split_planes_kernel<2, true>), grid, dim3(256), 0, s, in, out, n, plane, scale);
    else if (nplanes == 3) hipLaunchKernelGGL((split_planes_kernel<3, false>), grid, dim3(256), 0, s, in, out, n, plane, 1.0f);
    else if (nplanes == 2) hipLaunchKernelGGL((split_planes_kernel<2, false>), grid, dim3(256), 0, s, in, out, n, plane, 1.0f);
    else throw HipError("split_planes: 2 or 3 planes");
    IRSDE_HIP_CHECK(hipGetLastError());
}

}  // namespace irsde
