// Implicit-GEMM NHWC convolution for gfx950 on the matrix pipe, two arithmetic modes:
//   fp32 : v_mfma_f32_32x32x2_f32 — D = A(32x2) * B(2x32) + C, one f32 A and one f32 B operand per lane,
//          bit-for-bit an fmaf chain; 64 cycles per instruction per SIMD = the fp32 peak of 157 TFLOP/s.
//   bf16 : v_mfma_f32_32x32x16_bf16 — activations (fp32 in HBM) are rounded to bf16 (RNE) while they are
//          staged into LDS, weights come pre-rounded ([Cout][KH][KW][Cin] bf16), accumulation stays fp32;
//          32 cycles per 16-deep instruction = 16x the fp32 rate, so the kernel turns L2/HBM-bound.
//
// Replaces every nn.Conv2d of the reference score network (3x3, 1x1, 4x4 s2, 7x7 and the
// nearest-upsample + 3x3 pair): reference call sites module_util.py:93-105 (Upsample/Downsample/
// default_conv), :108-122 (Block), :150-161 (to_qkv / to_out), DenoisingUNet_arch.py:27,76.
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*(C0+C1).  Both operands are
// K-contiguous in HBM (NHWC activations, [Cout][KH][KW][Cin] weights), so both LDS tiles are
// [rows][32 k] with a 16-byte pad per row (row stride 144 B fp32 / 80 B bf16 = an odd number of 16-B
// slots => the 16-lane groups of ds_read_b128 hit 16 distinct slots: conflict-free).  A lane reads 16
// bytes of consecutive k with one ds_read_b128 (fp32: 4 k -> 4 MFMAs; bf16: 8 k -> one MFMA); lane half h
// owns the second 16 bytes of every 32-byte group, the same assignment for A and B, so the k-order
// inside the MFMA chain is a permutation (legal: sum over k).
//
// K loop: taps outer / 32-channel chunks inner.  The per-row pixel offset of a tap is computed once
// per tap (not per K-step); a K-step then costs one 64-bit mad + one load per staged row.
// global->register->LDS staging is split: the loads of K-step t+1 are issued before the MFMAs of
// step t and written to the other LDS buffer after them; one barrier per K-step.  Two blocks share a
// CU and cover each other's staging gaps.  (Interleaved schedules, wave priorities, LDS-DMA staging and
// AGPR accumulators were all measured and dropped: profiles/r01_conv_tuning_notes.md.)
//
// Epilogue: the accumulator tile is transposed through LDS (the A/B buffers are dead by then) so
// that every lane handles 4 consecutive output channels: bias / FiLM / residual / output move as
// 16-byte accesses, 512 contiguous bytes per output pixel row of a 128-wide tile.
#include "common.h"
#include <type_traits>
#include <mutex>
#include <vector>

namespace irsde {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

// The 16-bit operand type of the reduced-precision modes: bf16 (IRSDE_FLAG_BF16) or IEEE fp16 (IRSDE_FLAG_FP16) — the same
// 32x32x16 MFMA shape, the same fragment layout, fp32 accumulation; only the rounding (8 vs 11 significand bits) differs.
template <bool F16>
struct Op16 {
    using x8 = bf16x8;
    using x4 = bf16x4;
    static __device__ __forceinline__ floatx16 mfma(x8 a, x8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct Op16<true> {
    using x8 = f16x8;
    using x4 = f16x4;
    static __device__ __forceinline__ floatx16 mfma(x8 a, x8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

constexpr int BK = 32;  // k per K-step (channels of one tap)

template <int BM, int BN, int WAVES_M, int WAVES_N, bool BF16, bool ABF = false, int PL = 1>
struct Cfg {
    static constexpr int NT = 64 * WAVES_M * WAVES_N;
    static constexpr int ROW_BYTES = BK * (BF16 ? 2 : 4) + 16;  // LDS tile row: 32 k + 16-B pad
    static constexpr int TM = BM / WAVES_M / 32;
    static constexpr int TN = BN / WAVES_N / 32;
    // A (activations): fp32 in HBM = 8 x 16-B loads per row of 32 k; bf16 in HBM (ABF, IRSDE_FLAG_BF16_ACT) = 4
    static constexpr int A_CHUNKS = ABF ? 4 : 8;
    static constexpr int A_ROWS = NT / A_CHUNKS;
    static constexpr int A_PASSES = BM / A_ROWS;
    // B (weights): fp32 8 x 16 B per row, bf16 4 x 16 B per row
    static constexpr int B_CHUNKS = BF16 ? 4 : 8;
    static constexpr int B_ROWS = NT / B_CHUNKS;
    static constexpr int B_PASSES = (BN + B_ROWS - 1) / B_ROWS;
    static constexpr int LDS_C = BN + 4;  // epilogue tile row stride (floats): 16-B aligned, odd number of slots
    static constexpr int MAIN_BYTES = 2 * PL * (BM + BN) * ROW_BYTES;   // PL = 2: hi and lo operand planes (PAIR kernels)
    // the epilogue transposes EPI_ROWS tile rows per pass (bf16: smaller passes keep 3+ blocks per CU resident)
    static constexpr int EPI_ROWS = BF16 ? BM / WAVES_M : (BM > 128 ? 128 : BM);
    static constexpr int EPI_BYTES = EPI_ROWS * LDS_C * 4;
    static_assert((BM / WAVES_M) <= EPI_ROWS && EPI_ROWS % (BM / WAVES_M) == 0, "wave rows must tile an epilogue pass");
    static constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
    static_assert(BM % A_ROWS == 0, "tile/pass mismatch");
    static_assert(BN % B_ROWS == 0 || BN < B_ROWS, "tile/pass mismatch");
};

__device__ __forceinline__ float silu_f(float v) { return silu_hw(v); }

// compile-time loop: the index is a constant already in the front end, so register arrays indexed by it are promoted to
// registers no matter when the optimiser unrolls (a `#pragma unroll` loop over a staging array inside a lambda was
// seen to leave the array in scratch memory, with a vmcnt(0) after every load)
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// INSCALE: per-(batch, input channel) scale applied while staging (NAFNet SCA).  A template parameter, not a runtime
// test: a branch inside the staging code makes the compiler wait for every load where the paths join, which
// serialises the loads of a K-step and pulls the vmcnt(0) in front of the MFMAs.
// BUFA (f32 kernels): activations, weights and the SCA scale are read through buffer descriptors — per-thread 32-bit
// offsets that change only with the tap / concat source, the K-step's channel offset in an SGPR — so a K-step's loads cost
// no vector instruction.  On gfx950 the f32 MFMA shares the SIMD's vector ALU: vector instructions between MFMAs are paid
// in full (4+ cycles each) plus ~16 cycles per MFMA they follow (tools/probe/mfma_valu_samewave.hip), and the pointer
// arithmetic of the generic path was ~100 of them per K-step.  Needs every operand tensor < 2 GiB (launch_cfg checks).
// PAIR (r03, IRSDE_FLAG_SPLIT_BF16X2 / _F16X2): fp32 activations and fp32-derived weights, but every operand is split into a 16-bit
// hi + lo pair (hi = round(x), lo = round(x - hi): exact residual) and the three cross products hi.hi + hi.lo + lo.hi run on the
// 16-bit MFMA — the arithmetic of gemm_split.hip for the direct (implicit-GEMM) layers.  Activations are split while they are
// staged (two LDS planes per operand), the weights come pre-split (p.w_pair: two planes); fp16 pieces carry 22+ significand
// bits (fp32-equivalent), their weight plane is scaled by a power of two that p.pair_scale undoes on the accumulators.
template <int BM, int BN, int WAVES_M, int WAVES_N, int MIN_WAVES_PER_SIMD, bool BF16, bool INSCALE, bool ABF = false, bool F16 = false,
          bool BUFA = false, bool PAIR = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MIN_WAVES_PER_SIMD) void conv_igemm_kernel(
    const ConvParams pin, const int nblk_n, const int M, const int nk_total) {
    using C = Cfg<BM, BN, WAVES_M, WAVES_N, BF16, ABF, PAIR ? 2 : 1>;
    using H16 = Op16<F16>;
    static_assert(!PAIR || (BF16 && !ABF && !BUFA), "PAIR: 16-bit MFMA on fp32 activation storage, generic staging");
    constexpr int PL = PAIR ? 2 : 1;
    static_assert(!ABF || (BF16 && !INSCALE), "bf16 activation storage belongs to the bf16-MFMA mode");
    static_assert(!F16 || (BF16 && !ABF), "fp16 operands: the 16-bit MFMA mode with fp32 activation storage");
    // XCD-aware block remap (bijective): each XCD (linear block id % 8) walks a contiguous range of tiles so
    // that neighbouring tiles (same activation rows, other Cout slices / halo rows) share one L2.  Batched launches (Winograd
    // components, blockIdx.z) fold the component into the walk: an XCD then owns WHOLE components (A and B of a component are
    // fetched by one L2 instead of eight: 1.37 -> 0.35 GB of fabric reads for a 1024 x 1024 x 1024 x 36 layer, r03 PMC).
    int wgid, zc = 0;
    {
        const int tiles = gridDim.x;
        const bool fold = pin.nz > 1 && gridDim.y == 1;
        const int nwg = fold ? tiles * (int)gridDim.z : tiles;
        const int orig = fold ? (int)(blockIdx.x + tiles * blockIdx.z) : (int)blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        const int c = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        zc = fold ? c / tiles : (int)blockIdx.z;
        wgid = fold ? c - zc * tiles : c;
    }
    ConvParams p = pin;  // batched launch: component zc works on its own slice of in0 / w / out
    if (pin.nz > 1) {
        const long long z = zc;
        p.in0 = pin.in0 + z * pin.z_in;
        p.w = pin.w + z * pin.z_w;
        p.out = pin.out + z * pin.z_out;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As = reinterpret_cast<char*>(smem);
    char* Bs = As + 2 * PL * BM * C::ROW_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int l31 = lane & 31;
    const int h = lane >> 5;

    const int mblk = wgid / nblk_n;
    const int nblk = wgid - mblk * nblk_n;
    const int m0 = mblk * BM;
    const int n0 = nblk * BN;

    // split-K range of this block
    const int split = blockIdx.y;
    const int nsplit = gridDim.y;
    const int kt_begin = (int)(((long long)nk_total * split) / nsplit);
    const int kt_end = (int)(((long long)nk_total * (split + 1)) / nsplit);

    const int Ctot = p.C0 + p.C1;
    const int steps_per_tap = Ctot / BK;
    const int taps = p.KH * p.KW;
    const int Hv = p.Hin << p.in_shift;
    const int Wv = p.Win << p.in_shift;

    // ---- per-thread staging coordinates (fixed for the whole K loop) ----
    const int chunk = tid % C::A_CHUNKS;  // A: 16-B chunk (4 fp32 / 8 bf16 k) of the 32-k slice
    const int row0 = tid / C::A_CHUNKS;   // A: first staged row
    const int bchunk = tid % C::B_CHUNKS;
    const int brow0 = tid / C::B_CHUNKS;
    int a_iy0[C::A_PASSES], a_ix0[C::A_PASSES], a_pix[C::A_PASSES], a_b[C::A_PASSES];
#pragma unroll
    for (int ps = 0; ps < C::A_PASSES; ++ps) {
        const int m = m0 + row0 + ps * C::A_ROWS;
        const bool ok = m < M;
        const int mm = ok ? m : 0;
        const int ox = mm % p.Wo;
        const int t1 = mm / p.Wo;
        const int oy = t1 % p.Ho;
        const int b = t1 / p.Ho;
        // rows beyond M get an iy far outside the image: every tap is then "out of range" => zeros
        a_iy0[ps] = ok ? oy * p.stride - p.pad_y : -(1 << 28);
        a_ix0[ps] = ox * p.stride - p.pad_x;
        a_pix[ps] = b * p.Hin * p.Win;
        a_b[ps] = b;
    }
    // weight row (+ this lane's 16-B chunk) of every staged B row.  Rows past Cout are clamped to the last row, not
    // zeroed: they only feed output columns n >= Cout, which the epilogue never stores.
    const char* wrow[C::B_PASSES];
    constexpr int WESZ = BF16 ? 2 : 4;
    const char* wbase = PAIR ? reinterpret_cast<const char*>(p.w_pair) : BF16 ? reinterpret_cast<const char*>(p.w_bf) : reinterpret_cast<const char*>(p.w);
    if (BF16 && pin.nz > 1) wbase += (long long)zc * pin.z_w * WESZ;
#pragma unroll
    for (int ps = 0; ps < C::B_PASSES; ++ps) {
        const int n = n0 + brow0 + ps * C::B_ROWS;
        wrow[ps] = wbase + ((size_t)(n < p.Cout ? n : p.Cout - 1) * taps * Ctot) * WESZ + bchunk * 16;
    }

    // BUFA: descriptors + per-thread offsets
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rA0, rA1, rW, rS;
    [[maybe_unused]] unsigned a_voff[C::A_PASSES], b_voff[C::B_PASSES], s_voff[C::A_PASSES];
    constexpr unsigned kOOB = 0x80000000u;
    if constexpr (BUFA) {
        const unsigned in0_bytes = (unsigned)((((long long)p.B * p.Hin * p.Win - 1) * p.pix0 + p.C0) * 4);
        const unsigned in1_bytes = p.C1 ? (unsigned)((((long long)p.B * p.Hin * p.Win - 1) * p.pix1 + p.C1) * 4) : 0u;
        rA0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0), 0, in0_bytes, 0x00020000);
        rA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.C1 ? p.in1 : p.in0), 0, in1_bytes, 0x00020000);
        rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase), 0, (unsigned)((long long)p.Cout * taps * Ctot * WESZ), 0x00020000);
        rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(INSCALE ? p.in_scale : p.in0), 0,
                                               INSCALE ? (unsigned)((long long)p.B * p.C0 * 4) : 0u, 0x00020000);
#pragma unroll
        for (int ps = 0; ps < C::B_PASSES; ++ps) {
            const int n = n0 + brow0 + ps * C::B_ROWS;
            b_voff[ps] = (unsigned)(n < p.Cout ? n : p.Cout - 1) * (unsigned)(taps * Ctot * WESZ) + bchunk * 16;
        }
#pragma unroll
        for (int ps = 0; ps < C::A_PASSES; ++ps) s_voff[ps] = (unsigned)(a_b[ps] * p.C0 + chunk * 4) * 4u;
    }

    // ---- K-loop state: (tap, channel offset); per-tap pixel offsets of the staged rows ----
    int tap = kt_begin / steps_per_tap;
    int cc = (kt_begin - tap * steps_per_tap) * BK;
    int ky = tap / p.KW;
    int kx = tap - ky * p.KW;
    int a_poff[C::A_PASSES];  // pixel index into the source tensors for the current tap, or -1
    auto set_tap = [&]() {
#pragma unroll
        for (int ps = 0; ps < C::A_PASSES; ++ps) {
            const int iy = a_iy0[ps] + ky;
            const int ix = a_ix0[ps] + kx;
            const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
            a_poff[ps] = ok ? a_pix[ps] + (iy >> p.in_shift) * p.Win + (ix >> p.in_shift) : -1;
        }
    };
    auto advance = [&]() {
        cc += BK;
        if (cc >= Ctot) {
            cc = 0;
            ++tap;
            if (++kx == p.KW) {
                kx = 0;
                ++ky;
            }
            set_tap();
        }
    };

    // ---- staging pieces: A rows (A_PASSES) then B rows (B_PASSES), one 16-byte load each ----
    constexpr int NP = C::A_PASSES + C::B_PASSES;   // (the PAIR kernels stage through their own lambdas further down)
    floatx4 rs[NP];  // (ext_vector: HIP's float4 struct copies become memcpys that can pin the array to scratch)
    floatx4 rscale[INSCALE ? C::A_PASSES : 1];  // NAFNet SCA: the per-(image, channel) scales of the A pieces in flight
    constexpr int AESZ = ABF ? 2 : 4;        // bytes per activation element in HBM
    constexpr int ACE = 16 / AESZ;           // elements per 16-byte chunk
    const char* cur_src = nullptr;  // source pointer (+channel +chunk) of the K-step being staged
    int cur_pix = 0;                // bytes between consecutive pixels of that source
    size_t cur_wk = 0;  // byte offset of the K-step inside a weight row
    [[maybe_unused]] int cur_c = 0;  // BUFA: byte offset of the K-step's channels inside a pixel of the current source
    auto stage_setup = [&](const bool force_voff = false) {
        const float* src;
        int c;
        if (cc < p.C0) {
            src = p.in0; c = cc; cur_pix = p.pix0 * AESZ;
        } else {
            src = p.in1; c = cc - p.C0; cur_pix = p.pix1 * AESZ;
        }
        cur_src = reinterpret_cast<const char*>(src) + (size_t)(c + chunk * ACE) * AESZ;
        cur_wk = ((size_t)tap * Ctot + cc) * WESZ;
        if constexpr (BUFA) {
            cur_c = c * AESZ;
            if (c == 0 || force_voff) {  // first K-step of a (tap, source): one burst of vector instructions, then none until the next one
#pragma unroll
                for (int ps = 0; ps < C::A_PASSES; ++ps)
                    a_voff[ps] = a_poff[ps] >= 0 ? (unsigned)a_poff[ps] * (unsigned)cur_pix + (unsigned)(chunk * 16) : kOOB;
            }
        }
    };
    auto load_piece = [&](int q) {
        if constexpr (BUFA) {
            if (q < C::A_PASSES) {  // out-of-image taps and rows past M: offset out of range, the descriptor returns zeros
                rs[q] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(cc < p.C0 ? rA0 : rA1, (int)a_voff[q < C::A_PASSES ? q : 0], cur_c, 0));
                if constexpr (INSCALE)
                    rscale[q < C::A_PASSES ? q : 0] =
                        __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rS, (int)s_voff[q < C::A_PASSES ? q : 0], cc * 4, 0));
            } else {
                rs[q] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rW, (int)b_voff[q >= C::A_PASSES ? q - C::A_PASSES : 0], (int)cur_wk, 0));
            }
        } else if (q < C::A_PASSES) {
            // branch-free: out-of-image taps (zero padding, rows past M) read the zero page instead
            const char* g = a_poff[q] >= 0 ? cur_src + (size_t)a_poff[q] * cur_pix : reinterpret_cast<const char*>(p.zeros);
            rs[q] = *reinterpret_cast<const floatx4*>(g);
            if constexpr (INSCALE)  // single source (C1 == 0); multiplied when the piece is written to LDS (store_piece)
                rscale[q < C::A_PASSES ? q : 0] = *reinterpret_cast<const floatx4*>(p.in_scale + (size_t)a_b[q] * p.C0 + cc + chunk * 4);
        } else {
            rs[q] = *reinterpret_cast<const floatx4*>(wrow[q - C::A_PASSES] + cur_wk);
        }
    };
    auto store_piece = [&](int q, int buf) {
        if (q < C::A_PASSES) {
            char* dst = As + (buf * BM + row0 + q * C::A_ROWS) * C::ROW_BYTES;
            if (ABF) {
                *reinterpret_cast<floatx4*>(dst + chunk * 16) = rs[q];  // already bf16: 8 k per piece
            } else if (BF16) {
                floatx4 v = rs[q];
                if constexpr (INSCALE) v *= rscale[q < C::A_PASSES ? q : 0];
                *reinterpret_cast<typename H16::x4*>(dst + chunk * 8) = __builtin_convertvector(v, typename H16::x4);  // v_cvt_pk_bf16_f32 / v_cvt_f16_f32, RNE
            } else {
                if constexpr (INSCALE)
                    *reinterpret_cast<floatx4*>(dst + chunk * 16) = rs[q] * rscale[q < C::A_PASSES ? q : 0];
                else
                    *reinterpret_cast<floatx4*>(dst + chunk * 16) = rs[q];
            }
        } else {
            const int ps = q - C::A_PASSES;
            if (C::B_PASSES * C::B_ROWS == BN || brow0 + ps * C::B_ROWS < BN)
                *reinterpret_cast<floatx4*>(Bs + (buf * BN + brow0 + ps * C::B_ROWS) * C::ROW_BYTES + bchunk * 16) = rs[q];
        }
    };

    floatx16 acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (!PAIR) {
        if (kt_begin < kt_end) {
            set_tap();
            stage_setup(true);  // (a split-K range may start in the middle of a tap)
#pragma unroll
            for (int q = 0; q < NP; ++q) load_piece(q);
#pragma unroll
            for (int q = 0; q < NP; ++q) store_piece(q, 0);
        }
        __syncthreads();
    }

    // One K-step: next-tile loads issued first, then the MFMAs (the compiler streams the fragment reads
    // between them), then the LDS writes.  The other block on the CU covers the gaps.
    constexpr int NSB = BF16 ? 2 : 4;  // 32-byte groups per tile row
    auto k_step = [&](const int buf, const bool more) {
        if (more) {
            advance();
            stage_setup();
        }
        const char* a = As + (buf * PL * BM + wm * C::TM * 32 + l31) * C::ROW_BYTES + h * 16;
        const char* b = Bs + (buf * PL * BN + wn * C::TN * 32 + l31) * C::ROW_BYTES + h * 16;
        if constexpr (BF16) {
            if (more) {
#pragma unroll
                for (int q = 0; q < NP; ++q) load_piece(q);
            }
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                typename H16::x8 fa[C::TM], fb[C::TN];
#pragma unroll
                for (int i = 0; i < C::TM; ++i)
                    fa[i] = *reinterpret_cast<const typename H16::x8*>(a + i * 32 * C::ROW_BYTES + sb * 32);
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
                    fb[j] = *reinterpret_cast<const typename H16::x8*>(b + j * 32 * C::ROW_BYTES + sb * 32);
#pragma unroll
                for (int i = 0; i < C::TM; ++i)
#pragma unroll
                    for (int j = 0; j < C::TN; ++j)
                        acc[i][j] = H16::mfma(fa[i], fb[j], acc[i][j]);
            }
            if (more) {
#pragma unroll
                for (int q = 0; q < NP; ++q) store_piece(q, buf ^ 1);
            }
        } else {
            // f32: pinned schedule, as in gemm_zloop_kernel — 16 groups of TM x TN MFMAs (one k pair each), the next 8-k
            // sub-step's fragments read while the current one multiplies, ONE staging instruction behind each group (the
            // next K-step's loads behind groups 0..NP-1, their LDS writes behind groups 16-NP..15).  The staging is
            // unconditional — a branch around it splits the K-step into basic blocks and the wait insertion then drains
            // every load before the next is issued; the last K-step re-stages its own operands into the dead buffer.
            static_assert(NP <= 8, "one staging slot per MFMA group");
            // LDS write addresses of this K-step's pieces: formed here, in the vector-instruction burst in front of the
            // first MFMA; the stores behind the MFMA groups then only add immediates
            char* const st_a = As + ((buf ^ 1) * BM + row0) * C::ROW_BYTES + chunk * 16;
            char* const st_b = Bs + ((buf ^ 1) * BN + brow0) * C::ROW_BYTES + bchunk * 16;
            auto store_piece_f32 = [&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (q < C::A_PASSES) {
                    if constexpr (INSCALE)
                        *reinterpret_cast<floatx4*>(st_a + q * C::A_ROWS * C::ROW_BYTES) = rs[q] * rscale[q < C::A_PASSES ? q : 0];
                    else
                        *reinterpret_cast<floatx4*>(st_a + q * C::A_ROWS * C::ROW_BYTES) = rs[q];
                } else {
                    constexpr int ps = q - C::A_PASSES;
                    if (C::B_PASSES * C::B_ROWS == BN || brow0 + ps * C::B_ROWS < BN)
                        *reinterpret_cast<floatx4*>(st_b + ps * C::B_ROWS * C::ROW_BYTES) = rs[q];
                }
            };
            float4 fa[2][C::TM], fb[2][C::TN];
            auto read_frags = [&](auto sbc) {
                constexpr int sb = decltype(sbc)::value;
#pragma unroll
                for (int i = 0; i < C::TM; ++i) fa[sb & 1][i] = *reinterpret_cast<const float4*>(a + i * 32 * C::ROW_BYTES + sb * 32);
#pragma unroll
                for (int j = 0; j < C::TN; ++j) fb[sb & 1][j] = *reinterpret_cast<const float4*>(b + j * 32 * C::ROW_BYTES + sb * 32);
            };
            read_frags(std::integral_constant<int, 0>{});
            static_for<16>([&](auto gic) {
                constexpr int gi = decltype(gic)::value, sb = gi >> 2, q = gi & 3, cur = sb & 1;
                if constexpr (q == 0 && sb < 3) read_frags(std::integral_constant<int, sb + 1>{});
#pragma unroll
                for (int i = 0; i < C::TM; ++i)
#pragma unroll
                    for (int j = 0; j < C::TN; ++j) {
                        const float av = q == 0 ? fa[cur][i].x : q == 1 ? fa[cur][i].y : q == 2 ? fa[cur][i].z : fa[cur][i].w;
                        const float bv = q == 0 ? fb[cur][j].x : q == 1 ? fb[cur][j].y : q == 2 ? fb[cur][j].z : fb[cur][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                if constexpr (gi < NP) load_piece(gi);
                if constexpr (gi >= 16 - NP) store_piece_f32(std::integral_constant<int, gi - (16 - NP)>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    };

    if constexpr (PAIR) {
        // r03 (late): deeper staging for the direct PAIR layers.  The first PAIR loop staged A and B through registers ONE K step ahead
        // behind __syncthreads(), which on gfx9 drains vmcnt: a step cost the load latency (7.4k cycles at 3k of MFMAs, f16 pipe 0.27
        // busy on the NAFNet 1x1 layers).  Now A (activations: HBM / Infinity Cache latency) is loaded TWO steps ahead into a register
        // ping-pong, B (weights, pre-interleaved [Cout][K / 32][hi 32 | lo 32]: one 128-byte line per row and K step, L2-hot) one step
        // ahead; both are split / written to LDS one step ahead, and the step ends with s_waitcnt lgkmcnt(0) + s_barrier only — the
        // loads of the following steps stay in flight across the barrier (the compiler counts vmcnt per register).  Staging is
        // unconditional: the last steps re-stage into dead buffers (reads stay inside the tensors).
        constexpr int NA = C::A_PASSES, BP = BN * 8 / C::NT;   // 16-byte pieces per thread and step: A rows / B row pieces
        static_assert(BN * 8 % C::NT == 0 && C::NT % 8 == 0, "B tile pieces per pass");
        const int nkb = taps * steps_per_tap;     // 128-byte blocks per weight row
        const int bpiece = tid & 7, brow = tid >> 3;   // piece p of a row's line: plane p >> 2, k = 8 (p & 3) .. + 7
        const char* bsrc[BP];
#pragma unroll
        for (int ps = 0; ps < BP; ++ps) {
            const int n = n0 + brow + ps * (C::NT / 8);
            bsrc[ps] = reinterpret_cast<const char*>(p.w_pair) + (size_t)(n < p.Cout ? n : p.Cout - 1) * nkb * 128 + bpiece * 16;
        }
        floatx4 ra[2][NA], rb[BP];
        floatx4 rsc[INSCALE ? NA : 1];
        auto load_b = [&](const int kb) {
#pragma unroll
            for (int ps = 0; ps < BP; ++ps) rb[ps] = *reinterpret_cast<const floatx4*>(bsrc[ps] + (size_t)kb * 128);
        };
        auto store_b = [&](const int buf) {
#pragma unroll
            for (int ps = 0; ps < BP; ++ps)
                *reinterpret_cast<floatx4*>(Bs + ((buf * 2 + (bpiece >> 2)) * BN + brow + ps * (C::NT / 8)) * C::ROW_BYTES + (bpiece & 3) * 16) = rb[ps];
        };
        auto load_a = [&](auto phc) {
            constexpr int ph = decltype(phc)::value;
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const char* g = a_poff[q] >= 0 ? cur_src + (size_t)a_poff[q] * cur_pix : reinterpret_cast<const char*>(p.zeros);
                ra[ph][q] = *reinterpret_cast<const floatx4*>(g);
            }
        };
        auto load_scale = [&]() {   // (state = the step whose A pieces are written next)
            if constexpr (INSCALE) {
#pragma unroll
                for (int q = 0; q < NA; ++q) rsc[q] = *reinterpret_cast<const floatx4*>(p.in_scale + (size_t)a_b[q] * p.C0 + cc + chunk * 4);
            }
        };
        auto store_a = [&](auto phc, const int buf) {
            constexpr int ph = decltype(phc)::value;
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                floatx4 v = ra[ph][q];
                if constexpr (INSCALE) v *= rsc[q];
                const typename H16::x4 hi = __builtin_convertvector(v, typename H16::x4);          // RNE
                const floatx4 r = v - __builtin_convertvector(hi, floatx4);                        // exact in f32
                const typename H16::x4 lo = __builtin_convertvector(r, typename H16::x4);
                char* dst = As + ((buf * 2) * BM + row0 + q * C::A_ROWS) * C::ROW_BYTES + chunk * 8;
                *reinterpret_cast<typename H16::x4*>(dst) = hi;
                *reinterpret_cast<typename H16::x4*>(dst + BM * C::ROW_BYTES) = lo;
            }
        };
        auto mfma_step = [&](const int buf) {
            const char* a = As + (buf * 2 * BM + wm * C::TM * 32 + l31) * C::ROW_BYTES + h * 16;
            const char* b = Bs + (buf * 2 * BN + wn * C::TN * 32 + l31) * C::ROW_BYTES + h * 16;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                typename H16::x8 fa[2][C::TM], fb[2][C::TN];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int i = 0; i < C::TM; ++i)
                        fa[pl][i] = *reinterpret_cast<const typename H16::x8*>(a + (pl * BM + i * 32) * C::ROW_BYTES + sb * 32);
#pragma unroll
                    for (int j = 0; j < C::TN; ++j)
                        fb[pl][j] = *reinterpret_cast<const typename H16::x8*>(b + (pl * BN + j * 32) * C::ROW_BYTES + sb * 32);
                }
                // hi.lo, lo.hi, hi.hi: small terms first; consecutive MFMAs hit different accumulators
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int i = 0; i < C::TM; ++i)
#pragma unroll
                        for (int j = 0; j < C::TN; ++j)
                            acc[i][j] = H16::mfma(fa[pr == 1 ? 1 : 0][i], fb[pr == 0 ? 1 : 0][j], acc[i][j]);
            }
        };
        if (kt_begin < kt_end) {
            // prologue: A(k0) -> ra[0], scales(k0), B(k0), A(k0 + 1) -> ra[1]; then A(k0), B(k0) -> buffer 0
            set_tap();
            stage_setup(true);
            load_a(std::integral_constant<int, 0>{});
            load_scale();
            load_b(kt_begin);
            __builtin_amdgcn_sched_barrier(0);
            advance();
            stage_setup();
            load_a(std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            store_a(std::integral_constant<int, 0>{}, 0);
            store_b(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // step k (ph = parity): ra[ph] held A(k) (already in LDS) and is refilled with A(k + 2); ra[ph ^ 1] holds A(k + 1)
        // (What a step is made of, measured with run-time switches that have since been removed — they cost the production instance 10 %:
        //  NAFNet 512 -> 1024 @ 64^2: without the K loop's global loads -13 %, without MFMAs -40 %, without the LDS stores -11 %, without the
        //  epilogue -26 %: the parts add up, nothing overlaps; profiles/r03_pair_conv_sweep.txt.)
        auto pstep = [&](auto phc, const int kt) {
            constexpr int ph = decltype(phc)::value;
            const int buf = ph;
            load_b(kt + 1 < nk_total ? kt + 1 : nk_total - 1);
            load_scale();          // the state machine stands at step k + 1 here
            __builtin_amdgcn_sched_barrier(0);
            advance();
            stage_setup();
            load_a(phc);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(buf);
            store_a(std::integral_constant<int, ph ^ 1>{}, buf ^ 1);
            store_b(buf ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        int kt = kt_begin;
        for (; kt + 1 < kt_end; kt += 2) {
            pstep(std::integral_constant<int, 0>{}, kt);
            pstep(std::integral_constant<int, 1>{}, kt + 1);
        }
        if (kt < kt_end) pstep(std::integral_constant<int, 0>{}, kt);
        __syncthreads();   // (drains the stray loads of the last steps; the epilogue re-uses the LDS)
    } else {
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            k_step((kt - kt_begin) & 1, kt + 1 < kt_end);
            __syncthreads();
        }
    }
    if constexpr (PAIR && F16) {   // undo the power-of-two scale of the fp16 weight planes (exact)
        const float ps = p.pair_scale;
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= ps;
    }

    // ---- direct epilogue (BUFA kernels, plain output layouts): straight from the accumulator registers through buffer
    // descriptors, like gemm_zloop_kernel's write-back.  An MFMA register holds 2 rows x 32 consecutive columns: a store
    // (and a residual load) covers two full 128-byte lines; the per-thread offsets are formed once, the sub-tile position
    // goes in an SGPR, rows past M fall outside the descriptor.  bias -> FiLM (one row for the batch) -> SiLU -> channel
    // scale -> + residual, per column = per lane.  No LDS round trip, no per-row index arithmetic: ~4 vector instructions
    // per output instead of ~10 (vector instructions are paid in f32-MFMA time on gfx950).
    // (PAIR kernels too, late r03: their LDS-transposed epilogue was 26 % of a 512 -> 1024 NAFNet layer; the descriptors here only concern out / res)
    if constexpr (BUFA || PAIR) {
        if (p.splits == 1 && !p.ln_g && !p.gate && !p.shuffle && !(p.film && p.film_bstride != 0) && !p.out_bf16 && !p.no_direct_epi) {
            const int wm_s = __builtin_amdgcn_readfirstlane(wm), wn_s = __builtin_amdgcn_readfirstlane(wn);
            const int rowb = m0 + wm_s * C::TM * 32, colb = n0 + wn_s * C::TN * 32;
            const int rows = M - rowb;
            const unsigned nrows = rows <= 0 ? 0u : (unsigned)(rows < C::TM * 32 ? rows : C::TM * 32);
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out + (long long)rowb * p.out_stride, 0,
                                                                                nrows * (unsigned)p.out_stride * 4u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.res ? p.res + (long long)rowb * p.res_stride : p.out), 0,
                p.res ? nrows * (unsigned)p.res_stride * 4u : 0u, 0x00020000);
            unsigned o_voff[16], r_voff[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro_ = (r & 3) + 8 * (r >> 2) + 4 * h;
                o_voff[r] = (unsigned)(ro_ * p.out_stride + l31) * 4u;
                r_voff[r] = (unsigned)(ro_ * p.res_stride + l31) * 4u;
            }
#pragma unroll
            for (int j = 0; j < C::TN; ++j) {
                const int colu = colb + j * 32, col = colu + l31;
                const bool ok = col < p.Cout;
                const int cc_ = ok ? col : 0;
                const float bj = p.bias ? p.bias[cc_] : 0.f;
                const float scj = p.film ? p.film[cc_] + 1.0f : 1.0f, shj = p.film ? p.film[p.Cout + cc_] : 0.f;
                const float csj = p.ch_scale ? p.ch_scale[cc_] : 1.0f;
#pragma unroll
                for (int i = 0; i < C::TM; ++i) {
                    float rv[16];
                    if (p.res) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)r_voff[r], (i * 32 * p.res_stride + colu) * 4, 0));
                    }
                    if (ok) {
                        const int soff = (i * 32 * p.out_stride + colu) * 4;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float t = acc[i][j][r] + bj;
                            if (p.film) t = t * scj + shj;
                            if (p.silu) t = silu_f(t);
                            if (p.ch_scale) t *= csj;
                            if (p.res) t += rv[r];
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), ro, (int)o_voff[r], soff, 0);
                        }
                    }
                }
            }
            return;
        }
    }

    // ---- epilogue: transpose the accumulators through LDS (A/B buffers are dead after the last barrier) ----
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cs = smem;
    constexpr int NV = BN / 4;            // float4 columns of the tile
    constexpr int RSTEP = C::NT / NV;     // rows covered per sweep
    const int c4 = tid % NV;
    const int n = n0 + c4 * 4;
    const int HW = p.Ho * p.Wo;
    const bool vec_ok = (n + 3 < p.Cout) && ((p.out_stride & 3) == 0);
    float bias[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.splits == 1 && n < p.Cout) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < p.Cout) {
                if (p.bias) bias[e] = p.bias[n + e];
                if (p.film && p.film_bstride == 0) {
                    sc[e] = p.film[n + e] + 1.0f;
                    sh[e] = p.film[p.Cout + n + e];
                }
            }
    }
    constexpr int WAVE_ROWS = C::TM * 32;
#pragma unroll 1
    for (int pass = 0; pass < BM / C::EPI_ROWS; ++pass) {
        if (pass > 0) __syncthreads();  // previous pass fully stored before Cs is overwritten
        if ((wm * WAVE_ROWS) / C::EPI_ROWS == pass) {
            const int rbase = wm * WAVE_ROWS - pass * C::EPI_ROWS;
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        Cs[row * C::LDS_C + wn * C::TN * 32 + j * 32 + l31] = acc[i][j][r];
                    }
        }
        __syncthreads();
        if (p.ln_g) {
            // bias -> channel LayerNorm -> +res (LinearAttention.to_out + Residual): four rows per iteration, so that the ten
            // dependent cross-lane steps (ds_bpermute, ~100 cycles each) of the four rows overlap instead of queueing up
            constexpr int U = 4;
            const float4 g4 = *reinterpret_cast<const float4*>(p.ln_g + n);
            for (int rb = tid / NV; rb < C::EPI_ROWS; rb += U * RSTEP) {
                float v[U][4], sum[U], sq[U];
                long long mrow[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int row = rb + u * RSTEP;
                    const int m = m0 + pass * C::EPI_ROWS + row;
                    mrow[u] = (row < C::EPI_ROWS && m < M) ? m : -1;
                    const float4 cv = *reinterpret_cast<const float4*>(Cs + (row < C::EPI_ROWS ? row : 0) * C::LDS_C + c4 * 4);
                    v[u][0] = cv.x + bias[0]; v[u][1] = cv.y + bias[1]; v[u][2] = cv.z + bias[2]; v[u][3] = cv.w + bias[3];
                    sum[u] = (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
                }
#pragma unroll
                for (int o = NV / 2; o > 0; o >>= 1)
#pragma unroll
                    for (int u = 0; u < U; ++u) sum[u] += __shfl_xor(sum[u], o, 64);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float mean = sum[u] * (1.0f / (float)BN);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][e] -= mean;
                    sq[u] = (v[u][0] * v[u][0] + v[u][1] * v[u][1]) + (v[u][2] * v[u][2] + v[u][3] * v[u][3]);
                }
#pragma unroll
                for (int o = NV / 2; o > 0; o >>= 1)
#pragma unroll
                    for (int u = 0; u < U; ++u) sq[u] += __shfl_xor(sq[u], o, 64);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (mrow[u] < 0) continue;
                    const float rstd = 1.0f / sqrtf(sq[u] * (1.0f / (float)BN) + p.ln_eps);
                    float o4[4] = {v[u][0] * rstd * g4.x, v[u][1] * rstd * g4.y, v[u][2] * rstd * g4.z, v[u][3] * rstd * g4.w};
                    const size_t off = (size_t)mrow[u];
                    if (BF16 && p.out_bf16) {
                        if (p.res) {
                            const bf16x4 t4 = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.res) + off * p.res_stride + n);
                            o4[0] += (float)t4[0]; o4[1] += (float)t4[1]; o4[2] += (float)t4[2]; o4[3] += (float)t4[3];
                        }
                        const floatx4 fv = {o4[0], o4[1], o4[2], o4[3]};
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out) + off * p.out_stride + n) = __builtin_convertvector(fv, bf16x4);
                    } else {
                        if (p.res) {
                            const float4 t4 = *reinterpret_cast<const float4*>(p.res + off * p.res_stride + n);
                            o4[0] += t4.x; o4[1] += t4.y; o4[2] += t4.z; o4[3] += t4.w;
                        }
                        *reinterpret_cast<float4*>(p.out + off * p.out_stride + n) = make_float4(o4[0], o4[1], o4[2], o4[3]);
                    }
                }
            }
        } else if (n < p.Cout) {
        for (int row = tid / NV; row < C::EPI_ROWS; row += RSTEP) {
            const int m = m0 + pass * C::EPI_ROWS + row;
            if (m >= M) break;
            const float4 cv = *reinterpret_cast<const float4*>(Cs + row * C::LDS_C + c4 * 4);
            float v[4] = {cv.x, cv.y, cv.z, cv.w};
            if (p.splits > 1) {
                float* dst = p.partial + ((size_t)split * M + m) * p.Cout + n;
                if (n + 3 < p.Cout && (p.Cout & 3) == 0) {
                    *reinterpret_cast<float4*>(dst) = cv;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.Cout) dst[e] = v[e];
                }
                continue;
            }
            if (p.film && p.film_bstride != 0) {
                const float* f = p.film + (size_t)(m / HW) * p.film_bstride;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.Cout) {
                        sc[e] = f[n + e] + 1.0f;
                        sh[e] = f[p.Cout + n + e];
                    }
            }
            // output addressing: plain [m][n], gated [m][n/2], or pixel-shuffled [b][2y+dy][2x+dx][co]
            size_t opix = (size_t)m;
            int ocol = n;
            if (p.shuffle) {
                const int Cq = p.Cout >> 2;
                const int q = n / Cq;
                const int ox = m % p.Wo, t1 = m / p.Wo, oy = t1 % p.Ho, ob = t1 / p.Ho;
                opix = ((size_t)ob * 2 * p.Ho + 2 * oy + (q >> 1)) * (2 * p.Wo) + 2 * ox + (q & 1);
                ocol = n - q * Cq;
            } else if (p.gate) {
                ocol = n >> 1;
            }
            float cs[4] = {1.f, 1.f, 1.f, 1.f};
            if (p.ch_scale) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.Cout) cs[e] = p.ch_scale[n + e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[e] + bias[e];
                if (p.film) t = t * sc[e] + sh[e];
                if (p.silu) t = silu_f(t);
                v[e] = t * cs[e];
            }
            if (p.gate) {  // SimpleGate: pairs are adjacent by construction of the packed weights
                float* dst = p.out + opix * p.out_stride + ocol;
                float g0 = v[0] * v[1], g1 = v[2] * v[3];
                if (p.gate_film) {  // per-image FiLM on the gated channels (latent-bokeh cam_mlp)
                    const float* f = p.gate_film + (size_t)(m / HW) * p.gate_film_bstride;
                    const int ch = p.Cout >> 1;
                    g0 = g0 * (f[ocol] + 1.0f) + f[ch + ocol];
                    g1 = g1 * (f[ocol + 1] + 1.0f) + f[ch + ocol + 1];
                }
                *reinterpret_cast<float2*>(dst) = make_float2(g0, g1);
                continue;
            }
            if (BF16 && p.out_bf16) {  // bf16 activation storage: residual and output tensors are bf16
                if (p.res) {
                    const __bf16* rp = reinterpret_cast<const __bf16*>(p.res) + opix * p.res_stride + ocol;
                    if (n + 3 < p.Cout && (p.res_stride & 3) == 0) {
                        const bf16x4 t4 = *reinterpret_cast<const bf16x4*>(rp);
                        v[0] += (float)t4[0]; v[1] += (float)t4[1]; v[2] += (float)t4[2]; v[3] += (float)t4[3];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.Cout) v[e] += (float)rp[e];
                    }
                }
                __bf16* dst = reinterpret_cast<__bf16*>(p.out) + opix * p.out_stride + ocol;
                if (vec_ok) {
                    const floatx4 fv = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<bf16x4*>(dst) = __builtin_convertvector(fv, bf16x4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.Cout) dst[e] = (__bf16)v[e];
                }
                continue;
            }
            if (p.res) {
                const float* rp = p.res + opix * p.res_stride + ocol;
                if (n + 3 < p.Cout && (p.res_stride & 3) == 0) {
                    const float4 t4 = *reinterpret_cast<const float4*>(rp);
                    v[0] += t4.x; v[1] += t4.y; v[2] += t4.z; v[3] += t4.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.Cout) v[e] += rp[e];
                }
            }
            float* dst = p.out + opix * p.out_stride + ocol;
            if (vec_ok) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.Cout) dst[e] = v[e];
            }
        }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM with a tile loop INSIDE the block — for the layers whose K is short (2..32 K-steps per output tile):
//   * the (m+2)^2 component GEMMs of a Winograd layer, M_z[t][n] = sum_k V_z[t][k] * U_z[n][k]  (K = Cin);
//   * 1x1 convolutions (to_qkv, res_conv, to_out) with K = Cin = 64..384.
// Launched through conv_igemm_kernel every output tile is its own block with a cold prologue and an LDS-transposed epilogue
// around those few K-steps (55-95 TFLOP/s measured).  Here a block walks a sequence of output tiles — Winograd: the same
// (row tile, column tile) of ZB consecutive components; 1x1: all column tiles of several consecutive row tiles — as ONE
// software-pipelined K loop: the double-buffered staging runs straight across tile boundaries, and at a boundary the
// accumulators go to HBM directly from registers (per MFMA register: 2 rows x 32 consecutive columns = two full 128-byte
// lines per store instruction; + bias), are zeroed, and the MFMAs of the next tile start — no LDS round trip, no barrier.
// ---------------------------------------------------------------------------------------------------------------
struct ZLoopArgs {
    const float* a0; const float* a1;  // A sources, concatenated along K (a1 may be null): row m at a + m * lda (+ k)
    int C0, C1, lda0, lda1;
    const float* b;                    // B rows (weights): row n at b + n * K
    float* out; const float* bias;     // out[row * ldc + col] (+ bias[col])
    int M, N, K, ldc;                  // valid rows / columns, K = C0 + C1
    int n_inner, n_outer;              // tiles per block: inner (fastest) x outer
    long long pA, pB, pO;              // per inner step: plane offsets of A / B / out (floats)   (Winograd components)
    int col_step;                      // per inner step: column offset (1x1: 128 = next column tile)
    int row_step;                      // per outer step: row offset   (1x1: 128 = next row tile)
    int nblk_n;                        // column tiles across blockIdx.x (Winograd) or 1
};

template <int BM, int BN, int WAVES_M, int WAVES_N, int MIN_WAVES_PER_SIMD>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MIN_WAVES_PER_SIMD) void gemm_zloop_kernel(const ZLoopArgs g) {
    using C = Cfg<BM, BN, WAVES_M, WAVES_N, false>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As = reinterpret_cast<char*>(smem);
    char* Bs = As + 2 * BM * C::ROW_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, h = lane >> 5;
    int wgid, zg;   // XCD-aware walk over (component group, tile): an XCD owns whole component groups (see conv_igemm_kernel)
    {
        const int tiles = gridDim.x, nwg = tiles * (int)gridDim.y;
        const int orig = (int)(blockIdx.x + tiles * blockIdx.y);
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        const int c = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        zg = c / tiles;
        wgid = c - zg * tiles;
    }
    const int mblk = wgid / g.nblk_n, nblk = wgid - mblk * g.nblk_n;
    const int m0 = mblk * BM * (g.row_step ? g.n_outer : 1), n0 = nblk * BN;
    const int plane0 = zg * g.n_inner;  // Winograd: first component of this block
    const int nk = g.K / BK;
    const int steps = g.n_inner * g.n_outer * nk;

    const int chunk = tid % 8, row0 = tid / 8;
    // staging pointers of the tile being staged (rows past M / N are clamped: their products land in rows / columns that
    // are never stored); recomputed when the staging crosses into the next tile
    const float* arow0[C::A_PASSES];
    const float* arow1[C::A_PASSES];
    const float* brow[C::B_PASSES];
    int st_i = 0, st_o = 0;  // tile (inner, outer) being staged
    auto set_tile_ptrs = [&]() {
        const int rowb = m0 + st_o * g.row_step, colb = n0 + st_i * g.col_step;
        const long long pl = (long long)(plane0 + (g.col_step ? 0 : st_i));
        static_for<C::A_PASSES>([&](auto ps) {
            int m = rowb + row0 + ps() * C::A_ROWS;
            m = m < g.M ? m : g.M - 1;
            arow0[ps()] = g.a0 + pl * g.pA + (size_t)m * g.lda0 + chunk * 4;
            arow1[ps()] = g.a1 ? g.a1 + (size_t)m * g.lda1 + chunk * 4 - g.C0 : arow0[ps()];
        });
        static_for<C::B_PASSES>([&](auto ps) {
            int n = colb + row0 + ps() * C::B_ROWS;
            n = n < g.N ? n : g.N - 1;
            brow[ps()] = g.b + pl * g.pB + (size_t)n * g.K + chunk * 4;
        });
    };
    set_tile_ptrs();
    constexpr int NP = C::A_PASSES + C::B_PASSES;
    floatx4 rs[NP];  // (an ext_vector, not HIP's float4 struct: struct copies become memcpys that pin the array to scratch)
    int kk = 0;      // k position of the K-step being staged inside its tile
    auto advance = [&]() {
        kk += BK;
        if (kk == g.K) {
            kk = 0;
            if (++st_i == g.n_inner) { st_i = 0; ++st_o; }
            set_tile_ptrs();
        }
    };
    auto load_piece = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const bool first = kk < g.C0;  // wave-uniform: which concatenated source this K-step reads
        if constexpr (q < C::A_PASSES)
            rs[q] = *reinterpret_cast<const floatx4*>((first ? arow0[q] : arow1[q]) + kk);
        else
            rs[q] = *reinterpret_cast<const floatx4*>(brow[q - C::A_PASSES] + kk);
    };
    auto load_all = [&]() {
        const bool first = kk < g.C0;  // wave-uniform: which concatenated source this K-step reads
        static_for<C::A_PASSES>([&](auto q) { rs[q()] = *reinterpret_cast<const floatx4*>((first ? arow0[q()] : arow1[q()]) + kk); });
        static_for<C::B_PASSES>([&](auto q) { rs[C::A_PASSES + q()] = *reinterpret_cast<const floatx4*>(brow[q()] + kk); });
    };
    auto store_all = [&](int buf) {
        static_for<C::A_PASSES>([&](auto q) {
            *reinterpret_cast<floatx4*>(As + (buf * BM + row0 + q() * C::A_ROWS) * C::ROW_BYTES + chunk * 16) = rs[q()];
        });
        static_for<C::B_PASSES>([&](auto q) {
            *reinterpret_cast<floatx4*>(Bs + (buf * BN + row0 + q() * C::B_ROWS) * C::ROW_BYTES + chunk * 16) = rs[C::A_PASSES + q()];
        });
    };

    floatx16 acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_all();
    store_all(0);
    __syncthreads();

    // The finished tile is written at the START of the following K-step, after that step's loads have been issued: the
    // stores then have a whole MFMA phase to drain before the next vmcnt wait.
    // bias of every column this block will produce, kept in LDS behind the staging buffers: a global load inside (or
    // one K-step before) the store sequence makes the compiler put vmcnt waits in front of every store, which serialises
    // the stores themselves; LDS reads are counted separately (lgkmcnt)
    float* bias_s = reinterpret_cast<float*>(Bs + 2 * BN * C::ROW_BYTES);
    if (g.bias) {  // kernel-uniform
        const int ncols = g.col_step ? g.n_inner * g.col_step : BN;
        for (int c = tid; c < ncols; c += C::NT) bias_s[c] = (n0 + c < g.N) ? g.bias[n0 + c] : 0.f;
    }
    // Tile write-back: buffer stores with per-thread offsets formed ONCE (row pattern of the MFMA accumulator x ldc) and
    // the (i, j) sub-tile position in an SGPR; rows past M fall outside the descriptor and are dropped by the hardware.
    // (With 64-bit index arithmetic and two compares per element the 64 stores of a tile cost ~300 vector instructions —
    // more than the K loop of a short-K tile, and vector instructions are paid in f32-MFMA time.)
    const int wm_s = __builtin_amdgcn_readfirstlane(wm), wn_s = __builtin_amdgcn_readfirstlane(wn);
    unsigned o_voff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) o_voff[r] = (unsigned)(((r & 3) + 8 * (r >> 2) + 4 * h) * g.ldc + l31) * 4u;
    auto flush = [&](int fi, int fo) {
        const int rowb = m0 + fo * g.row_step + wm_s * C::TM * 32, colb = n0 + fi * g.col_step + wn_s * C::TN * 32;
        float* ob = g.out + (long long)(plane0 + (g.col_step ? 0 : fi)) * g.pO + (long long)rowb * g.ldc;
        const int rows = g.M - rowb;  // valid rows of this wave's C::TM * 32
        const unsigned nrec = rows <= 0 ? 0u : (unsigned)(rows < C::TM * 32 ? rows : C::TM * 32) * (unsigned)g.ldc * 4u;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ob, 0, nrec, 0x00020000);
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j) {
                const int colu = colb + j * 32;
                const float bvj = g.bias ? bias_s[colu + l31 - n0] : 0.f;
                if (colu + l31 < g.N) {
                    const int soff = (i * 32 * g.ldc + colu) * 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[i][j][r] + bvj), ro, (int)o_voff[r], soff, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
    };
    int kdone = 0;          // K-steps finished inside the current tile
    int cu_i = 0, cu_o = 0;  // tile being computed
    int fl_i = 0, fl_o = 0;
    bool pending = false;
    for (int st = 0; st < steps; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < steps;
        if (more) advance();
        if (pending) {
            flush(fl_i, fl_o);
            pending = false;
        }
        const char* a = As + (buf * BM + wm * C::TM * 32 + l31) * C::ROW_BYTES + h * 16;
        const char* b = Bs + (buf * BN + wn * C::TN * 32 + l31) * C::ROW_BYTES + h * 16;
        // Pinned schedule: 16 groups of TM x TN MFMAs (one k pair each); the fragments of the next 8-k sub-step are read
        // while the current one multiplies, and ONE staging instruction rides behind each group — the loads of the next
        // K-step behind groups 0..NP-1, their LDS writes behind groups 16-NP..15 (>= 8 groups = 2k MFMA cycles later).
        // Left to itself hipcc puts all loads in front of the 64 MFMAs and all LDS writes behind them: the wave then spends
        // ~3k of every ~7k cycles outside the matrix pipe (1 wave/SIMD: 0.55 MFMA-busy, profiles/r02_pmc_gemm_*.txt).
        char* const st_a = As + ((buf ^ 1) * BM + row0) * C::ROW_BYTES + chunk * 16;  // LDS write addresses, formed once per K-step
        char* const st_b = Bs + ((buf ^ 1) * BN + row0) * C::ROW_BYTES + chunk * 16;
        auto store_piece_at = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q < C::A_PASSES)
                *reinterpret_cast<floatx4*>(st_a + q * C::A_ROWS * C::ROW_BYTES) = rs[q];
            else
                *reinterpret_cast<floatx4*>(st_b + (q - C::A_PASSES) * C::B_ROWS * C::ROW_BYTES) = rs[q];
        };
        float4 fa[2][C::TM], fb[2][C::TN];
        auto read_frags = [&](auto sbc) {
            constexpr int sb = decltype(sbc)::value;
#pragma unroll
            for (int i = 0; i < C::TM; ++i) fa[sb & 1][i] = *reinterpret_cast<const float4*>(a + i * 32 * C::ROW_BYTES + sb * 32);
#pragma unroll
            for (int j = 0; j < C::TN; ++j) fb[sb & 1][j] = *reinterpret_cast<const float4*>(b + j * 32 * C::ROW_BYTES + sb * 32);
        };
        read_frags(std::integral_constant<int, 0>{});
        static_for<16>([&](auto gic) {
            constexpr int gi = decltype(gic)::value, sb = gi >> 2, q = gi & 3, cur = sb & 1;
            if constexpr (q == 0 && sb < 3) read_frags(std::integral_constant<int, sb + 1>{});
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j) {
                    const float av = q == 0 ? fa[cur][i].x : q == 1 ? fa[cur][i].y : q == 2 ? fa[cur][i].z : fa[cur][i].w;
                    const float bv = q == 0 ? fb[cur][j].x : q == 1 ? fb[cur][j].y : q == 2 ? fb[cur][j].z : fb[cur][j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                }
            // (unconditional: a branch around the staging instructions splits the K-step into basic blocks, and the wait
            //  insertion then drains every load before the next one is issued.  The last K-step re-stages its own operands
            //  into the buffer nobody reads any more.)
            if constexpr (gi < NP) load_piece(std::integral_constant<int, gi>{});
            if constexpr (gi >= 16 - NP) store_piece_at(std::integral_constant<int, gi - (16 - NP)>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        if (++kdone == nk) {  // tile finished
            kdone = 0;
            fl_i = cu_i; fl_o = cu_o;
            if (++cu_i == g.n_inner) { cu_i = 0; ++cu_o; }
            pending = true;
        }
        __syncthreads();
    }
    flush(fl_i, fl_o);
}

// final_conv (DenoisingUNet_arch.py:76, 64 -> 3 channels): with Cout <= 4 an MFMA tile is > 90 % padding (8 TFLOP/s, 0.44 ms).
// Vector pipe instead: 16 lanes share a pixel, lane l owns channels 4l..4l+3 (one coalesced 256-byte row per tap), keeps its
// 27 weight float4s in registers, walks 8 consecutive pixels, and the 16 partial sums are reduced once per pixel.
// (One thread per pixel with its own 256-byte rows was 5x SLOWER than the MFMA kernel: 64 cache lines per load instruction.)
constexpr int kNarrowPPG = 8;  // pixels per 16-lane group
__global__ __launch_bounds__(256) void conv3x3_narrow_kernel(const ConvParams p, const int M) {
    const int l = threadIdx.x & 15;
    // XCD-aware walk: consecutive blocks (= neighbouring half rows) stay on one XCD, so the rows above / below a block's pixels come
    // out of the same L2 (round-robin placement had every input row fetched by ~4 XCDs: 1.05 GB of fabric reads for a 268 MB input)
    int wgid;
    {
        const int orig = blockIdx.x, nwg = gridDim.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int group = wgid * 16 + (threadIdx.x >> 4);
    const int C = p.C0;  // == 64
    float4 w[3][9];
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            w[co][t] = co < p.Cout ? *reinterpret_cast<const float4*>(p.w + ((size_t)co * 9 + t) * C + l * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < kNarrowPPG; ++i) {
        const int m = group * kNarrowPPG + i;
        const bool ok = m < M;  // all 16 lanes of a group agree; the shuffles below need every lane
        const int mm = ok ? m : 0;
        const int ox = mm % p.Wo, t1 = mm / p.Wo, oy = t1 % p.Ho, b = t1 / p.Ho;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy - 1 + ky, ix = ox - 1 + kx;
                const bool in = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                const float* g = in ? p.in0 + ((size_t)(b * p.Hin + iy) * p.Win + ix) * p.pix0 + l * 4 : p.zeros;
                const float4 v = *reinterpret_cast<const float4*>(g);
#pragma unroll
                for (int co = 0; co < 3; ++co) {
                    const float4 wv = w[co][ky * 3 + kx];
                    acc[co] = fmaf(v.x, wv.x, acc[co]);
                    acc[co] = fmaf(v.y, wv.y, acc[co]);
                    acc[co] = fmaf(v.z, wv.z, acc[co]);
                    acc[co] = fmaf(v.w, wv.w, acc[co]);
                }
            }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
#pragma unroll
            for (int co = 0; co < 3; ++co) acc[co] += __shfl_xor(acc[co], o, 64);
        if (ok && l < p.Cout) p.out[(size_t)m * p.out_stride + l] = (l == 0 ? acc[0] : l == 1 ? acc[1] : acc[2]) + (p.bias ? p.bias[l] : 0.f);
    }
}

// split-K second stage: sum partials, run the epilogue (memory-bound, tiny layers only)
__global__ void conv_splitk_reduce(const ConvParams p, const int M) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)M * p.Cout;
    if (idx >= total) return;
    const int n = (int)(idx % p.Cout);
    const int m = (int)(idx / p.Cout);
    float v = 0.f;
    for (int s = 0; s < p.splits; ++s) v += p.partial[(size_t)s * total + idx];
    if (p.bias) v += p.bias[n];
    if (p.film) {
        const float* f = p.film + (size_t)(p.film_bstride ? m / (p.Ho * p.Wo) : 0) * p.film_bstride;
        v = v * (f[n] + 1.0f) + f[p.Cout + n];
    }
    if (p.silu) v = silu_f(v);
    if (p.ch_scale) v *= p.ch_scale[n];
    if (p.out_bf16) {
        if (p.res) v += (float)reinterpret_cast<const __bf16*>(p.res)[(size_t)m * p.res_stride + n];
        reinterpret_cast<__bf16*>(p.out)[(size_t)m * p.out_stride + n] = (__bf16)v;
        return;
    }
    if (p.res) v += p.res[(size_t)m * p.res_stride + n];
    p.out[(size_t)m * p.out_stride + n] = v;
}

// Debug / cross-check path: direct convolution, one thread per output element (VALU fmaf chain).
__global__ void conv_naive_kernel(const ConvParams p, const int M) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * p.Cout) return;
    const int n = (int)(idx % p.Cout);
    const int m = (int)(idx / p.Cout);
    const int ox = m % p.Wo, t1 = m / p.Wo, oy = t1 % p.Ho, b = t1 / p.Ho;
    const int Ctot = p.C0 + p.C1;
    const int Hv = p.Hin << p.in_shift, Wv = p.Win << p.in_shift;
    float acc = 0.f;
    for (int ky = 0; ky < p.KH; ++ky)
        for (int kx = 0; kx < p.KW; ++kx) {
            const int iy = oy * p.stride - p.pad_y + ky, ix = ox * p.stride - p.pad_x + kx;
            if ((unsigned)iy >= (unsigned)Hv || (unsigned)ix >= (unsigned)Wv) continue;
            const size_t pixel = (size_t)b * p.Hin * p.Win + (size_t)(iy >> p.in_shift) * p.Win + (ix >> p.in_shift);
            const float* wr = p.w + ((size_t)n * p.KH * p.KW + ky * p.KW + kx) * Ctot;
            const float* s0 = p.in0 + pixel * p.pix0;
            for (int c = 0; c < p.C0; ++c) acc = fmaf(s0[c], wr[c], acc);
            if (p.C1) {
                const float* s1 = p.in1 + pixel * p.pix1;
                for (int c = 0; c < p.C1; ++c) acc = fmaf(s1[c], wr[p.C0 + c], acc);
            }
        }
    float v = acc;
    if (p.bias) v += p.bias[n];
    if (p.film) {
        const float* f = p.film + (size_t)(p.film_bstride ? b : 0) * p.film_bstride;
        v = v * (f[n] + 1.0f) + f[p.Cout + n];
    }
    if (p.silu) v = silu_f(v);
    if (p.res) v += p.res[(size_t)m * p.res_stride + n];
    p.out[(size_t)m * p.out_stride + n] = v;
}

int g_variant = 0;  // tuning experiments only (irsde_bench_conv); 6 = generic pointer staging instead of buffer descriptors

// PAIR kernels (split-operand arithmetic on fp32 storage): bf16 or fp16 pieces by p.f16
template <int BM, int BN, int WAVES_M, int WAVES_N, int MINW, bool INSCALE>
void launch_cfg_pair(const ConvParams& p, int M, int nk_total, hipStream_t s) {
    using C = Cfg<BM, BN, WAVES_M, WAVES_N, true, false, 2>;
    const int nblk_m = (M + BM - 1) / BM;
    const int nblk_n = (p.Cout + BN - 1) / BN;
    dim3 grid(nblk_m * nblk_n, p.splits, 1);
    if (p.f16)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, MINW, true, INSCALE, false, true, false, true>), grid, dim3(C::NT),
                           C::LDS_BYTES, s, p, nblk_n, M, nk_total);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, MINW, true, INSCALE, false, false, false, true>), grid, dim3(C::NT),
                           C::LDS_BYTES, s, p, nblk_n, M, nk_total);
    IRSDE_HIP_CHECK(hipGetLastError());
}
template <int BM, int BN, int WAVES_M, int WAVES_N, int MINW, bool INSCALE>
void init_cfg_pair() {
    IRSDE_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, MINW, true, INSCALE, false, true, false, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, MINW, true, INSCALE, false, false, false, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MINW, bool BF16, bool INSCALE = false, bool ABF = false, bool F16 = false>
void launch_cfg(const ConvParams& p, int M, int nk_total, hipStream_t s, int lds_override = 0) {
    using C = Cfg<BM, BN, WAVES_M, WAVES_N, BF16, ABF>;
    if constexpr (BF16 && !ABF && !F16) {
        if (p.f16) return launch_cfg<BM, BN, WAVES_M, WAVES_N, MINW, true, INSCALE, false, true>(p, M, nk_total, s, lds_override);
    }
    auto kern = conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, MINW, BF16, INSCALE, ABF, F16>;
    if constexpr (!BF16) {  // f32: the buffer-descriptor staging when every operand tensor is below 2 GiB
        const long long npix = (long long)p.B * p.Hin * p.Win;
        const long long e0 = ((npix - 1) * p.pix0 + p.C0) * 4, e1 = p.C1 ? ((npix - 1) * p.pix1 + p.C1) * 4 : 0;
        const long long ew = (long long)p.Cout * p.KH * p.KW * (p.C0 + p.C1) * 4;
        static const int nobuf = tuning_env_int("IRSDE_NO_BUFA", 0);
        if (!nobuf && e0 < (1ll << 31) && e1 < (1ll << 31) && ew < (1ll << 31) && g_variant != 6)
            kern = conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, MINW, BF16, INSCALE, ABF, F16, true>;
    }
    const int nblk_m = (M + BM - 1) / BM;
    const int nblk_n = (p.Cout + BN - 1) / BN;
    dim3 grid(nblk_m * nblk_n, p.splits, p.nz);
    ConvParams pk = p;
    static const int no_direct = tuning_env_int("IRSDE_NO_DIRECT_EPI", 0);
    if (no_direct || g_variant == 7) pk.no_direct_epi = 1;
    hipLaunchKernelGGL(kern, grid, dim3(C::NT), lds_override ? lds_override : C::LDS_BYTES, s, pk, nblk_n, M, nk_total);
    IRSDE_HIP_CHECK(hipGetLastError());
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MINW, bool BF16, bool INSCALE = false, bool ABF = false>
void init_cfg() {
    IRSDE_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, MINW, BF16, INSCALE, ABF>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if constexpr (BF16 && !ABF)  // the fp16 twin of every bf16-operand kernel
        IRSDE_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, MINW, true, INSCALE, false, true>),
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if constexpr (!BF16)  // the buffer-descriptor twin of every f32 kernel
        IRSDE_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, MINW, false, INSCALE, false, false, true>),
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}


// 1x1 convolutions with a short K on the tile-loop kernel: row tiles per block, or 0 = generic kernel.
int zloop_1x1_rows(const ConvParams& p) {
    if (p.nz != 1 || p.w_bf || p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad_y || p.pad_x || p.in_shift || p.film || p.silu ||
        p.res || p.splits != 1 || p.gate || p.shuffle || p.ch_scale || p.in_scale || p.ln_g || p.in_bf16 || p.out_bf16)
        return 0;
    static const int env = tuning_env_int("IRSDE_ZLOOP_1X1", -1);  // tuning: 0 = off, n = rows tiles
    if (g_variant == 70 || env == 0) return 0;
    const int M = p.B * p.Ho * p.Wo, K = p.C0 + p.C1;
    const int mtiles = (M + 127) / 128, ntiles = (p.Cout + 127) / 128, nk = K / 32;
    if (g_variant == 73) return 2;  // test hook: force the path on small shapes
    if (env > 0) return env;
    if (nk > 12 || mtiles < 1024) return 0;  // long K amortises the per-tile overhead already; small M: not enough blocks
    int mb = 1;
    while (mb < 8 && mtiles / (mb * 2) >= 1024 && mb * ntiles * nk < 24) mb *= 2;
    return mb;
}

// few rows (single image, deep levels): 64 x 128 tiles — a 128-row tile would be half empty or the grid under-filled
bool zloop_small_m(const ConvParams& p) {
    const long long b128 = (long long)((p.Wo + 127) / 128) * ((p.Cout + 127) / 128) * p.nz;
    if (p.Wo <= 64 || b128 < 512) return true;
    // r05: a ragged last round of 128-row tiles (the small-batch shards: 512 tiles x 512 channels x 36 components = 576 blocks on 512 block slots — one full round
    // and 64 blocks alone) against 64-row tiles (1152 blocks: three rounds of half the work, ~7 % less efficient per tile); only where rounds are few
    // (measured, one box: B = 2 / 4 / 8 x 256^2 3.11 / 3.68 / 4.03 -> 3.16 / 3.77 / 4.07 img/s; the T = 512 layers 98 -> 111 TFLOP/s.)  IRSDE_ZLOOP_RAGGED64 = the block
    // count below which the rule applies (0 = off)
    static const int cutoff = tuning_env_int("IRSDE_ZLOOP_RAGGED64", 2048);
    if (cutoff <= 0 || b128 >= cutoff) return false;
    const long long b64 = (long long)((p.Wo + 63) / 64) * ((p.Cout + 127) / 128) * p.nz;
    const long long slots = 2ll * device_cu_count();   // resident block slots: two 256-thread blocks per CU
    const double e128 = (double)b128 / ((double)slots * (double)((b128 + slots - 1) / slots));
    const double e64 = 0.93 * (double)b64 / ((double)slots * (double)((b64 + slots - 1) / slots));
    return e64 > e128;
}

// components per block for gemm_zloop_kernel, or 0 = use one block per (component, tile)
int zloop_batch(const ConvParams& p) {
    if (p.nz <= 1 || p.w_bf || p.KH != 1 || p.KW != 1 || p.C1 || p.bias || p.film || p.silu || p.res || p.splits != 1 ||
        p.gate || p.shuffle || p.ch_scale || p.in_scale || p.stride != 1 || p.out_stride != p.Cout || p.pix0 != p.C0 ||
        p.B != 1 || p.Ho != 1)
        return 0;
    static const int env_force = tuning_env_int("IRSDE_ZLOOP", -1);  // tuning: 0 = off, n = fixed batch
    // test hooks: variant 71 = all components in one block, 72 = batches of 2 (F2: 16, F4: 36 components), 70 = off
    const int force = g_variant == 70 ? 0 : g_variant == 71 ? p.nz : g_variant == 72 ? 2 : env_force;
    if (force == 0) return 0;
    const int nk = p.C0 / 32;
    const long long tiles = (long long)((p.Wo + 127) / 128) * ((p.Cout + 127) / 128);
    if (force > 0) return p.nz % force == 0 ? force : 0;
    if (zloop_small_m(p)) return nk >= 4 ? 1 : 0;  // single-image deep levels: 64-row tiles, one component per block
    static const int max_nk = tuning_env_int("IRSDE_ZLOOP_MAXNK", 32);
    static const int min_blocks = tuning_env_int("IRSDE_ZLOOP_MINBLK", 1024);
    if (nk > max_nk) return 0;  // long K: the per-block overhead is already amortised (measured: 12 / 32 / 48 -> 2.82 / 2.87 / 2.83 img/s)
    static const int deep_rule = tuning_env_int("IRSDE_ZLOOP_DEEP", 1);
    if (deep_rule && nk > 16 && (p.Wo + 127) / 128 < 32) return 0;  // few row tiles + long K (32x32 level): generic kernel is ahead
    int best = 0;
    for (int zb = p.nz; zb >= 2; --zb) {  // largest batch that still fills the 2 x 256 block slots evenly
        if (p.nz % zb) continue;
        const long long blocks = tiles * (p.nz / zb);
        if (blocks >= 512 && (blocks % 512 == 0 || blocks >= min_blocks)) { best = zb; break; }
    }
    return best;
}

}  // namespace

void conv_set_variant(int v) { g_variant = v; }

// CU count of the CURRENT device, cached per device ordinal (a process may hold parts with different CU counts)
int device_cu_count() {
    static std::mutex mu;
    static std::vector<int> cache;
    int dev = 0;
    IRSDE_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if ((int)cache.size() <= dev) cache.resize(dev + 1, 0);
    if (cache[dev] == 0) {
        int n = 0;
        IRSDE_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        cache[dev] = n > 0 ? n : 256;
    }
    return cache[dev];
}

void conv_global_init() {
    conv_halo_global_init();
    wino_fused_global_init();
    wino_fused_t_global_init();
    naf_chain_global_init();
    attention_global_init();
    naf_lnconv_global_init();
    gemm_split_global_init();
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_zloop_kernel<128, 128, 2, 2, 2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_zloop_kernel<128, 64, 2, 2, 2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_zloop_kernel<64, 128, 1, 4, 2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    init_cfg_pair<256, 256, 2, 4, 2, false>();
    init_cfg_pair<256, 256, 2, 4, 2, true>();
    init_cfg_pair<256, 128, 4, 2, 2, false>();
    init_cfg_pair<256, 128, 4, 2, 2, true>();
    init_cfg_pair<128, 64, 2, 2, 2, false>();
    init_cfg_pair<128, 64, 2, 2, 2, true>();
    init_cfg<128, 128, 2, 2, 2, false>();
    init_cfg<128, 64, 2, 2, 2, false>();
    init_cfg<128, 32, 4, 1, 2, false>();
    init_cfg<256, 128, 4, 2, 2, false>();
    init_cfg<256, 256, 2, 4, 2, false>();
    init_cfg<128, 128, 2, 2, 2, true>();
    init_cfg<128, 64, 2, 2, 2, true>();
    init_cfg<128, 32, 4, 1, 2, true>();
    init_cfg<256, 256, 2, 4, 2, true>();
    init_cfg<128, 128, 2, 2, 2, true, false, true>();
    init_cfg<128, 64, 2, 2, 2, true, false, true>();
    init_cfg<128, 32, 4, 1, 2, true, false, true>();
    init_cfg<256, 256, 2, 4, 2, true, false, true>();
    init_cfg<128, 128, 2, 2, 2, false, true>();
    init_cfg<128, 64, 2, 2, 2, false, true>();
    init_cfg<128, 32, 4, 1, 2, false, true>();
    init_cfg<128, 128, 2, 2, 2, true, true>();
    init_cfg<128, 64, 2, 2, 2, true, true>();
    init_cfg<128, 32, 4, 1, 2, true, true>();
}

double conv_flops(const ConvParams& p) {
    return 2.0 * (double)p.B * p.Ho * p.Wo * (double)p.Cout * (double)(p.KH * p.KW) * (double)(p.C0 + p.C1);
}

// 256x256 tiles (8 waves, 128x64 per wave) halve the staging instructions per MFMA (+5..8 % on deep layers) but
// need Cout % 256 == 0 and a tile count that fills the 256 CUs without a ragged last round.
bool conv_use_tile256(int M, int Cout, int splits, int nk_total) {
    if (splits != 1 || Cout % 256) return false;
    const long long nb = (long long)((M + 255) / 256) * (Cout / 256);
    return nb % 256 == 0 || nb >= 1024;
}

// r06 (measurement knob, default off): the under-filled-grid rule of the 16-bit-operand kernels for the f32 direct kernels — column-tile width 128 / 64 / 32
static int f32_narrow(const ConvParams& p, int M) {
    int bn = p.Cout >= 128 ? 128 : p.Cout > 32 ? 64 : 32;
    static const int k = tuning_env_int("IRSDE_SMALL_BN_F32", 0);
    if (k && g_variant == 0) {
        const long long slots = (long long)k * device_cu_count();
        auto blocks = [&](int n) { return (long long)((M + 127) / 128) * ((p.Cout + n - 1) / n) * p.splits * p.nz; };
        while (bn > 32 && blocks(bn) < slots) bn >>= 1;
    }
    return bn;
}

void launch_conv(const ConvParams& p, hipStream_t s) {
    const int M = p.B * p.Ho * p.Wo;
    const int Ctot = p.C0 + p.C1;
    if (p.C0 % 32 || p.C1 % 32 || Ctot == 0)
        throw HipError("launch_conv: channel counts must be multiples of 32 (got " + std::to_string(p.C0) + "+" +
                       std::to_string(p.C1) + ")");
    if (p.splits > 1 && !p.partial) throw HipError("launch_conv: split-K needs a partial buffer");
    if (p.ln_g && (p.splits != 1 || (p.Cout != 64 && p.Cout != 128) || p.nz != 1 || (p.out_stride & 3) || (p.res && (p.res_stride & 3))))
        throw HipError("launch_conv: fused LayerNorm needs Cout == 64 or 128 in one tile, no split-K");
    if (!p.zeros) throw HipError("launch_conv: ConvParams::zeros (zero page for out-of-image taps) is not set");
    if (p.w_pair) {   // split-operand arithmetic (IRSDE_FLAG_SPLIT_BF16X2 / _F16X2): the PAIR kernels on fp32 storage
        if (p.w_bf || p.in_bf16 || p.out_bf16 || p.nz != 1) throw HipError("launch_conv: split-operand pairs go with fp32 storage, one component");
        if (p.in_scale && p.C1) throw HipError("launch_conv: in_scale needs a single source");
        if (p.ln_g) throw HipError("launch_conv: the fused LayerNorm epilogue (BN == Cout) is not available on the PAIR tiles");
        const int nk = p.KH * p.KW * (Ctot / 32);
        static const int t256 = tuning_env_int("IRSDE_PAIR_TILE256", 1);
        // 256 x 256: half the staging work per MFMA (160 KB of LDS: one block per CU) — where it still fills the 256 CUs
        // (the INSCALE instance of that tile spills since the two-step A staging: SCA-scaled layers take 256 x 128 unless IRSDE_PAIR_INSCALE256=1)
        static const int inscale256 = tuning_env_int("IRSDE_PAIR_INSCALE256", 0);
        // (measured and dropped: a 128 x 128 tile at two blocks per CU, so that one block's write-back overlaps the other's K loop — 10-20 %
        //  SLOWER on every layer class, profiles/r03_pair_conv_sweep.txt: the doubled L2 -> CU operand traffic costs more than the overlap buys)
        if (t256 && (!p.in_scale || inscale256) && p.Cout % 256 == 0 && (long long)((M + 255) / 256) * (p.Cout / 256) * p.splits >= 256 && g_variant != 61) {
            if (p.in_scale) launch_cfg_pair<256, 256, 2, 4, 2, true>(p, M, nk, s);
            else launch_cfg_pair<256, 256, 2, 4, 2, false>(p, M, nk, s);
        } else if (p.Cout >= 128 && M >= 256) {
            if (p.in_scale) launch_cfg_pair<256, 128, 4, 2, 2, true>(p, M, nk, s);
            else launch_cfg_pair<256, 128, 4, 2, 2, false>(p, M, nk, s);
        } else {
            if (p.in_scale) launch_cfg_pair<128, 64, 2, 2, 2, true>(p, M, nk, s);
            else launch_cfg_pair<128, 64, 2, 2, 2, false>(p, M, nk, s);
        }
        if (p.splits > 1) {
            const size_t total = (size_t)M * p.Cout;
            hipLaunchKernelGGL(conv_splitk_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, M);
            IRSDE_HIP_CHECK(hipGetLastError());
        }
        return;
    }
    if (!p.w_bf && g_variant == 0 && p.Cout <= 3 && p.C0 == 64 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_y == 1 && p.pad_x == 1 &&
        !p.in_shift && !p.C1 && !p.film && !p.silu && !p.res && p.splits == 1 && p.nz == 1 && !p.gate && !p.shuffle &&
        !p.ch_scale && !p.in_scale && !p.ln_g && M >= 65536) {  // final_conv: vector-pipe kernel (big feature maps only)
        hipLaunchKernelGGL(conv3x3_narrow_kernel, dim3((M + 16 * kNarrowPPG - 1) / (16 * kNarrowPPG)), dim3(256), 0, s, p, M);
        IRSDE_HIP_CHECK(hipGetLastError());
        return;
    }
    if (const int zb = zloop_batch(p)) {  // short-K Winograd component GEMMs: component loop inside the block
        using C = Cfg<128, 128, 2, 2, false>;
        ZLoopArgs g{};
        g.a0 = p.in0; g.a1 = nullptr; g.C0 = p.C0; g.C1 = 0; g.lda0 = p.C0; g.lda1 = 0;
        g.b = p.w; g.out = p.out; g.bias = nullptr;
        g.M = p.Wo; g.N = p.Cout; g.K = p.C0; g.ldc = p.Cout;
        g.n_inner = zb; g.n_outer = 1; g.pA = p.z_in; g.pB = p.z_w; g.pO = p.z_out; g.col_step = 0; g.row_step = 0;
        g.nblk_n = (p.Cout + 127) / 128;
        if (zloop_small_m(p)) {  // 64 x 128 tiles, 4 waves of 64 x 32 (all four SIMDs busy)
            using C64 = Cfg<64, 128, 1, 4, false>;
            dim3 grid64(((p.Wo + 63) / 64) * g.nblk_n, p.nz / zb);
            hipLaunchKernelGGL((gemm_zloop_kernel<64, 128, 1, 4, 2>), grid64, dim3(C64::NT), C64::MAIN_BYTES, s, g);
            IRSDE_HIP_CHECK(hipGetLastError());
            return;
        }
        if (p.Cout <= 64) {  // 128 x 64 tiles: a 128-wide tile would be half padding (the 64-channel full-resolution layers)
            using C64n = Cfg<128, 64, 2, 2, false>;
            g.nblk_n = 1;
            dim3 grid_n((p.Wo + 127) / 128, p.nz / zb);
            hipLaunchKernelGGL((gemm_zloop_kernel<128, 64, 2, 2, 2>), grid_n, dim3(C64n::NT), C64n::MAIN_BYTES, s, g);
            IRSDE_HIP_CHECK(hipGetLastError());
            return;
        }
        dim3 grid(((p.Wo + 127) / 128) * g.nblk_n, p.nz / zb);
        hipLaunchKernelGGL((gemm_zloop_kernel<128, 128, 2, 2, 2>), grid, dim3(C::NT), C::MAIN_BYTES, s, g);
        IRSDE_HIP_CHECK(hipGetLastError());
        return;
    }
    if (const int mb = zloop_1x1_rows(p)) {  // short-K 1x1 convolutions: (row tiles x all column tiles) loop inside the block
        using C = Cfg<128, 128, 2, 2, false>;
        const int M = p.B * p.Ho * p.Wo;
        ZLoopArgs g{};
        g.a0 = p.in0; g.a1 = p.C1 ? p.in1 : nullptr; g.C0 = p.C0; g.C1 = p.C1; g.lda0 = p.pix0; g.lda1 = p.pix1;
        g.b = p.w; g.out = p.out; g.bias = p.bias;
        g.M = M; g.N = p.Cout; g.K = p.C0 + p.C1; g.ldc = p.out_stride;
        g.n_inner = (p.Cout + 127) / 128; g.n_outer = mb; g.pA = g.pB = g.pO = 0; g.col_step = 128; g.row_step = 128;
        g.nblk_n = 1;
        const int mtiles = (M + 127) / 128;
        dim3 grid((mtiles + mb - 1) / mb, 1);
        if (p.Cout <= 64) {  // one 64-wide column tile
            using C64n = Cfg<128, 64, 2, 2, false>;
            g.col_step = 64;
            hipLaunchKernelGGL((gemm_zloop_kernel<128, 64, 2, 2, 2>), grid, dim3(C64n::NT),
                               C64n::MAIN_BYTES + (p.bias ? 64 * 4 : 0), s, g);
            IRSDE_HIP_CHECK(hipGetLastError());
            return;
        }
        const int lds = C::MAIN_BYTES + (p.bias ? g.n_inner * 128 * 4 : 0);  // + the block's bias columns
        hipLaunchKernelGGL((gemm_zloop_kernel<128, 128, 2, 2, 2>), grid, dim3(C::NT), lds, s, g);
        IRSDE_HIP_CHECK(hipGetLastError());
        return;
    }
    const int nk_total = p.KH * p.KW * (Ctot / 32);
    if ((p.in_bf16 || p.out_bf16) && !p.w_bf) throw HipError("launch_conv: bf16 activation storage needs the bf16-MFMA mode");
    if (p.f16 && (!p.w_bf || p.in_bf16 || p.out_bf16)) throw HipError("launch_conv: fp16 operands go with fp32 activation storage");
    if (p.in_bf16 && (p.in_scale || p.gate || p.shuffle)) throw HipError("launch_conv: NAFNet fusions are fp32-storage only");
    if (p.in_scale) {  // NAFNet SCA fused into the staging (128-row tiles only)
        if (p.C1) throw HipError("launch_conv: in_scale needs a single source");
        if (p.w_bf) {
            int bn = p.Cout >= 128 ? 128 : p.Cout > 32 ? 64 : 32;
            static const int small_bn = tuning_env_int("IRSDE_SMALL_BN", 2);   // block slots per CU the grid should fill (0 = rule off)
            if (small_bn && g_variant == 0) {   // (see the rule at the plain 16-bit-operand branch below)
                const long long slots = (long long)small_bn * device_cu_count();
                auto blocks = [&](int n) { return (long long)((M + 127) / 128) * ((p.Cout + n - 1) / n) * p.splits * p.nz; };
                while (bn > 32 && blocks(bn) < slots) bn >>= 1;
            }
            if (bn == 128) launch_cfg<128, 128, 2, 2, 2, true, true>(p, M, nk_total, s);
            else if (bn == 64) launch_cfg<128, 64, 2, 2, 2, true, true>(p, M, nk_total, s);
            else launch_cfg<128, 32, 4, 1, 2, true, true>(p, M, nk_total, s);
        } else {
            if (p.Cout >= 128) launch_cfg<128, 128, 2, 2, 2, false, true>(p, M, nk_total, s);
            else if (p.Cout > 32) launch_cfg<128, 64, 2, 2, 2, false, true>(p, M, nk_total, s);
            else launch_cfg<128, 32, 4, 1, 2, false, true>(p, M, nk_total, s);
        }
    } else if (p.w_bf && g_variant != 60 && g_variant != 61 && conv_halo_eligible(p)) {
        launch_conv_halo(p, s, g_variant == 64 ? 1 : g_variant == 65 ? -1 : 0);  // 3x3 s1 p1: LDS-resident halo tile (conv_halo.hip)
        return;
    } else if (p.w_bf && p.in_bf16) {  // bf16 operands AND bf16 activation storage
        if (p.Cout >= 128) {
            if (g_variant != 61 && (g_variant == 60 || conv_use_tile256(M, p.Cout, p.splits, nk_total)))
                launch_cfg<256, 256, 2, 4, 2, true, false, true>(p, M, nk_total, s);
            else
                launch_cfg<128, 128, 2, 2, 2, true, false, true>(p, M, nk_total, s);
        } else if (p.Cout > 32) {
            launch_cfg<128, 64, 2, 2, 2, true, false, true>(p, M, nk_total, s);
        } else {
            launch_cfg<128, 32, 4, 1, 2, true, false, true>(p, M, nk_total, s);
        }
    } else if (p.w_bf) {  // bf16 operands, fp32 accumulation
        // r06: an under-filled grid of 16-bit-operand tiles is latency-bound — one block per CU keeps ~24 KB of loads in flight against 2 us of cross-XCD
        // latency (the small-batch NAFNet levels of configs[4]: every 1x1 layer took 17-19 us whatever its size) — narrower column tiles put 2-4 blocks on a CU
        // (the A rows are re-read from L2).  IRSDE_SMALL_BN = block slots per CU to fill (0 switches the rule off).
        int bn = p.Cout >= 128 ? 128 : p.Cout > 32 ? 64 : 32;
        static const int small_bn = tuning_env_int("IRSDE_SMALL_BN", 2);   // block slots per CU the grid should fill (0 = rule off)
        if (small_bn && g_variant == 0) {
            const long long slots = (long long)small_bn * device_cu_count();
            auto blocks = [&](int n) { return (long long)((M + 127) / 128) * ((p.Cout + n - 1) / n) * p.splits * p.nz; };
            while (bn > 32 && blocks(bn) < slots) bn >>= 1;
        }
        if (bn == 128) {
            if (g_variant != 61 && (g_variant == 60 || conv_use_tile256(M, p.Cout, p.splits, nk_total)))
                launch_cfg<256, 256, 2, 4, 2, true>(p, M, nk_total, s);
            else
                launch_cfg<128, 128, 2, 2, 2, true>(p, M, nk_total, s);
        } else if (bn == 64) {
            launch_cfg<128, 64, 2, 2, 2, true>(p, M, nk_total, s);
        } else {
            launch_cfg<128, 32, 4, 1, 2, true>(p, M, nk_total, s);
        }
    } else if (p.Cout >= 128 && f32_narrow(p, M) == 128) {
        if (g_variant == 3)
            launch_cfg<256, 128, 4, 2, 2, false>(p, M, nk_total, s);
        else if (g_variant == 50)
            launch_cfg<256, 256, 2, 4, 2, false>(p, M, nk_total, s);
        else if (g_variant == 5)
            launch_cfg<128, 128, 2, 2, 2, false>(p, M, nk_total, s, 120 * 1024);  // diagnostic: force 1 block/CU
        else if (g_variant == 0 && conv_use_tile256(M, p.Cout, p.splits, nk_total))
            launch_cfg<256, 256, 2, 4, 2, false>(p, M, nk_total, s);
        else
            launch_cfg<128, 128, 2, 2, 2, false>(p, M, nk_total, s);
    } else if (p.Cout > 32 && f32_narrow(p, M) >= 64) {
        launch_cfg<128, 64, 2, 2, 2, false>(p, M, nk_total, s);
    } else {
        launch_cfg<128, 32, 4, 1, 2, false>(p, M, nk_total, s);
    }
    if (p.splits > 1) {
        const size_t total = (size_t)M * p.Cout;
        hipLaunchKernelGGL(conv_splitk_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, M);
        IRSDE_HIP_CHECK(hipGetLastError());
    }
}

void launch_conv_naive(const ConvParams& p, hipStream_t s) {
    const int M = p.B * p.Ho * p.Wo;
    const size_t total = (size_t)M * p.Cout;
    hipLaunchKernelGGL(conv_naive_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, M);
    IRSDE_HIP_CHECK(hipGetLastError());
}

}  // namespace irsde
