// Implicit-GEMM NHWC convolution for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: D = A(32x2) * B(2x32) + C, one f32 A and one f32 B operand per lane,
//  bit-for-bit an fmaf chain; 64 cycles per instruction per SIMD = the fp32 peak of 157 TFLOP/s).
//
// Replaces every nn.Conv2d of the reference score network (3x3, 1x1, 4x4 s2, 7x7 and the
// nearest-upsample + 3x3 pair): reference call sites module_util.py:93-105 (Upsample/Downsample/
// default_conv), :108-122 (Block), :150-161 (to_qkv / to_out), DenoisingUNet_arch.py:27,76.
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*(C0+C1).  Both operands are
// K-contiguous in HBM (NHWC activations, [Cout][KH][KW][Cin] weights), so both LDS tiles are
// [rows][BK] with a 16-byte pad per row (row stride 36 floats = 9 x 16-B slots, odd => the
// 16-lane groups of ds_read_b128 hit 16 distinct slots: conflict-free).  A lane reads 4 consecutive
// k with one ds_read_b128 and feeds 4 MFMAs; lane half h owns k = 8*sb + 4*h + {0..3}, the same
// assignment for A and B, so the k-order inside the MFMA chain is a permutation (legal: sum over k).
//
// K loop: taps outer / 32-channel chunks inner.  The per-row pixel offset of a tap is computed once
// per tap (not per K-step); a K-step then costs one 64-bit mad + one load per staged row.
// global->register->LDS staging is split: the loads of K-step t+1 are issued before the MFMAs of
// step t and written to the other LDS buffer after them; one barrier per K-step.
//
// Epilogue: the accumulator tile is transposed through LDS (the A/B buffers are dead by then) so
// that every lane handles 4 consecutive output channels: bias / FiLM / residual / output move as
// 16-byte accesses, 512 contiguous bytes per output pixel row of a 128-wide tile.
#include "common.h"

namespace irsde {

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int DMA = 0>
struct Cfg {
    static constexpr int NT = 64 * WAVES_M * WAVES_N;
    // register staging: rows padded by 16 B (conflict-free b128 reads); LDS-DMA staging: dense 128-B rows,
    // 16-B chunks XOR-swizzled by ((row >> 1) & 7) on the SOURCE side (the DMA destination is lane-linear)
    static constexpr int LDS_K = DMA ? BK : BK + 4;
    static constexpr int TM = BM / WAVES_M / 32;
    static constexpr int TN = BN / WAVES_N / 32;
    static constexpr int CHUNKS = BK / 4;
    static constexpr int ROWS = NT / CHUNKS;
    static constexpr int A_PASSES = BM / ROWS;
    static constexpr int B_PASSES = (BN + ROWS - 1) / ROWS;
    static constexpr int LDS_C = BN + 4;  // epilogue tile row stride (floats): 16-B aligned, odd number of slots
    static constexpr int MAIN_BYTES = 2 * (BM + BN) * LDS_K * 4;
    static constexpr int EPI_ROWS = BM > 128 ? 128 : BM;  // the epilogue transposes EPI_ROWS tile rows per pass
    static constexpr int EPI_BYTES = EPI_ROWS * LDS_C * 4;
    static_assert((BM / WAVES_M) <= EPI_ROWS && EPI_ROWS % (BM / WAVES_M) == 0, "wave rows must tile an epilogue pass");
    static constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
    static_assert(BM % ROWS == 0, "tile/pass mismatch");
    static_assert(BN % ROWS == 0 || BN < ROWS, "tile/pass mismatch");
};

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }

// MFMA with the accumulator pinned to the AGPR half of the register file (experiment: keeps the 16-register
// C/D traffic off the arch-VGPR ports that LDS stores / VMEM address reads use).
template <bool AGPR>
__device__ __forceinline__ void mfma32(floatx16& acc, float a, float b) {
    if (AGPR) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int MIN_WAVES_PER_SIMD, int SCHED = 0, int DMA = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MIN_WAVES_PER_SIMD) void conv_igemm_kernel(
    const ConvParams pin, const int nblk_n, const int M, const int nk_total) {
    using C = Cfg<BM, BN, WAVES_M, WAVES_N, BK, DMA>;
    static_assert(!DMA || BK == 32, "LDS-DMA staging assumes 128-byte rows");
    ConvParams p = pin;  // batched launch: component blockIdx.z works on its own slice of in0 / w / out
    if (pin.nz > 1) {
        const long long z = blockIdx.z;
        p.in0 = pin.in0 + z * pin.z_in;
        p.w = pin.w + z * pin.z_w;
        p.out = pin.out + z * pin.z_out;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * C::LDS_K;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // Two blocks share a CU (one wave of each per SIMD).  Running identical code they phase-lock: they
    // share the MFMA pipe evenly, so both leave the MFMA phase together and both sit in the staging /
    // barrier gap together, idling the pipe (measured: 18 % idle).  Breaking the symmetry with a static
    // priority by hardware wave slot (HW_REG_HW_ID.WAVE_ID parity: co-resident waves of one SIMD hold
    // different slots) lets one block run its MFMA phase at full rate while the other fills its gaps.
    if (SCHED & 2) {
        const unsigned hwid = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_ID[3:0] = wave slot
        if (hwid & 1) __builtin_amdgcn_s_setprio(2);
    }
    if (SCHED & 8 && !(SCHED & 4)) {  // experiment: initial half-period phase offset for odd wave slots
        const unsigned hwid = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
        if (hwid & 1)
            for (int i = 0; i < 40; ++i) __builtin_amdgcn_s_sleep(1);
    }

    // XCD-aware block remap (bijective): each XCD (block id % 8) walks a contiguous range of tiles so
    // that neighbouring tiles (same activation rows, other Cout slices / halo rows) share one L2.
    int wgid;
    {
        const int orig = blockIdx.x, nwg = gridDim.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
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
    const int chunk = tid % C::CHUNKS;
    const int row0 = tid / C::CHUNKS;
    // logical 16-B chunk of the K slice this lane fetches (DMA: inverse of the read-side swizzle)
    const int cchunk = DMA ? (chunk ^ ((row0 >> 1) & 7)) : chunk;
    int a_iy0[C::A_PASSES], a_ix0[C::A_PASSES], a_pix[C::A_PASSES], a_b[C::A_PASSES];
#pragma unroll
    for (int ps = 0; ps < C::A_PASSES; ++ps) {
        const int m = m0 + row0 + ps * C::ROWS;
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
    const float* wrow[C::B_PASSES];
    bool b_ok[C::B_PASSES];
#pragma unroll
    for (int ps = 0; ps < C::B_PASSES; ++ps) {
        const int r = row0 + ps * C::ROWS;
        const int n = n0 + r;
        b_ok[ps] = n < p.Cout && r < BN;
        wrow[ps] = p.w + (size_t)(b_ok[ps] ? n : 0) * taps * Ctot + cchunk * 4;
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

    // ---- staging pieces: A rows (A_PASSES) then B rows (B_PASSES), one 16-byte load/store each ----
    constexpr int NP = C::A_PASSES + C::B_PASSES;
    float4 rs[NP];
    const float* cur_src = nullptr;  // source pointer (+channel +chunk) of the K-step being staged
    int cur_pix = 0;
    size_t cur_wk = 0;
    auto stage_setup = [&]() {
        const float* src;
        int c;
        if (cc < p.C0) {
            src = p.in0; c = cc; cur_pix = p.pix0;
        } else {
            src = p.in1; c = cc - p.C0; cur_pix = p.pix1;
        }
        cur_src = src + c + cchunk * 4;
        cur_wk = (size_t)tap * Ctot + cc;
    };
    // LDS-DMA: one global_load_lds_dwordx4 per piece writes 64 lanes x 16 B = 8 rows x 128 B straight into the
    // LDS tile (destination = wave-uniform base + lane*16); out-of-image taps / rows read a zero page.
    auto dma_piece = [&](int q, int buf) {
        if (q < C::A_PASSES) {
            const bool ok = a_poff[q] >= 0;
            const float* g = ok ? cur_src + (size_t)a_poff[q] * cur_pix : p.zeros;
            float* dst = As + buf * BM * C::LDS_K + (q * C::ROWS + wave * (64 / C::CHUNKS)) * C::LDS_K;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)dst, 16, 0, 0);
        } else {
            const int ps = q - C::A_PASSES;
            if (C::B_PASSES * C::ROWS == BN || ps * C::ROWS + wave * (64 / C::CHUNKS) < BN) {
                const float* g = b_ok[ps] ? wrow[ps] + cur_wk : p.zeros;
                float* dst = Bs + buf * BN * C::LDS_K + (ps * C::ROWS + wave * (64 / C::CHUNKS)) * C::LDS_K;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)dst, 16, 0, 0);
            }
        }
    };
    auto load_piece = [&](int q) {
        if (q < C::A_PASSES) {
            const bool ok = a_poff[q] >= 0;  // branch-free: out-of-image taps read pixel 0 and are zeroed
            float4 v = *reinterpret_cast<const float4*>(cur_src + (size_t)(ok ? a_poff[q] : 0) * cur_pix);
            if (p.in_scale) {  // per-(batch, input channel) scale applied while staging (NAFNet SCA), single source
                const float4 sc4 = *reinterpret_cast<const float4*>(p.in_scale + (size_t)a_b[q] * p.C0 + cc + cchunk * 4);
                v.x *= sc4.x; v.y *= sc4.y; v.z *= sc4.z; v.w *= sc4.w;
            }
            rs[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const int ps = q - C::A_PASSES;
            const float4 v = *reinterpret_cast<const float4*>(wrow[ps] + cur_wk);
            rs[q] = b_ok[ps] ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_piece = [&](int q, int buf) {
        if (q < C::A_PASSES) {
            *reinterpret_cast<float4*>(As + buf * BM * C::LDS_K + (row0 + q * C::ROWS) * C::LDS_K + chunk * 4) = rs[q];
        } else {
            const int ps = q - C::A_PASSES;
            if (C::B_PASSES * C::ROWS == BN || row0 + ps * C::ROWS < BN)
                *reinterpret_cast<float4*>(Bs + buf * BN * C::LDS_K + (row0 + ps * C::ROWS) * C::LDS_K + chunk * 4) =
                    rs[q];
        }
    };

    floatx16 acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt_begin < kt_end) {
        set_tap();
        stage_setup();
        if (DMA) {
#pragma unroll
            for (int q = 0; q < NP; ++q) dma_piece(q, 0);
        } else {
#pragma unroll
            for (int q = 0; q < NP; ++q) load_piece(q);
#pragma unroll
            for (int q = 0; q < NP; ++q) store_piece(q, 0);
        }
    }
    __syncthreads();

    // One K-step = NG groups of 4 chained MFMAs (one accumulator tile x 8 k).  All LDS fragments of the
    // step are read up front into distinct registers (no LDS wait inside the MFMA stream); the staging
    // of the next K-step rides in the MFMA shadows: one global load per group in the first half of the
    // step, one LDS write per group in the second half (counted vmcnt waits), order pinned with
    // sched_barrier so the compiler cannot re-serialise it.  Only lgkmcnt(0)+barrier remain at the end.
    constexpr int NSB = BK / 8;
    constexpr int NG = C::TM * C::TN * NSB;
    constexpr int HALF = NG / 2;
    constexpr int PPG = (NP + HALF - 1) / HALF;  // staging pieces per group
    static_assert(NG >= 2 && NG % 2 == 0, "group count");

    auto k_step = [&](const int buf, const bool more) {
        float4 fa[NSB][C::TM], fb[NSB][C::TN];
        const float* a = As + buf * BM * C::LDS_K + (wm * C::TM * 32 + l31) * C::LDS_K + h * 4;
        const float* b = Bs + buf * BN * C::LDS_K + (wn * C::TN * 32 + l31) * C::LDS_K + h * 4;
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
                fa[sb][i] = *reinterpret_cast<const float4*>(a + i * 32 * C::LDS_K + sb * 8);
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
                fb[sb][j] = *reinterpret_cast<const float4*>(b + j * 32 * C::LDS_K + sb * 8);
        }
        if (more) {
            advance();
            stage_setup();
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int sb = g / (C::TM * C::TN);
            const int i = (g % (C::TM * C::TN)) / C::TN;
            const int j = g % C::TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sb][i].x, fb[sb][j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sb][i].y, fb[sb][j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sb][i].z, fb[sb][j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sb][i].w, fb[sb][j].w, acc[i][j], 0, 0, 0);
            if (more) {
                if (g < HALF) {
#pragma unroll
                    for (int q = g * PPG; q < (g + 1) * PPG && q < NP; ++q) load_piece(q);
                } else {
#pragma unroll
                    for (int q = (g - HALF) * PPG; q < (g - HALF + 1) * PPG && q < NP; ++q) store_piece(q, buf ^ 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // SCHED 0: lumped schedule — next-tile loads issued first, then the MFMAs (the compiler streams the
    // fragment reads between them), then the LDS writes.  The other block on the CU covers the gaps.
    constexpr int ABL = (SCHED >> 4) & 7;  // ablation experiments only (results are wrong when ABL != 0)
    // float offset of the 16-B fragment chunk (k = 8*sb + 4*h .. +3) inside a tile row
    const int rsw = (l31 >> 1) & 7;
    auto koff = [&](int sb) { return DMA ? (((sb * 2 + h) ^ rsw) * 4) : (sb * 8 + h * 4); };
    auto k_step_lumped = [&](const int buf, const bool more) {
        if (more) {
            advance();
            stage_setup();
            if (DMA) {
#pragma unroll
                for (int q = 0; q < NP; ++q) dma_piece(q, buf ^ 1);  // lands during this step's MFMAs
            } else if (ABL != 1 && ABL != 2 && ABL != 3) {
#pragma unroll
                for (int q = 0; q < NP; ++q) load_piece(q);
            }
        }
        const float* a = As + buf * BM * C::LDS_K + (wm * C::TM * 32 + l31) * C::LDS_K;
        const float* b = Bs + buf * BN * C::LDS_K + (wn * C::TN * 32 + l31) * C::LDS_K;
        if (SCHED & 128) __builtin_amdgcn_s_setprio(1);
        if (SCHED & 4) {
            // all fragments of the step in distinct registers: no LDS wait inside the MFMA stream
            float4 fa[NSB][C::TM], fb[NSB][C::TN];
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
#pragma unroll
                for (int i = 0; i < C::TM; ++i)
                    fa[sb][i] = *reinterpret_cast<const float4*>(a + i * 32 * C::LDS_K + koff(sb));
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
                    fb[sb][j] = *reinterpret_cast<const float4*>(b + j * 32 * C::LDS_K + koff(sb));
            }
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                if (SCHED & 8) {
                    // rotate accumulators: consecutive MFMAs are independent (a dependent 32x32x2 chain from one
                    // wave issues at ~72 instead of 64 cycles)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < C::TM; ++i)
#pragma unroll
                            for (int j = 0; j < C::TN; ++j) {
                                const float av = e == 0 ? fa[sb][i].x : e == 1 ? fa[sb][i].y : e == 2 ? fa[sb][i].z : fa[sb][i].w;
                                const float bv = e == 0 ? fb[sb][j].x : e == 1 ? fb[sb][j].y : e == 2 ? fb[sb][j].z : fb[sb][j].w;
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                            }
                } else {
#pragma unroll
                    for (int i = 0; i < C::TM; ++i)
#pragma unroll
                        for (int j = 0; j < C::TN; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sb][i].x, fb[sb][j].x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sb][i].y, fb[sb][j].y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sb][i].z, fb[sb][j].z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sb][i].w, fb[sb][j].w, acc[i][j], 0, 0, 0);
                        }
                }
            }
        } else {
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
            float4 fa[C::TM], fb[C::TN];
            if (ABL == 4) {
#pragma unroll
                for (int i = 0; i < C::TM; ++i) fa[i] = make_float4(lane * 0.001f, 0.5f, 0.25f, 0.125f);
#pragma unroll
                for (int j = 0; j < C::TN; ++j) fb[j] = make_float4(0.3f, lane * 0.002f, 0.1f, 0.7f);
            } else {
#pragma unroll
                for (int i = 0; i < C::TM; ++i)
                    fa[i] = *reinterpret_cast<const float4*>(a + i * 32 * C::LDS_K + koff(sb));
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
                    fb[j] = *reinterpret_cast<const float4*>(b + j * 32 * C::LDS_K + koff(sb));
            }
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j) {
                    mfma32<(SCHED & 256) != 0>(acc[i][j], fa[i].x, fb[j].x);
                    mfma32<(SCHED & 256) != 0>(acc[i][j], fa[i].y, fb[j].y);
                    mfma32<(SCHED & 256) != 0>(acc[i][j], fa[i].z, fb[j].z);
                    mfma32<(SCHED & 256) != 0>(acc[i][j], fa[i].w, fb[j].w);
                }
        }
        }  // SCHED & 4
        if (SCHED & 128) __builtin_amdgcn_s_setprio(0);
        if (more && !DMA && ABL != 2 && ABL != 3) {
#pragma unroll
            for (int q = 0; q < NP; ++q) store_piece(q, buf ^ 1);
        }
    };

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (SCHED & 1)
            k_step((kt - kt_begin) & 1, kt + 1 < kt_end);
        else
            k_step_lumped((kt - kt_begin) & 1, kt + 1 < kt_end);
        if (ABL != 3) __syncthreads();
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
        if (n < p.Cout) {
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
                *reinterpret_cast<float2*>(dst) = make_float2(v[0] * v[1], v[2] * v[3]);
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

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int MINW, int SCHED = 0, int DMA = 0>
void launch_cfg(const ConvParams& p, int M, int nk_total, hipStream_t s, int lds_override = 0) {
    using C = Cfg<BM, BN, WAVES_M, WAVES_N, BK, DMA>;
    if (DMA && !p.zeros) throw HipError("launch_conv: DMA staging needs ConvParams::zeros");
    auto kern = conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, BK, MINW, SCHED, DMA>;
    const int nblk_m = (M + BM - 1) / BM;
    const int nblk_n = (p.Cout + BN - 1) / BN;
    dim3 grid(nblk_m * nblk_n, p.splits, p.nz);
    hipLaunchKernelGGL(kern, grid, dim3(C::NT), lds_override ? lds_override : C::LDS_BYTES, s, p, nblk_n, M, nk_total);
    IRSDE_HIP_CHECK(hipGetLastError());
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int MINW, int SCHED = 0, int DMA = 0>
void init_cfg() {
    IRSDE_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, BK, MINW, SCHED, DMA>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}

int g_variant = 0;  // tuning experiments only (irsde_bench_conv)

}  // namespace

void conv_set_variant(int v) { g_variant = v; }

void conv_global_init() {
    init_cfg<128, 128, 2, 2, 32, 2>();
    init_cfg<128, 64, 2, 2, 32, 2>();
    init_cfg<128, 32, 4, 1, 32, 2>();
    init_cfg<256, 128, 4, 2, 32, 2>();
    init_cfg<128, 128, 2, 2, 32, 2, 1>();
    init_cfg<128, 128, 2, 2, 32, 2, 2>();
    init_cfg<128, 128, 2, 2, 32, 2, 3>();
    init_cfg<256, 256, 2, 4, 32, 2>();
    init_cfg<128, 128, 2, 2, 32, 2, 128>();
    init_cfg<128, 128, 2, 2, 32, 2, 256>();
    init_cfg<128, 128, 2, 2, 32, 2, 0, 1>();
    init_cfg<256, 128, 4, 2, 32, 2, 0, 1>();
    init_cfg<128, 128, 2, 2, 32, 2, 4>();
    init_cfg<128, 128, 2, 2, 32, 2, 8>();
    init_cfg<128, 128, 2, 2, 32, 2, 10>();
    init_cfg<128, 128, 2, 2, 32, 2, 12>();
    init_cfg<128, 128, 2, 2, 32, 2, 16>();
    init_cfg<128, 128, 2, 2, 32, 2, 32>();
    init_cfg<128, 128, 2, 2, 32, 2, 48>();
    init_cfg<128, 128, 2, 2, 32, 2, 64>();
}

double conv_flops(const ConvParams& p) {
    return 2.0 * (double)p.B * p.Ho * p.Wo * (double)p.Cout * (double)(p.KH * p.KW) * (double)(p.C0 + p.C1);
}

// 256x256 tiles (8 waves, 128x64 per wave) halve the staging instructions per MFMA (+5..8 % on deep layers) but
// need Cout % 256 == 0 and a tile count that fills the 256 CUs without a ragged last round.
bool conv_use_tile256(int M, int Cout, int splits, int nk_total) {
    static const int min_nk = getenv("IRSDE_T256_MIN_NK") ? atoi(getenv("IRSDE_T256_MIN_NK")) : 0;
    if (splits != 1 || Cout % 256 || nk_total < min_nk) return false;
    const long long nb = (long long)((M + 255) / 256) * (Cout / 256);
    return nb % 256 == 0 || nb >= 1024;
}

void launch_conv(const ConvParams& p, hipStream_t s) {
    const int M = p.B * p.Ho * p.Wo;
    const int Ctot = p.C0 + p.C1;
    if (p.C0 % 32 || p.C1 % 32 || Ctot == 0)
        throw HipError("launch_conv: channel counts must be multiples of 32 (got " + std::to_string(p.C0) + "+" +
                       std::to_string(p.C1) + ")");
    if (p.splits > 1 && !p.partial) throw HipError("launch_conv: split-K needs a partial buffer");
    const int nk_total = p.KH * p.KW * (Ctot / 32);
    if (p.Cout >= 128) {
        if (g_variant == 3)
            launch_cfg<256, 128, 4, 2, 32, 2>(p, M, nk_total, s);
        else if (g_variant == 1)
            launch_cfg<128, 128, 2, 2, 32, 2, 1>(p, M, nk_total, s);
        else if (g_variant == 6)
            launch_cfg<128, 128, 2, 2, 32, 2, 2>(p, M, nk_total, s);
        else if (g_variant == 7)
            launch_cfg<128, 128, 2, 2, 32, 2, 3>(p, M, nk_total, s);
        else if (g_variant == 50)
            launch_cfg<256, 256, 2, 4, 32, 2>(p, M, nk_total, s);
        else if (g_variant == 41)
            launch_cfg<128, 128, 2, 2, 32, 2, 256>(p, M, nk_total, s);
        else if (g_variant == 40)
            launch_cfg<128, 128, 2, 2, 32, 2, 128>(p, M, nk_total, s);
        else if (g_variant == 30)
            launch_cfg<128, 128, 2, 2, 32, 2, 0, 1>(p, M, nk_total, s);
        else if (g_variant == 31)
            launch_cfg<256, 128, 4, 2, 32, 2, 0, 1>(p, M, nk_total, s);
        else if (g_variant == 20)
            launch_cfg<128, 128, 2, 2, 32, 2, 8>(p, M, nk_total, s);
        else if (g_variant == 21)
            launch_cfg<128, 128, 2, 2, 32, 2, 10>(p, M, nk_total, s);
        else if (g_variant == 8)
            launch_cfg<128, 128, 2, 2, 32, 2, 4>(p, M, nk_total, s);
        else if (g_variant == 9)
            launch_cfg<128, 128, 2, 2, 32, 2, 12>(p, M, nk_total, s);
        else if (g_variant == 11)
            launch_cfg<128, 128, 2, 2, 32, 2, 16>(p, M, nk_total, s);
        else if (g_variant == 12)
            launch_cfg<128, 128, 2, 2, 32, 2, 32>(p, M, nk_total, s);
        else if (g_variant == 13)
            launch_cfg<128, 128, 2, 2, 32, 2, 48>(p, M, nk_total, s);
        else if (g_variant == 14)
            launch_cfg<128, 128, 2, 2, 32, 2, 64>(p, M, nk_total, s);
        else if (g_variant == 5)
            launch_cfg<128, 128, 2, 2, 32, 2>(p, M, nk_total, s, 120 * 1024);  // diagnostic: force 1 block/CU
        else if (g_variant == 0 && conv_use_tile256(M, p.Cout, p.splits, nk_total))
            launch_cfg<256, 256, 2, 4, 32, 2>(p, M, nk_total, s);
        else
            launch_cfg<128, 128, 2, 2, 32, 2>(p, M, nk_total, s);
    } else if (p.Cout > 32) {
        launch_cfg<128, 64, 2, 2, 32, 2>(p, M, nk_total, s);
    } else {
        launch_cfg<128, 32, 4, 1, 32, 2>(p, M, nk_total, s);
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
