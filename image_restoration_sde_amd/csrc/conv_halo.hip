// 3x3 stride-1 pad-1 convolution with an LDS-resident halo tile — the bf16 mode's (IRSDE_FLAG_BF16) kernel for the
// layers where the generic implicit GEMM (conv_igemm.hip) is bound by L2 -> CU traffic: there every one of the 9 taps
// re-reads its 128 x 32-channel activation slice through L2, 24 KB per 1.05 MFLOP.  Here a block owns a 16 x 16 pixel
// tile x BN output channels; per 32-channel chunk it stages the 18 x 18 halo ONCE (fp32 in HBM -> bf16 in LDS, RNE) and
// the 9 taps read their A fragments from it at shifted pixel addresses, so only the weight slice of a tap (BN x 32 bf16)
// moves per K-step: ~6 KB per MFLOP, 4x less.  Reference call sites: Block.proj (module_util.py:108-122) and the
// 3x3 default_conv layers (DenoisingUNet_arch.py:60,67).
//
// 8 waves = 4 (pixel rows of 64) x 2 (BN/2 channels), each 2 x TN tiles of v_mfma_f32_32x32x16_bf16, fp32 accumulate.
// LDS rows are 32 bf16 + 16 B pad (80 B: an odd number of 16-byte slots => conflict-free ds_read_b128, as in
// conv_igemm.hip); lane half h owns the second 16 bytes of each 32-byte group for A and B alike.
// Pipeline: three weight slices in flight — the slice of K-step s+3 and one sixth of the next chunk's halo are loaded
// before the MFMAs of step s and written to LDS (2 slice buffers, 2 halo buffers) after the MFMAs of step s+2; one barrier
// per tap.  The slice stored at step s+2 goes to the buffer step s+1 read: every wave is past the barrier that ended that
// step by then.  Two slice buffers (not three) keep BN = 128 at 72,320 B of LDS: two blocks per CU, so one block's
// barrier stalls are covered by the other's MFMAs.
#include "common.h"

namespace irsde {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

// 16-bit operand type: bf16 (IRSDE_FLAG_BF16) or IEEE fp16 (IRSDE_FLAG_FP16, fp32 activation storage only)
template <bool F16>
struct HOp16 {
    using x8 = bf16x8;
    using x4 = bf16x4;
    static __device__ __forceinline__ floatx16 mfma(x8 a, x8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct HOp16<true> {
    using x8 = f16x8;
    using x4 = f16x4;
    static __device__ __forceinline__ floatx16 mfma(x8 a, x8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

constexpr int TS = 16;                 // output tile edge (pixels)
constexpr int HW_ = TS + 2;            // halo edge
constexpr int HALO_PIX = HW_ * HW_;    // 324
constexpr int ROWB = 80;               // bytes per LDS row (32 bf16 + pad)
constexpr int NT = 512;
constexpr int HALO_BYTES = HALO_PIX * ROWB;             // 25920

// SiLU for the 16-bit-operand modes: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of expf + IEEE division: 5 instead of ~25
// vector instructions per output, on operands that were rounded to 8 / 11 significant bits two instructions earlier.  The
// epilogue's vector work is comparable to the main loop's MFMA time on the 64..192-channel layers.
__device__ __forceinline__ float silu_h(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
}

template <int BN>
struct HCfg {
    static constexpr int TN = BN / 2 / 32;  // MFMA tiles per wave along N (wave tile 64 x BN/2)
    static constexpr int B_BYTES = BN * ROWB;
    static constexpr int MAIN_BYTES = 2 * HALO_BYTES + 2 * B_BYTES;
    static constexpr int LDS_C = BN + 4;
    static constexpr int EPI_BYTES = 64 * LDS_C * 4;
    static constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
};

// ABF: the activations are bf16 in HBM (IRSDE_FLAG_BF16_ACT): 4 instead of 8 16-byte pieces per halo pixel, no conversion.
template <int BN, bool ABF, bool F16 = false>
__global__ __launch_bounds__(NT, 4) void conv3x3_halo_bf16_kernel(const ConvParams p, const int tiles_x, const int tiles_y,
                                                                  const int nblk_n, const int n_slow) {
    using C = HCfg<BN>;
    using H16 = HOp16<F16>;
    static_assert(!F16 || !ABF, "fp16 operands go with fp32 activation storage");
    constexpr int PPP = ABF ? 4 : 8;                            // 16-byte pieces per halo pixel (32 channels)
    constexpr int PSH = ABF ? 2 : 3;
    constexpr int NPIECE = HALO_PIX * PPP;
    constexpr int A_PASSES = (NPIECE + NT - 1) / NT;           // loads per thread and chunk: 6 (fp32) / 3 (bf16)
    constexpr int AESZ = ABF ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    char* Ah = lds;                    // 2 halo buffers
    char* Bs = lds + 2 * HALO_BYTES;   // 2 weight-slice buffers (K-step parity)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1;  // 0..3: pixel rows 64*wm .. +63 of the 256-pixel tile
    const int wn = wave & 1;
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // XCD-aware bijective remap, N-blocks fastest (they share the halo in L2)
    int wgid;
    {
        const int orig = blockIdx.x, nwg = gridDim.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    // Tile order inside an XCD's contiguous range.  Few output-channel blocks: N-blocks fastest (they share the halo in
    // L2).  Wide layers whose weights exceed an L2 (n_slow): N-blocks SLOWEST, so that an XCD keeps one weight slice
    // resident and streams the (much smaller) activation tiles of every image past it.
    const int ntile = gridDim.x / nblk_n;
    const int nblk = n_slow ? wgid / ntile : wgid % nblk_n;
    int t = n_slow ? wgid % ntile : wgid / nblk_n;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int n0 = nblk * BN;
    const int H = p.Ho, W = p.Wo;  // output = virtual input size (in_shift = 1: fused nearest x2 upsample of the source)
    const int Ctot = p.C0 + p.C1;
    const int nch = Ctot / 32;

    // ---- halo staging coordinates: pass q covers 16-byte piece q*512 + tid = (halo pixel, 4-channel group) ----
    // (the last pass wraps around: its surplus threads re-stage pieces of pass 0 with identical data — no branch)
    // The source pixel of a piece is recomputed when it is loaded (a handful of integer ops per tap, off the MFMA
    // path) instead of living in 6 registers: the kernel has to fit 128 VGPRs for 2 blocks per CU.
    const int hp0 = tid >> PSH;
    auto a_src_pix = [&](int q) {  // pixel index into the sources, or -1 outside the image (zero padding)
        const int hp = q == A_PASSES - 1 ? ((q * NT + tid) % NPIECE) >> PSH : q * (NT / PPP) + hp0;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int y = ty * TS - 1 + hy, x = tx * TS - 1 + hx;
        return ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                   ? (b * p.Hin + (y >> p.in_shift)) * p.Win + (x >> p.in_shift) : -1;
    };
    const int c8 = tid & (PPP - 1);  // (q * 512 + tid) % NPIECE keeps the low bits: 512 and NPIECE are multiples of PPP
    // LDS byte offset of pass q: halo pixel q*(512/PPP) + tid/PPP (no wrap before the last pass) => compile-time steps
    constexpr int PB = 64 / PPP;  // LDS bytes per piece (fp32 pieces shrink to 8 bytes of bf16)
    const int a_lds0 = (tid >> PSH) * ROWB + c8 * PB;
    const int a_lds_last = (((A_PASSES - 1) * NT + tid) % NPIECE >> PSH) * ROWB + c8 * PB;
    // ---- weight-slice staging: row = tid / 4, 16-byte piece tid % 4 (BN*4 pieces) ----
    // (BN = 64: the upper half of the block duplicates the lower half's pieces; rows past Cout are clamped — they only
    //  feed output columns that are never stored)
    const int brow = (tid >> 2) % BN, bchunk = tid & 3;
    const int bn = n0 + brow;
    const char* wrow = reinterpret_cast<const char*>(p.w_bf) + ((size_t)(bn < p.Cout ? bn : p.Cout - 1) * 9 * Ctot) * 2 + bchunk * 16;
    const int b_lds = brow * ROWB + bchunk * 16;

    float4 ra[2];  // halo pieces in flight (pass q lives in ra[q % 2])
    float4 rb0, rb1, rb2;  // weight slices in flight: the slice of K-step s+3 is loaded during step s (register slot = tap % 3; 9 taps
                   // per chunk keep it aligned), written to LDS during step s+2: two K-steps of cover for an L2 round trip
    auto a_load = [&](int q, int chunk) {
        const int cc = chunk * 32;
        const float* src;
        int c, pst;
        if (cc < p.C0) { src = p.in0; c = cc; pst = p.pix0; } else { src = p.in1; c = cc - p.C0; pst = p.pix1; }
        const int pix = a_src_pix(q);
        const char* g = pix >= 0 ? reinterpret_cast<const char*>(src) + ((size_t)pix * pst + c + c8 * (16 / AESZ)) * AESZ
                                 : reinterpret_cast<const char*>(p.zeros);
        ra[q % 2] = *reinterpret_cast<const float4*>(g);
    };
    auto a_store = [&](int q, int buf) {
        const float4 v = ra[q % 2];
        const int off = q == A_PASSES - 1 ? a_lds_last : a_lds0 + q * (NT / PPP) * ROWB;
        if (ABF) {
            *reinterpret_cast<float4*>(Ah + buf * HALO_BYTES + off) = v;
        } else {
            const floatx4 fv = {v.x, v.y, v.z, v.w};
            *reinterpret_cast<typename H16::x4*>(Ah + buf * HALO_BYTES + off) = __builtin_convertvector(fv, typename H16::x4);
        }
    };
    const int last_step = nch * 9 - 1;
    auto b_load = [&](int slot, int step) {  // step = chunk * 9 + tap; steps past the end re-read the last slice (unused)
        const int st = step < last_step ? step : last_step;
        const int chunk = st / 9, tap = st - chunk * 9;
        (slot == 0 ? rb0 : slot == 1 ? rb1 : rb2) = *reinterpret_cast<const float4*>(wrow + ((size_t)tap * Ctot + chunk * 32) * 2);
    };
    auto b_store = [&](int ring, int slot) { *reinterpret_cast<float4*>(Bs + ring * C::B_BYTES + b_lds) = (slot == 0 ? rb0 : slot == 1 ? rb1 : rb2); };

    floatx16 acc[2][C::TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-lane fragment bases: MFMA row r = wm*64 + i*32 + l31 -> tile pixel (R, x) = (r / 16, (r % 16 - 2 R) mod 16).
    // The rotation by 2R makes the halo index R*18 + x congruent to l31 (mod 16) for both 16-lane halves: ds_read_b128 is
    // served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (MI355X_MICROARCH.md, LDS), and with 80-byte rows
    // sixteen rows that are distinct mod 16 hit sixteen distinct 16-byte slots.  With x = r % 16 the second half sat 2
    // rows off and every A read had 2-way conflicts (SQ_LDS_BANK_CONFLICT = 36 % of the LDS cycles).
    int a_base[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + l31;
        const int R = r >> 4;
        a_base[i] = (R * HW_ + ((r - 2 * R) & 15)) * ROWB + h * 16;
    }
    const int b_base = (wn * (BN / 2) + l31) * ROWB + h * 16;

    // ---- prologue: halo of chunk 0, weight slice of (chunk 0, tap 0) ----
#pragma unroll
    for (int q0 = 0; q0 < A_PASSES; q0 += 2) {
#pragma unroll
        for (int q = q0; q < q0 + 2 && q < A_PASSES; ++q) a_load(q, 0);
#pragma unroll
        for (int q = q0; q < q0 + 2 && q < A_PASSES; ++q) a_store(q, 0);
    }
    b_load(0, 0);
    b_store(0, 0);
    b_load(1, 1);  // slices of steps 1 and 2: written to LDS during steps 0 and 1
    b_load(2, 2);
    __syncthreads();

    for (int ci = 0; ci < nch; ++ci) {
        const bool more_chunks = ci + 1 < nch;
        const char* ah = Ah + (ci & 1) * HALO_BYTES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            b_load(tap % 3, ci * 9 + tap + 3);
            if (more_chunks && tap < A_PASSES) a_load(tap, ci + 1);
            const int toff = ((tap / 3) * HW_ + (tap % 3)) * ROWB;
            const char* bs = Bs + ((ci + tap) & 1) * C::B_BYTES + b_base;  // step = 9 ci + tap: parity (ci + tap) & 1
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                typename H16::x8 fa[2], fb[C::TN];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const typename H16::x8*>(ah + a_base[i] + toff + sb * 32);
#pragma unroll
                for (int j = 0; j < C::TN; ++j) fb[j] = *reinterpret_cast<const typename H16::x8*>(bs + j * 32 * ROWB + sb * 32);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < C::TN; ++j)
                        acc[i][j] = H16::mfma(fa[i], fb[j], acc[i][j]);
            }
            // pass q is loaded at tap q and written one tap later (ra[q % 2] is free again before pass q + 2 loads)
            if (more_chunks && tap >= 1 && tap - 1 < A_PASSES) a_store(tap - 1, (ci + 1) & 1);
            b_store((ci + tap + 1) & 1, (tap + 1) % 3);  // slice of step + 1 -> the buffer step - 1 read
            __syncthreads();
        }
    }

    // ---- epilogue: 4 passes of 64 tile rows through LDS; bias -> FiLM -> SiLU -> +res, 16-byte stores ----
    float* Cs = smem;
    constexpr int NV = BN / 4;
    constexpr int RSTEP = NT / NV;
    const int c4 = tid % NV;
    const int n = n0 + c4 * 4;
    float bias[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < p.Cout) {
        const float* f = p.film ? p.film + (size_t)b * p.film_bstride : nullptr;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < p.Cout) {
                if (p.bias) bias[e] = p.bias[n + e];
                if (f) {
                    sc[e] = f[n + e] + 1.0f;
                    sh[e] = f[p.Cout + n + e];
                }
            }
    }
    const bool vec_ok = (n + 3 < p.Cout) && ((p.out_stride & 3) == 0);
    const bool res_vec = (n + 3 < p.Cout) && ((p.res_stride & 3) == 0);
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        if (pass > 0) __syncthreads();
        if (wm == pass) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        Cs[row * C::LDS_C + wn * (BN / 2) + j * 32 + l31] = acc[i][j][r];
                    }
        }
        __syncthreads();
        if (n < p.Cout) {
            for (int row = tid / NV; row < 64; row += RSTEP) {
                const int r = pass * 64 + row;
                const int y = ty * TS + (r >> 4), x = tx * TS + ((r - 2 * (r >> 4)) & 15);  // same rotation as a_base
                if (y >= H || x >= W) continue;
                const size_t m = ((size_t)b * H + y) * W + x;
                const float4 cv = *reinterpret_cast<const float4*>(Cs + row * C::LDS_C + c4 * 4);
                float v[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float tv = v[e] + bias[e];
                    if (p.film) tv = tv * sc[e] + sh[e];
                    if (p.silu) tv = silu_h(tv);
                    v[e] = tv;
                }
                if (p.out_bf16) {
                    if (p.res) {
                        const __bf16* rp = reinterpret_cast<const __bf16*>(p.res) + m * p.res_stride + n;
                        if (res_vec) {
                            const bf16x4 t4 = *reinterpret_cast<const bf16x4*>(rp);
                            v[0] += (float)t4[0]; v[1] += (float)t4[1]; v[2] += (float)t4[2]; v[3] += (float)t4[3];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n + e < p.Cout) v[e] += (float)rp[e];
                        }
                    }
                    __bf16* dst = reinterpret_cast<__bf16*>(p.out) + m * p.out_stride + n;
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
                    const float* rp = p.res + m * p.res_stride + n;
                    if (res_vec) {
                        const float4 t4 = *reinterpret_cast<const float4*>(rp);
                        v[0] += t4.x; v[1] += t4.y; v[2] += t4.z; v[3] += t4.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.Cout) v[e] += rp[e];
                    }
                }
                float* dst = p.out + m * p.out_stride + n;
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

// ================================================================================================================
// r05: conv3x3_halo2_kernel — the same LDS-resident-halo 3x3 convolution with twice the register tile per wave.
//
// What bounds conv3x3_halo_bf16_kernel (0.40 - 0.48 of the 16-bit MFMA roof, 75 % of a configs[2] evaluation): a wave owns 64 pixels x 64 output channels
// (2 x 2 MFMA tiles), so every v_mfma_f32_32x32x16 (32 cycles) is fed by one kilobyte of ds_read_b128 — with the weight-slice and halo writes ~0.75 of the
// LDS's cycles at full MFMA rate — and the block meets at a barrier after every tap, 256 MFMA cycles per wave apart.  Here:
//   * block = 16 x 32 output pixels (512) x 128 output channels, 8 waves = 4 (pixel rows 4 wm .. 4 wm + 3) x 2 (64 channels): a wave holds 4 x 2 MFMA tiles
//     (128 accumulator registers, 2 waves per SIMD, one block per CU): 6 fragment reads per 8 MFMAs instead of 4 per 4, and a weight slice is written
//     once per 512 pixels instead of once per 256;
//   * an MFMA row tile is one output row of 32 consecutive pixels: tap (dy, dx) reads the 32 halo pixels (row + dy) * 34 + dx + l31 — consecutive 80-byte
//     rows (an odd number of 16-byte slots), conflict-free for every shift, no rotation trick;
//   * weights are staged three taps (one kernel row) at a time: one barrier per 1536 MFMA cycles of a wave; the stage after next is requested at the top
//     of a stage and written behind its MFMAs into the buffer the previous stage read; the next chunk's halo rides along in the same way (two halo buffers).
// LDS: 2 x 612 x 80 (halos) + 2 x 3 x 128 x 80 (weight stages) = 159 360 bytes; the epilogue transposes 128 rows x 132 floats per pass through it.
// Activations, weights, residual and output move through buffer descriptors (32-bit offsets: every tensor < 2 GiB, the launcher checks; pixels outside the
// image carry an out-of-range offset and read zeros).  Reference call sites: Block.proj (module_util.py:108-122), default_conv (DenoisingUNet_arch.py:60,67).
// ================================================================================================================
constexpr int H2_TH = 16, H2_TW = 32, H2_HW = H2_TW + 2, H2_HPIX = (H2_TH + 2) * H2_HW;   // 612 halo pixels
constexpr int H2_BN = 128;
constexpr int H2_HALO_BYTES = H2_HPIX * ROWB;           // 48 960
constexpr int H2_WTAP_BYTES = H2_BN * ROWB;             // 10 240: one tap's slice
constexpr int H2_WB_BYTES = 3 * H2_WTAP_BYTES;          // 30 720: one stage
constexpr int H2_MAIN_BYTES = 2 * H2_HALO_BYTES + 2 * H2_WB_BYTES;   // 159 360
constexpr int H2_LDS_C = H2_BN + 4;
constexpr int H2_EPI_BYTES = 128 * H2_LDS_C * 4;        // 67 584
constexpr int H2_LDS_BYTES = H2_MAIN_BYTES > H2_EPI_BYTES ? H2_MAIN_BYTES : H2_EPI_BYTES;
constexpr unsigned H2_OOB = 0x80000000u;

template <bool ABF, bool F16 = false>
__global__ __launch_bounds__(NT, 2) void conv3x3_halo2_kernel(const ConvParams p, const int tiles_x, const int tiles_y, const int nblk_n, const int n_slow,
                                                              const unsigned in0_bytes, const unsigned in1_bytes, const unsigned w_bytes) {
    using H16 = HOp16<F16>;
    static_assert(!F16 || !ABF, "fp16 operands go with fp32 activation storage");
    constexpr int PPP = ABF ? 4 : 8;             // 16-byte pieces per halo pixel and 32-channel chunk
    constexpr int PSH = ABF ? 2 : 3;
    constexpr int NPIECE = H2_HPIX * PPP;        // 2448 / 4896
    constexpr int NPASS = (NPIECE + NT - 1) / NT;   // 5 / 10 loads per thread and chunk
    constexpr int AESZ = ABF ? 2 : 4;
    constexpr int PB = 64 / PPP;                 // LDS bytes per piece (fp32 pieces shrink to 8 bytes of 16-bit operands)
    // halo passes requested in stage g of the previous chunk: [first, last)
    constexpr int G0 = ABF ? 2 : 4, G1 = ABF ? 4 : 7;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    char* Ah = lds;                              // 2 halo buffers
    char* Ws = lds + 2 * H2_HALO_BYTES;          // 2 weight stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    int wgid;   // XCD-aware bijective remap (as in conv3x3_halo_bf16_kernel)
    {
        const int orig = blockIdx.x, nwg = gridDim.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int ntile = gridDim.x / nblk_n;
    const int nblk = n_slow ? wgid / ntile : wgid % nblk_n;
    int t = n_slow ? wgid % ntile : wgid / nblk_n;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int n0 = nblk * H2_BN;
    const int H = p.Ho, W = p.Wo;
    const int Ctot = p.C0 + p.C1;
    const int nch = Ctot / 32;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0), 0, in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.C1 ? p.in1 : p.in0), 0, p.C1 ? in1_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w_bf), 0, w_bytes, 0x00020000);

    // ---- halo staging: pass q covers piece q * 512 + tid = (halo pixel, 16-byte group); the last pass wraps (duplicates re-stage identical data) ----
    const int c8 = tid & (PPP - 1);
    int a_pix[NPASS];      // source pixel index, or -1 outside the image
    int a_lds[NPASS];
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        const int piece = q == NPASS - 1 ? (q * NT + tid) % NPIECE : q * NT + tid;
        const int hp = piece >> PSH;
        const int hy = hp / H2_HW, hx = hp - hy * H2_HW;
        const int y = ty * H2_TH - 1 + hy, x = tx * H2_TW - 1 + hx;
        a_pix[q] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? (b * p.Hin + (y >> p.in_shift)) * p.Win + (x >> p.in_shift) : -1;
        a_lds[q] = hp * ROWB + c8 * PB;
    }
    float4 ra[NPASS > 5 ? 4 : 2];   // halo pieces in flight (the passes of one stage)
    auto a_load = [&](const int q, const int slot, const int chunk) {
        const int cc = chunk * 32;
        const bool second = cc >= p.C0;
        const int pst = (second ? p.pix1 : p.pix0) * AESZ;
        const int soff = (second ? cc - p.C0 : cc) * AESZ;
        const unsigned voff = a_pix[q] >= 0 ? (unsigned)a_pix[q] * (unsigned)pst + (unsigned)(c8 * 16) : H2_OOB;
        ra[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(second ? rs1 : rs0, (int)voff, soff, 0));
    };
    auto a_store = [&](const int q, const int slot, const int buf) {
        const float4 v = ra[slot];
        if constexpr (ABF) {
            *reinterpret_cast<float4*>(Ah + buf * H2_HALO_BYTES + a_lds[q]) = v;
        } else {
            const floatx4 fv = {v.x, v.y, v.z, v.w};
            *reinterpret_cast<typename H16::x4*>(Ah + buf * H2_HALO_BYTES + a_lds[q]) = __builtin_convertvector(fv, typename H16::x4);
        }
    };
    // ---- weight staging: a stage = 3 taps x 128 rows x 64 bytes; thread = (row tid / 4, 16-byte piece tid % 4) of each tap ----
    const int brow = tid >> 2, bchunk = tid & 3;
    const int bn = n0 + brow;
    const unsigned w_voff = (unsigned)(bn < p.Cout ? bn : p.Cout - 1) * (unsigned)(9 * Ctot * 2) + (unsigned)(bchunk * 16);
    const int b_lds = brow * ROWB + bchunk * 16;
    float4 rb[3];
    const int nstage = nch * 3;
    auto b_load = [&](const int stage) {   // stages past the end re-read the last one (never stored)
        const int st = stage < nstage ? stage : nstage - 1;
        const int chunk = st / 3, g = st - chunk * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)w_voff, ((3 * g + j) * Ctot + chunk * 32) * 2, 0));
    };
    auto b_store = [&](const int buf) {
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<float4*>(Ws + buf * H2_WB_BYTES + j * H2_WTAP_BYTES + b_lds) = rb[j];
    };

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_base = ((4 * wm) * H2_HW + l31) * ROWB + h * 16;   // halo pixel (row 4 wm + i + dy, column l31 + dx)
    const int b_base = (wn * 64 + l31) * ROWB + h * 16;

    // ---- prologue: halo of chunk 0, weight stage 0 ----
    b_load(0);
    constexpr int NSLOT = NPASS > 5 ? 4 : 2;
#pragma unroll
    for (int q0 = 0; q0 < NPASS; q0 += NSLOT) {
#pragma unroll
        for (int q = q0; q < q0 + NSLOT && q < NPASS; ++q) a_load(q, q - q0, 0);
#pragma unroll
        for (int q = q0; q < q0 + NSLOT && q < NPASS; ++q) a_store(q, q - q0, 0);
    }
    b_store(0);
    __syncthreads();

    for (int ci = 0; ci < nch; ++ci) {
        // (unconditional staging: a branch around the loads splits the stage into basic blocks and the wait insertion then drains every load before the
        //  next is issued; the last chunk re-stages itself into the halo buffer nobody reads any more)
        const int cnext = ci + 1 < nch ? ci + 1 : ci;
        const char* ah = Ah + (ci & 1) * H2_HALO_BYTES + a_base;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int stage = ci * 3 + g;
            const int q_lo = g == 0 ? 0 : g == 1 ? G0 : G1, q_hi = g == 0 ? G0 : g == 1 ? G1 : NPASS;
            b_load(stage + 1);
#pragma unroll
            for (int q = q_lo; q < q_hi; ++q) a_load(q, q - q_lo, cnext);
            __builtin_amdgcn_sched_barrier(0);   // the requests stay in front of the stage's MFMAs (left alone the scheduler sinks them to the LDS writes: one exposed L2 / HBM round trip per stage)
            const char* ws = Ws + (stage & 1) * H2_WB_BYTES + b_base;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int toff = (g * H2_HW + j) * ROWB;
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    typename H16::x8 fa[4], fb[2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const typename H16::x8*>(ah + i * H2_HW * ROWB + toff + sb * 32);
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) fb[jn] = *reinterpret_cast<const typename H16::x8*>(ws + j * H2_WTAP_BYTES + jn * 32 * ROWB + sb * 32);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn) acc[i][jn] = H16::mfma(fa[i], fb[jn], acc[i][jn]);
                }
            }
            // the next stage's weights go to the buffer the PREVIOUS stage read (every wave is past the barrier that ended it); the next chunk's halo
            // pieces to the other halo buffer (last read in chunk ci - 1)
            __builtin_amdgcn_sched_barrier(0);
            b_store((stage + 1) & 1);
#pragma unroll
            for (int q = q_lo; q < q_hi; ++q) a_store(q, q - q_lo, (ci + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue: 4 passes of 128 tile rows (= the 4 output rows of wave row wm) through LDS; bias -> FiLM -> SiLU -> +res, 16-byte accesses ----
    float* Cs = smem;
    constexpr int NV = H2_BN / 4;        // 32 float4 columns
    constexpr int RSTEP = NT / NV;       // 16 rows per sweep
    const int c4 = tid % NV;
    const int n = n0 + c4 * 4;
    float bias[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < p.Cout) {
        const float* f = p.film ? p.film + (size_t)b * p.film_bstride : nullptr;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < p.Cout) {
                if (p.bias) bias[e] = p.bias[n + e];
                if (f) {
                    sc[e] = f[n + e] + 1.0f;
                    sh[e] = f[p.Cout + n + e];
                }
            }
    }
    const bool vec_ok = (n + 3 < p.Cout) && ((p.out_stride & 3) == 0);
    const bool res_vec = (n + 3 < p.Cout) && ((p.res_stride & 3) == 0);
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        if (pass > 0) __syncthreads();
        if (wm == pass) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // MFMA tile i = output row 4 wm + i; accumulator row (r&3) + 8 (r>>2) + 4 h = pixel column, lane l31 = output channel
                        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        Cs[row * H2_LDS_C + wn * 64 + j * 32 + l31] = acc[i][j][r];
                    }
        }
        __syncthreads();
        if (n < p.Cout) {
            for (int row = tid / NV; row < 128; row += RSTEP) {
                const int y = ty * H2_TH + 4 * pass + (row >> 5), x = tx * H2_TW + (row & 31);
                if (y >= H || x >= W) continue;
                const size_t m = ((size_t)b * H + y) * W + x;
                const float4 cv = *reinterpret_cast<const float4*>(Cs + row * H2_LDS_C + c4 * 4);
                float v[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float tv = v[e] + bias[e];
                    if (p.film) tv = tv * sc[e] + sh[e];
                    if (p.silu) tv = silu_h(tv);
                    v[e] = tv;
                }
                if (p.out_bf16) {
                    if (p.res) {
                        const __bf16* rp = reinterpret_cast<const __bf16*>(p.res) + m * p.res_stride + n;
                        if (res_vec) {
                            const bf16x4 t4 = *reinterpret_cast<const bf16x4*>(rp);
                            v[0] += (float)t4[0]; v[1] += (float)t4[1]; v[2] += (float)t4[2]; v[3] += (float)t4[3];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n + e < p.Cout) v[e] += (float)rp[e];
                        }
                    }
                    __bf16* dst = reinterpret_cast<__bf16*>(p.out) + m * p.out_stride + n;
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
                    const float* rp = p.res + m * p.res_stride + n;
                    if (res_vec) {
                        const float4 t4 = *reinterpret_cast<const float4*>(rp);
                        v[0] += t4.x; v[1] += t4.y; v[2] += t4.z; v[3] += t4.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.Cout) v[e] += rp[e];
                    }
                }
                float* dst = p.out + m * p.out_stride + n;
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

// 32-bit buffer offsets: every tensor the kernel addresses through a descriptor stays below 2 GiB
inline bool halo2_fits(const ConvParams& p) {
    const long long aesz = p.in_bf16 ? 2 : 4;
    const long long npix = (long long)p.B * p.Hin * p.Win;
    const long long in0 = ((npix - 1) * p.pix0 + p.C0) * aesz, in1 = p.C1 ? ((npix - 1) * p.pix1 + p.C1) * aesz : 0;
    const long long wb = (long long)p.Cout * 9 * (p.C0 + p.C1) * 2;
    return in0 < (1ll << 31) && in1 < (1ll << 31) && wb < (1ll << 31);
}

template <bool ABF, bool F16 = false>
void launch_halo2(const ConvParams& p, hipStream_t s) {
    const int tiles_x = (p.Wo + H2_TW - 1) / H2_TW, tiles_y = (p.Ho + H2_TH - 1) / H2_TH;
    const int nblk_n = (p.Cout + H2_BN - 1) / H2_BN;
    static const int env = tuning_env_int("IRSDE_HALO_NSLOW", -1);
    const double wbytes = 2.0 * p.Cout * 9.0 * (p.C0 + p.C1);
    const int n_slow = env >= 0 ? (env && nblk_n > 1) : (nblk_n >= 2 && wbytes > 4.0e6);
    const long long aesz = ABF ? 2 : 4;
    const long long npix = (long long)p.B * p.Hin * p.Win;
    const unsigned in0_bytes = (unsigned)(((npix - 1) * p.pix0 + p.C0) * aesz);
    const unsigned in1_bytes = p.C1 ? (unsigned)(((npix - 1) * p.pix1 + p.C1) * aesz) : 0u;
    const unsigned w_bytes = (unsigned)((long long)p.Cout * 9 * (p.C0 + p.C1) * 2);
    hipLaunchKernelGGL((conv3x3_halo2_kernel<ABF, F16>), dim3((unsigned)(p.B * tiles_y * tiles_x * nblk_n)), dim3(NT), H2_LDS_BYTES, s, p, tiles_x, tiles_y,
                       nblk_n, n_slow, in0_bytes, in1_bytes, w_bytes);
    IRSDE_HIP_CHECK(hipGetLastError());
}

template <int BN, bool ABF, bool F16 = false>
void launch_halo(const ConvParams& p, hipStream_t s) {
    const int tiles_x = (p.Wo + TS - 1) / TS, tiles_y = (p.Ho + TS - 1) / TS;
    const int nblk_n = (p.Cout + BN - 1) / BN;
    static const int env = tuning_env_int("IRSDE_HALO_NSLOW", -1);
    const double wbytes = 2.0 * p.Cout * 9.0 * (p.C0 + p.C1);
    const int n_slow = env >= 0 ? (env && nblk_n > 1) : (nblk_n >= 2 && wbytes > 4.0e6);  // one XCD L2 = 4 MiB
    hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<BN, ABF, F16>), dim3((unsigned)(p.B * tiles_y * tiles_x * nblk_n)), dim3(NT),
                       HCfg<BN>::LDS_BYTES, s, p, tiles_x, tiles_y, nblk_n, n_slow);
    IRSDE_HIP_CHECK(hipGetLastError());
}

}  // namespace

void conv_halo_global_init() {
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_bf16_kernel<128, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_bf16_kernel<64, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_bf16_kernel<128, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_bf16_kernel<64, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_bf16_kernel<128, false, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_bf16_kernel<64, false, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo2_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}

bool conv_halo_eligible(const ConvParams& p) {
    return p.w_bf && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_y == 1 && p.pad_x == 1 && (p.in_shift == 0 || p.in_shift == 1) &&
           p.splits == 1 && !p.gate && !p.shuffle && !p.ch_scale && !p.in_scale && !p.ln_g && p.nz == 1 && p.Cout >= 64 &&
           p.Ho == (p.Hin << p.in_shift) && p.Wo == (p.Win << p.in_shift) && (!p.film || p.film_bstride >= 0) && p.zeros;
}

// r05: the 512-pixel x 128-channel kernel where it fills the chip (one block per CU: >= 256 blocks) without padding the feature map by more than a
// quarter; force: 1 = always (if the layer fits its descriptors), -1 = never (conv_set_variant 64 / 65: irsde_debug_conv 162 / 262 / 166 and 163 / 263 / 167, irsde_bench_conv 64 .. 67)
bool conv_halo2_wanted(const ConvParams& p, int force) {
    if (force < 0 || p.Cout < 128 || !halo2_fits(p)) return false;
    if (force > 0) return true;
    static const int env = tuning_env_int("IRSDE_HALO2", 1);
    if (!env) return false;
    const long long tx = (p.Wo + H2_TW - 1) / H2_TW, ty = (p.Ho + H2_TH - 1) / H2_TH, nb = (p.Cout + H2_BN - 1) / H2_BN;
    const long long blocks = (long long)p.B * tx * ty * nb;
    // measured at the 16 x 256^2 layer shapes (profiles/r05_halo2_sweep.txt): one block per CU exposes the epilogue that the 256-pixel kernel's second
    // block hides, so the larger register tile only pays where the K loop is long — +7 .. 12 % from 768 input channels (+12 / +18 % at 384 / 256 on fp32
    // tensors without a residual), -8 .. -30 % on the 128- and 256-channel residual layers
    const int Ctot = p.C0 + p.C1;
    const bool deep = Ctot >= 768 || (!p.in_bf16 && !p.res && Ctot >= 256);
    return deep && blocks >= 256 && tx * H2_TW * ty * H2_TH * 4 <= (long long)p.Ho * p.Wo * 5;
}

void launch_conv_halo(const ConvParams& p, hipStream_t s, int force_halo2) {
    if (!conv_halo_eligible(p)) throw HipError("launch_conv_halo: layer not eligible");
    if (conv_halo2_wanted(p, force_halo2)) {
        if (p.f16) {
            if (p.in_bf16) throw HipError("launch_conv_halo: fp16 operands go with fp32 activation storage");
            launch_halo2<false, true>(p, s);
        } else if (p.in_bf16) {
            launch_halo2<true>(p, s);
        } else {
            launch_halo2<false>(p, s);
        }
        return;
    }
    if (p.f16) {
        if (p.in_bf16) throw HipError("launch_conv_halo: fp16 operands go with fp32 activation storage");
        if (p.Cout >= 128) launch_halo<128, false, true>(p, s); else launch_halo<64, false, true>(p, s);
    } else if (p.in_bf16) {
        if (p.Cout >= 128) launch_halo<128, true>(p, s); else launch_halo<64, true>(p, s);
    } else {
        if (p.Cout >= 128) launch_halo<128, false>(p, s); else launch_halo<64, false>(p, s);
    }
}

}  // namespace irsde
