// naf_chain_kernel — a run of consecutive NAFBlocks of the latent score network as ONE launch: one work-group per image walks the whole chain
// (NAFBlock.forward, codes/config/latent-bokeh/models/modules/DenoisingNAFNet_arch.py:56-83 and the deraining / latent-dehazing twins) with the
// activations resident in registers + LDS and the 16-bit weights streamed from L2.
//
// Why (profiles/r04_latent_bench_kernel_trace_stats.txt): BASELINE configs[4] samples 64 x 64 x 4 latents, so 28 of the 36 NAFBlocks run on
// 8 x 8 pixels x 512 channels.  Per image that is 64 x 512 activations: every op of the block (LayerNorm, 1x1 convolutions, depthwise 3x3, SimpleGate,
// the SCA pool, the beta / gamma residuals) is per image, and the block took ~190 us in 8 - 11 launches of 4 - 25 us each (GEMMs of M = 4096, K = 512:
// launch- and ramp-bound at 0.04 of the fp16 MFMA roof) for ~20 us of MFMA work.
//
// Work-group = 512 threads = 8 waves, image b = blockIdx.x.  Wave w owns output channels [64 w, 64 w + 64) of every 512-wide tensor and the gate pairs
// (j, j + 512), j in that range, of the two 1024-wide ones.  All GEMMs run on v_mfma_f32_16x16x32_f16 with A = weights (16 output channels x 32 k),
// B = activations (32 k x 16 pixels): a lane (n = lane & 15, q = lane >> 4) ends with 4 consecutive channels (4 q .. 4 q + 3 of the 16-channel tile) of
// pixel n of the 16-pixel tile.
//   registers  x[4 channel tiles][4 pixel tiles] (64): the fp32 residual stream of the block chain; acc (64); the weight ring (16 fragments = 64)
//   LDS        bufA, bufB: [64 px][512 k] fp16 operand images (LayerNorm output / gated tensors), 16-byte chunks XOR-swizzled by the pixel so that both the
//              B-fragment reads (ds_read_b128) and the accumulator write-back (ds_write_b64) are conflict-free; a 10 x 10 zero-bordered fp16 staging
//              grid per wave for the depthwise 3x3; LayerNorm partials; the SCA mean / scale vectors
//   weights    one contiguous fp16 stream per wave in exactly the order of use (pack_naf_chain in engine_weights.hip), 448 fragments of 1 KB per block:
//              a lane's 16 bytes of fragment (tile, k step) are W[tile * 16 + n][32 ks + 8 q .. + 7]; ring of 16 fragments, refilled 16 ahead.
// Arithmetic: the 16-bit operand mode's (IRSDE_FLAG_FP16): fp16 operands, fp32 accumulation, fp32 residual stream / LayerNorm / depthwise conv / gates.
// Differences to the per-layer path, all at operand-rounding level: the conv1 output passes through fp16 before the depthwise conv (staging grid),
// and the SCA 1x1 conv runs on the MFMA pipe with fp16 operands.
//
// r06 — G work-groups per image (G = 2 / 4; VERDICT r05 "a chain kernel that uses more than one CU per image").  What bounds the one-group kernel is ONE CU
// streaming the block's 3.67 MB of fp16 weights through its ~30 B/clk vector-memory return path (49 us per block whatever the batch): at the per-GPU
// shard of BASELINE configs[4] (8 images) 8 of 256 CUs work for 57 % of the step.  Group g of an image owns the output channels [512 g / G, 512 (g + 1) / G)
// of every 512-wide tensor (gate pairs (j, j + 512) of the 1024-wide ones) and streams 1 / G of the weights; every GEMM still needs the FULL operand image,
// so per block the groups of an image trade slices through L2 six times — the fp32 residual stream in front of both LayerNorms, the LayerNorm rows (group g
// normalises the pixel tiles g in the one-group lane layout and summation order), the gated tensor + the group's sca.1 partial sums behind conv1, the gated
// tensor behind conv4 — each closed by an arrive / spin barrier on a per-image counter (agent-scope release / acquire fences).  The groups of an
// image get block ids of one residue mod 8 = one XCD (the exchange stays in that XCD's L2).  Co-residency: a spinning group holds its CU, so every group of
// an image must be resident at the same time: the launch keeps the grid <= the CUs the caller says are free (launch_naf_chain_split), and a spin that
// exceeds ~1 s sets an error flag instead of hanging the GPU (irsde_sample checks it and fails the call).
#include "common.h"

namespace irsde {

typedef float nc_f4 __attribute__((ext_vector_type(4)));
typedef _Float16 nc_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 nc_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 nc_h2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int NC_C = 512;                 // channels
constexpr int NC_PX = 64;                 // pixels per image (8 x 8)
constexpr int NC_ROW = NC_C * 2;          // bytes per pixel row of an operand image
constexpr int NC_BUF = NC_PX * NC_ROW;    // 65 536
constexpr int NC_GRID = 100 * 32;         // staging grid per wave: 10 x 10 positions x 16 channels fp16
constexpr int NC_OFF_B = NC_BUF;
constexpr int NC_OFF_GRID = 2 * NC_BUF;
constexpr int NC_OFF_RED1 = NC_OFF_GRID + 8 * NC_GRID;     // [64 px][8 waves] f32
constexpr int NC_OFF_RED2 = NC_OFF_RED1 + 2048;
constexpr int NC_OFF_MEAN = NC_OFF_RED2 + 2048;            // [512] fp16
constexpr int NC_OFF_S = NC_OFF_MEAN + 1024;               // [512] fp16
constexpr int NC_LDS_BYTES = NC_OFF_S + 1024;              // 162 816
static_assert(NC_LDS_BYTES <= 160 * 1024, "LDS budget");

// fp32 vectors of a block (floats): pack_naf_chain writes them in this order
// NV_TAP: the depthwise 3x3 taps as fp16, [1024 channels][16 halves] (taps 0 .. 8 = ky * 3 + kx, then zeros): the rows of the depthwise MFMAs' A operand
constexpr int NV_G1 = 0, NV_G2 = 512, NV_B1 = 1024, NV_DWB = 2048, NV_TAP = 3072, NV_SCAB = 11264, NV_B3 = 11776, NV_BETA = 12288, NV_B4 = 12800,
              NV_B5 = 13824, NV_GAMMA = 14336, NV_TOTAL = 14848;
constexpr int NC_FRAGS_PER_BLOCK = 448;   // per wave: conv1 128, sca 64, conv3 64, conv4 128, conv5 64

struct NafChainArgs {
    const float* x;            // [B][64][512] fp32 in
    float* out;                // [B][64][512]
    const unsigned short* w;   // fp16 fragment streams: [8 waves][nblocks][448][512 halves]
    const float* vecs;         // [nblocks][NV_TOTAL]
    const float* film;         // FiLM rows of the step: row of image b at film + b * film_bstride; block i at + film_off + i * 2048: [shift_att | scale_att | shift_ffn | scale_ffn]
    const float* cam;          // latent-bokeh lens FiLM (or nullptr): row b at cam + b * cam_bstride; block i at + cam_off + i * 1024: [scale | shift]
    int film_bstride, film_off, cam_bstride, cam_off;
    int nblocks;
    unsigned w_bytes;
    unsigned long long* dbg;   // STAMP twin only
    // G > 1 (naf_chain_kernel<.., G>): exchange buffers of the image's groups
    unsigned short* xgate;     // [B][64 px][512] fp16: the gated tensors
    unsigned short* xnorm;     // [B][64 px][512] fp16: the LayerNorm outputs (operand image of conv1 / conv4)
    float* xvec;               // [B][4 quarters][512] fp32: the sca.1 partial sums
    unsigned* ctr;             // [B][4] barrier counters (zero between launches: the kernel restores them) | [4 B]: error word
    int B;
    int sabotage;              // PROBES build, test of the spin limit: group 1 of every image leaves at once, the others must time out and end
};

// Barrier of the G groups of one image (all 512 threads of each call it).  The exchanged tensors are written with 16-byte sc1 (write-through) stores and read
// with sc1 loads (guides/cdna_hip_programming.md 6, Guideline 16: valid under ANY placement of the groups; a release / acquire fence pair per barrier —
// buffer_wbl2 + buffer_inv — measured 4 - 7x slower here, profiles/r06_notes.md): every wave drains its stores, one lane arrives on a counter of the image
// and polls it relaxed.
// Counters: three per image, barrier i on c[i % 3] (target G); behind barrier i group 0 zeroes c[(i + 2) % 3] — last used by barrier i - 1, which every group
// has left (it arrived at i), next used by barrier i + 2, which nobody reaches before group 0 (its store drained) arrives at i + 1.  At the end every group
// counts itself out on a fourth word and the last one zeroes all four: the launch leaves the state it found, with no memset node in the step graph (with a
// hipMemsetAsync in front of the kernel the engine's replayed step graph showed counters and error word 0x01010101; an isolated probe replays memset nodes
// correctly — tools/probe/graph_memset_probe.hip — so the cause was not identified).
// A spin past NC_SPIN_LIMIT polls (~1 s: a group that is not resident) raises the launch's error word (0x10000 | barrier index); once it is up nobody waits
// any more — the results are garbage and the next irsde_sample call on the engine fails and re-zeroes the state.
constexpr int NC_SPIN_LIMIT = 1 << 20;
constexpr int NC_SC1 = 16;   // cache-policy bit of the raw buffer builtins on gfx950: sc1
__device__ __forceinline__ void nc_group_barrier(unsigned* cnt, unsigned* err, const unsigned index, const unsigned G, const bool zeroer, const int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned* c = cnt + index % 3u;
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G) {
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 255) == 0) {
                if (spins >= NC_SPIN_LIMIT) {
                    __hip_atomic_store(err, 0x10000u | (index & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // another group gave up: nobody waits any more
            }
        }
        if (zeroer) __hip_atomic_store(cnt + (index + 2u) % 3u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}
// the end of the launch: the last group of the image to leave restores the counters
__device__ __forceinline__ void nc_group_exit(unsigned* cnt, const unsigned G, const int tid) {
    if (tid == 0) {
        if (__hip_atomic_fetch_add(cnt + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) __hip_atomic_store(cnt + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Cross-lane sums on the vector pipe (r05; kernels_misc.hip has the same helpers and the story of why the swaps are inline assembly): the LayerNorm
// partials are summed over the four lane quarters q = lane >> 4 (xor 16, xor 32), the SCA pool over the 16 pixel lanes n = lane & 15 of a quarter
// (DPP quad permutes + row mirrors) — instead of __shfl_xor's ds_bpermute round trips through the LDS the GEMM passes need.
struct NcSwap { float a, b; };
__device__ __forceinline__ float nc_sum_quarters(float v) {
    NcSwap r{v, v};
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(r.a), "+v"(r.b));
    v = r.a + r.b;
    NcSwap t{v, v};
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(t.a), "+v"(t.b));
    return t.a + t.b;
}
__device__ __forceinline__ float nc_sum_row16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

__device__ __forceinline__ nc_h4 cvt4(const nc_f4 v) {
    nc_h4 h;
    h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
    return h;
}

// ---- the weight stream of one wave: ring[i] holds fragment (pos + i); slot i is refilled with fragment pos + 16 + i right after its last use ----
struct WStream {
    __amdgpu_buffer_rsrc_t rs;
    int lane_off;   // wave region + lane * 16 (bytes)
    int pos;        // fragment index of ring slot 0 (bytes / 1024), multiple of 16
};

// One GEMM pass: acc[t][pt] += sum_k W_t[., k] B[k, pixel tile pt] over K = 512 for the NT weight tiles t the stream delivers per k step (16 NT fragments:
// 16 NT / NC_RING rounds of the fragment ring).  NPT: pixel tiles (4: the image; 1: the SCA matvec, every column carries the same vector).  SCALE: B fragments
// are multiplied by the fp16 vector at sv (the SCA scale per input channel, DenoisingNAFNet_arch.py:68) on their way into the MFMA.
// bsrc[c]: LDS address of this lane's B fragment of k steps ks with (ks & 3) == c (see the swizzle); + (ks >> 2) * 256, pixel tile stride 16 rows.
template <int NT, int NPT, bool SCALE, int NC_RING>
__device__ __forceinline__ void gemm_pass(nc_f4 (&acc)[NT][NPT], const char* const* bsrc, const char* sv, nc_f4 (&ring)[NC_RING], WStream& ws) {
    constexpr int KSU = NC_RING / NT;   // k steps per round of the ring (a multiple of 4)
    static_assert(KSU % 4 == 0 && 16 % KSU == 0, "ring rounds");
#pragma unroll 1
    for (int ko = 0; ko < 16 / KSU; ++ko) {
#pragma unroll
        for (int kk = 0; kk < KSU; ++kk) {
            nc_h8 bq[NPT];
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) bq[pt] = *reinterpret_cast<const nc_h8*>(bsrc[kk & 3] + ((KSU / 4) * ko + (kk >> 2)) * 256 + pt * (16 * NC_ROW));
            if constexpr (SCALE) {
                const nc_h8 s8 = *reinterpret_cast<const nc_h8*>(sv + (ko * KSU + kk) * 64);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) bq[pt] = bq[pt] * s8;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt)
                    acc[t][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(nc_h8, ring[kk * NT + t]), bq[pt], acc[t][pt], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                ring[kk * NT + t] = __builtin_bit_cast(nc_f4, __builtin_amdgcn_raw_buffer_load_b128(ws.rs, ws.lane_off, (ws.pos + NC_RING + kk * NT + t) * 1024, 0));
            __builtin_amdgcn_sched_barrier(0);
        }
        ws.pos += NC_RING;
    }
}

// NC_RING: weight fragments in flight per wave (x 4 registers).  XG: the fp32 residual stream lives in the output tensor (L2) instead of 64 registers —
// every lane re-reads / updates exactly the elements it wrote itself — which frees the registers for a ring of 32 fragments (a whole GEMM pass):
// with 64 work-groups streaming the same weights in lock-step nearly every line is a first touch for its XCD (MALL / HBM latency, not an L2 hit),
// and a ring of 8 covers only ~0.5k cycles of it.
// STAMP (irsde_bench_naf_chain variant 11): per-wave cycle totals per phase into a.dbg[(block * 8 + wave) * 16 ..]: 0 norm1, 1 conv1 GEMM passes,
// 2 depthwise conv + gate, 3 SCA pool barrier, 4 sca.1 GEMM, 5 conv3 GEMM + residual, 6 norm2, 7 conv4 GEMM + gate, 8 conv5 GEMM + residual, 9 barriers
// behind sca / conv4, 15 whole kernel
// G: work-groups per image (1: the r04 kernel, unchanged; 2 / 4: see the file header).  NTW = 4 / G 16-channel tiles per wave.
template <int NC_RING, bool XG, bool STAMP = false, int G = 1>
__global__ __launch_bounds__(512, 2) void naf_chain_kernel(const NafChainArgs a) {
    static_assert(G == 1 || G == 2 || G == 4, "groups per image");
    static_assert(G == 1 || !XG, "the split kernel keeps the residual stream in registers");
    constexpr int NTW = 4 / G;                 // 16-channel tiles per wave of a 512-wide tensor (gate pairs of a 1024-wide one)
    constexpr int NT3 = NTW >= 2 ? 2 : 1;      // weight tiles per k step of the 512 -> 512 passes (sca.1, conv3, conv5)
    constexpr int NP3 = NTW / NT3;             // ... and passes
    unsigned long long st[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0, st_t0 = 0;   // (10 .. 12, G > 1: group barriers, gated fetch + publish, residual-stream publish)
    if constexpr (STAMP) st_t0 = st_t = __builtin_amdgcn_s_memtime();
#define NC_STAMP(K)                                                       \
    if constexpr (STAMP) {                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        st[K] += now_ - st_t;                                             \
        st_t = now_;                                                      \
    }
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    // image and group of this work-group.  G > 1: block id v runs on XCD v & 7 — the G groups of an image share a residue (one L2): image = 8 slot + xcd
    int b = blockIdx.x, grp = 0;
    if constexpr (G > 1) {
        const int v = blockIdx.x, idx = v >> 3;
        grp = idx % G;
        b = (idx / G) * 8 + (v & 7);
        if (b >= a.B) return;
#ifdef IRSDE_PROBES
        if (a.sabotage && grp == 1) return;
#endif
    }
    const int cown = grp * (NC_C / G) + wave * (16 * NTW);   // first of this wave's 16 NTW channels
    unsigned* const ctr = G > 1 ? a.ctr + 4 * b : nullptr;
    unsigned* const err = G > 1 ? a.ctr + 4 * a.B : nullptr;
    unsigned nbar = 0;
    auto group_barrier = [&]() {
        if constexpr (G > 1) {
            NC_STAMP(11)
            nc_group_barrier(ctr, err, nbar, G, grp == 0, tid);
            nbar += 1;
            NC_STAMP(10)
        }
    };

    // ---- LDS addressing ----
    // operand image element (px, k): byte px * 1024 + (((k >> 3) ^ (px & 15)) << 4) + (k & 7) * 2   (the XOR touches the low 4 bits of the chunk index only)
    // B fragment of k step ks = 4 k4 + kk, pixel tile pt: lane reads chunk (4 ks + q) of pixel 16 pt + n: chunk' = (k4 << 4) | (((kk << 2) | q) ^ n)
    const char* bsrcA[4];
    const char* bsrcB[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {   // (kk = ks & 3)
        const int off = n * NC_ROW + ((((kk << 2) | q) ^ n) << 4);
        bsrcA[kk] = lds + off;
        bsrcB[kk] = lds + NC_OFF_B + off;
    }
    // accumulator write-back: this lane's 4 channels (tile channel base cb, + 4 q) of pixel 16 pt + n: 8 bytes at chunk (cb >> 3) + (q >> 1), half (q & 1)
    auto wr_off = [&](const int cb, const int pt) {
        const int chunk = ((cb >> 3) + (q >> 1)) ^ n;   // cb is a multiple of 16: (cb >> 3) has bit 0 clear, adding q >> 1 cannot carry
        return (16 * pt + n) * NC_ROW + (chunk << 4) + ((q & 1) << 3);
    };
    char* const grid = lds + NC_OFF_GRID + wave * NC_GRID;
    float* const red1 = reinterpret_cast<float*>(lds + NC_OFF_RED1);
    float* const red2 = reinterpret_cast<float*>(lds + NC_OFF_RED2);
    // zero the staging grid once: its border is never written again
    for (int i = lane; i < NC_GRID / 4; i += 64) reinterpret_cast<unsigned*>(grid)[i] = 0u;
    // grid position of pixel (16 pt + n): (y + 1) * 10 + x + 1 with y = 2 pt + (n >> 3), x = n & 7; the 3 x 3 window starts one row / column earlier
    const int gpos0 = ((n >> 3) * 10 + (n & 7)) * 32 + q * 8;   // byte offset of the window origin for pt = 0 (pt adds 20 positions)
    // depthwise conv as MFMAs (see conv1): B fragment of k step ks = 8 channels (8 (q & 1) ..) of the neighbour for tap t = 2 ks + (q >> 1) of pixel n
    const int gposB = ((n >> 3) * 10 + (n & 7)) * 32 + (q & 1) * 16;
    int dw_toff[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int t = min(2 * ks + (q >> 1), 8);   // (tap 9 does not exist: its A column is zero, read tap 8's data)
        dw_toff[ks] = ((t / 3) * 10 + (t % 3)) * 32;
    }
    // A fragment: row = channel n of the tile; its single non-zero half sits at k = channel n - 8 (q & 1) of the lane's 8, if that is this quarter's range
    const bool dw_row_active = (n >> 3) == (q & 1);
    const int dw_widx = (n & 7) >> 1, dw_wsh = (n & 1) * 16;

    // ---- weight stream: [G groups][8 waves][nblocks][448 / G fragments] ----
    WStream ws;
    ws.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w), 0, a.w_bytes, 0x00020000);
    ws.lane_off = (grp * 8 + wave) * (a.nblocks * (NC_FRAGS_PER_BLOCK / G) * 1024) + lane * 16;
    ws.pos = 0;
    nc_f4 ring[NC_RING];
#pragma unroll
    for (int i = 0; i < NC_RING; ++i) ring[i] = __builtin_bit_cast(nc_f4, __builtin_amdgcn_raw_buffer_load_b128(ws.rs, ws.lane_off, i * 1024, 0));

    // ---- residual stream: x[ct][pt] = channels cown + 16 ct + 4 q .. + 3 of pixel 16 pt + n ----
    nc_f4 x[XG ? 1 : NTW][XG ? 1 : 4];
    const float* xin = a.x + (size_t)b * NC_PX * NC_C;
    float* const xg = a.out + (size_t)b * NC_PX * NC_C + cown + 4 * q + n * NC_C;   // + 16 ct + pt * 16 * NC_C
    if constexpr (XG) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
                *reinterpret_cast<nc_f4*>(xg + 16 * ct + pt * 16 * NC_C) = *reinterpret_cast<const nc_f4*>(xin + (16 * pt + n) * NC_C + 64 * wave + 16 * ct + 4 * q);
    } else {
#pragma unroll
        for (int ct = 0; ct < NTW; ++ct)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) x[ct][pt] = *reinterpret_cast<const nc_f4*>(xin + (16 * pt + n) * NC_C + cown + 16 * ct + 4 * q);
    }
    // G > 1: the exchange tensors of the image through buffer descriptors (sc1 stores / loads)
    typedef unsigned nc_u4x __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * NC_PX * NC_C, 0, G > 1 ? NC_PX * NC_C * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_gate =
        __builtin_amdgcn_make_buffer_rsrc(G > 1 ? a.xgate + (size_t)b * NC_PX * NC_C : const_cast<unsigned short*>(a.w), 0, G > 1 ? NC_PX * NC_C * 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_norm =
        __builtin_amdgcn_make_buffer_rsrc(G > 1 ? a.xnorm + (size_t)b * NC_PX * NC_C : const_cast<unsigned short*>(a.w), 0, G > 1 ? NC_PX * NC_C * 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_vec =
        __builtin_amdgcn_make_buffer_rsrc(G > 1 ? a.xvec + (size_t)b * 4 * NC_C : const_cast<float*>(a.vecs), 0, G > 1 ? 4 * NC_C * 4 : 0, 0x00020000);
    const int chl = cown + 4 * q;        // + 16 ct: this lane's channels of a 512-wide tensor
    const int chl1 = 64 * wave + 4 * q;  // + 16 ct: the lane's channels in the LayerNorm over the WHOLE image (every group normalises all 512 channels)

    // LayerNorm over the channels (module_util.py:20-26: biased variance, eps 1e-5) * g, then the block's FiLM x * (scale + 1) + shift
    // (DenoisingNAFNet_arch.py:63-64,74-75), result as the fp16 operand image in bufA.  Two passes like layernorm_kernel.
    // (the epilogue vectors of the GEMM passes are requested BEFORE the pass and return under it: a load issued where it is used waits for its own L2
    // latency behind the weight ring's loads — ~45 such waits per block were most of the kernel's time, profiles/r04_naf_chain_bench_a.txt.  The LayerNorm's
    // own vectors are not: 48 more live registers across a GEMM pass spill the residual stream)
    // G > 1: the whole image's residual stream is read from `src` (the input tensor before the first block, the output tensor — where every group
    // stores its slice in front of the barrier — afterwards): the same lane layout, operations and summation order as the one-group kernel.
    struct LnVec { nc_f4 g[4], fs[4], fh[4]; };
    auto ln_prefetch = [&](LnVec& v, const float* g, const float* fscale, const float* fshift) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            v.g[ct] = *reinterpret_cast<const nc_f4*>(g + chl1 + 16 * ct);
            v.fs[ct] = *reinterpret_cast<const nc_f4*>(fscale + chl1 + 16 * ct);
            v.fh[ct] = *reinterpret_cast<const nc_f4*>(fshift + chl1 + 16 * ct);
        }
    };
    // G > 1: group grp normalises the NPO = 4 / G pixel tiles pt0 .. pt0 + NPO - 1 of the image (all 512 channels — the one-group lane layout, operations and summation
    // order per pixel: bit-identical rows), publishes its rows of the fp16 operand image and fetches the other groups' rows behind a group barrier: a quarter of the
    // LayerNorm work and 80 instead of 128 KB of exchange reads per group for one more barrier (every group normalising the whole image: 15-16k cycles per
    // LayerNorm of the block's 82k, profiles/r06_zz_chain_stamps.txt).
    constexpr int NPO = G > 1 ? 4 / G : 4;
    const int pt0 = G > 1 ? grp * NPO : 0;
    auto layernorm_to_A = [&](const float* g, const float* fscale, const float* fshift, const bool from_input) {
        LnVec lv;
        ln_prefetch(lv, g, fscale, fshift);
        nc_f4 xf[G > 1 ? 4 : 1][G > 1 ? NPO : 1];
        if constexpr (G > 1) {
            if (from_input) {   // (the first block: the launch's input tensor, plain loads)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int pp = 0; pp < NPO; ++pp) xf[ct][pp] = *reinterpret_cast<const nc_f4*>(xin + (16 * (pt0 + pp) + n) * NC_C + chl1 + 16 * ct);
            } else {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int pp = 0; pp < NPO; ++pp)
                        xf[ct][pp] = __builtin_bit_cast(nc_f4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, ((16 * (pt0 + pp) + n) * NC_C + chl1 + 16 * ct) * 4, 0, NC_SC1));
            }
        }
        // (pp = index into the group's own pixel tiles; G = 1: pp = pt)
        // XG: every pass re-reads the lane's 16 vectors from L2 instead of holding them (the ring of 32 fragments owns the registers)
        auto X = [&](const int ct, const int pp) -> nc_f4 {
            if constexpr (G > 1) return xf[ct][pp];
            else if constexpr (XG) return *reinterpret_cast<const nc_f4*>(xg + 16 * ct + pp * 16 * NC_C);
            else return x[ct][pp];
        };
        float s[NPO];
#pragma unroll
        for (int pp = 0; pp < NPO; ++pp) {
            float t = 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) { const nc_f4 v = X(ct, pp); t += (v[0] + v[1]) + (v[2] + v[3]); }
            s[pp] = nc_sum_quarters(t);
        }
        if (q == 0) {
#pragma unroll
            for (int pp = 0; pp < NPO; ++pp) red1[(16 * (pt0 + pp) + n) * 8 + wave] = s[pp];
        }
        __syncthreads();
        float mean[NPO], rstd[NPO];
#pragma unroll
        for (int pp = 0; pp < NPO; ++pp) {
            const int px = 16 * (pt0 + pp) + n;
            const nc_f4 r0 = *reinterpret_cast<const nc_f4*>(red1 + px * 8), r1 = *reinterpret_cast<const nc_f4*>(red1 + px * 8 + 4);
            mean[pp] = (((r0[0] + r0[1]) + (r0[2] + r0[3])) + ((r1[0] + r1[1]) + (r1[2] + r1[3]))) * (1.0f / NC_C);
            float t = 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const nc_f4 d = X(ct, pp) - mean[pp];
                t += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
            s[pp] = nc_sum_quarters(t);
        }
        if (q == 0) {
#pragma unroll
            for (int pp = 0; pp < NPO; ++pp) red2[(16 * (pt0 + pp) + n) * 8 + wave] = s[pp];
        }
        __syncthreads();
#pragma unroll
        for (int pp = 0; pp < NPO; ++pp) {
            const int px = 16 * (pt0 + pp) + n;
            const nc_f4 r0 = *reinterpret_cast<const nc_f4*>(red2 + px * 8), r1 = *reinterpret_cast<const nc_f4*>(red2 + px * 8 + 4);
            const float var = (((r0[0] + r0[1]) + (r0[2] + r0[3])) + ((r1[0] + r1[1]) + (r1[2] + r1[3]))) * (1.0f / NC_C);
            rstd[pp] = __builtin_amdgcn_rsqf(var + 1e-5f);   // v_rsq_f32 (1 ulp)
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const nc_f4 gg = lv.g[ct], fs = lv.fs[ct] + 1.0f, fh = lv.fh[ct];
#pragma unroll
            for (int pp = 0; pp < NPO; ++pp) {
                const nc_f4 v = ((X(ct, pp) - mean[pp]) * rstd[pp] * gg) * fs + fh;
                *reinterpret_cast<nc_h4*>(lds + wr_off(64 * wave + 16 * ct, pt0 + pp)) = cvt4(v);
            }
        }
        __syncthreads();
        if constexpr (G > 1) {
            // the group's rows (16 NPO pixels x 64 chunks of 16 bytes) out, barrier, the other groups' rows in
            constexpr int NOWN = (16 * NPO * 64) / 512, NOTH = ((NC_PX - 16 * NPO) * 64) / 512;
#pragma unroll
            for (int i = 0; i < NOWN; ++i) {
                const int idx = tid + 512 * i, px = 16 * pt0 + (idx >> 6), c8 = idx & 63;
                const nc_u4x v = *reinterpret_cast<const nc_u4x*>(lds + px * NC_ROW + ((c8 ^ (px & 15)) << 4));
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_norm, px * NC_ROW + c8 * 16, 0, NC_SC1);
            }
            group_barrier();
            nc_u4x c[NOTH];
#pragma unroll
            for (int i = 0; i < NOTH; ++i) {
                const int idx = tid + 512 * i, r = idx >> 6, px = r < 16 * pt0 ? r : r + 16 * NPO, c8 = idx & 63;
                c[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_norm, px * NC_ROW + c8 * 16, 0, NC_SC1);
            }
#pragma unroll
            for (int i = 0; i < NOTH; ++i) {
                const int idx = tid + 512 * i, r = idx >> 6, px = r < 16 * pt0 ? r : r + 16 * NPO, c8 = idx & 63;
                *reinterpret_cast<nc_u4x*>(lds + px * NC_ROW + ((c8 ^ (px & 15)) << 4)) = c[i];
            }
            __syncthreads();
        }
    };
    // G > 1: the gated tensor.  Every group writes its channel slice into its own bufB (like the one-group kernel), publishes it — 16-byte chunks, LDS ->
    // exchange buffer, plain [px][512] fp16 — and behind the group barrier fetches the other groups' slices into bufB (+ the pooled means into the LDS mean vector)
    constexpr int CH8 = 64 / G;   // 16-byte chunks (8 channels) per pixel of a group's slice
    auto publish_gated = [&]() {
        if constexpr (G > 1) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < (NC_PX * CH8) / 512; ++i) {
                const int idx = tid + 512 * i, px = idx / CH8, c8 = grp * CH8 + idx % CH8;
                const nc_u4x v = *reinterpret_cast<const nc_u4x*>(lds + NC_OFF_B + px * NC_ROW + ((c8 ^ (px & 15)) << 4));
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_gate, px * NC_ROW + c8 * 16, 0, NC_SC1);
            }
        }
    };
    auto fetch_gated = [&](const float* sca_bias) {   // sca_bias != nullptr: + the scale vector from the groups' sca.1 partials
        if constexpr (G > 1) {
            constexpr int NO = 64 - CH8, NI = (NC_PX * NO) / 512;   // chunks per pixel of the other groups; per thread
            nc_u4x c[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int idx = tid + 512 * i, px = idx / NO, r = idx % NO, c8 = r < grp * CH8 ? r : r + CH8;
                c[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_gate, px * NC_ROW + c8 * 16, 0, NC_SC1);
            }
            nc_f4 pq[4], sb = {0.f, 0.f, 0.f, 0.f};
            if (sca_bias && tid < 128) {   // thread t: channels 4 t .. 4 t + 3
#pragma unroll
                for (int k = 0; k < 4; ++k) pq[k] = __builtin_bit_cast(nc_f4, __builtin_amdgcn_raw_buffer_load_b128(rs_vec, (k * NC_C + 4 * tid) * 4, 0, NC_SC1));
                sb = *reinterpret_cast<const nc_f4*>(sca_bias + 4 * tid);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int idx = tid + 512 * i, px = idx / NO, r = idx % NO, c8 = r < grp * CH8 ? r : r + CH8;
                *reinterpret_cast<nc_u4x*>(lds + NC_OFF_B + px * NC_ROW + ((c8 ^ (px & 15)) << 4)) = c[i];
            }
            if (sca_bias && tid < 128) *reinterpret_cast<nc_h4*>(lds + NC_OFF_S + 4 * tid * 2) = cvt4((((pq[0] + pq[1]) + pq[2]) + pq[3]) + sb);
            __syncthreads();
        }
    };
    auto put_gated = [&](const int cb, const int pt, const nc_h4 hv) { *reinterpret_cast<nc_h4*>(lds + NC_OFF_B + wr_off(cb, pt)) = hv; };
    // G > 1: this lane's slice of the residual stream into the output tensor (the exchange buffer of the LayerNorms, and the result)
    auto put_x = [&]() {
        if constexpr (G > 1) {
#pragma unroll
            for (int ct = 0; ct < NTW; ++ct)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(nc_u4x, x[ct][pt]), rs_x, ((16 * pt + n) * NC_C + chl + 16 * ct) * 4, 0, NC_SC1);
        }
    };

    for (int blk = 0; blk < a.nblocks; ++blk) {
        const float* vec = a.vecs + (size_t)blk * NV_TOTAL;
        const float* film = a.film + (size_t)b * a.film_bstride + a.film_off + blk * (4 * NC_C);
        // ===== norm1 + time FiLM -> bufA =====
        layernorm_to_A(vec + NV_G1, film + NC_C, film, blk == 0);
        NC_STAMP(0)

        // ===== conv1 (1x1, 512 -> 1024) + conv2 (depthwise 3x3) + SimpleGate -> bufB; channel means for the SCA pool =====
        // NTW passes of one gate pair of 16-channel tiles: lo = channels j = cown + 16 ps (+ 4 q + i), hi = j + 512
#pragma unroll 1
        for (int ps = 0; ps < NTW; ++ps) {
            nc_f4 acc[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[t][pt] = nc_f4{0.f, 0.f, 0.f, 0.f};
            // the pass's epilogue vectors (conv1 bias, depthwise bias in the accumulator layout; the 9 fp16 taps of tile row channel n): the lo half's are
            // requested before the GEMM and return under it, the hi half's in one batch right behind the GEMM and return under the lo half's depthwise conv
            // (loaded where they are used they cost ~7 exposed L2 latencies per half: 24k of the block's 135k cycles, profiles/r04_naf_chain_stamps_a.txt)
            typedef unsigned nc_u4 __attribute__((ext_vector_type(4)));
            nc_f4 b1v[2], dbv[2];
            nc_u4 tapa[2], tapb[2];
            auto dw_prefetch = [&](const int hi) {
                const int cb = (hi ? NC_C : 0) + cown + 16 * ps;
                b1v[hi] = *reinterpret_cast<const nc_f4*>(vec + NV_B1 + cb + 4 * q);
                dbv[hi] = *reinterpret_cast<const nc_f4*>(vec + NV_DWB + cb + 4 * q);
                tapa[hi] = *reinterpret_cast<const nc_u4*>(vec + NV_TAP + (cb + n) * 8);
                tapb[hi] = *reinterpret_cast<const nc_u4*>(vec + NV_TAP + (cb + n) * 8 + 4);
            };
            dw_prefetch(0);
            gemm_pass<2, 4, false, NC_RING>(acc, bsrcA, nullptr, ring, ws);
            NC_STAMP(1)
            dw_prefetch(1);
            __builtin_amdgcn_sched_barrier(0);
            nc_f4 dwlo[4];
            nc_f4 cs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const nc_f4 b1 = b1v[hi], db = dbv[hi];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) *reinterpret_cast<nc_h4*>(grid + gpos0 + pt * 640 + 11 * 32) = cvt4(acc[hi][pt] + b1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                // The depthwise conv on the MFMA pipe (idle in this phase; on the vector pipe it was 1152 multiply-adds per lane and block, a fifth of the
                // kernel): out[ch][px] = sum over k = (tap, ch') of A[ch][k] B[k][px] with A[ch][(tap, ch')] = w[ch][tap] if ch' == ch else 0 and
                // B[(tap, ch')][px] = the staged tile at the tap's neighbour of px — an im2col read of 8 channels = one ds_read_b128.  K = 9 taps x 16 channels,
                // k step ks = taps 2 ks, 2 ks + 1 (lane quarter q: tap 2 ks + (q >> 1), channels 8 (q & 1) .. + 7); the 10th tap has zero weights.
                // Products fp16 x fp16, accumulation f32: what v_fma_mix_f32 computed.  The result lands in the accumulator layout of the GEMMs.
                nc_f4 o[4] = {db, db, db, db};
#pragma unroll
                for (int ks = 0; ks < 5; ++ks) {
                    const unsigned word = ks < 4 ? tapa[hi][ks] : tapb[hi][0];        // taps 2 ks (low half) and 2 ks + 1 (high half) of channel n
                    const unsigned w16 = (q >> 1) ? (word >> 16) : (word & 0xffffu);
                    const unsigned wv = (dw_row_active && 2 * ks + (q >> 1) < 9) ? (w16 << dw_wsh) : 0u;
                    nc_u4 af;
                    af[0] = dw_widx == 0 ? wv : 0u; af[1] = dw_widx == 1 ? wv : 0u; af[2] = dw_widx == 2 ? wv : 0u; af[3] = dw_widx == 3 ? wv : 0u;
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt) {
                        const nc_h8 bf = *reinterpret_cast<const nc_h8*>(grid + gposB + pt * 640 + dw_toff[ks]);
                        o[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(nc_h8, af), bf, o[pt], 0, 0, 0);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    if (hi == 0) {
                        dwlo[pt] = o[pt];
                    } else {
                        const nc_f4 gv = dwlo[pt] * o[pt];
                        cs += gv;
                        put_gated(cown + 16 * ps, pt, cvt4(gv));
                    }
                }
            }
            // SCA pool (AdaptiveAvgPool2d(1)): the lane's 4 pixels are in cs; sum the 16 pixel lanes of the channel group
            cs[0] = nc_sum_row16(cs[0]); cs[1] = nc_sum_row16(cs[1]); cs[2] = nc_sum_row16(cs[2]); cs[3] = nc_sum_row16(cs[3]);
            if (n == 0) {
                const nc_h4 mh = cvt4(cs * (1.0f / NC_PX));
                *reinterpret_cast<nc_h4*>(lds + NC_OFF_MEAN + (cown + 16 * ps + 4 * q) * 2) = mh;   // (G > 1: the group's own 512 / G channels — the K range of its sca.1 partial)
            }
            NC_STAMP(2)
        }
        if constexpr (G > 1) publish_gated();   // (starts with a barrier of the work-group: the pooled means are in LDS)
        else __syncthreads();
        NC_STAMP(3)
        // ===== sca.1 (1x1 conv on the pooled vector): s = W mean + b -> the fp16 scale vector =====
        // r06: the K = 512 sum in four quarters of 128 input channels (4 k steps), each accumulated from zero, s = ((p0 + p1) + p2) + p3 + b; wave w owns the output
        // channels 64 w .. 64 w + 63 (four 16-channel tiles).  One group: all four quarters in turn.  G groups: group g computes the quarters of ITS gated channels —
        // their pooled means never leave the group — and publishes the fp32 partials with the gated slice: one exchange (and barrier) less per block, the same bits.
        {
            const char* msrc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) msrc[c] = lds + NC_OFF_MEAN + (c * 32 + 8 * q) * 2;   // every column reads the same 8 k values of its k step
            constexpr int NQ = 4 / G;   // quarters per group
            nc_f4 tot[4];
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) {
                const int qg = grp * NQ + qq;   // quarter = k steps 4 qg .. 4 qg + 3
                nc_f4 cur[4] = {nc_f4{0.f, 0.f, 0.f, 0.f}, nc_f4{0.f, 0.f, 0.f, 0.f}, nc_f4{0.f, 0.f, 0.f, 0.f}, nc_f4{0.f, 0.f, 0.f, 0.f}};
                constexpr int KPR = NC_RING / 4;   // k steps (x 4 tiles) per round of the ring
                static_assert(KPR == 2 || KPR == 4, "ring of 8 or 16 fragments");
#pragma unroll
                for (int hf = 0; hf < 4 / KPR; ++hf) {
#pragma unroll
                    for (int k2 = 0; k2 < KPR; ++k2) {
                        const nc_h8 bq = *reinterpret_cast<const nc_h8*>(msrc[KPR * hf + k2] + qg * 256);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            cur[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(nc_h8, ring[k2 * 4 + t]), bq, cur[t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            ring[k2 * 4 + t] = __builtin_bit_cast(nc_f4, __builtin_amdgcn_raw_buffer_load_b128(ws.rs, ws.lane_off, (ws.pos + NC_RING + k2 * 4 + t) * 1024, 0));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    ws.pos += NC_RING;
                }
                if constexpr (G > 1) {
                    if (n == 0) {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(nc_u4x, cur[t]), rs_vec, (qg * NC_C + 64 * wave + 16 * t + 4 * q) * 4, 0, NC_SC1);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) tot[t] = qq == 0 ? cur[t] : tot[t] + cur[t];
                }
            }
            if constexpr (G == 1) {
                if (n == 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const nc_f4 sb = *reinterpret_cast<const nc_f4*>(vec + NV_SCAB + 64 * wave + 16 * t + 4 * q);
                        *reinterpret_cast<nc_h4*>(lds + NC_OFF_S + (64 * wave + 16 * t + 4 * q) * 2) = cvt4(tot[t] + sb);
                    }
                }
            }
        }
        NC_STAMP(4)
        if constexpr (G > 1) {
            group_barrier();        // every group's gated slice and sca.1 partials are out
            fetch_gated(vec + NV_SCAB);
        } else {
            __syncthreads();
        }
        NC_STAMP(9)
        // ===== conv3 (1x1, 512 -> 512) on x * sca(x); y = inp + conv3 * beta =====
#pragma unroll
        for (int ps = 0; ps < NP3; ++ps) {
            nc_f4 acc[NT3][4];
#pragma unroll
            for (int t = 0; t < NT3; ++t)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[t][pt] = nc_f4{0.f, 0.f, 0.f, 0.f};
            nc_f4 b3v[NT3], bev[NT3];
#pragma unroll
            for (int t = 0; t < NT3; ++t) {
                b3v[t] = *reinterpret_cast<const nc_f4*>(vec + NV_B3 + chl + 16 * (NT3 * ps + t));
                bev[t] = *reinterpret_cast<const nc_f4*>(vec + NV_BETA + chl + 16 * (NT3 * ps + t));
            }
            gemm_pass<NT3, 4, true, NC_RING>(acc, bsrcB, lds + NC_OFF_S + 8 * q * 2, ring, ws);
#pragma unroll
            for (int t = 0; t < NT3; ++t) {
                const int ct = NT3 * ps + t;
                const nc_f4 b3 = b3v[t], be = bev[t];
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    if constexpr (XG) {
                        nc_f4* xp = reinterpret_cast<nc_f4*>(xg + 16 * ct + pt * 16 * NC_C);
                        *xp = *xp + (acc[t][pt] + b3) * be;
                    } else {
                        x[ct][pt] = x[ct][pt] + (acc[t][pt] + b3) * be;
                    }
                }
            }
        }
        NC_STAMP(5)
        if constexpr (G > 1) {
            put_x();
            group_barrier();        // the whole residual stream is in the output tensor
        }
        // ===== norm2 + time FiLM -> bufA (its barriers also fence conv3's reads of bufB) =====
        layernorm_to_A(vec + NV_G2, film + 3 * NC_C, film + 2 * NC_C, false);
        NC_STAMP(6)
        // ===== conv4 (1x1, 512 -> 1024) + SimpleGate (+ lens FiLM) -> bufB =====
#pragma unroll 1
        for (int ps = 0; ps < NTW; ++ps) {
            nc_f4 acc[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[t][pt] = nc_f4{0.f, 0.f, 0.f, 0.f};
            const int ch = cown + 16 * ps + 4 * q;
            const nc_f4 blo = *reinterpret_cast<const nc_f4*>(vec + NV_B4 + ch), bhi = *reinterpret_cast<const nc_f4*>(vec + NV_B4 + NC_C + ch);
            nc_f4 cs = {0.f, 0.f, 0.f, 0.f}, cf = {0.f, 0.f, 0.f, 0.f};
            if (a.cam) {
                const float* cam = a.cam + (size_t)b * a.cam_bstride + a.cam_off + blk * (2 * NC_C);
                cs = *reinterpret_cast<const nc_f4*>(cam + ch);
                cf = *reinterpret_cast<const nc_f4*>(cam + NC_C + ch);
            }
            gemm_pass<2, 4, false, NC_RING>(acc, bsrcA, nullptr, ring, ws);
            cs = cs + 1.0f;
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const nc_f4 v = ((acc[0][pt] + blo) * (acc[1][pt] + bhi)) * cs + cf;
                put_gated(cown + 16 * ps, pt, cvt4(v));
            }
        }
        NC_STAMP(7)
        if constexpr (G > 1) {
            publish_gated();
            group_barrier();        // every group's gated slice is out
            fetch_gated(nullptr);
        } else {
            __syncthreads();
        }
        NC_STAMP(9)
        // ===== conv5 (1x1, 512 -> 512); out = y + conv5 * gamma =====
#pragma unroll
        for (int ps = 0; ps < NP3; ++ps) {
            nc_f4 acc[NT3][4];
#pragma unroll
            for (int t = 0; t < NT3; ++t)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[t][pt] = nc_f4{0.f, 0.f, 0.f, 0.f};
            nc_f4 b5v[NT3], gav[NT3];
#pragma unroll
            for (int t = 0; t < NT3; ++t) {
                b5v[t] = *reinterpret_cast<const nc_f4*>(vec + NV_B5 + chl + 16 * (NT3 * ps + t));
                gav[t] = *reinterpret_cast<const nc_f4*>(vec + NV_GAMMA + chl + 16 * (NT3 * ps + t));
            }
            gemm_pass<NT3, 4, false, NC_RING>(acc, bsrcB, nullptr, ring, ws);
#pragma unroll
            for (int t = 0; t < NT3; ++t) {
                const int ct = NT3 * ps + t;
                const nc_f4 b5 = b5v[t], ga = gav[t];
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    if constexpr (XG) {
                        nc_f4* xp = reinterpret_cast<nc_f4*>(xg + 16 * ct + pt * 16 * NC_C);
                        *xp = *xp + (acc[t][pt] + b5) * ga;
                    } else {
                        x[ct][pt] = x[ct][pt] + (acc[t][pt] + b5) * ga;
                    }
                }
            }
        }
        NC_STAMP(8)
        // (the next block's norm1 writes bufA: every wave has passed the barrier behind conv4; its barriers fence conv5's reads of bufB)
        if constexpr (G > 1) {
            put_x();                                        // (behind the last block: the result)
            if (blk + 1 < a.nblocks) group_barrier();       // the whole residual stream is in the output tensor: the next block's norm1 reads it
        }
    }
    if constexpr (G > 1) nc_group_exit(ctr, G, tid);
    if constexpr (!XG && G == 1) {
        float* xout = a.out + (size_t)b * NC_PX * NC_C;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) *reinterpret_cast<nc_f4*>(xout + (16 * pt + n) * NC_C + 64 * wave + 16 * ct + 4 * q) = x[ct][pt];
    }
    if constexpr (STAMP) {
        if (lane == 0 && a.dbg) {
            unsigned long long* d = a.dbg + ((size_t)(G > 1 ? (int)blockIdx.x : b) * 8 + wave) * 16;
            for (int i = 0; i < 12; ++i) d[i] = st[i];
            d[15] = __builtin_amdgcn_s_memtime() - st_t0;
        }
    }
#undef NC_STAMP
}

}  // namespace

bool naf_chain_shape_ok(int H, int W, int c) { return H * W == NC_PX && H == 8 && c == NC_C; }
size_t naf_chain_weight_halves(int nblocks) { return (size_t)8 * nblocks * NC_FRAGS_PER_BLOCK * 512; }
size_t naf_chain_vec_floats(int nblocks) { return (size_t)nblocks * NV_TOTAL; }

static unsigned long long* g_nc_dbg = nullptr;
void naf_chain_set_debug(unsigned long long* buf) { g_nc_dbg = buf; }
static int g_nc_sabotage = 0;   // (PROBES build only reads it: irsde_bench_naf_chain variant 26)
void naf_chain_set_sabotage(int on) { g_nc_sabotage = on; }

void naf_chain_global_init() {
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(naf_chain_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(naf_chain_kernel<8, false, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(naf_chain_kernel<8, false, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef IRSDE_PROBES
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(naf_chain_kernel<8, false, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(naf_chain_kernel<8, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    IRSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(naf_chain_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
}

// variant: 0 production (= 1); 1 residual stream in registers + ring of 8 weight fragments; 2 residual stream in L2 + ring of 16: measured 1.4x slower
// (profiles/r04_naf_chain_bench_a.txt; the ring of 32 was 1.55x slower): kept as the measurement twin  (IRSDE_NAF_CHAIN_VARIANT under IRSDE_TUNING=1)
void launch_naf_chain(const float* x, float* out, const unsigned short* w, const float* vecs, int nblocks, int B, const float* film, int film_bstride,
                      int film_off, const float* cam, int cam_bstride, int cam_off, hipStream_t s, int variant) {
    NafChainArgs a;
    a.x = x; a.out = out; a.w = w; a.vecs = vecs; a.film = film; a.cam = cam;
    a.film_bstride = film_bstride; a.film_off = film_off; a.cam_bstride = cam_bstride; a.cam_off = cam_off;
    a.nblocks = nblocks;
    a.dbg = g_nc_dbg;
    a.xgate = nullptr; a.xnorm = nullptr; a.xvec = nullptr; a.ctr = nullptr; a.B = B; a.sabotage = 0;
    const size_t wb = naf_chain_weight_halves(nblocks) * 2;
    if (wb >= 0x7fff0000ull) throw HipError("launch_naf_chain: weight stream too large for 32-bit buffer offsets");
    a.w_bytes = (unsigned)wb;
    static const int env_variant = tuning_env_int("IRSDE_NAF_CHAIN_VARIANT", 0);
    if (variant == 0) variant = env_variant ? env_variant : 1;
    if (x == out && variant != 1) throw HipError("launch_naf_chain: in-place call needs the register variant");
    switch (variant) {
        case 1: hipLaunchKernelGGL((naf_chain_kernel<8, false>), dim3((unsigned)B), dim3(512), NC_LDS_BYTES, s, a); break;
#ifdef IRSDE_PROBES   // the cycle-stamp twin and the measured-slower residual-stream-in-L2 variant: measurement build only (make PROBES=1)
        case 11: hipLaunchKernelGGL((naf_chain_kernel<8, false, true>), dim3((unsigned)B), dim3(512), NC_LDS_BYTES, s, a); break;
        case 2: hipLaunchKernelGGL((naf_chain_kernel<16, true>), dim3((unsigned)B), dim3(512), NC_LDS_BYTES, s, a); break;
#endif
        default: throw HipError("launch_naf_chain: bad variant (11 / 2 are measurement variants: make PROBES=1)");
    }
    IRSDE_HIP_CHECK(hipGetLastError());
}

// ---- G groups per image (r06) ----
// The split kernel's weight streams are a permutation of the one-group streams' 1 KB fragments: order[i] = index (in fragments) into the G = 1 buffer of the
// i-th fragment of the [G][8 waves][nblocks][448 / G] buffer.  Same enumeration as pack_naf_chain (engine_weights.hip) and the kernel's passes.
std::vector<int> naf_chain_split_order(int nblocks, int G) {
    if (G != 2 && G != 4) throw HipError("naf_chain_split_order: G must be 2 or 4");
    const int NTW = 4 / G, NT3 = NTW >= 2 ? 2 : 1, NP3 = NTW / NT3;
    // fragment id = (conv 0..4, 16-channel tile 0..63, k step): position in the one-group buffer
    auto pos1 = [&](int conv, int tile, int ks, int blk) -> int {
        // one-group stream of wave w1, block blk: [conv1: 4 ps x 16 ks x (lo, hi)] [sca: 2 ps x 16 x 2] [conv3] [conv4 as conv1] [conv5]
        const bool gated = conv == 0 || conv == 3;
        const int base = conv == 0 ? 0 : conv == 1 ? 128 : conv == 2 ? 192 : conv == 3 ? 256 : 384;
        int w1, f;
        if (gated) {
            const int hi = tile >= 32 ? 1 : 0, t = tile & 31;   // 32 lo tiles, then 32 hi tiles
            w1 = t >> 2;
            f = base + ((t & 3) * 16 + ks) * 2 + hi;
        } else if (conv == 1) {   // sca.1: [quarter][k step of the quarter][tile of the wave]
            w1 = tile >> 2;
            f = base + ks * 4 + (tile & 3);
        } else {
            w1 = tile >> 2;
            const int tt = tile & 3;
            f = base + ((tt >> 1) * 16 + ks) * 2 + (tt & 1);
        }
        return (w1 * nblocks + blk) * NC_FRAGS_PER_BLOCK + f;
    };
    std::vector<int> order;
    order.reserve((size_t)8 * nblocks * NC_FRAGS_PER_BLOCK);
    for (int g = 0; g < G; ++g)
        for (int w = 0; w < 8; ++w)
            for (int blk = 0; blk < nblocks; ++blk) {
                const int tile0 = (g * (NC_C / G) + w * 16 * NTW) / 16;
                auto gated = [&](int conv) {
                    for (int ps = 0; ps < NTW; ++ps)
                        for (int ks = 0; ks < 16; ++ks) { order.push_back(pos1(conv, tile0 + ps, ks, blk)); order.push_back(pos1(conv, 32 + tile0 + ps, ks, blk)); }
                };
                auto plain = [&](int conv) {
                    for (int ps = 0; ps < NP3; ++ps)
                        for (int ks = 0; ks < 16; ++ks)
                            for (int t = 0; t < NT3; ++t) order.push_back(pos1(conv, tile0 + NT3 * ps + t, ks, blk));
                };
                auto sca = [&]() {   // group g: the quarters of its own gated channels, every output tile of wave w
                    for (int qq = 0; qq < 4 / G; ++qq)
                        for (int kk = 0; kk < 4; ++kk)
                            for (int t = 0; t < 4; ++t) order.push_back(pos1(1, 4 * w + t, 4 * (g * (4 / G) + qq) + kk, blk));
                };
                gated(0); sca(); plain(2); gated(3); plain(4);
            }
    if (order.size() != (size_t)8 * nblocks * NC_FRAGS_PER_BLOCK) throw HipError("naf_chain_split_order: stream length mismatch");
    return order;
}

namespace {
__global__ void naf_chain_permute_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const int* __restrict__ order, const int nfrag) {
    const int f = blockIdx.x;
    if (f >= nfrag) return;
    dst[(size_t)f * 64 + threadIdx.x] = src[(size_t)order[f] * 64 + threadIdx.x];
}
}  // namespace

// dst = the [G][8][nblocks][448 / G] fragment streams built from the one-group buffer w1 (both naf_chain_weight_halves(nblocks) halves)
void naf_chain_build_split_weights(const unsigned short* w1, unsigned short* dst, int nblocks, int G, hipStream_t s) {
    const std::vector<int> order = naf_chain_split_order(nblocks, G);
    int* dorder = nullptr;
    IRSDE_HIP_CHECK(hipMalloc(&dorder, order.size() * sizeof(int)));
    IRSDE_HIP_CHECK(hipMemcpy(dorder, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(naf_chain_permute_kernel, dim3((unsigned)order.size()), dim3(64), 0, s, reinterpret_cast<const uint4*>(w1), reinterpret_cast<uint4*>(dst),
                       dorder, (int)order.size());
    IRSDE_HIP_CHECK(hipGetLastError());
    IRSDE_HIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(dorder);
}

size_t naf_chain_split_scratch_bytes(int B) { return (size_t)B * (2 * NC_PX * NC_C * 2 + 4 * NC_C * 4) + (4 * (size_t)B + 1) * 4 + 64; }

// Work-groups the split launch needs resident at the same time: 8 ceil(B / 8) G (an image's groups share a block-id residue mod 8 = an XCD)
int naf_chain_split_groups(int B, int G) { return 8 * ((B + 7) / 8) * G; }

// G = 2 / 4 groups per image.  wG: the split weight streams (naf_chain_build_split_weights); scratch: naf_chain_split_scratch_bytes(B) bytes, 16-byte aligned —
// [gated tensors][LayerNorm outputs][vectors][4 B counters + error word], ZERO before the first launch (the kernel leaves the counters zero; naf_chain_split_reset after an error).
// The caller guarantees naf_chain_split_groups(B, G) <= the CUs no other resident kernel holds for long (see the file header).
void launch_naf_chain_split(const float* x, float* out, const unsigned short* wG, const float* vecs, int nblocks, int B, const float* film, int film_bstride,
                            int film_off, const float* cam, int cam_bstride, int cam_off, int G, void* scratch, hipStream_t s) {
    if (G != 2 && G != 4) throw HipError("launch_naf_chain_split: G must be 2 or 4");
    if (naf_chain_split_groups(B, G) > device_cu_count()) throw HipError("launch_naf_chain_split: more work-groups than compute units (they must be co-resident)");
    NafChainArgs a;
    a.x = x; a.out = out; a.w = wG; a.vecs = vecs; a.film = film; a.cam = cam;
    a.film_bstride = film_bstride; a.film_off = film_off; a.cam_bstride = cam_bstride; a.cam_off = cam_off;
    a.nblocks = nblocks;
    a.dbg = g_nc_dbg;   // (PROBES build, G = 4: the cycle-stamp twin when a stamp buffer is set)
    char* sc = reinterpret_cast<char*>(scratch);
    a.xgate = reinterpret_cast<unsigned short*>(sc);
    a.xnorm = reinterpret_cast<unsigned short*>(sc + (size_t)B * NC_PX * NC_C * 2);
    a.xvec = reinterpret_cast<float*>(sc + (size_t)B * 2 * NC_PX * NC_C * 2);
    a.ctr = reinterpret_cast<unsigned*>(sc + (size_t)B * (2 * NC_PX * NC_C * 2 + 4 * NC_C * 4));
    a.B = B;
    a.sabotage = g_nc_sabotage;
    const size_t wb = naf_chain_weight_halves(nblocks) * 2;
    if (wb >= 0x7fff0000ull) throw HipError("launch_naf_chain_split: weight stream too large for 32-bit buffer offsets");
    a.w_bytes = (unsigned)wb;
    const dim3 grid((unsigned)naf_chain_split_groups(B, G));
#ifdef IRSDE_PROBES
    if (G == 4 && a.dbg) {
        hipLaunchKernelGGL((naf_chain_kernel<8, false, true, 4>), grid, dim3(512), NC_LDS_BYTES, s, a);
        IRSDE_HIP_CHECK(hipGetLastError());
        return;
    }
#endif
    if (G == 2) hipLaunchKernelGGL((naf_chain_kernel<8, false, false, 2>), grid, dim3(512), NC_LDS_BYTES, s, a);
    else hipLaunchKernelGGL((naf_chain_kernel<8, false, false, 4>), grid, dim3(512), NC_LDS_BYTES, s, a);
    IRSDE_HIP_CHECK(hipGetLastError());
}

// the error word of a split launch's scratch buffer (device pointer): non-zero after a run whose groups were not co-resident
const unsigned* naf_chain_split_error_flag(const void* scratch, int B) {
    return reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(scratch) + (size_t)B * (2 * NC_PX * NC_C * 2 + 4 * NC_C * 4)) + 4 * B;
}
// counters + error word back to zero (synchronous; after an error was reported)
void naf_chain_split_reset(void* scratch, int B) {
    IRSDE_HIP_CHECK(hipMemset(reinterpret_cast<char*>(scratch) + (size_t)B * (2 * NC_PX * NC_C * 2 + 4 * NC_C * 4), 0, (4 * (size_t)B + 1) * 4));
}

}  // namespace irsde
