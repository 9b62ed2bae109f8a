// Shared declarations for the IR-SDE gfx950 engine (internal; the public C ABI is include/irsde_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace irsde {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct HipOutOfMemory : HipError {   // a device allocation failed: the plan cache evicts and retries (engine_plan.hip, get_plan)
    using HipError::HipError;
};

// Tuning knobs (A/B experiments logged under profiles/) are honoured only when IRSDE_TUNING=1 is set: a stray
// environment variable must not change the launch plan the parity tests validated.  Returns the integer value of
// `name`, or `dflt` when tuning is off or the variable is unset.
inline int tuning_env_int(const char* name, int dflt) {
    static const bool on = [] { const char* t = getenv("IRSDE_TUNING"); return t && atoi(t) == 1; }();
    if (!on) return dflt;
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

#define IRSDE_HIP_CHECK(expr)                                                                  \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            throw ::irsde::HipError(std::string(#expr) + " failed: " + hipGetErrorString(_e) + \
                                    " (" __FILE__ ":" + std::to_string(__LINE__) + ")");       \
        }                                                                                      \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM convolution, NHWC fp32, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//   out[m][n] = epilogue( sum_{ky,kx,c} in[(b, oy*stride-pad_y+ky, ox*stride-pad_x+kx), c] * w[n][ky][kx][c] )
// with m = (b*Ho + oy)*Wo + ox.  The input may be the channel-concatenation of two tensors
// (torch.cat([x, skip], 1) is never materialised) and may be read through a nearest x2 upsample
// (in_shift = 1: virtual pixel (y,x) reads physical (y>>1, x>>1)).
// Epilogue order: +bias[n] -> *(film_scale[n]+1)+film_shift[n] -> SiLU -> +res[m][n].
// ---------------------------------------------------------------------------------------------
struct ConvParams {
    const float* in0 = nullptr;
    const float* in1 = nullptr;  // second concat source or null
    int C0 = 0, C1 = 0;          // channels looped per source (multiples of 32)
    int pix0 = 0, pix1 = 0;      // floats between consecutive pixels of each source
    int Hin = 0, Win = 0;        // physical input height/width (both sources)
    int in_shift = 0;            // 1: fused nearest x2 upsample
    const float* w = nullptr;    // [Cout][KH*KW][C0+C1]
    const unsigned short* w_bf = nullptr;  // same layout rounded to bf16: selects the bf16-MFMA kernel (fp32 accumulate)
    int f16 = 0;                 // 1: w_bf / w_pair hold IEEE fp16 (IRSDE_FLAG_FP16 / _SPLIT_F16X2): v_mfma_f32_32x32x16_f16
    // split-operand arithmetic on fp32 storage (IRSDE_FLAG_SPLIT_BF16X2 / _F16X2; conv_igemm.hip PAIR kernels): the weights as two
    // pair-interleaved 16-bit pieces [Cout][KH*KW*(C0+C1) / 32][hi 32 | lo 32] (split_pairs_kernel); fp16: pieces of w * 2^k, pair_scale = 2^-k
    const unsigned short* w_pair = nullptr;
    float pair_scale = 1.f;
    int Cout = 0;
    int KH = 1, KW = 1, stride = 1, pad_y = 0, pad_x = 0;
    int B = 0, Ho = 0, Wo = 0;
    float* out = nullptr;
    int out_stride = 0;          // floats between consecutive output pixels (>= Cout)
    const float* bias = nullptr;
    const float* film = nullptr;  // row r: [scale(Cout) | shift(Cout)]
    int film_bstride = 0;         // floats between rows of different batch items (0: one shared row)
    int silu = 0;
    const float* res = nullptr;   // [M][res_stride]
    int res_stride = 0;
    // split-K (small-M layers): partial sums are written to `partial` [splits][M][Cout] and the
    // epilogue runs in conv_splitk_reduce.
    int splits = 1;
    float* partial = nullptr;
    const float* zeros = nullptr;  // >= 16 zero bytes: out-of-image taps are loaded from here (address select, no branch)
    // --- NAFNet (Refusion) fusions -------------------------------------------------------------------------
    const float* ch_scale = nullptr;  // [Cout]: v *= ch_scale[n] after the activation, before +res  (beta / gamma)
    const float* in_scale = nullptr;  // [B][C0]: input element (b, k) is multiplied by in_scale[b][k] while staged (SCA)
    int gate = 0;      // SimpleGate epilogue: weight rows are interleaved (2j <- j, 2j+1 <- j + Cout/2); writes the Cout/2
                       // products out[m][j] = v[2j] * v[2j+1]  (out_stride counts the gated width)
    const float* gate_film = nullptr;  // gate epilogue only: product j *= (gate_film[b][j] + 1), += gate_film[b][Cout/2 + j]
    int gate_film_bstride = 0;         // floats between the rows of different batch items (latent-bokeh lens FiLM)
    int shuffle = 0;   // PixelShuffle(2) epilogue: weight rows ordered n' = (dy*2+dx)*Cout/4 + co; writes (and adds res at)
                       // out[b][2y+dy][2x+dx][co]  (out/res tensors are [B][2Ho][2Wo][Cout/4])
    // batched launch (Winograd: 16 independent GEMMs): blockIdx.z = k offsets the three tensors (floats)
    int nz = 1;
    long long z_in = 0, z_w = 0, z_out = 0;
    int no_direct_epi = 0;  // tuning (irsde_bench_conv variant 7 / IRSDE_NO_DIRECT_EPI): LDS-transposed epilogue in the f32 BUFA kernels too
    // bf16 activation storage (IRSDE_FLAG_BF16_ACT, bf16-MFMA kernels only): in0/in1 resp. out/res point at bf16 tensors
    // (strides in elements); accumulation and the epilogue arithmetic stay fp32
    int in_bf16 = 0, out_bf16 = 0;
    // channel LayerNorm fused into the epilogue (LinearAttention.to_out = Conv2d + LayerNorm, module_util.py:158-161, then the
    // Residual add): v = (v - mean_n v) * rsqrt(var_n v + eps) * ln_g[n], applied after bias and before +res.  Needs the
    // whole output row in one tile: Cout == 64 or 128, no split-K.
    const float* ln_g = nullptr;
    float ln_eps = 1e-5f;
};

// Winograd F(m x m,3x3) transforms (wino.hip).  Tiles: T = B * TH * TW with TH = Ho/m, TW = Wo/m.
struct WinoParams {
    const float* in0 = nullptr;
    const float* in1 = nullptr;
    int C0 = 0, C1 = 0, Hin = 0, Win = 0, in_shift = 0;
    int B = 0, TH = 0, TW = 0, T = 0;
    int tile = 2;               // output tile edge m of F(m x m, 3x3): 2 or 4; components = (m+2)^2
    float* V = nullptr;         // [(m+2)^2][T][C0+C1]
    // split-operand mode (gemm_split.hip): V is written as `nplanes` bf16 planes instead, element (p, k, t, c) at
    // Vs + p * v_plane + (k * T + t) * (C0 + C1) + c
    unsigned short* Vs = nullptr;
    int nplanes = 0;
    long long v_plane = 0;
    int v_f16 = 0;              // pairs only: 1 = IEEE fp16 pieces of V * v_scale (v_scale a power of two), 0 = bf16 pieces of V
    float v_scale = 1.f;
    int v_pairs = 0;            // 1: two planes, pair-interleaved: element (k, t, c, p) at Vs + ((k * T + t) * (Ctot / 32) + c / 32) * 64 + p * 32 + c % 32
    const float* M = nullptr;   // [(m+2)^2][T][Cout]
    int Cout = 0;
    float* out = nullptr;
    int out_stride = 0;
    const float* bias = nullptr;
    const float* film = nullptr;
    int film_bstride = 0;
    int silu = 0;
    const float* res = nullptr;
    int res_stride = 0;
};
void launch_wino_input(const WinoParams& p, hipStream_t s);
// fp32-equivalent GEMMs on the bf16 MFMA pipe (gemm_split.hip): C_z[m][n] = sum_k A_z[m][k] B_z[n][k] with both operands given
// as 2 or 3 bf16 planes (hi / mid / lo pieces of the f32 value), f32 accumulate, z = 0 .. ncomp-1.
struct SplitGemmArgs {
    const unsigned short* a = nullptr;  // element (plane p, z, m, k) at a + p * plA + z * pA + m * lda + k
    const unsigned short* b = nullptr;  // element (plane q, z, n, k) at b + q * plB + z * pB + n * K + k
    float* out = nullptr;               // element (z, m, n) at out + z * pO + m * ldc + n
    long long plA = 0, plB = 0, pA = 0, pB = 0, pO = 0;
    int M = 0, N = 0, K = 0, lda = 0, ldc = 0;
    float out_scale = 1.f;              // fp16 pairs: the power of two that undoes the operand scales (applied to the accumulators)
    int n_inner = 1;                    // components walked by one block (gemm_split_inner)
    int nblk_n = 0;                     // set by the launcher
};
void gemm_split_global_init();
int gemm_split_inner(int M, int N, int ncomp);
// two planes in the pair-interleaved layout [row][k / 32][plane][32 k] (v3 kernel: LDS-DMA, 256 x 256 tiles); pA / pB = elements
// per component and plane (M * K resp. N * K), plA / plB unused
// f16: IEEE binary16 pieces (hi + lo = 22+ significand bits: fp32-equivalent products) instead of bf16 (16 bits); a.out_scale undoes
// the power-of-two scales the operand writers applied to stay inside fp16's range
void launch_gemm_split_pairs(const SplitGemmArgs& a, int ncomp, hipStream_t s, int abl = 0, bool f16 = false);
void launch_split_pairs(const float* in, unsigned short* out, size_t rows, int K, hipStream_t s, bool f16 = false,
                        float scale = 1.0f);  // f32 [rows][K] -> pair-interleaved hi / lo (f16: of in * scale)
void launch_gemm_split(const SplitGemmArgs& a, int nplanes, int ncomp, hipStream_t s);
void launch_split_planes(const float* in, unsigned short* out, size_t n, size_t plane, int nplanes, hipStream_t s, bool f16 = false,
                         float scale = 1.0f);  // f32 -> bf16 pieces (f16: two IEEE fp16 pieces of in * scale), plane-major
void launch_wino_output(const WinoParams& p, hipStream_t s);
void wino_transform_weights(const float* w_packed, int Cout, int Cin, float* U, int tile);  // host
// Fused Winograd F(4x4,3x3) convolution (wino_fused.hip): transforms and the 36 component GEMMs in one kernel.  `p` is the
// plain 3x3 s1 p1 convolution (as for launch_conv), Uf the weights from wino_fused_pack_weights.
bool wino_fused_eligible(const ConvParams& p);
void wino_fused_pack_weights(const float* U, int Cout, int Cin, float* Uf);  // host: U[36][Cout][Cin] -> fragment order
void launch_wino_fused(const ConvParams& p, const float* Uf, hipStream_t s, unsigned long long* dbg = nullptr, int dflags = 0);
int wino_fused_num_blocks(const ConvParams& p);
// r03: 16 tiles x 64 couts per block (Cout and Cin multiples of 64); its own weight fragment order
bool wino_fused64_eligible(const ConvParams& p);
void wino_fused64_pack_weights(const float* U, int Cout, int Cin, float* Uf);
void launch_wino_fused64(const ConvParams& p, const float* Uf, hipStream_t s, int variant = 0);
void wino_fused64_set_debug(unsigned long long* buf);   // stamp buffer of launch variant 25 (irsde_bench_conv 435)
void wino_fused64_set_opt(int opt);                     // OPT value of the tuning twins (launch variants 26 / 27; irsde_bench_conv 436 / 437)
// the fp16-pair twin (variant 4): V is split as V / 16, the weights as scale * U (scale a power of two), p.pair_scale undoes both
constexpr float kWinoFused64PairVScale = 0.0625f;
void launch_wino_fused64_split_weights(const float* Uf, unsigned short* out, size_t nfloats, float scale, hipStream_t s);
int wino_fused64_num_blocks(const ConvParams& p);
bool wino_fused64_xcd_nb(const ConvParams& p);   // the launch maps cout blocks to XCDs (layers whose input is small next to U x rounds)
void wino_fused_global_init();
int device_cu_count();   // compute units of the CURRENT device, cached per device ordinal (conv_igemm.hip)
// r06: 32 tiles x 64 couts per work item, every weight fragment feeds two tile groups (wino_fused_t.hip); weights = wino_fused64_pack_weights order
bool wino_fused64t_eligible(const ConvParams& p);
long long wino_fused64t_num_items(const ConvParams& p);
void launch_wino_fused64t(const ConvParams& p, const float* Uf, hipStream_t s, int variant = 0);
void wino_fused64t_set_debug(unsigned long long* buf);   // stamp buffer of launch variant 5
void wino_fused64t_set_skew(int cycles);                 // tuning: start skew per phase class
void wino_fused_t_global_init();
// fused NAFBlock chain (naf_chain.hip): consecutive 512-channel NAFBlocks on an 8 x 8 feature map, one work-group per image
void naf_chain_global_init();
void naf_chain_set_debug(unsigned long long* buf);   // stamp buffer of launch variant 11: [B][8 waves][16]
bool naf_chain_shape_ok(int H, int W, int c);
size_t naf_chain_weight_halves(int nblocks);   // fp16 fragment streams [8 waves][nblocks][448 fragments][512]
size_t naf_chain_vec_floats(int nblocks);      // fp32 per-channel vectors [nblocks][NV_TOTAL = 14848]
void launch_naf_chain(const float* x, float* out, const unsigned short* w, const float* vecs, int nblocks, int B, const float* film, int film_bstride,
                      int film_off, const float* cam, int cam_bstride, int cam_off, hipStream_t s, int variant = 0);
// r06: G = 2 / 4 work-groups per image (the groups trade operand slices through L2; see naf_chain.hip)
std::vector<int> naf_chain_split_order(int nblocks, int G);   // fragment permutation: split stream position -> one-group stream position
void naf_chain_build_split_weights(const unsigned short* w1, unsigned short* dst, int nblocks, int G, hipStream_t s);
size_t naf_chain_split_scratch_bytes(int B);
int naf_chain_split_groups(int B, int G);                     // work-groups that must be co-resident
void launch_naf_chain_split(const float* x, float* out, const unsigned short* wG, const float* vecs, int nblocks, int B, const float* film, int film_bstride,
                            int film_off, const float* cam, int cam_bstride, int cam_off, int G, void* scratch, hipStream_t s);
const unsigned* naf_chain_split_error_flag(const void* scratch, int B);
void naf_chain_split_reset(void* scratch, int B);
void naf_chain_set_sabotage(int on);   // PROBES build: one group per image never arrives (test of the spin limit)
void attention_global_init();

double conv_flops(const ConvParams& p);  // 2*M*Cout*K (algorithmic)
void launch_conv(const ConvParams& p, hipStream_t s);
void conv_global_init();
// bf16 mode: 3x3 s1 p1 layers with an LDS-resident halo tile (conv_halo.hip)
void conv_halo_global_init();
bool conv_halo_eligible(const ConvParams& p);
bool conv_halo2_wanted(const ConvParams& p, int force);   // true: launch_conv_halo runs the layer on the 512-pixel x 128-channel kernel (irsde_plan_describe names it)
void launch_conv_halo(const ConvParams& p, hipStream_t s, int force_halo2 = 0);   // force_halo2: 1 = the 512-pixel x 128-channel kernel, -1 = the 256-pixel one, 0 = by shape
void conv_set_variant(int v);  // tuning experiments (irsde_bench_conv)
void launch_fill_random(float* p, size_t n, unsigned seed, float scale, hipStream_t s);  // sets the dynamic-LDS attribute of every tile configuration (call before graph capture)
// SiLU for the epilogues of the f32 MFMA kernels: v_exp_f32 + v_rcp_f32 (1 ulp each) — 5 vector instructions instead of the
// ~25 of expf() + IEEE division.  On gfx950 the vector instructions of an f32-MFMA kernel are paid in matrix-pipe time
// (tools/probe/mfma_valu_*.hip); the epilogue SiLU was a third of the fused Winograd kernel's vector instructions.
// |result - v / (1 + exp(-v))| <= ~3e-7 |result| (the exponent product adds |v| 2^-24 relative to e^-v, which only matters
// where SiLU itself is ~0); e^-v = inf gives v * 0.
__device__ __forceinline__ float silu_hw(const float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
}

void launch_f32_to_bf16(const float* in, unsigned short* out, size_t n, hipStream_t s);  // round-to-nearest-even
void launch_f32_to_f16(const float* in, unsigned short* out, size_t n, hipStream_t s);   // IEEE binary16, round-to-nearest-even
void launch_bf16_to_f32(const unsigned short* in, float* out, size_t n, hipStream_t s);
// naive direct convolution on VALU (one thread per output) — debug / cross-check path only
void launch_conv_naive(const ConvParams& p, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Memory-bound helpers (kernels_misc.hip)
// ---------------------------------------------------------------------------------------------
// Channel LayerNorm over C per pixel (gain only, eps inside rsqrt), optional residual add.
// bf16 = true: x / res / out are bf16 tensors behind the float* (IRSDE_FLAG_BF16_ACT); arithmetic stays fp32.
void launch_layernorm(const float* x, const float* g, const float* res, float* out, int64_t M, int C,
                      float eps, hipStream_t s, bool bf16 = false);
// NAFNet: y = LN(x) * g * (scale + 1) + shift with per-channel FiLM rows (row stride film_bstride per batch item, 0 = shared)
void launch_layernorm_film(const float* x, const float* g, const float* scale, const float* shift, int film_bstride,
                           int64_t pixels_per_image, float* out, int64_t M, int C, float eps, hipStream_t s);
// NAFNet: depthwise 3x3 (pad 1, bias) over u [B][H][W][2c] fused with SimpleGate -> out [B][H][W][c], plus per-tile
// channel sums partial[b][tile][c] (deterministic two-stage global average pool).  w: [9][2c], bias: [2c].
int dwgate_tiles(int H, int W, int c);
// r06: NAFBlock norm + FiLM + 1x1 convolution (+ SimpleGate) in one launch, fp16 operand mode, c = 64 / 128 / 256 (kernels_misc.hip)
bool naf_lnconv_ok(int c, int Cout, long long M);
void launch_naf_pwconv(const float* x, const float* in_scale, int64_t pixels_per_image, const unsigned short* w16, const float* bias, const float* ch_scale,
                       const float* res, float* out, int64_t M, int c, int Cout, hipStream_t s);
void naf_lnconv_global_init();
void launch_naf_lnconv(const float* x, const float* g, const float* fscale, const float* fshift, int film_bstride, int64_t pixels_per_image,
                       const unsigned short* w16, const float* bias, float* out, int64_t M, int c, int Cout, int gate, const float* gate_film,
                       int gate_film_bstride, hipStream_t s);  // tiles of launch_dwconv_gate = rows of its `partial` buffer per image
void launch_dwconv_gate(const float* u, const float* w, const float* bias, float* out, float* partial, int B, int H, int W,
                        int c, hipStream_t s);
// NAFNet SCA: s[b][o] = bias[o] + sum_k W[o][k] * mean_hw(gated)[b][k]
void launch_sca(const float* partial, int ntiles, const float* W, const float* bias, float* mean, float* s_out, int B,
                int c, int HW, hipStream_t s);  // mean: scratch [B][c]
// out[r][j] = in[r][j] * in[r][j + h]  (SimpleGate on time-embedding rows)
void launch_row_gate(const float* in, float* out, int rows, int h, hipStream_t s);

struct AttnWorkspace {
    float* pmax = nullptr;   // [B][nch][128]
    float* pctx = nullptr;   // [B][4][nch][1024]
    float* psum = nullptr;   // [B][4][nch][32]
    float* ctx = nullptr;    // [B][4][32][32]
    int nch = 0;             // N-chunks per image
};
int attn_num_chunks(int N, int B);
// qkv: [B][N][384] (q | k | v, 4 heads x 32 each).  out: [B][N][128].
void launch_linear_attention(const float* qkv, float* out, int B, int N, const AttnWorkspace& ws, hipStream_t s,
                             bool bf16 = false);

// fp32 fused form of the same block: k, v and their softmax / context from the LayerNorm output and the k | v rows of
// to_qkv.weight ([256][C]) without a k / v tensor in HBM; q is a separate [B][N][128] tensor (its own 1x1 convolution).
// ln_g != nullptr: `xn` is the un-normalised block input and PreNorm's LayerNorm (* ln_g, eps) runs inside the kernel.
// wkv_pair != nullptr (IRSDE_FLAG_SPLIT_F16X2; C = 64 / 128 / 256): the projection on fp16 hi + lo operand pairs — wkv_pair = the two
// planes [2][256][C] of wkv * 2^k, w_inv_scale = 2^-k
void launch_attention_kv_context(const float* xn, const float* wkv, int B, int N, int C, const AttnWorkspace& ws, hipStream_t s,
                                 const float* ln_g = nullptr, float ln_eps = 1e-5f, const unsigned short* wkv_pair = nullptr,
                                 float w_inv_scale = 1.f, int op16 = 0, bool act_bf16 = false);
// op16 (r05): 0 = f32 (or, with *_pair planes, the fp16 hi + lo pairs); 2 / 3 = ONE bf16 / fp16 operand plane in *_pair (IRSDE_FLAG_BF16 / _FP16:
// [256][C] k | v rows, [128][C] q rows, [C][128] to_out rows, unscaled); act_bf16 (op16 == 2 only): xn / x / y are bf16 tensors (IRSDE_FLAG_BF16_ACT)
void launch_attention_q_out(const float* q, float* out, int B, int N, const AttnWorkspace& ws, hipStream_t s);
// C = 64 / 128 / 256: q projection, softmax over d, context product, to_out (+ bias), LayerNorm (* g2) and the residual in one
// kernel: y = LayerNorm(Wout . (ctx^T softmax(Wq . xn)) + bias) * g2 + x.  wq = rows 0..127 of to_qkv.weight ([128][C]),
// wout = to_out.0.weight ([C][128]); needs ws.ctx from launch_attention_kv_context.
void launch_attention_q_out_fused(const float* xn, const float* x, const float* wq, const float* wout, const float* bias,
                                  const float* g2, float* y, int B, int N, int C, float eps, const AttnWorkspace& ws, hipStream_t s,
                                  const float* ln_g = nullptr, const unsigned short* wq_pair = nullptr, const unsigned short* wout_pair = nullptr,
                                  float wq_inv = 1.f, float wout_inv = 1.f, int op16 = 0, bool act_bf16 = false);   // *_pair: fp16 hi / lo planes (IRSDE_FLAG_SPLIT_F16X2), see kernels_misc.hip

// Full softmax attention over N tokens (denoising-sde bottleneck): qkv [B][N][384] -> out [B][N][128].
void launch_full_attention(const float* qkv, float* out, int B, int N, hipStream_t s);

// xt, cond: NCHW [B][3][H][W] (cond may be null: unconditional variant, channels = xt only).  x0: [B][Hp+6][Wp+6][8] (+8 floats slack), zero border of 3,
// channels {xt-cond (3), cond (3), 0, 0}, reflect-padded right/bottom from (H,W) to (Hp,Wp).
void launch_prep_input(const float* xt, const float* cond, float* x0, int B, int in_nc, int H, int W, int Hp, int Wp,
                       hipStream_t s, int reflect = 1);

// FiLM / time-embedding path.
//   temb0[r][i] = sin/cos(t_r * freq[i])          (SinusoidalPosEmb)
//   generic row-linear: out[r][o] = act_out( sum_i act_in(in[r][i]) * W[o][i] + b[o] )
enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2 };
void launch_sinusoid(const float* tvals, const float* freqs, float* out, int rows, int half, hipStream_t s);
void launch_row_linear(const float* in, int in_stride, const float* W, const float* b, float* out, int out_stride,
                       int rows, int in_dim, int out_dim, int act_in, int act_out, hipStream_t s);

// Per-step device state (lets one captured hipGraph replay for every t).
struct StepState {
    int t;           // current step (T..1)
    int t_next;      // step the next step_begin will take
    int pad0, pad1;
    float coef[12];  // row t of the coefficient table (see include/irsde_hip.h)
};
// cur.t = cur.t_next--, copy film_table[t] -> film_cur and coef_table[t] -> state.coef
void launch_step_begin(StepState* st, const float* film_table, int film_row, float* film_cur,
                       const float* coef_table, hipStream_t s);
void launch_set_step(StepState* st, int t_next, hipStream_t s);

// Per-call sampler arguments kept in device memory so that a captured graph stays call-invariant.
struct SampleCtl {
    int mode;
    int pad;
    const float* noise;
    long long noise_tstride;
    unsigned long long seed;
    unsigned long long image_offset;
};
void launch_set_ctl(SampleCtl* ctl, int mode, const float* noise, long long noise_tstride, unsigned long long seed,
                    unsigned long long image_offset, hipStream_t s);

// eps_hat [B][Hp][Wp][stride] (NHWC, padded) -> out [B][C][H][W] (crop)
void launch_unpack_pred(const float* pred, float* out, int B, int C, int H, int W, int Hp, int Wp, int stride, hipStream_t s);
void launch_add(const float* a, const float* b, float* out, size_t n, hipStream_t s);  // n % 4 == 0
// NCHW -> NHWC with channel padding to Cp (zeros) and spatial padding to (Hp, Wp) (reflect or zero)
void launch_nchw_to_nhwc_pad(const float* in, float* out, int B, int C, int H, int W, int Hp, int Wp, int Cp, int reflect,
                             hipStream_t s);
// NHWC [B][H][W][C] -> NCHW (debug taps)
void launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int H, int W, hipStream_t s, bool bf16 = false);

// Reverse-step update x <- f(x, mu, eps_hat, z) on NCHW state; eps_hat addressed by strides.
struct UpdateParams {
    float* x;            // [B][C][H][W] in/out
    const float* mu;     // [B][C][H][W]
    const float* pred;   // eps_hat, element (b,c,y,x) at pred[b*sb + c*sc + y*sy + x*sx]
    int64_t sb, sc, sy, sx;
    const float* noise;  // injected noise base [T+1][B][C][H][W] or null (=> Philox)
    int64_t noise_tstride;
    const StepState* st;  // device step state, or null => use t_imm / coef_imm
    const SampleCtl* ctl; // device per-call arguments (override mode/noise/seed/image_offset) or null
    int t_imm;
    float coef_imm[12];
    int mode;            // IRSDE: 0 sde, 1 ode, 2 posterior; DenoisingSDE: 3 sde, 4 ode
    int B, C, H, W;
    uint64_t seed;
    uint64_t image_offset;  // global index of image 0 of this shard (RNG invariance to sharding)
    int batch0;          // r05 sub-batch plans: x / mu / pred hold images [batch0, batch0 + B) of the call's batch — the injected noise tensor and the
                         // Philox image index are addressed with the call-level index b + batch0
};
void launch_sde_update(const UpdateParams& p, hipStream_t s);
// fills out[B][C][H][W] with the Philox N(0,1) draw for step t (test hook for the RNG)
void launch_philox_normal(float* out, int B, int CHW, int t, uint64_t seed, uint64_t image_offset, hipStream_t s);

// Evaluation tail (eval_metrics.hip): per-image sums for PSNR / SSIM on RGB and Y.  sums_host: [B][4] =
// {squared-error sum RGB, SSIM-map sum RGB, squared-error sum Y, SSIM-map sum Y}.  Synchronises the stream.
void eval_metrics(const float* out, const float* gt, int B, int C, int H, int W, int crop, double* sums_host, hipStream_t s);
void tensor2img_u8(const float* in, unsigned char* out, int B, int C, int H, int W, hipStream_t s);

}  // namespace irsde
