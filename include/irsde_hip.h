/* irsde_hip.h — C ABI of libirsde_hip.so: the MI355X (gfx950) IR-SDE reverse-diffusion sampler.
 *
 * The reference (Algolzw/image-restoration-sde) is pure Python/PyTorch and has no FFI; its seam for
 * this path is Python duck typing at two objects:
 *   - the `sde` object  (codes/utils/sde_utils.py:80-361, class IRSDE) and
 *   - the score model   (codes/config/deraining/models/modules/DenoisingUNet_arch.py:18-134,
 *                        class ConditionalUNet), driven by DenoisingModel.test()
 *                        (codes/config/deraining/models/denoising_model.py:150-160).
 * Each entry point below names the reference interface it replaces.  All pointers are plain
 * device (or, where stated, host) pointers to float32; no torch types cross this boundary.
 * Image tensors are NCHW [B][C][H][W] contiguous exactly as the reference passes them; the NHWC
 * working layout is internal.  `stream` is a hipStream_t (NULL = the legacy default stream); every
 * call is stream-ordered with respect to it and does not synchronise the device unless stated.
 *
 * Return value: 0 on success, negative IRSDE_ERR_* otherwise; irsde_last_error() gives the text.
 *
 * ABI changelog (irsde_version()):
 *   100  round 1.
 *   101  irsde_sample: T == 0 runs NO step and copies xT to out (the reference's `range(1, T + 1)` is empty); only T < 0
 *        selects the full schedule.  (Version 100 treated T <= 0 as "full schedule".)
 *   102  IRSDE_FLAG_SPLIT_BF16X2 / IRSDE_FLAG_SPLIT_F16X2 (split-operand GEMMs on the 16-bit MFMA pipe for the deep Winograd layers).
 *   103  IRSDE_FLAG_NO_NAF_CHAIN (the fused NAFBlock chain of the fp16 ConditionalNAFNet on 8 x 8 feature maps is on by default).
 *   104  irsde_latent_encode / irsde_latent_decode accept hidden == NULL (the skips stay resident in the engine between the two calls: no
 *        NCHW round trip of 4.8 GB per 64 images); irsde_latent_hidden exports one resident skip.
 *   105  irsde_sample on a ConditionalNAFNet whose plan contains the per-image NAFBlock chain (fp16 mode, 8 x 8 level) runs batches of >= 64 images as
 *        2 concurrent sub-batches, each with its own step graph on its own stream (same results up to the tilings the smaller plans choose; noise / Philox
 *        streams stay keyed by the call-level image index).  Debug header: irsde_debug_force_subbatches.  The measurement kernels (stamp / ablation /
 *        superseded twins) moved to the PROBES build (libirsde_hip_probes.so): the product library refuses their selectors.
 *   106  IRSDE_FLAG_BF16 / _BF16_ACT / _FP16: the LinearAttention blocks (C = 64 / 128 / 256) run on the fused attention kernels with 16-bit
 *        projection operands (results differ from ABI 105 by the roundings of the q | k | v / attention-output tensors that no longer exist).
 *        (ABI 107: from 8 images on.)
 *   107  the per-image NAFBlock chain of the fp16 ConditionalNAFNet runs on 2 / 4 work-groups per image where they fit the compute units (results equal
 *        to the one-group kernel's: same operations in the same order); a split launch whose groups were not co-resident fails the NEXT irsde_sample
 *        call on the engine instead of hanging the GPU.  Debug header: irsde_debug_force_chain_groups; irsde_bench_naf_chain variants 22 / 24.
 */
#ifndef IRSDE_HIP_H
#define IRSDE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct irsde_engine irsde_engine;

enum {
    IRSDE_OK = 0,
    IRSDE_ERR_INVALID = -1, /* bad argument / unsupported shape */
    IRSDE_ERR_HIP = -2,     /* a HIP runtime call failed */
    IRSDE_ERR_STATE = -3,   /* call order violated (weights not finalized, schedule not set ...) */
    IRSDE_ERR_WEIGHT = -4   /* unknown weight name / wrong shape / missing weights */
};

enum { IRSDE_MODE_SDE = 0, IRSDE_MODE_ODE = 1, IRSDE_MODE_POSTERIOR = 2,
       /* DenoisingSDE.reverse_sde / reverse_ode (sde_utils.py:488-528): no mu term; coef[9] = exp(-2*thetas_cumsum[t]*dt) */
       IRSDE_MODE_DSDE_SDE = 3, IRSDE_MODE_DSDE_ODE = 4 };

/* engine flags (irsde_config.flags) */
enum {
    IRSDE_FLAG_KEEP_ACTIVATIONS = 1, /* never recycle activation buffers: enables irsde_debug_tap */
    IRSDE_FLAG_NAIVE_CONV = 2,       /* debug: run every convolution on the naive VALU kernel */
    IRSDE_FLAG_NO_WINOGRAD = 4,      /* run every 3x3 layer as a direct implicit GEMM (no Winograd) */
    IRSDE_FLAG_UNCOND_FULLATTN = 16, /* the denoising-sde variant of ConditionalUNet (codes/config/denoising-sde/models/modules/
                                        DenoisingUNet_arch.py:20-130): forward(x, time) without a condition input (init_conv takes
                                        in_nc channels) and full softmax Attention at the bottleneck (module_util.py:182-204) */
    IRSDE_FLAG_BF16 = 32,            /* reduced-precision mode (BASELINE configs[2]; the reference is fp32 only): every convolution
                                        runs on v_mfma_f32_32x32x16_bf16 with operands rounded to bf16 (RNE) and fp32
                                        accumulation; no Winograd; everything else (state, LayerNorm, attention, FiLM,
                                        update step) stays fp32.  ABI 106: the LinearAttention blocks with 64 / 128 / 256 channels run as the
                                        two fused kernels of the fp32 path with their three projections (to_qkv rows, to_out) on the 16-bit MFMA
                                        (operands rounded once, RNE; LayerNorm, both softmaxes, the context and every accumulation in fp32):
                                        q | k | v and the attention output are no longer rounded to a stored tensor.
                                        IRSDE_FLAG_NO_FUSED_ATTN restores the to_qkv convolution + q | k | v tensor */
    IRSDE_FLAG_BF16_ACT = 128,       /* IRSDE_FLAG_BF16 plus bf16 storage of every activation tensor between the prepped input
                                        and eps_hat (conditional ConditionalUNet only): halves the HBM/L2 traffic of the
                                        bandwidth-bound layers; LayerNorm / attention / epilogue arithmetic stays fp32 */
    IRSDE_FLAG_NO_FUSED_LN = 256,    /* keep LinearAttention.to_out's LayerNorm + residual as a separate kernel (default: fused into the
                                        1x1 conv's epilogue when the channel row fits one tile, C = 64 or 128) */
    IRSDE_FLAG_NAF_LENS = 512,       /* ConditionalNAFNet of latent-bokeh (codes/config/latent-bokeh/models/modules/DenoisingNAFNet_arch.py):
                                        time_mlp.{0,2} / block time_mlp.1 names, a lens-information embedding (cam_mlp) and a
                                        per-block FiLM on the gated FFN activation; needs irsde_set_lens_info before forward/sample */
    IRSDE_FLAG_NAF_INTRO_SKIP = 64,  /* ConditionalNAFNet of the latent tasks (codes/config/latent-dehazing/models/modules/
                                        DenoisingNAFNet_arch.py:162-176): ending(x + intro(x)) instead of ending(x) */
    IRSDE_FLAG_FP16 = 1024,          /* IRSDE_FLAG_BF16's mode with IEEE fp16 operands instead (BASELINE configs[4] names fp16; the reference is
                                        fp32 only): every convolution runs on v_mfma_f32_32x32x16_f16 with activations and weights
                                        rounded to binary16 (RNE, 11 significand bits vs 8 for bf16) and fp32 accumulation;
                                        activation storage, LayerNorm, attention, FiLM and the update step stay fp32.  Not
                                        combinable with IRSDE_FLAG_BF16_ACT.  Operands beyond +-65504 would overflow: the score
                                        networks' activations are O(1..100) */
    IRSDE_FLAG_NO_WINOGRAD_FUSED = 2048, /* keep every Winograd layer on the three-launch path (input transform, component GEMMs, output
                                        transform); default: the big feature maps (>= 4096 tiles, Cin <= 256, Cout <= 256) run the
                                        fused kernel of csrc/wino_fused.hip, whose transformed tensors never reach HBM */
    IRSDE_FLAG_NO_FUSED_ATTN = 4096, /* LinearAttention with the whole to_qkv convolution and a q | k | v tensor in HBM (default for fp32, C <= 256:
                                        k / v projection + softmax over the pixels + context in one kernel, only q is a convolution) */
    IRSDE_FLAG_SPLIT_BF16X2 = 16384, /* r03, opt-in, new behaviour (the reference is plain fp32): split-operand arithmetic on the 16-bit MFMA pipe with
                                        fp32 storage everywhere.  Every f32 GEMM operand is split into a 16-bit pair hi + lo (hi = round(x), lo =
                                        round(x - hi): the residual is exact) and the three cross products hi.hi + hi.lo + lo.hi are accumulated in
                                        f32, at 3/16 of the f32-MFMA cycles.  Covered: the component GEMMs of the three-launch Winograd layers with
                                        >= 256 input channels (csrc/gemm_split.hip: V written as pairs, U split at weight load) and the direct
                                        implicit-GEMM layers with K = KH KW Cin >= 128 and >= 64 output channels (csrc/conv_igemm.hip PAIR kernels:
                                        activations split while staged) incl. every NAFNet 1x1 GEMM and its epilogues.  Transforms, the fused
                                        Winograd kernel, attention, LayerNorm, epilogues and the sampler state stay native f32.  With bf16 pieces
                                        (this flag) an operand carries 16 significand bits and f32's exponent range.  Error / speed tables:
                                        profiles/r03_split_*.txt, r03_pair_conv_sweep.txt.  Not combinable with the 16-bit modes. */
    IRSDE_FLAG_SPLIT_F16X2 = 32768,  /* r03, opt-in: the same paths with IEEE fp16 pieces (hi + lo = 22+ of f32's 24 significand bits: fp32-equivalent
                                        per product and per layer — measured 0.5 .. 1.0 x the native kernels' error against float64).  fp16's
                                        exponent range is handled by exact power-of-two scales: Winograd V is written as V / 16 (|V| <= 1e6 stays
                                        finite), weights are scaled per tensor to max |w| in (256, 512], the kernels undo both on the accumulators;
                                        activations of the direct layers are used unscaled (|x| < 65504; larger values become inf).  This flag
                                        also moves the k / v, q and to_out projections of the fused LinearAttention kernels onto fp16 pairs, and the
                                        fused Winograd kernel of the big feature maps onto its fp16-pair twin (all four hi / lo cross products).
                                        Small end: an fp16 pair resolves 2^-25 ABSOLUTE (after the scale), so the 22+ bits hold for operands of
                                        magnitude >~ 0.1 — the network's O(1) activations; a layer fed |x| ~ 0.02 is at 3e-5 instead of 5e-6
                                        (tests/test_gpu_split.py), |x| ~ 1e-3 at 4e-4: use IRSDE_FLAG_SPLIT_BF16X2 (f32 exponent range) or the
                                        native mode for data scaled that far from unit range.  Wins over IRSDE_FLAG_SPLIT_BF16X2 when both are set. */
    IRSDE_FLAG_NO_NAF_CHAIN = 8192,  /* r04: keep every NAFBlock on the per-layer path.  Default with IRSDE_FLAG_FP16: a run of consecutive 512-channel
                                        NAFBlocks on an 8 x 8 feature map (the 28-block level of BASELINE configs[4]'s 64 x 64 latents) runs as ONE
                                        launch, one work-group per image, activations in registers + LDS, fp16 weights streamed from L2
                                        (csrc/naf_chain.hip).  Same operand mode, and these additional roundings to fp16 (all at operand level;
                                        chain vs per-layer path 2e-4, chain vs oracle.naf_block per block <= 1e-3, tests/test_gpu_parity.py): the conv1
                                        output before the depthwise conv, the depthwise taps, the pooled SCA mean, the SCA 1x1 conv's operands and
                                        its output (the scale vector), and the product x * sca(x) — formed in fp16 from the fp16 gated tensor on its
                                        way into conv3, i.e. that operand is rounded twice */
    IRSDE_FLAG_NO_WINOGRAD_F43 = 8   /* Winograd F(2x2,3x3) only (>= 256 channels); default also uses F(4x4,3x3) from 128
                                        channels up where H, W are multiples of 4 */
};
/* per-call flags (irsde_sample) */
enum {
    IRSDE_SAMPLE_GRAPH = 1,  /* replay one captured hipGraph per step instead of eager launches */
    IRSDE_SAMPLE_PROFILE = 2 /* eager + hipEvent pairs around every kernel (see irsde_get_profile) */
};

/* ConditionalUNet(in_nc, out_nc, nf, depth) — DenoisingUNet_arch.py:19-20 (upscale is unused there). */
typedef struct irsde_config {
    int32_t in_nc;
    int32_t out_nc;
    int32_t nf;
    int32_t depth;
    int32_t device; /* HIP device ordinal */
    int32_t flags;  /* IRSDE_FLAG_* */
} irsde_config;

/* ConditionalNAFNet(img_channel, width, middle_blk_num, enc_blk_nums, dec_blk_nums) — the Refusion score network,
 * codes/config/deraining/models/modules/DenoisingNAFNet_arch.py:85-147 (refusion.yml: width 64, enc [1,1,1,28],
 * middle 1, dec [1,1,1,1]).  Everything after creation (weights, schedule, forward, sample) uses the same entry points. */
typedef struct irsde_nafnet_config {
    int32_t img_channel;
    int32_t width;
    int32_t middle_blk_num;
    int32_t n_enc;
    int32_t enc_blk_nums[8];
    int32_t n_dec;
    int32_t dec_blk_nums[8];
    int32_t device;
    int32_t flags;
} irsde_nafnet_config;

/* UNet(in_ch, out_ch, ch, ch_mult, embed_dim) — the frozen latent compressor of the latent tasks,
 * codes/config/latent-dehazing/models/modules/UNet_arch.py:17-57 (nasde.yml: ch 8, ch_mult [4,8,8,16], embed_dim 8). */
typedef struct irsde_latent_unet_config {
    int32_t in_ch;
    int32_t out_ch;
    int32_t ch;
    int32_t n_mult;
    int32_t ch_mult[8];
    int32_t embed_dim;
    int32_t device;
    int32_t flags;
} irsde_latent_unet_config;

/* Row layout of the per-step coefficient table passed to irsde_set_schedule (floats per row). */
#define IRSDE_COEF_STRIDE 12
/*  [0] thetas[t]  [1] sigmas[t]  [2] sigma_bars[t]  [3] dt  [4] sqrt(dt)
 *  [5] exp(thetas_cumsum[t]*dt)                      (get_init_state_from_noise, sde_utils.py:237-239)
 *  [6] posterior term1  [7] posterior term2          (reverse_optimum_step,      sde_utils.py:197-205)
 *  [8] posterior std                                 (reverse_optimum_std,       sde_utils.py:207-217)
 *  [9] exp(-2*thetas_cumsum[t]*dt)  (DenoisingSDE drifts, sde_utils.py:448-454)   [10..11] reserved (0)   */

const char* irsde_last_error(void);
int irsde_version(void);

/* Replaces ConditionalUNet.__init__ (DenoisingUNet_arch.py:19-76).  Host-only; no GPU work. */
int irsde_create(const irsde_config* cfg, irsde_engine** out);
/* Replaces ConditionalNAFNet.__init__ (DenoisingNAFNet_arch.py:87-147).  Host-only. */
int irsde_create_nafnet(const irsde_nafnet_config* cfg, irsde_engine** out);
void irsde_destroy(irsde_engine* e);

/* Weight inventory = the reference state_dict (151 tensors for nf=64, depth=4). */
int irsde_num_weights(const irsde_engine* e);
const char* irsde_weight_name(const irsde_engine* e, int index);
int irsde_weight_shape(const irsde_engine* e, int index, int64_t shape[4], int* ndim);

/* Replaces load_state_dict / BaseModel.load_network (deraining/models/base_model.py:92-105):
 * `data` is a HOST pointer to the tensor in the reference's own layout (OIHW conv weights,
 * [out][in] linear weights, (1,C,1,1) LayerNorm gains).  Repacked to the kernel layout
 * ([Cout][KH][KW][Cin]) and uploaded by irsde_finalize_weights. */
int irsde_load_weight(irsde_engine* e, const char* name, const float* data, const int64_t* shape, int ndim);
int irsde_finalize_weights(irsde_engine* e);

/* Replaces IRSDE._initialize's device tables (sde_utils.py:145-148).  `coef` is a HOST table of
 * (T+1) rows x IRSDE_COEF_STRIDE floats (row 0 unused).  Also builds the [T+1] x sum(2*Cout) FiLM
 * table: time_mlp (DenoisingUNet_arch.py:42-47) and every ResBlock.mlp (module_util.py:127-141)
 * evaluated for t = 0..T.  Needs finalized weights.  Synchronises the device. */
int irsde_set_schedule(irsde_engine* e, int T, const float* coef);

/* Replaces ConditionalUNet.forward(xt, cond, time) (DenoisingUNet_arch.py:85-134).
 * xt, cond, out: device NCHW [B][in_nc|out_nc][H][W].  t_host: HOST array of nt timesteps,
 * nt == 1 (python int / tensor([t]): shared by the batch) or nt == B (training-style [B] tensor,
 * denoising_model.py:135).  Any H, W >= 2 (reflect pad to a multiple of 2^depth, :78-83). */
int irsde_unet_forward(irsde_engine* e, const float* xt, const float* cond, const int64_t* t_host, int nt, int B,
                       int H, int W, float* out, void* stream);

/* Replaces IRSDE.reverse_sde / reverse_ode / reverse_posterior (sde_utils.py:252-299) with the score
 * network evaluated by this engine: for t = T..1: eps = net(x, mu, t); x = step(x, mu, eps, z_t).
 *   xT, mu, out : device NCHW [B][C][H][W]; xT is not modified (the reference clones it, :254).
 *   noise       : device [T_sched+1][B][C][H][W] injected N(0,1) draws indexed by t (parity runs), or
 *                 NULL => Philox4x32-10 keyed by (seed, image_offset + b, t, element).
 *   T           : first step (<= the T given to irsde_set_schedule; < 0 means that T; 0 runs no step and copies
 *                 xT to out, as the reference's empty range(1, 1) does), like the
 *                 reference's `T` argument; t_stop: last step NOT taken (0 = run down to t = 1); running
 *                 T..t_stop+1 and then t_stop..1 equals one T..1 call (used for save_states dumps).
 *   flags       : IRSDE_SAMPLE_* */
int irsde_sample(irsde_engine* e, int mode, const float* xT, const float* mu, const float* noise, uint64_t seed,
                 uint64_t image_offset, int B, int H, int W, int T, int t_stop, float* out, void* stream,
                 int flags);

/* One reverse step with an externally computed eps_hat (NCHW, e.g. from a foreign nn.Module):
 * replaces SDE.reverse_sde_step / reverse_ode_step / IRSDE.reverse_posterior_step
 * (sde_utils.py:44-48, 219-223).  x is updated in place.  `coef_row` is a HOST row (IRSDE_COEF_STRIDE
 * floats).  noise_t: device [B][C][H][W] or NULL (Philox).  No engine state is needed (e may be NULL). */
int irsde_sde_step(int mode, int t, const float* coef_row, float* x, const float* mu, const float* eps_hat,
                   const float* noise_t, uint64_t seed, uint64_t image_offset, int B, int C, int H, int W,
                   void* stream);

/* The device RNG on its own: out[B][CHW] = the N(0,1) draws irsde_sample would use at step t. */
int irsde_philox_normal(float* out, int B, int CHW, int t, uint64_t seed, uint64_t image_offset, void* stream);

/* Timing of the last IRSDE_SAMPLE_PROFILE call (hipEvents on the engine's own stream).
 * out[0] conv kernel ms (sum)      out[1] conv algorithmic FLOPs (sum)  out[2] conv launches
 * out[3] conv algorithmic bytes    out[4] layernorm ms                   out[5] attention ms
 * out[6] other kernels ms          out[7] wall ms of the whole call      out[8] network evaluations
 * out[9] Winograd transform kernels ms   out[10] FLOPs actually issued by the conv kernels (Winograd layers
 * issue 2.25x fewer than the algorithmic direct-convolution count in out[1])   out[11] reserved */
int irsde_get_profile(const irsde_engine* e, double out[12]);

/* Debug (IRSDE_FLAG_KEEP_ACTIVATIONS): copy a named intermediate activation of the last forward to
 * HOST memory as NCHW.  Names follow the reference module paths ("downs.0.0", "mid_attn", ...).
 * dims receives {B, C, H, W}; pass dst == NULL to query dims only. */
int irsde_debug_tap(irsde_engine* e, const char* name, float* dst, int64_t dims[4]);

/* Algorithmic work of one network evaluation at (B,H,W): flops[0] = conv FLOPs, flops[1] = ideal-fusion
 * conv bytes (inputs + outputs + weights once). */
int irsde_work_model(irsde_engine* e, int B, int H, int W, double out[2]);

/* Per-launch-group average ms of the last IRSDE_SAMPLE_PROFILE call, one text line per group. */
int irsde_op_profile(irsde_engine* e, char* buf, int buflen);

/* Text description (one line per launch group) of the network-evaluation plan at (B,H,W): debugging/analysis. */
int irsde_plan_describe(irsde_engine* e, int B, int H, int W, char* buf, int buflen);

/* Latent wrapper.  irsde_create_latent_unet replaces UNet.__init__ (UNet_arch.py:18-50); weights load
 * through irsde_load_weight / irsde_finalize_weights with the reference's state_dict names (69 tensors for nasde.yml).
 * irsde_latent_shapes: latent (C,h,w) and the 2*depth+1 hidden skips (C,h,w each, reference list order) for an H x W input.
 * irsde_latent_encode replaces UNet.encode(x) (:59-77): x device NCHW [B][in_ch][H][W] -> latent [B][embed][h][w] and the
 * hidden list (device NCHW tensors, caller-allocated).  irsde_latent_decode replaces UNet.decode(x, h) (:79-91) and crops
 * to H x W.  Used as latent_denoising_model.py:50-51,177-189: encode once, sample in the latent, decode once. */
int irsde_create_latent_unet(const irsde_latent_unet_config* cfg, irsde_engine** out);
int irsde_latent_shapes(irsde_engine* e, int H, int W, int64_t latent_chw[3], int64_t* hidden_chw, int* n_hidden);
int irsde_latent_encode(irsde_engine* e, const float* x, int B, int H, int W, float* latent, float* const* hidden,
                        void* stream);
int irsde_latent_decode(irsde_engine* e, const float* latent, const float* const* hidden, int B, int H, int W, float* out,
                        void* stream);
/* ABI 104: `hidden` may be NULL in both calls.  encode(hidden = NULL) leaves the skips in the engine's working layout; decode(hidden = NULL)
 * reads them there — valid until the next irsde_latent_encode / irsde_latent_decode(hidden != NULL) of the same B x H x W on this engine (error
 * otherwise).  The reference only threads `hidden` through (latent-dehazing/test.py:90-95, latent_denoising_model.py:177-189).
 * irsde_latent_hidden writes resident skip k (reference list order) as a device NCHW tensor [B][C_k][h_k][w_k]. */
int irsde_latent_hidden(irsde_engine* e, int B, int H, int W, int k, float* out, void* stream);

/* latent-bokeh only (IRSDE_FLAG_NAF_LENS): replaces the `lens_info` kwargs of ConditionalNAFNet.forward(inp, cond, time,
 * lens_info=[src_lens, tgt_lens, disparity]) that latent_denoising_model.py:183-189 threads through sde.reverse_sde(**kwargs).
 * info: HOST [B][3] = (src_lens, tgt_lens, disparity) of every image; evaluates cam_mlp and every block's cam_mlp once; the
 * rows stay valid for all following forward / sample calls with batch <= B. */
int irsde_set_lens_info(irsde_engine* e, const float* info, int B);

/* Evaluation tail on the device — replaces, per image of a batch, the metric block of
 * codes/config/deraining/test.py:131-178: util.tensor2img on output and GT (codes/utils/img_utils.py:136-163),
 * util.calculate_psnr / calculate_ssim (:182-234) on the uint8 images cropped by crop_border, and the same two on
 * the Y channel of bgr2ycbcr (codes/data/util.py:177-198).  out, gt: device NCHW fp32 [B][C][H][W], C = 1 or 3 (RGB).
 * metrics: HOST [B][4] = {psnr, ssim, psnr_y, ssim_y} (the Y entries are NaN for C == 1).  Synchronises `stream`. */
int irsde_eval_metrics(const float* out, const float* gt, int B, int C, int H, int W, int crop_border, double* metrics,
                       void* stream);

/* util.tensor2img (img_utils.py:136-163) for a batch: device NCHW fp32 -> device [B][H][W][C] uint8 with the channel
 * order reversed (RGB -> BGR, what cv2.imwrite expects): clamp to [0,1], x255, round half to even. */
int irsde_tensor2img(const float* in, unsigned char* out, int B, int C, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IRSDE_HIP_H */
