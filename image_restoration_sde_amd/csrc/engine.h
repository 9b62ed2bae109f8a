// libirsde_hip.so — engine internals shared by engine_weights.hip / engine_plan.hip / engine_api.hip.
//
// Host-side structure (all C++; PyTorch never appears here):
//   Engine      weights in kernel layout, FiLM/time table, coefficient table, plans
//   Plan        per (B,H,W): static activation arena + the launch list of ONE network evaluation
//               (ConditionalUNet.forward, DenoisingUNet_arch.py:85-134) built once, replayed T times
//   sample()    the reverse loop (sde_utils.py:252-299): [step_begin, prep, net, update] per t, either
//               eager or as one captured hipGraph replayed T times; the step index lives in device
//               memory (StepState) so the graph is t-invariant.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/irsde_hip.h"
#include "common.h"

namespace irsde {

// Winograd only where the transforms' extra HBM traffic is small next to the GEMM: F(2x2) moves 4x the input and
// 4x the output through HBM and pays from 256 channels, F(4x4) 2.25x and pays from 128 — from 64 (+1.6 %) since the
// component GEMMs run on the batch-loop kernel (measured, profiles/).
// IRSDE_WINO2_MINC / IRSDE_WINO4_MINC override the thresholds (tuning experiments, only with IRSDE_TUNING=1).
// The fused Winograd kernel (wino_fused.hip) streams a layer's whole U slice per block: it pays where the feature map is
// large (many tiles) and the channel counts are moderate; beyond these limits the three-launch path keeps the layer.
constexpr int kWinoFusedMaxCin = 256, kWinoFusedMaxCout = 256;  // measured crossover (profiles/r02_wino_fused_sweep.txt): Cin 384+ is faster on the three-launch path
inline long long wino_fused_min_tiles() { return tuning_env_int("IRSDE_WINO_FUSED_MINT", 4096); }
// r03: the 64-cout fused kernel (16 tiles x 64 couts per block) takes the layers whose channel counts are multiples of 64;
// IRSDE_WINO_FUSED64=0 keeps them on the 32-cout kernel, IRSDE_WINO_FUSED64_MAXCIN / _MAXCOUT move its crossover (tuning only)
inline bool wino_fused64_enabled() { return tuning_env_int("IRSDE_WINO_FUSED64", 1) != 0; }
inline bool wino_fused64_pair_enabled() { return tuning_env_int("IRSDE_WINO_FUSED64_PAIR", 1) != 0; }   // fp32_split_f16: the kernel's fp16-pair twin
inline int wino_fused64_max_cin() { return tuning_env_int("IRSDE_WINO_FUSED64_MAXCIN", 512); }
inline int wino_fused64_max_cout() { return tuning_env_int("IRSDE_WINO_FUSED64_MAXCOUT", 512); }
inline long long wino_fused64_min_tiles() { return tuning_env_int("IRSDE_WINO_FUSED64_MINT", 1024); }
// r06: the two-tile-group kernel (wino_fused_t.hip) on the layers it measured faster on: 0 never, 1 by the rule of Plan::push_wino_fused, 2 wherever eligible
inline int wino_fused64t_mode() { return tuning_env_int("IRSDE_WINO_FUSED64T", 1); }
// r06: work-groups per image of the NAFBlock chain kernel: 0 = as many (4, 2) as fit the compute units next to the call's other sub-batches, 1 = the one-group kernel, 2 / 4 forced (if they fit)
// r06: NAFBlock norm + 1x1 convolution as one launch on the small-channel levels of the fp16 operand mode (1; 0 = LayerNorm kernel + convolution kernel)
inline bool naf_lnconv_enabled() { return tuning_env_int("IRSDE_NAF_LNCONV", 1) != 0; }
inline int naf_chain_split_mode() { return tuning_env_int("IRSDE_NAF_CHAIN_SPLIT", 0); }

inline int wino_min_c(int tile) {
    return tuning_env_int(tile == 4 ? "IRSDE_WINO4_MINC" : "IRSDE_WINO2_MINC", tile == 4 ? 64 : 256);
}

// Selects the engine's device for the scope of one C-ABI call and restores the caller's current device afterwards
// (a multi-GPU host process — and PyTorch inside it — must not see its current device change under its feet).
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) IRSDE_HIP_CHECK(hipSetDevice(dev));
        else prev = -1;
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    bool loaded = false;
};

struct ConvW {
    float* w = nullptr;  // device [Cout][KH*KW][Cin]
    float* bias = nullptr;
    int Cout = 0, Cin = 0, KH = 1, KW = 1;
    float* wino_u2 = nullptr;  // device [16][Cout][Cin] = G g G^T of F(2x2,3x3)  (3x3 layers with Cin,Cout >= 256)
    float* wino_u4 = nullptr;  // device [36][Cout][Cin]              F(4x4,3x3)  (3x3 layers with Cin,Cout >= 128)
    float* wino_uf = nullptr;  // the same F(4x4,3x3) weights in the fused kernel's fragment order (wino_fused.hip)
    float* wino_uf64 = nullptr;  // ... in the 64-cout fused kernel's fragment order (Cout, Cin multiples of 64)
    unsigned short* wino_uf64p = nullptr;   // IRSDE_FLAG_SPLIT_F16X2: wino_uf64 as fp16 hi / lo halves (wf64_split_weights_kernel), scaled by wino_uf64p_scale
    float wino_uf64p_scale = 1.f;
    unsigned short* wino_up = nullptr;  // IRSDE_FLAG_SPLIT_BF16X2 / _F16X2: the F(4x4,3x3) weights as hi / lo pairs, [36][Cout][Cin / 32][2][32]
    float wino_up_scale = 1.f;          // fp16 pairs: the power of two U was multiplied by (max |U| * scale <= 512)
};
struct ResW {
    ConvW b1, b2, res;
    bool has_res = false;
    float* mlp_w = nullptr;  // [2*Cout][time_dim]
    float* mlp_b = nullptr;
    int film_off = 0;
    int Cout = 0;
};
struct AttnW {
    float* g1 = nullptr;
    ConvW qkv, out;
    float* g2 = nullptr;
    int C = 0;
};

// NAFBlock weights (DenoisingNAFNet_arch.py:15-49) in kernel layout
struct NafBlockW {
    int c = 0;
    float *g1 = nullptr, *g2 = nullptr;          // norm1.g / norm2.g
    ConvW conv1, conv3, conv4, conv5;            // 1x1: c->2c, c->c, c->2c (rows interleaved for the gate), c->c
    float *dw_w = nullptr, *dw_b = nullptr;      // conv2 depthwise 3x3: [9][2c], [2c]
    float *sca_w = nullptr, *sca_b = nullptr;    // sca.1: [c][c], [c]
    float *beta = nullptr, *gamma = nullptr;
    float *mlp_w = nullptr, *mlp_b = nullptr;    // mlp.1: Linear(time_dim/2, 4c)
    int film_off = 0;                            // [shift_att | scale_att | shift_ffn | scale_ffn]
    float *cam_w = nullptr, *cam_b = nullptr;    // latent-bokeh: cam_mlp.1: Linear(time_dim/2, 2c)
    int cam_off = 0;                             // [cam_scale | cam_shift] inside a row of the lens table
};

// a run of consecutive 512-channel NAFBlocks packed for naf_chain_kernel (fp16 mode): fragment streams + fp32 vectors
struct NafChainW {
    unsigned short* w = nullptr;
    float* vecs = nullptr;
    int nblocks = 0, film_off = 0, cam_off = 0;
    unsigned short* wsplit[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // [G]: the fragment streams of the G-groups-per-image kernel, built on first use
};

enum OpKind { OP_CONV = 0, OP_LN = 1, OP_ATTN = 2, OP_OTHER = 3, OP_WINO = 4, OP_NKINDS = 5 };

struct Op {
    std::function<void(hipStream_t)> fn;
    OpKind kind;
    double flops = 0, bytes = 0;  // algorithmic (direct-convolution) work attributed to this launch group
    double exec_flops = 0;        // multiply-adds actually issued to the MFMA pipe (differs for Winograd)
    std::string desc;
};

// Winograd F(2x2,3x3) launch triple derived from the direct-form parameters of a 3x3 stride-1 pad-1 convolution
struct WinoPlan {
    WinoParams in, out;
    ConvParams gemm;
};
inline WinoPlan make_wino(const ConvParams& d, const float* U, float* V, float* Mb, int tile) {
    WinoPlan w;
    const int Ctot = d.C0 + d.C1;
    const int TH = d.Ho / tile, TW = d.Wo / tile, T = d.B * TH * TW;
    const int ncomp = (tile + 2) * (tile + 2);
    w.in.tile = tile;
    w.in.in0 = d.in0; w.in.in1 = d.in1; w.in.C0 = d.C0; w.in.C1 = d.C1; w.in.Hin = d.Hin; w.in.Win = d.Win;
    w.in.in_shift = d.in_shift; w.in.B = d.B; w.in.TH = TH; w.in.TW = TW; w.in.T = T; w.in.V = V;
    w.out = w.in;
    w.out.M = Mb; w.out.Cout = d.Cout; w.out.out = d.out; w.out.out_stride = d.out_stride; w.out.bias = d.bias;
    w.out.film = d.film; w.out.film_bstride = d.film_bstride; w.out.silu = d.silu; w.out.res = d.res;
    w.out.res_stride = d.res_stride;
    ConvParams& g = w.gemm;
    g.in0 = V; g.C0 = Ctot; g.pix0 = Ctot; g.Hin = 1; g.Win = T; g.w = U; g.Cout = d.Cout;
    g.KH = g.KW = 1; g.stride = 1; g.pad_y = g.pad_x = 0; g.B = 1; g.Ho = 1; g.Wo = T;
    g.out = Mb; g.out_stride = d.Cout; g.zeros = d.zeros;
    g.nz = ncomp; g.z_in = (long long)T * Ctot; g.z_w = (long long)d.Cout * Ctot; g.z_out = (long long)T * d.Cout;
    return w;
}
// Split-operand variant (gemm_split.hip): V as `nplanes` bf16 planes (Vs: nplanes * 36 * T * Ctot elements), Us the weights'
// planes (nplanes * 36 * Cout * Ctot), the component GEMMs on the bf16 MFMA pipe, M and the output transform unchanged (f32)
struct WinoSplitPlan {
    WinoParams in, out;
    SplitGemmArgs gemm;
    int nplanes = 0;
    bool f16 = false;
};
inline WinoSplitPlan make_wino_split(const ConvParams& d, const unsigned short* Us, unsigned short* Vs, float* Mb, int nplanes) {
    const WinoPlan w = make_wino(d, nullptr, nullptr, Mb, 4);
    WinoSplitPlan sp;
    sp.in = w.in; sp.out = w.out; sp.nplanes = nplanes;
    const int Ctot = d.C0 + d.C1;
    const long long T = w.in.T;
    sp.in.Vs = Vs; sp.in.nplanes = nplanes; sp.in.v_plane = 36ll * T * Ctot;
    SplitGemmArgs& g = sp.gemm;
    g.a = Vs; g.b = Us; g.out = Mb;
    g.plA = 36ll * T * Ctot; g.plB = 36ll * d.Cout * Ctot;
    g.pA = T * Ctot; g.pB = (long long)d.Cout * Ctot; g.pO = T * d.Cout;
    g.M = (int)T; g.N = d.Cout; g.K = Ctot; g.lda = Ctot; g.ldc = d.Cout;
    g.n_inner = gemm_split_inner(g.M, g.N, 36);
    return sp;
}
// IRSDE_FLAG_SPLIT_BF16X2: pair-interleaved operands for launch_gemm_split_pairs (Vs: 36 * T * Ctot * 2 elements)
constexpr float kSplitF16VScale = 1.0f / 16.0f;   // fp16 pairs: V is written as V / 16 (finite up to |V| = 1e6)
inline WinoSplitPlan make_wino_pairs(const ConvParams& d, const unsigned short* Up, unsigned short* Vs, float* Mb, bool f16 = false,
                                     float u_scale = 1.f) {
    const WinoPlan w = make_wino(d, nullptr, nullptr, Mb, 4);
    WinoSplitPlan sp;
    sp.in = w.in; sp.out = w.out; sp.nplanes = 2;
    const int Ctot = d.C0 + d.C1;
    const long long T = w.in.T;
    sp.in.Vs = Vs; sp.in.nplanes = 2; sp.in.v_pairs = 1;
    sp.in.v_f16 = f16 ? 1 : 0; sp.in.v_scale = f16 ? kSplitF16VScale : 1.f;
    SplitGemmArgs& g = sp.gemm;
    g.out_scale = f16 ? 1.0f / (kSplitF16VScale * u_scale) : 1.f;
    sp.f16 = f16;
    g.a = Vs; g.b = Up; g.out = Mb;
    g.pA = T * Ctot; g.pB = (long long)d.Cout * Ctot; g.pO = T * d.Cout;
    g.M = (int)T; g.N = d.Cout; g.K = Ctot; g.lda = Ctot; g.ldc = d.Cout;
    return sp;
}
// split mode: three-launch Winograd layers with at least this many input channels run the pair GEMM (below it the fused
// f32 kernel is faster: profiles/r03_split_gemm_bench.txt); IRSDE_SPLIT_MINC moves the crossover (tuning only)
inline int split_min_cin() { return tuning_env_int("IRSDE_SPLIT_MINC", 256); }
// split mode: direct (implicit-GEMM) layers with K = KH * KW * Cin of at least this and >= 64 output channels run the PAIR kernel
// (conv_igemm.hip); below it the layer is HBM-bound and the native f32 kernel is as fast.  IRSDE_SPLIT_DIRECT_MINK (tuning only)
inline int split_direct_min_k() { return tuning_env_int("IRSDE_SPLIT_DIRECT_MINK", 128); }   // measured crossover: profiles/r03_pair_conv_sweep.txt
inline bool wino_shape_ok(const ConvParams& d, int tile) {
    return d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad_y == 1 && d.pad_x == 1 && d.Ho % tile == 0 && d.Wo % tile == 0 &&
           (d.C0 + d.C1) % 32 == 0 && d.Cout % 4 == 0 && d.out_stride % 4 == 0 && (!d.res || d.res_stride % 4 == 0);
}

struct Tensor {
    float* p = nullptr;  // bf16 == true: really a bf16 tensor (IRSDE_FLAG_BF16_ACT)
    int B = 0, H = 0, W = 0, C = 0;
    bool bf16 = false;
    size_t numel() const { return (size_t)B * H * W * C; }
};

struct PoolBlock {
    float* p;
    size_t n;
    bool free;
};

struct Plan {
    int B = 0, H = 0, W = 0, Hp = 0, Wp = 0;
    bool per_sample_film = false;
    StepState* own_step = nullptr;   // sub-batch plans (slot >= 1) carry their own step counter / coefficient row and FiLM row: their step graphs replay
    float* own_film = nullptr;       // on their own stream with no per-step dependency on the other parts (both live in the plan's arena)
    int slot = 0, b0 = 0;     // r05 sub-batch plans: slot >= 1 = part `slot - 1` of a split batch, holding images [b0, b0 + B) of the call (own arena, own state buffers;
                              // per-image tables — lens FiLM rows, per-sample time rows, noise / Philox index — are addressed from b0)
    std::vector<PoolBlock> pool;
    std::vector<Op> net_ops;  // prep + network (one evaluation)
    float* xin = nullptr;     // [B][in_nc][H][W]  state x / xt
    float* cin = nullptr;     // [B][in_nc][H][W]  mu / cond
    float* x0 = nullptr;      // prepped NHWC input
    float* pred = nullptr;    // [B][Hp][Wp][pred_stride]
    int pred_stride = 4;      // roundup(out_nc, 4)
    std::map<std::string, Tensor> taps;
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    double conv_flops = 0, conv_bytes = 0, conv_exec_flops = 0;
    uint64_t last_use = 0;
    std::vector<std::pair<void*, int>> chain_scratch;   // (scratch, images) of the split chain launches, for the reset after an error
    std::vector<const unsigned*> chain_err;   // error flags of the split chain launches (naf_chain.hip): non-zero = the groups of an image were not co-resident

    ~Plan() {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (graph) (void)hipGraphDestroy(graph);
        for (auto& b : pool) (void)hipFree(b.p);
    }
    float* alloc(size_t n, bool reuse) {
        if (reuse) {
            int best = -1;
            for (int i = 0; i < (int)pool.size(); ++i)
                if (pool[i].free && pool[i].n >= n && (best < 0 || pool[i].n < pool[best].n)) best = i;
            if (best >= 0 && pool[best].n <= n + n / 2 + 1024) {
                pool[best].free = false;
                return pool[best].p;
            }
        }
        float* p = nullptr;
        const hipError_t err = hipMalloc(&p, std::max<size_t>(n, 64) * sizeof(float));
        if (err == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            throw HipOutOfMemory("plan arena: hipMalloc of " + std::to_string(std::max<size_t>(n, 64) * sizeof(float)) + " bytes failed (out of memory)");
        }
        IRSDE_HIP_CHECK(err);
        pool.push_back({p, n, false});
        return p;
    }
    void release(float* p) {
        for (auto& b : pool)
            if (b.p == p) {
                b.free = true;
                return;
            }
    }
};

// One direction (encode or decode) of the latent UNet at a fixed (B,H,W): a Plan whose ops run on NHWC buffers, plus
// the NHWC tensors that are read from / written to the caller's NCHW tensors around it.
struct LatentPlan {
    bool decode = false;
    std::unique_ptr<Plan> plan;
    Tensor image;                // encode: padded NHWC input image; decode: final_conv output [B][Hp][Wp][4]
    Tensor latent;               // NHWC latent (channels padded to 32)
    std::vector<Tensor> hidden;  // NHWC skips in the reference's list order h[0..2*depth]
    std::vector<int> hidden_c;   // logical channel counts
    bool resident = false;       // encode plan: the skips of the last encode are in place (decode plan: it SHARES the encode plan's skip tensors)
};

}  // namespace irsde

using namespace irsde;  // internal header: included only by the three engine_*.hip translation units

struct irsde_engine {
    irsde_config cfg{};
    int time_dim = 0;
    std::vector<std::string> names;  // weight inventory, reference state_dict order-independent
    std::map<std::string, HostTensor> host;
    bool finalized = false;
    std::vector<float*> dev_allocs;

    // packed weights
    ConvW init_conv, final_conv;
    float *tm_w1 = nullptr, *tm_b1 = nullptr, *tm_w3 = nullptr, *tm_b3 = nullptr, *freqs = nullptr;
    std::vector<ResW> down_res;   // 2 per level
    std::vector<AttnW> down_attn;
    std::vector<ConvW> down_conv;
    ResW mid1, mid2;
    AttnW mid_attn;
    std::vector<ResW> up_res;
    std::vector<AttnW> up_attn;
    std::vector<ConvW> up_conv;
    ResW final_res;
    std::vector<ResW*> all_res;
    int film_row = 0;

    // ConditionalNAFNet (arch == 1)
    int arch = 0;
    std::vector<int> naf_enc_nums, naf_dec_nums;
    int naf_mid_num = 0;
    std::vector<std::vector<NafBlockW>> naf_enc, naf_dec;
    std::vector<NafBlockW> naf_mid;
    std::vector<ConvW> naf_downs, naf_ups;
    ConvW naf_intro, naf_ending;
    std::vector<NafBlockW*> naf_all;
    std::vector<NafChainW> naf_chain_enc, naf_chain_dec;   // per level (nblocks == 0: none)
    NafChainW naf_chain_mid;
    // latent-bokeh variant (IRSDE_FLAG_NAF_LENS): lens-information FiLM, one row per image of the batch
    float *cm_w1 = nullptr, *cm_b1 = nullptr, *cm_w3 = nullptr, *cm_b3 = nullptr;  // cam_mlp.0 / cam_mlp.2
    int cam_row = 0;            // sum over blocks of 2c
    float* cam_cur = nullptr;   // [cam_rows][cam_row]
    int cam_rows = 0;           // capacity (images)
    int cam_set = 0;            // images covered by the last irsde_set_lens_info

    // latent UNet (arch == 2): codes/config/latent-dehazing/models/modules/UNet_arch.py
    int lat_in = 0, lat_out = 0, lat_ch = 0, lat_embed = 0;
    std::vector<int> lat_mult;
    ConvW lat_init, lat_latent, lat_post, lat_final;
    std::vector<ResW> lat_enc_res, lat_dec_res;  // 2 per level (decoder in module order: deepest first)
    AttnW lat_enc_attn, lat_dec_attn;            // deepest level only
    std::vector<ConvW> lat_down, lat_up;
    std::vector<std::unique_ptr<LatentPlan>> lat_plans;

    // schedule / FiLM tables
    int T = 0;
    float* coef_table = nullptr;  // [(T+1)][12]
    float* film_table = nullptr;  // [(T+1)][film_row]
    float* film_cur = nullptr;    // [max_rows][film_row]
    int film_cur_rows = 0;
    StepState* step = nullptr;
    SampleCtl* ctl = nullptr;

    float* zeros = nullptr;        // zero page: the branch-free source of out-of-image conv taps
    hipStream_t stream = nullptr;  // engine stream (graph capture needs a non-default stream)
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    // r05: concurrent sub-batches of a NAFNet sampler step (engine_api.hip: sample_split): branch i > 0 runs on sub_stream[i - 1] between ev_fork and ev_join[i - 1]
    static constexpr int kMaxSub = 4;
    hipStream_t sub_stream[kMaxSub - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxSub - 1] = {nullptr, nullptr, nullptr};
    std::vector<std::unique_ptr<Plan>> plans;
    uint64_t use_counter = 0;
    double profile[12] = {0};
    std::vector<double> op_ms;          // per launch group of the last profiled plan (summed over steps)
    std::vector<std::string> op_desc;
    int op_steps = 0;
    int plan_parts = 1;   // concurrent sub-batch plans the sampler call being prepared runs (set by irsde_sample around get_plan: the split chain's CU budget)
    int op_split = 1;   // concurrent sub-batches the sampler would run the profiled batch as (the event-instrumented pass times the un-split plan)
    std::vector<hipEvent_t> ev_pool;
    std::mutex mu;

    float* dmalloc(size_t n) {
        float* p = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&p, std::max<size_t>(n, 16) * sizeof(float)));
        dev_allocs.push_back(p);
        return p;
    }
    // bf16 / fp16 (RNE) copy of a packed fp32 weight tensor, made once per tensor (IRSDE_FLAG_BF16 / IRSDE_FLAG_FP16)
    std::map<const float*, unsigned short*> bf16_copies;
    const unsigned short* bf16_copy(const float* w, size_t n) {
        auto it = bf16_copies.find(w);
        if (it != bf16_copies.end()) return it->second;
        unsigned short* d = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&d, n * sizeof(unsigned short)));
        if (cfg.flags & IRSDE_FLAG_FP16) launch_f32_to_f16(w, d, n, stream);  // IEEE fp16 operands (RNE)
        else launch_f32_to_bf16(w, d, n, stream);
        IRSDE_HIP_CHECK(hipStreamSynchronize(stream));
        bf16_copies[w] = d;
        return d;
    }
    // hi / lo operand planes of a packed fp32 weight tensor for the PAIR convolution kernels (IRSDE_FLAG_SPLIT_BF16X2 / _F16X2), made
    // once per tensor.  fp16 pieces: the tensor is scaled by the power of two that brings max |w| into (256, 512]; `scale` returns
    // its inverse (what the kernel multiplies the accumulators by).
    struct PairCopy { unsigned short* p; float inv_scale; };
    std::map<const float*, PairCopy> pair_copies;
    PairCopy pair_copy(const float* w, size_t n) {
        auto it = pair_copies.find(w);
        if (it != pair_copies.end()) return it->second;
        const bool f16 = (cfg.flags & IRSDE_FLAG_SPLIT_F16X2) != 0;
        float sc = 1.f;
        if (f16) {
            std::vector<float> h(n);
            IRSDE_HIP_CHECK(hipMemcpy(h.data(), w, n * sizeof(float), hipMemcpyDeviceToHost));
            float mx = 0.f;
            for (float v : h) mx = std::max(mx, std::fabs(v));
            if (mx > 0.f) sc = std::exp2(std::floor(std::log2(512.0f / mx)));
        }
        unsigned short* d = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&d, 2 * n * sizeof(unsigned short)));
        dev_allocs.push_back(reinterpret_cast<float*>(d));
        launch_split_planes(w, d, n, n, 2, stream, f16, sc);
        IRSDE_HIP_CHECK(hipStreamSynchronize(stream));
        const PairCopy pc{d, 1.0f / sc};
        pair_copies[w] = pc;
        return pc;
    }
    // The PAIR kernels of conv_igemm.hip read their weights pair-interleaved: [rows][K / 32][hi 32 | lo 32] (split_pairs_kernel), one
    // 128-byte line per 32-k block, fetched global -> LDS by global_load_lds.  K must be a multiple of 32.
    std::map<const float*, PairCopy> pair_copies_il;
    PairCopy pair_copy_il(const float* w, size_t rows, int K) {
        auto it = pair_copies_il.find(w);
        if (it != pair_copies_il.end()) return it->second;
        const bool f16 = (cfg.flags & IRSDE_FLAG_SPLIT_F16X2) != 0;
        const size_t n = rows * (size_t)K;
        float sc = 1.f;
        if (f16) {
            std::vector<float> h(n);
            IRSDE_HIP_CHECK(hipMemcpy(h.data(), w, n * sizeof(float), hipMemcpyDeviceToHost));
            float mx = 0.f;
            for (float v : h) mx = std::max(mx, std::fabs(v));
            if (mx > 0.f) sc = std::exp2(std::floor(std::log2(512.0f / mx)));
        }
        unsigned short* d = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&d, 2 * n * sizeof(unsigned short)));
        dev_allocs.push_back(reinterpret_cast<float*>(d));
        launch_split_pairs(w, d, rows, K, stream, f16, sc);
        IRSDE_HIP_CHECK(hipStreamSynchronize(stream));
        const PairCopy pc{d, 1.0f / sc};
        pair_copies_il[w] = pc;
        return pc;
    }
    float* upload(const std::vector<float>& v) {
        float* p = dmalloc(v.size());
        IRSDE_HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        return p;
    }
};

namespace irsde {

inline bool naf_lens(const irsde_engine* e) { return (e->cfg.flags & IRSDE_FLAG_NAF_LENS) != 0; }
inline int rup32(int c) { return (c + 31) & ~31; }

// engine_weights.hip: weight inventory (reference state_dict names), packing into kernel layouts, FiLM rows
void build_inventory(irsde_engine* e);
void build_inventory_naf(irsde_engine* e);
void build_inventory_latent(irsde_engine* e);
void finalize(irsde_engine* e);
void compute_film_rows(irsde_engine* e, const float* tvals, int rows, float* dst, hipStream_t s);
void ensure_film_cur(irsde_engine* e, int rows);

// engine_plan.hip: one network evaluation as a static launch list over a static arena
Plan* get_plan(irsde_engine* e, int B, int H, int W, bool per_sample_film, int slot = 0, int b0 = 0);
int naf_subbatches(const irsde_engine* e, int B, int H, int W);
void set_force_chain_groups(int g);   // irsde_debug_force_chain_groups
int forced_chain_groups();
void set_force_subbatches(int n);                                 // irsde_debug_force_subbatches   // how many concurrent sub-batches the sampler splits a NAFNet batch into (1 = none)
LatentPlan* get_latent_plan(irsde_engine* e, int B, int H, int W, bool decode);

}  // namespace irsde
