// libirsde_hip.so — engine + C ABI (include/irsde_hip.h).
//
// Host-side structure (all C++; PyTorch never appears here):
//   Engine      weights in kernel layout, FiLM/time table, coefficient table, plans
//   Plan        per (B,H,W): static activation arena + the launch list of ONE network evaluation
//               (ConditionalUNet.forward, DenoisingUNet_arch.py:85-134) built once, replayed T times
//   sample()    the reverse loop (sde_utils.py:252-299): [step_begin, prep, net, update] per t, either
//               eager or as one captured hipGraph replayed T times; the step index lives in device
//               memory (StepState) so the graph is t-invariant.
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

namespace {

thread_local std::string g_last_error;
// Winograd only where the transforms' extra HBM traffic is small next to the GEMM: F(2x2) moves 4x the input and
// 4x the output through HBM and pays from 256 channels, F(4x4) 2.25x and pays from 128 — from 64 (+1.6 %) since the
// component GEMMs run on the batch-loop kernel (measured, profiles/).
// IRSDE_WINO2_MINC / IRSDE_WINO4_MINC override the thresholds (tuning experiments).
static int wino_min_c(int tile) {
    const char* v = getenv(tile == 4 ? "IRSDE_WINO4_MINC" : "IRSDE_WINO2_MINC");
    return v ? atoi(v) : (tile == 4 ? 64 : 256);
}

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
        IRSDE_HIP_CHECK(hipMalloc(&p, std::max<size_t>(n, 64) * sizeof(float)));
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
};

}  // namespace

}  // namespace irsde

using namespace irsde;

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

    float* zeros = nullptr;        // zero page for LDS-DMA staging of out-of-image taps
    hipStream_t stream = nullptr;  // engine stream (graph capture needs a non-default stream)
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    std::vector<std::unique_ptr<Plan>> plans;
    uint64_t use_counter = 0;
    double profile[12] = {0};
    std::vector<double> op_ms;          // per launch group of the last profiled plan (summed over steps)
    std::vector<std::string> op_desc;
    int op_steps = 0;
    std::vector<hipEvent_t> ev_pool;
    std::mutex mu;

    float* dmalloc(size_t n) {
        float* p = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&p, std::max<size_t>(n, 16) * sizeof(float)));
        dev_allocs.push_back(p);
        return p;
    }
    // bf16 (RNE) copy of a packed fp32 weight tensor, made once per tensor (IRSDE_FLAG_BF16)
    std::map<const float*, unsigned short*> bf16_copies;
    const unsigned short* bf16_copy(const float* w, size_t n) {
        auto it = bf16_copies.find(w);
        if (it != bf16_copies.end()) return it->second;
        unsigned short* d = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&d, n * sizeof(unsigned short)));
        launch_f32_to_bf16(w, d, n, stream);
        IRSDE_HIP_CHECK(hipStreamSynchronize(stream));
        bf16_copies[w] = d;
        return d;
    }
    float* upload(const std::vector<float>& v) {
        float* p = dmalloc(v.size());
        IRSDE_HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        return p;
    }
};

namespace irsde {
namespace {

// ---------------------------------------------------------------------------------------------
// Weight inventory (DenoisingUNet_arch.py:19-76; names = reference state_dict keys)
// ---------------------------------------------------------------------------------------------
void finalize_common(irsde_engine* e);
void add_w(irsde_engine* e, const std::string& name, std::vector<int64_t> shape) {
    e->names.push_back(name);
    HostTensor t;
    t.shape = std::move(shape);
    e->host[name] = std::move(t);
}
void inv_resblock(irsde_engine* e, const std::string& p, int ci, int co) {
    const int td = e->time_dim;
    add_w(e, p + "mlp.1.weight", {2 * co, td});
    add_w(e, p + "mlp.1.bias", {2 * co});
    add_w(e, p + "block1.proj.weight", {co, ci, 3, 3});
    add_w(e, p + "block2.proj.weight", {co, co, 3, 3});
    if (ci != co) add_w(e, p + "res_conv.weight", {co, ci, 1, 1});
}
void inv_attn(irsde_engine* e, const std::string& p, int c) {
    add_w(e, p + "fn.norm.g", {1, c, 1, 1});
    add_w(e, p + "fn.fn.to_qkv.weight", {384, c, 1, 1});
    add_w(e, p + "fn.fn.to_out.0.weight", {c, 128, 1, 1});
    add_w(e, p + "fn.fn.to_out.0.bias", {c});
    add_w(e, p + "fn.fn.to_out.1.g", {1, c, 1, 1});
}
void build_inventory(irsde_engine* e) {
    const int nf = e->cfg.nf, depth = e->cfg.depth;
    const bool uncond = (e->cfg.flags & IRSDE_FLAG_UNCOND_FULLATTN) != 0;  // denoising-sde variant
    add_w(e, "init_conv.weight", {nf, (uncond ? 1 : 2) * e->cfg.in_nc, 7, 7});
    add_w(e, "time_mlp.1.weight", {e->time_dim, nf});
    add_w(e, "time_mlp.1.bias", {e->time_dim});
    add_w(e, "time_mlp.3.weight", {e->time_dim, e->time_dim});
    add_w(e, "time_mlp.3.bias", {e->time_dim});
    for (int i = 0; i < depth; ++i) {
        const int di = nf << i, dout = nf << (i + 1);
        const std::string d = "downs." + std::to_string(i) + ".";
        inv_resblock(e, d + "0.", di, di);
        inv_resblock(e, d + "1.", di, di);
        inv_attn(e, d + "2.", di);
        if (i != depth - 1) {
            add_w(e, d + "3.weight", {dout, di, 4, 4});
            add_w(e, d + "3.bias", {dout});
        } else {
            add_w(e, d + "3.weight", {dout, di, 3, 3});
        }
        const std::string u = "ups." + std::to_string(depth - 1 - i) + ".";
        inv_resblock(e, u + "0.", dout + di, dout);
        inv_resblock(e, u + "1.", dout + di, dout);
        inv_attn(e, u + "2.", dout);
        if (i != 0) {
            add_w(e, u + "3.1.weight", {di, dout, 3, 3});
            add_w(e, u + "3.1.bias", {di});
        } else {
            add_w(e, u + "3.weight", {di, dout, 3, 3});
        }
    }
    const int mid = nf << depth;
    inv_resblock(e, "mid_block1.", mid, mid);
    if (uncond) {  // full Attention: to_out is a bare Conv2d, no LayerNorm (module_util.py:182-191)
        add_w(e, "mid_attn.fn.norm.g", {1, mid, 1, 1});
        add_w(e, "mid_attn.fn.fn.to_qkv.weight", {384, mid, 1, 1});
        add_w(e, "mid_attn.fn.fn.to_out.weight", {mid, 128, 1, 1});
        add_w(e, "mid_attn.fn.fn.to_out.bias", {mid});
    } else {
        inv_attn(e, "mid_attn.", mid);
    }
    inv_resblock(e, "mid_block2.", mid, mid);
    inv_resblock(e, "final_res_block.", 2 * nf, nf);
    add_w(e, "final_conv.weight", {e->cfg.out_nc, nf, 3, 3});
    add_w(e, "final_conv.bias", {e->cfg.out_nc});
}

const HostTensor& need(irsde_engine* e, const std::string& n) {
    auto it = e->host.find(n);
    if (it == e->host.end() || !it->second.loaded) throw HipError("missing weight: " + n);
    return it->second;
}

// OIHW -> [O][KH][KW][I]
ConvW pack_conv(irsde_engine* e, const std::string& wname, const std::string& bname) {
    const HostTensor& t = need(e, wname);
    const int O = (int)t.shape[0], I = (int)t.shape[1], KH = (int)t.shape[2], KW = (int)t.shape[3];
    std::vector<float> p((size_t)O * KH * KW * I);
    for (int o = 0; o < O; ++o)
        for (int i = 0; i < I; ++i)
            for (int ky = 0; ky < KH; ++ky)
                for (int kx = 0; kx < KW; ++kx)
                    p[(((size_t)o * KH + ky) * KW + kx) * I + i] = t.data[(((size_t)o * I + i) * KH + ky) * KW + kx];
    ConvW c;
    c.w = e->upload(p);
    c.Cout = O; c.Cin = I; c.KH = KH; c.KW = KW;
    if (!bname.empty()) c.bias = e->upload(need(e, bname).data);
    if (KH == 3 && KW == 3 && I % 32 == 0 && !(e->cfg.flags & (IRSDE_FLAG_NO_WINOGRAD | IRSDE_FLAG_BF16))) {
        for (int tile : {2, 4}) {
            if (tile == 4 && (e->cfg.flags & IRSDE_FLAG_NO_WINOGRAD_F43)) continue;
            if (I < wino_min_c(tile) || O < wino_min_c(tile)) continue;
            std::vector<float> U((size_t)(tile + 2) * (tile + 2) * O * I);
            wino_transform_weights(p.data(), O, I, U.data(), tile);
            (tile == 4 ? c.wino_u4 : c.wino_u2) = e->upload(U);
        }
    }
    return c;
}

// init 7x7 conv as a 7-tap (ky) conv over rows of 7 pixels x P channels: weight [O][7][CK], CK = roundup(7*P,32),
// element (kx, c) at kx*P + c, zeros elsewhere (the kernel over-reads into the next pixels; zero weights).
ConvW pack_init_conv(irsde_engine* e) {
    const HostTensor& t = need(e, "init_conv.weight");
    const int O = (int)t.shape[0], I = (int)t.shape[1];
    const int P = (I + 3) & ~3;
    const int CK = (7 * P + 31) & ~31;
    std::vector<float> p((size_t)O * 7 * CK, 0.f);
    for (int o = 0; o < O; ++o)
        for (int i = 0; i < I; ++i)
            for (int ky = 0; ky < 7; ++ky)
                for (int kx = 0; kx < 7; ++kx)
                    p[((size_t)o * 7 + ky) * CK + kx * P + i] = t.data[(((size_t)o * I + i) * 7 + ky) * 7 + kx];
    ConvW c;
    c.w = e->upload(p);
    c.Cout = O; c.Cin = CK; c.KH = 7; c.KW = 1;
    return c;
}

ResW pack_res(irsde_engine* e, const std::string& p) {
    ResW r;
    r.b1 = pack_conv(e, p + "block1.proj.weight", "");
    r.b2 = pack_conv(e, p + "block2.proj.weight", "");
    r.Cout = r.b1.Cout;
    r.has_res = e->host.count(p + "res_conv.weight") > 0;
    if (r.has_res) r.res = pack_conv(e, p + "res_conv.weight", "");
    r.mlp_w = e->upload(need(e, p + "mlp.1.weight").data);
    r.mlp_b = e->upload(need(e, p + "mlp.1.bias").data);
    return r;
}
AttnW pack_attn(irsde_engine* e, const std::string& p) {
    AttnW a;
    a.g1 = e->upload(need(e, p + "fn.norm.g").data);
    a.qkv = pack_conv(e, p + "fn.fn.to_qkv.weight", "");
    a.out = pack_conv(e, p + "fn.fn.to_out.0.weight", p + "fn.fn.to_out.0.bias");
    a.g2 = e->upload(need(e, p + "fn.fn.to_out.1.g").data);
    a.C = a.out.Cout;
    return a;
}

// ---------------------------------------------------------------------------------------------
// ConditionalNAFNet (Refusion): inventory, packing — DenoisingNAFNet_arch.py:85-147
// ---------------------------------------------------------------------------------------------
inline bool naf_lens(const irsde_engine* e) { return (e->cfg.flags & IRSDE_FLAG_NAF_LENS) != 0; }

void inv_nafblock(irsde_engine* e, const std::string& p, int c) {
    const int td = e->time_dim;
    // latent-bokeh names the block's time MLP `time_mlp` and adds `cam_mlp` (latent-bokeh DenoisingNAFNet_arch.py:18-24)
    const std::string tm = naf_lens(e) ? "time_mlp.1." : "mlp.1.";
    add_w(e, p + tm + "weight", {4 * c, td / 2});
    add_w(e, p + tm + "bias", {4 * c});
    if (naf_lens(e)) {
        add_w(e, p + "cam_mlp.1.weight", {2 * c, td / 2});
        add_w(e, p + "cam_mlp.1.bias", {2 * c});
    }
    add_w(e, p + "conv1.weight", {2 * c, c, 1, 1});
    add_w(e, p + "conv1.bias", {2 * c});
    add_w(e, p + "conv2.weight", {2 * c, 1, 3, 3});
    add_w(e, p + "conv2.bias", {2 * c});
    add_w(e, p + "conv3.weight", {c, c, 1, 1});
    add_w(e, p + "conv3.bias", {c});
    add_w(e, p + "sca.1.weight", {c, c, 1, 1});
    add_w(e, p + "sca.1.bias", {c});
    add_w(e, p + "conv4.weight", {2 * c, c, 1, 1});
    add_w(e, p + "conv4.bias", {2 * c});
    add_w(e, p + "conv5.weight", {c, c, 1, 1});
    add_w(e, p + "conv5.bias", {c});
    add_w(e, p + "norm1.g", {1, c, 1, 1});
    add_w(e, p + "norm2.g", {1, c, 1, 1});
    add_w(e, p + "beta", {1, c, 1, 1});
    add_w(e, p + "gamma", {1, c, 1, 1});
}
void build_inventory_naf(irsde_engine* e) {
    const int width = e->cfg.nf, ic = e->cfg.in_nc, td = e->time_dim;
    // latent-bokeh keeps SinusoidalPosEmb outside the Sequential: indices 0 / 2 instead of 1 / 3 (:103-108)
    const std::string t1 = naf_lens(e) ? "time_mlp.0." : "time_mlp.1.", t3 = naf_lens(e) ? "time_mlp.2." : "time_mlp.3.";
    add_w(e, t1 + "weight", {td * 2, width});
    add_w(e, t1 + "bias", {td * 2});
    add_w(e, t3 + "weight", {td, td});
    add_w(e, t3 + "bias", {td});
    if (naf_lens(e)) {
        add_w(e, "cam_mlp.0.weight", {td * 2, 3 * width});
        add_w(e, "cam_mlp.0.bias", {td * 2});
        add_w(e, "cam_mlp.2.weight", {td, td});
        add_w(e, "cam_mlp.2.bias", {td});
    }
    add_w(e, "intro.weight", {width, 2 * ic, 3, 3});
    add_w(e, "intro.bias", {width});
    add_w(e, "ending.weight", {ic, width, 3, 3});
    add_w(e, "ending.bias", {ic});
    int chan = width;
    for (size_t i = 0; i < e->naf_enc_nums.size(); ++i) {
        for (int j = 0; j < e->naf_enc_nums[i]; ++j)
            inv_nafblock(e, "encoders." + std::to_string(i) + "." + std::to_string(j) + ".", chan);
        add_w(e, "downs." + std::to_string(i) + ".weight", {2 * chan, chan, 2, 2});
        add_w(e, "downs." + std::to_string(i) + ".bias", {2 * chan});
        chan *= 2;
    }
    for (int j = 0; j < e->naf_mid_num; ++j) inv_nafblock(e, "middle_blks." + std::to_string(j) + ".", chan);
    for (size_t i = 0; i < e->naf_dec_nums.size(); ++i) {
        add_w(e, "ups." + std::to_string(i) + ".0.weight", {chan * 2, chan, 1, 1});
        chan /= 2;
        for (int j = 0; j < e->naf_dec_nums[i]; ++j)
            inv_nafblock(e, "decoders." + std::to_string(i) + "." + std::to_string(j) + ".", chan);
    }
}

// 1x1 / KxK conv with an output-row permutation: packed row n' = original row perm[n']
ConvW pack_conv_perm(irsde_engine* e, const std::string& wname, const std::string& bname, const std::vector<int>& perm) {
    const HostTensor& t = need(e, wname);
    const int O = (int)t.shape[0], I = (int)t.shape[1], KH = (int)t.shape[2], KW = (int)t.shape[3];
    std::vector<float> p((size_t)O * KH * KW * I);
    for (int o = 0; o < O; ++o) {
        const int so = perm.empty() ? o : perm[o];
        for (int i = 0; i < I; ++i)
            for (int ky = 0; ky < KH; ++ky)
                for (int kx = 0; kx < KW; ++kx)
                    p[(((size_t)o * KH + ky) * KW + kx) * I + i] = t.data[(((size_t)so * I + i) * KH + ky) * KW + kx];
    }
    ConvW c;
    c.w = e->upload(p);
    c.Cout = O; c.Cin = I; c.KH = KH; c.KW = KW;
    if (!bname.empty()) {
        const HostTensor& b = need(e, bname);
        std::vector<float> pb(O);
        for (int o = 0; o < O; ++o) pb[o] = b.data[perm.empty() ? o : perm[o]];
        c.bias = e->upload(pb);
    }
    return c;
}

NafBlockW pack_nafblock(irsde_engine* e, const std::string& p, int c) {
    NafBlockW b;
    b.c = c;
    b.g1 = e->upload(need(e, p + "norm1.g").data);
    b.g2 = e->upload(need(e, p + "norm2.g").data);
    b.conv1 = pack_conv_perm(e, p + "conv1.weight", p + "conv1.bias", {});
    b.conv3 = pack_conv_perm(e, p + "conv3.weight", p + "conv3.bias", {});
    std::vector<int> gate_perm(2 * c);  // SimpleGate pairs (j, j + c) made adjacent: 2j <- j, 2j+1 <- j + c
    for (int j = 0; j < c; ++j) {
        gate_perm[2 * j] = j;
        gate_perm[2 * j + 1] = j + c;
    }
    b.conv4 = pack_conv_perm(e, p + "conv4.weight", p + "conv4.bias", gate_perm);
    b.conv5 = pack_conv_perm(e, p + "conv5.weight", p + "conv5.bias", {});
    {
        const HostTensor& t = need(e, p + "conv2.weight");  // [2c][1][3][3] -> [9][2c]
        std::vector<float> w((size_t)9 * 2 * c);
        for (int ch = 0; ch < 2 * c; ++ch)
            for (int k = 0; k < 9; ++k) w[(size_t)k * 2 * c + ch] = t.data[(size_t)ch * 9 + k];
        b.dw_w = e->upload(w);
        b.dw_b = e->upload(need(e, p + "conv2.bias").data);
    }
    b.sca_w = e->upload(need(e, p + "sca.1.weight").data);
    b.sca_b = e->upload(need(e, p + "sca.1.bias").data);
    b.beta = e->upload(need(e, p + "beta").data);
    b.gamma = e->upload(need(e, p + "gamma").data);
    const std::string tm = naf_lens(e) ? "time_mlp.1." : "mlp.1.";
    b.mlp_w = e->upload(need(e, p + tm + "weight").data);
    b.mlp_b = e->upload(need(e, p + tm + "bias").data);
    if (naf_lens(e)) {
        b.cam_w = e->upload(need(e, p + "cam_mlp.1.weight").data);
        b.cam_b = e->upload(need(e, p + "cam_mlp.1.bias").data);
    }
    return b;
}

void finalize_naf(irsde_engine* e) {
    const int width = e->cfg.nf, ic = e->cfg.in_nc;
    const std::string t1 = naf_lens(e) ? "time_mlp.0." : "time_mlp.1.", t3 = naf_lens(e) ? "time_mlp.2." : "time_mlp.3.";
    e->tm_w1 = e->upload(need(e, t1 + "weight").data);
    e->tm_b1 = e->upload(need(e, t1 + "bias").data);
    e->tm_w3 = e->upload(need(e, t3 + "weight").data);
    e->tm_b3 = e->upload(need(e, t3 + "bias").data);
    if (naf_lens(e)) {
        e->cm_w1 = e->upload(need(e, "cam_mlp.0.weight").data);
        e->cm_b1 = e->upload(need(e, "cam_mlp.0.bias").data);
        e->cm_w3 = e->upload(need(e, "cam_mlp.2.weight").data);
        e->cm_b3 = e->upload(need(e, "cam_mlp.2.bias").data);
    }
    {
        const int half = width / 2;
        std::vector<float> f(half);
        const double emb = std::log(10000.0) / (half - 1);
        for (int i = 0; i < half; ++i) f[i] = expf((float)i * (float)(-emb));
        e->freqs = e->upload(f);
    }
    {   // intro 3x3 (2*ic -> width, bias) as a 3-tap (ky) conv over rows of 3 pixels x P channels (+ zero K padding)
        const HostTensor& t = need(e, "intro.weight");
        const int O = (int)t.shape[0], I = (int)t.shape[1];
        const int P = (I + 3) & ~3;
        const int CK = (3 * P + 31) & ~31;
        std::vector<float> p((size_t)O * 3 * CK, 0.f);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i)
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx)
                        p[((size_t)o * 3 + ky) * CK + kx * P + i] = t.data[(((size_t)o * I + i) * 3 + ky) * 3 + kx];
        e->naf_intro.w = e->upload(p);
        e->naf_intro.Cout = O; e->naf_intro.Cin = CK; e->naf_intro.KH = 3; e->naf_intro.KW = 1;
        e->naf_intro.bias = e->upload(need(e, "intro.bias").data);
    }
    e->naf_ending = pack_conv_perm(e, "ending.weight", "ending.bias", {});
    (void)ic;
    int chan = width;
    e->naf_enc.resize(e->naf_enc_nums.size());
    for (size_t i = 0; i < e->naf_enc_nums.size(); ++i) {
        for (int j = 0; j < e->naf_enc_nums[i]; ++j)
            e->naf_enc[i].push_back(pack_nafblock(e, "encoders." + std::to_string(i) + "." + std::to_string(j) + ".", chan));
        e->naf_downs.push_back(pack_conv_perm(e, "downs." + std::to_string(i) + ".weight", "downs." + std::to_string(i) + ".bias", {}));
        chan *= 2;
    }
    for (int j = 0; j < e->naf_mid_num; ++j) e->naf_mid.push_back(pack_nafblock(e, "middle_blks." + std::to_string(j) + ".", chan));
    e->naf_dec.resize(e->naf_dec_nums.size());
    for (size_t i = 0; i < e->naf_dec_nums.size(); ++i) {
        // ups.i.0: 1x1 chan -> 2 chan, then PixelShuffle(2): row n' = q * Cq + co  <-  co * 4 + q
        const int Cq = chan / 2;
        std::vector<int> perm(2 * chan);
        for (int q = 0; q < 4; ++q)
            for (int co = 0; co < Cq; ++co) perm[q * Cq + co] = co * 4 + q;
        e->naf_ups.push_back(pack_conv_perm(e, "ups." + std::to_string(i) + ".0.weight", "", perm));
        chan /= 2;
        for (int j = 0; j < e->naf_dec_nums[i]; ++j)
            e->naf_dec[i].push_back(pack_nafblock(e, "decoders." + std::to_string(i) + "." + std::to_string(j) + ".", chan));
    }
    e->naf_all.clear();
    for (auto& v : e->naf_enc) for (auto& b : v) e->naf_all.push_back(&b);
    for (auto& b : e->naf_mid) e->naf_all.push_back(&b);
    for (auto& v : e->naf_dec) for (auto& b : v) e->naf_all.push_back(&b);
    int off = 0, coff = 0;
    for (NafBlockW* b : e->naf_all) {
        b->film_off = off;
        off += 4 * b->c;
        b->cam_off = coff;
        coff += 2 * b->c;
    }
    e->film_row = off;
    e->cam_row = coff;
}

// ---------------------------------------------------------------------------------------------
// Latent UNet (arch == 2): inventory / packing — latent-dehazing/models/modules/UNet_arch.py:17-57.
// Channel counts there are small and irregular (8, 40, 96 ...); every NHWC tensor is stored with its channel count
// rounded up to 32 and the packed weights carry zero rows / zero K columns for the padding, so the padding channels
// hold exact zeros everywhere and the implicit-GEMM kernel (32-channel K chunks) needs no special case.
// ---------------------------------------------------------------------------------------------
inline int rup32(int c) { return (c + 31) & ~31; }

void build_inventory_latent(irsde_engine* e) {
    const int depth = (int)e->lat_mult.size(), ch = e->lat_ch;
    auto dim = [&](int i) { return i == 0 ? ch : ch * e->lat_mult[i - 1]; };
    auto resb = [&](const std::string& p, int ci, int co) {
        add_w(e, p + "block1.proj.weight", {co, ci, 3, 3});
        add_w(e, p + "block2.proj.weight", {co, co, 3, 3});
        if (ci != co) add_w(e, p + "res_conv.weight", {co, ci, 1, 1});
    };
    add_w(e, "init_conv.weight", {ch, e->lat_in, 3, 3});
    for (int i = 0; i < depth; ++i) {
        const int di = dim(i), dout = dim(i + 1);
        const std::string en = "encoder." + std::to_string(i) + ".";
        resb(en + "0.", di, di);
        resb(en + "1.", di, di);
        if (i == depth - 1) inv_attn(e, en + "2.", di);
        if (i != depth - 1) {
            add_w(e, en + "3.weight", {dout, di, 4, 4});
            add_w(e, en + "3.bias", {dout});
        } else {
            add_w(e, en + "3.weight", {dout, di, 3, 3});
        }
        const std::string de = "decoder." + std::to_string(depth - 1 - i) + ".";
        resb(de + "0.", dout + di, dout);
        resb(de + "1.", dout + di, dout);
        if (i == depth - 1) inv_attn(e, de + "2.", dout);
        if (i != 0) {
            add_w(e, de + "3.1.weight", {di, dout, 3, 3});
            add_w(e, de + "3.1.bias", {di});
        } else {
            add_w(e, de + "3.weight", {di, dout, 3, 3});
        }
    }
    const int mid = dim(depth);
    add_w(e, "latent_conv.weight", {e->lat_embed, mid, 1, 1});
    add_w(e, "post_latent_conv.weight", {mid, e->lat_embed, 1, 1});
    add_w(e, "final_conv.weight", {e->lat_out, ch, 3, 3});
    add_w(e, "final_conv.bias", {e->lat_out});
}

// OIHW -> [O_p][KH][KW][sum rup32(split)] with zero padding; `splits` = logical channels of each concatenated source
ConvW pack_conv_pad(irsde_engine* e, const std::string& wname, const std::string& bname, const std::vector<int>& splits,
                    bool pad_out) {
    const HostTensor& t = need(e, wname);
    const int O = (int)t.shape[0], I = (int)t.shape[1], KH = (int)t.shape[2], KW = (int)t.shape[3];
    int isum = 0, Ip = 0;
    for (int c : splits) { isum += c; Ip += rup32(c); }
    if (isum != I) throw HipError("pack_conv_pad: channel split mismatch for " + wname);
    const int Op = pad_out ? rup32(O) : O;
    std::vector<float> p((size_t)Op * KH * KW * Ip, 0.f);
    for (int o = 0; o < O; ++o) {
        int src = 0, dst = 0;
        for (int c : splits) {
            for (int i = 0; i < c; ++i)
                for (int ky = 0; ky < KH; ++ky)
                    for (int kx = 0; kx < KW; ++kx)
                        p[(((size_t)o * KH + ky) * KW + kx) * Ip + dst + i] = t.data[(((size_t)o * I + src + i) * KH + ky) * KW + kx];
            src += c;
            dst += rup32(c);
        }
    }
    ConvW cw;
    cw.w = e->upload(p);
    cw.Cout = Op; cw.Cin = Ip; cw.KH = KH; cw.KW = KW;
    if (!bname.empty()) {
        std::vector<float> pb(Op, 0.f);
        const HostTensor& b = need(e, bname);
        for (int o = 0; o < O; ++o) pb[o] = b.data[o];
        cw.bias = e->upload(pb);
    }
    return cw;
}

void finalize_latent(irsde_engine* e) {
    const int depth = (int)e->lat_mult.size(), ch = e->lat_ch;
    auto dim = [&](int i) { return i == 0 ? ch : ch * e->lat_mult[i - 1]; };
    auto resb = [&](const std::string& p, const std::vector<int>& in_splits, int co) {
        ResW r;
        r.b1 = pack_conv_pad(e, p + "block1.proj.weight", "", in_splits, true);
        r.b2 = pack_conv_pad(e, p + "block2.proj.weight", "", {co}, true);
        r.Cout = r.b1.Cout;
        r.has_res = e->host.count(p + "res_conv.weight") > 0;
        if (r.has_res) r.res = pack_conv_pad(e, p + "res_conv.weight", "", in_splits, true);
        return r;
    };
    e->lat_init = pack_conv_pad(e, "init_conv.weight", "", {e->lat_in}, true);
    e->lat_dec_res.resize(2 * depth);
    e->lat_up.resize(depth);
    for (int i = 0; i < depth; ++i) {
        const int di = dim(i), dout = dim(i + 1);
        const std::string en = "encoder." + std::to_string(i) + ".";
        e->lat_enc_res.push_back(resb(en + "0.", {di}, di));
        e->lat_enc_res.push_back(resb(en + "1.", {di}, di));
        if (i == depth - 1) {
            if (di % 32) throw HipError("latent UNet: the attention level needs a channel count that is a multiple of 32");
            e->lat_enc_attn = pack_attn(e, en + "2.");
        }
        e->lat_down.push_back(pack_conv_pad(e, en + "3.weight", i != depth - 1 ? en + "3.bias" : "", {di}, true));
        const int j = depth - 1 - i;
        const std::string de = "decoder." + std::to_string(j) + ".";
        e->lat_dec_res[2 * j] = resb(de + "0.", {dout, di}, dout);
        e->lat_dec_res[2 * j + 1] = resb(de + "1.", {dout, di}, dout);
        if (i == depth - 1) {
            if (dout % 32) throw HipError("latent UNet: the attention level needs a channel count that is a multiple of 32");
            e->lat_dec_attn = pack_attn(e, de + "2.");
        }
        e->lat_up[j] = i != 0 ? pack_conv_pad(e, de + "3.1.weight", de + "3.1.bias", {dout}, true)
                              : pack_conv_pad(e, de + "3.weight", "", {dout}, true);
    }
    e->lat_latent = pack_conv_pad(e, "latent_conv.weight", "", {dim(depth)}, true);
    e->lat_post = pack_conv_pad(e, "post_latent_conv.weight", "", {e->lat_embed}, true);
    e->lat_final = pack_conv_pad(e, "final_conv.weight", "final_conv.bias", {ch}, false);
}

void finalize(irsde_engine* e) {
    IRSDE_HIP_CHECK(hipSetDevice(e->cfg.device));
    for (auto& n : e->names)
        if (!e->host[n].loaded) throw HipError("missing weight: " + n);
    if (e->arch == 2) {
        finalize_latent(e);
        finalize_common(e);
        return;
    }
    if (e->arch == 1) {
        finalize_naf(e);
        finalize_common(e);
        return;
    }
    const int depth = e->cfg.depth;
    e->init_conv = pack_init_conv(e);
    e->tm_w1 = e->upload(need(e, "time_mlp.1.weight").data);
    e->tm_b1 = e->upload(need(e, "time_mlp.1.bias").data);
    e->tm_w3 = e->upload(need(e, "time_mlp.3.weight").data);
    e->tm_b3 = e->upload(need(e, "time_mlp.3.bias").data);
    {
        // SinusoidalPosEmb frequencies (module_util.py:35-38), fp32 like the reference
        const int half = e->cfg.nf / 2;
        std::vector<float> f(half);
        const double emb = std::log(10000.0) / (half - 1);
        for (int i = 0; i < half; ++i) f[i] = expf((float)i * (float)(-emb));
        e->freqs = e->upload(f);
    }
    e->down_res.reserve(2 * depth);
    e->up_res.reserve(2 * depth);
    for (int i = 0; i < depth; ++i) {
        const std::string d = "downs." + std::to_string(i) + ".";
        e->down_res.push_back(pack_res(e, d + "0."));
        e->down_res.push_back(pack_res(e, d + "1."));
        e->down_attn.push_back(pack_attn(e, d + "2."));
        e->down_conv.push_back(pack_conv(e, d + "3.weight", i != depth - 1 ? d + "3.bias" : ""));
    }
    e->mid1 = pack_res(e, "mid_block1.");
    if (e->cfg.flags & IRSDE_FLAG_UNCOND_FULLATTN) {
        AttnW a;
        a.g1 = e->upload(need(e, "mid_attn.fn.norm.g").data);
        a.qkv = pack_conv(e, "mid_attn.fn.fn.to_qkv.weight", "");
        a.out = pack_conv(e, "mid_attn.fn.fn.to_out.weight", "mid_attn.fn.fn.to_out.bias");
        a.C = a.out.Cout;
        e->mid_attn = a;  // g2 == nullptr marks the full-attention block
    } else {
        e->mid_attn = pack_attn(e, "mid_attn.");
    }
    e->mid2 = pack_res(e, "mid_block2.");
    for (int j = 0; j < depth; ++j) {
        const std::string u = "ups." + std::to_string(j) + ".";
        e->up_res.push_back(pack_res(e, u + "0."));
        e->up_res.push_back(pack_res(e, u + "1."));
        e->up_attn.push_back(pack_attn(e, u + "2."));
        if (j != depth - 1)
            e->up_conv.push_back(pack_conv(e, u + "3.1.weight", u + "3.1.bias"));
        else
            e->up_conv.push_back(pack_conv(e, u + "3.weight", ""));
    }
    e->final_res = pack_res(e, "final_res_block.");
    e->final_conv = pack_conv(e, "final_conv.weight", "final_conv.bias");

    e->all_res.clear();
    for (auto& r : e->down_res) e->all_res.push_back(&r);
    e->all_res.push_back(&e->mid1);
    e->all_res.push_back(&e->mid2);
    for (auto& r : e->up_res) e->all_res.push_back(&r);
    e->all_res.push_back(&e->final_res);
    int off = 0;
    for (ResW* r : e->all_res) {
        r->film_off = off;
        off += 2 * r->Cout;
    }
    e->film_row = off;
    finalize_common(e);
}

void finalize_common(irsde_engine* e) {
    conv_global_init();
    e->zeros = e->dmalloc(256);
    IRSDE_HIP_CHECK(hipMemset(e->zeros, 0, 1024));
    IRSDE_HIP_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    IRSDE_HIP_CHECK(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
    IRSDE_HIP_CHECK(hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming));
    e->step = reinterpret_cast<StepState*>(e->dmalloc(sizeof(StepState) / 4 + 4));
    e->ctl = reinterpret_cast<SampleCtl*>(e->dmalloc(sizeof(SampleCtl) / 4 + 4));
    IRSDE_HIP_CHECK(hipMemset(e->step, 0, sizeof(StepState)));
    IRSDE_HIP_CHECK(hipMemset(e->ctl, 0, sizeof(SampleCtl)));
    // free host copies
    for (auto& kv : e->host) std::vector<float>().swap(kv.second.data);
    e->finalized = true;
}

// FiLM rows for `rows` timesteps (device array tvals[rows]) -> dst[rows][film_row]
void compute_film_rows(irsde_engine* e, const float* tvals, int rows, float* dst, hipStream_t s) {
    const int nf = e->cfg.nf, td = e->time_dim;
    float* emb = nullptr;
    float* h1 = nullptr;
    float* h2 = nullptr;
    IRSDE_HIP_CHECK(hipMalloc(&emb, (size_t)rows * nf * 4));
    IRSDE_HIP_CHECK(hipMalloc(&h1, (size_t)rows * td * 4));
    IRSDE_HIP_CHECK(hipMalloc(&h2, (size_t)rows * td * 4));
    launch_sinusoid(tvals, e->freqs, emb, rows, nf / 2, s);
    if (e->arch == 1) {
        // time_mlp: Linear(width, 2*td) -> SimpleGate -> Linear(td, td); block mlp: SimpleGate -> Linear(td/2, 4c)
        // (DenoisingNAFNet_arch.py:93-98, 18-20)
        float *w1 = nullptr, *g1 = nullptr, *g2 = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&w1, (size_t)rows * 2 * td * 4));
        IRSDE_HIP_CHECK(hipMalloc(&g1, (size_t)rows * td * 4));
        IRSDE_HIP_CHECK(hipMalloc(&g2, (size_t)rows * (td / 2) * 4));
        launch_row_linear(emb, nf, e->tm_w1, e->tm_b1, w1, 2 * td, rows, nf, 2 * td, ACT_NONE, ACT_NONE, s);
        launch_row_gate(w1, g1, rows, td, s);
        launch_row_linear(g1, td, e->tm_w3, e->tm_b3, h2, td, rows, td, td, ACT_NONE, ACT_NONE, s);
        launch_row_gate(h2, g2, rows, td / 2, s);
        for (NafBlockW* b : e->naf_all)
            launch_row_linear(g2, td / 2, b->mlp_w, b->mlp_b, dst + b->film_off, e->film_row, rows, td / 2, 4 * b->c,
                              ACT_NONE, ACT_NONE, s);
        IRSDE_HIP_CHECK(hipStreamSynchronize(s));
        (void)hipFree(w1); (void)hipFree(g1); (void)hipFree(g2);
    } else {
    launch_row_linear(emb, nf, e->tm_w1, e->tm_b1, h1, td, rows, nf, td, ACT_NONE, ACT_GELU, s);
    launch_row_linear(h1, td, e->tm_w3, e->tm_b3, h2, td, rows, td, td, ACT_NONE, ACT_NONE, s);
    for (ResW* r : e->all_res)
        launch_row_linear(h2, td, r->mlp_w, r->mlp_b, dst + r->film_off, e->film_row, rows, td, 2 * r->Cout, ACT_SILU,
                          ACT_NONE, s);
    }
    IRSDE_HIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(emb);
    (void)hipFree(h1);
    (void)hipFree(h2);
}

void ensure_film_cur(irsde_engine* e, int rows) {
    if (rows <= e->film_cur_rows) return;
    e->film_cur = e->dmalloc((size_t)rows * e->film_row);
    e->film_cur_rows = rows;
    // plans bake the film_cur pointer: drop them
    e->plans.clear();
}

// ---------------------------------------------------------------------------------------------
// Plan construction: one network evaluation as a static launch list over a static arena
// ---------------------------------------------------------------------------------------------
struct Builder {
    irsde_engine* e;
    Plan* pl;
    bool reuse;
    bool naive;
    int film_bstride;
    const float* fused_ln_g = nullptr;  // set around a conv() call: LayerNorm gain applied in that conv's epilogue

    bool act_bf16() const { return (e->cfg.flags & IRSDE_FLAG_BF16_ACT) != 0; }
    Tensor talloc(int B, int H, int W, int C, int force_f32 = 0) {
        Tensor t;
        t.B = B; t.H = H; t.W = W; t.C = C;
        t.bf16 = act_bf16() && !force_f32;
        t.p = pl->alloc(t.bf16 ? (t.numel() + 1) / 2 : t.numel(), reuse);
        return t;
    }
    void tfree(const Tensor& t) {
        if (reuse) pl->release(t.p);
    }
    void tap(const std::string& name, const Tensor& t) { pl->taps[name] = t; }

    void push_conv(ConvParams p) {
        const int M = p.B * p.Ho * p.Wo;
        // split-K for under-filled grids (small batch / deep levels)
        const int bn = p.Cout >= 128 ? 128 : (p.Cout > 32 ? 64 : 32);
        const int blocks = ((M + 127) / 128) * ((p.Cout + bn - 1) / bn);
        const int nk = p.KH * p.KW * ((p.C0 + p.C1) / 32);
        int splits = 1;
        if (!naive && blocks < 256 && nk >= 16 && !p.gate && !p.shuffle) {
            splits = std::min(std::min(nk / 8, (512 + blocks - 1) / blocks), 16);
            if (splits < 2) splits = 1;
        }
        if (splits > 1) {
            p.splits = splits;
            p.partial = pl->alloc((size_t)splits * M * p.Cout, true);
        }
        p.zeros = e->zeros;
        if (!naive && (e->cfg.flags & IRSDE_FLAG_BF16))
            p.w_bf = e->bf16_copy(p.w, (size_t)p.Cout * p.KH * p.KW * (p.C0 + p.C1));
        Op op;
        op.kind = OP_CONV;
        op.flops = conv_flops(p);
        const double in_bytes = (p.in_bf16 ? 2.0 : 4.0) * (double)p.B * (p.Hin) * (p.Win) * (double)(p.C0 + p.C1);
        op.bytes = in_bytes + (p.out_bf16 ? 2.0 : 4.0) * (double)M * p.Cout +
                   (p.w_bf ? 2.0 : 4.0) * (double)p.Cout * p.KH * p.KW * (p.C0 + p.C1);
        op.exec_flops = op.flops;
        pl->conv_flops += op.flops;
        pl->conv_exec_flops += op.exec_flops;
        pl->conv_bytes += op.bytes;
        {
            char buf[256];
            snprintf(buf, sizeof buf, "conv%s M=%d Cout=%d Cin=%d k=%dx%d s=%d up=%d splits=%d blocks=%d flops=%.4g",
                     p.w_bf ? "(bf16)" : "", M, p.Cout, p.C0 + p.C1, p.KH, p.KW, p.stride, p.in_shift, splits, blocks, op.flops);
            op.desc = buf;
        }
        const bool nv = naive;
        op.fn = [p, nv](hipStream_t s) {
            if (nv)
                launch_conv_naive(p, s);
            else
                launch_conv(p, s);
        };
        pl->net_ops.push_back(std::move(op));
        // the split-K scratch is dead once this op's reduce kernel has run (same stream): recycle it
        if (p.partial) pl->release(p.partial);
    }

    // generic KxK conv over (in0 | in1)
    Tensor conv(const ConvW& w, const Tensor& in0, const Tensor* in1, int stride, int pad, int in_shift,
                const float* film, int silu, const Tensor* res, int out_stride = 0) {
        ConvParams p;
        p.in0 = in0.p; p.C0 = in0.C; p.pix0 = in0.C;
        if (in1) { p.in1 = in1->p; p.C1 = in1->C; p.pix1 = in1->C; }
        if (p.C0 + p.C1 != w.Cin) throw HipError("conv: channel mismatch");
        p.Hin = in0.H; p.Win = in0.W; p.in_shift = in_shift;
        p.w = w.w; p.Cout = w.Cout; p.KH = w.KH; p.KW = w.KW; p.stride = stride; p.pad_y = pad; p.pad_x = pad;
        const int Hv = in0.H << in_shift, Wv = in0.W << in_shift;
        p.B = in0.B;
        p.Ho = (Hv + 2 * pad - w.KH) / stride + 1;
        p.Wo = (Wv + 2 * pad - w.KW) / stride + 1;
        const int ostr = out_stride ? out_stride : w.Cout;
        Tensor out = talloc(p.B, p.Ho, p.Wo, ostr, out_stride != 0);  // an explicit stride = the fp32 eps_hat tensor
        p.out = out.p; p.out_stride = ostr;
        p.in_bf16 = in0.bf16; p.out_bf16 = out.bf16;
        p.ln_g = fused_ln_g;
        if ((in1 && in1->bf16 != in0.bf16) || (res && res->bf16 != out.bf16)) throw HipError("conv: mixed activation storage types");
        p.bias = w.bias;
        p.film = film; p.film_bstride = film ? film_bstride : 0;
        p.silu = silu;
        if (res) { p.res = res->p; p.res_stride = res->C; }
        if (!naive) {  // prefer F(4x4,3x3), then F(2x2,3x3), then the direct implicit GEMM
            if (w.wino_u4 && wino_shape_ok(p, 4) && push_wino(p, w.wino_u4, 4)) return out;
            if (w.wino_u2 && wino_shape_ok(p, 2) && push_wino(p, w.wino_u2, 2)) return out;
        }
        push_conv(p);
        return out;
    }

    // Winograd F(2x2,3x3): input transform -> 16 batched GEMMs on the MFMA kernel -> output transform + epilogue
    bool push_wino(const ConvParams& d, const float* U, int tile) {
        const int Ctot = d.C0 + d.C1;
        const int ncomp = (tile + 2) * (tile + 2);
        const long long T = (long long)d.B * (d.Ho / tile) * (d.Wo / tile);
        const long long gemm_blocks = ncomp * ((T + 127) / 128) * ((d.Cout + 127) / 128);
        if (gemm_blocks < 256) return false;  // tiny layers: direct conv + split-K
        ConvParams dd = d;
        dd.zeros = e->zeros;
        float* V = pl->alloc((size_t)ncomp * T * Ctot, true);
        float* Mb = pl->alloc((size_t)ncomp * T * d.Cout, true);
        const WinoPlan wp = make_wino(dd, U, V, Mb, tile);
        const double direct = conv_flops(d);
        {
            Op op;
            op.kind = OP_WINO;
            op.desc = "wino_input T=" + std::to_string(T) + " C=" + std::to_string(Ctot);
            const WinoParams ip = wp.in;
            op.fn = [ip](hipStream_t s) { launch_wino_input(ip, s); };
            pl->net_ops.push_back(std::move(op));
        }
        {
            Op op;
            op.kind = OP_CONV;
            op.flops = direct;
            op.exec_flops = ncomp * 2.0 * (double)T * Ctot * d.Cout;
            const double in_bytes = 4.0 * (double)d.B * d.Hin * d.Win * Ctot;
            op.bytes = in_bytes + 4.0 * (double)d.B * d.Ho * d.Wo * d.Cout + 4.0 * 9.0 * (double)d.Cout * Ctot;
            pl->conv_flops += op.flops;
            pl->conv_exec_flops += op.exec_flops;
            pl->conv_bytes += op.bytes;
            char buf[256];
            snprintf(buf, sizeof buf, "conv(winograd F%d gemm x%d) T=%lld Cout=%d Cin=%d flops=%.4g exec=%.4g", tile, ncomp, T,
                     d.Cout, Ctot, op.flops, op.exec_flops);
            op.desc = buf;
            const ConvParams g = wp.gemm;
            op.fn = [g](hipStream_t s) { launch_conv(g, s); };
            pl->net_ops.push_back(std::move(op));
        }
        {
            Op op;
            op.kind = OP_WINO;
            op.desc = "wino_output T=" + std::to_string(T) + " Cout=" + std::to_string(d.Cout);
            const WinoParams oparm = wp.out;
            op.fn = [oparm](hipStream_t s) { launch_wino_output(oparm, s); };
            pl->net_ops.push_back(std::move(op));
        }
        pl->release(V);
        pl->release(Mb);
        return true;
    }

    struct ConvOpts {
        int stride = 1, pad = 0;
        const Tensor* res = nullptr;
        const float* ch_scale = nullptr;
        const float* in_scale = nullptr;
        const float* gate_film = nullptr;
        int gate = 0, shuffle = 0, out_stride = 0;
    };
    // 1x1 / KxK conv with the NAFNet fusions (bias from the ConvW; no FiLM / SiLU in NAFNet convs)
    Tensor conv_naf(const ConvW& w, const Tensor& in, const ConvOpts& o) {
        ConvParams p;
        p.in0 = in.p; p.C0 = in.C; p.pix0 = in.C;
        if (in.C != w.Cin) throw HipError("conv_naf: channel mismatch");
        p.Hin = in.H; p.Win = in.W;
        p.w = w.w; p.Cout = w.Cout; p.KH = w.KH; p.KW = w.KW; p.stride = o.stride; p.pad_y = o.pad; p.pad_x = o.pad;
        p.B = in.B;
        p.Ho = (in.H + 2 * o.pad - w.KH) / o.stride + 1;
        p.Wo = (in.W + 2 * o.pad - w.KW) / o.stride + 1;
        Tensor out;
        if (o.shuffle)
            out = talloc(p.B, 2 * p.Ho, 2 * p.Wo, w.Cout / 4);
        else if (o.gate)
            out = talloc(p.B, p.Ho, p.Wo, w.Cout / 2);
        else
            out = talloc(p.B, p.Ho, p.Wo, o.out_stride ? o.out_stride : w.Cout);
        p.out = out.p; p.out_stride = out.C;
        p.bias = w.bias;
        p.ch_scale = o.ch_scale; p.in_scale = o.in_scale; p.gate = o.gate; p.shuffle = o.shuffle;
        p.gate_film = o.gate_film; p.gate_film_bstride = o.gate_film ? e->cam_row : 0;
        if (o.res) { p.res = o.res->p; p.res_stride = o.res->C; }
        push_conv(p);
        return out;
    }

    // NAFBlock.forward — DenoisingNAFNet_arch.py:56-82
    Tensor nafblock(const NafBlockW& w, const Tensor& x) {
        const int64_t M = (int64_t)x.B * x.H * x.W;
        const int64_t ppi = (int64_t)x.H * x.W;
        const int c = w.c;
        const float* film = e->film_cur + w.film_off;  // [shift_att | scale_att | shift_ffn | scale_ffn]
        const int fb = film_bstride;
        Tensor t1 = talloc(x.B, x.H, x.W, c);
        {
            const float *xp = x.p, *g = w.g1;
            float* o = t1.p;
            push_other(OP_LN, [=](hipStream_t s) { launch_layernorm_film(xp, g, film + c, film, fb, ppi, o, M, c, 1e-5f, s); });
        }
        Tensor u = conv_naf(w.conv1, t1, ConvOpts());
        tfree(t1);
        Tensor gt = talloc(x.B, x.H, x.W, c);
        const int nt = dwgate_tiles(x.H * x.W);
        float* partial = pl->alloc((size_t)x.B * nt * c, true);
        float* sca = pl->alloc((size_t)x.B * c, true);
        float* mean = pl->alloc((size_t)x.B * c, true);
        {
            const float *up = u.p, *dw = w.dw_w, *db = w.dw_b, *sw = w.sca_w, *sb = w.sca_b;
            float* gp = gt.p;
            const int B = x.B, H = x.H, W = x.W;
            push_other(OP_OTHER, [=](hipStream_t s) {
                launch_dwconv_gate(up, dw, db, gp, partial, B, H, W, c, s);
                launch_sca(partial, nt, sw, sb, mean, sca, B, c, H * W, s);
            });
        }
        tfree(u);
        ConvOpts o3;
        o3.in_scale = sca; o3.ch_scale = w.beta; o3.res = &x;
        Tensor y = conv_naf(w.conv3, gt, o3);
        tfree(gt);
        pl->release(partial);
        pl->release(sca);
        pl->release(mean);
        Tensor t2 = talloc(x.B, x.H, x.W, c);
        {
            const float *yp = y.p, *g = w.g2;
            float* o = t2.p;
            push_other(OP_LN, [=](hipStream_t s) { launch_layernorm_film(yp, g, film + 3 * c, film + 2 * c, fb, ppi, o, M, c, 1e-5f, s); });
        }
        ConvOpts o4;
        o4.gate = 1;
        if (naf_lens(e)) o4.gate_film = e->cam_cur + w.cam_off;  // x * (cam_scale + 1) + cam_shift after the gate (:82-83)
        Tensor v = conv_naf(w.conv4, t2, o4);
        tfree(t2);
        ConvOpts o5;
        o5.ch_scale = w.gamma; o5.res = &y;
        Tensor out = conv_naf(w.conv5, v, o5);
        tfree(v);
        tfree(y);
        return out;
    }

    // ResBlock.forward — module_util.py:136-146
    Tensor resblock(const ResW& w, const Tensor& in0, const Tensor* in1) {
        Tensor R;
        if (w.has_res)
            R = conv(w.res, in0, in1, 1, 0, 0, nullptr, 0, nullptr);
        else
            R = in0;
        // latent UNet ResBlocks have no time MLP (UNet_arch.py:23): plain conv -> SiLU
        Tensor h1 = conv(w.b1, in0, in1, 1, 1, 0, w.mlp_w ? e->film_cur + w.film_off : nullptr, 1, nullptr);
        Tensor out = conv(w.b2, h1, nullptr, 1, 1, 0, nullptr, 1, &R);
        tfree(h1);
        if (w.has_res) tfree(R);
        return out;
    }

    // Residual(PreNorm(dim, LinearAttention(dim))) — module_util.py:20-26,82-90,150-178
    Tensor attn(const AttnW& w, const Tensor& x) {
        const int64_t M = (int64_t)x.B * x.H * x.W;
        const int N = x.H * x.W;
        Tensor xn = talloc(x.B, x.H, x.W, x.C);
        {
            const float *xp = x.p, *g = w.g1;
            float* o = xn.p;
            const int C = x.C;
            const bool bf = x.bf16;
            push_other(OP_LN, [=](hipStream_t s) { launch_layernorm(xp, g, nullptr, o, M, C, 1e-5f, s, bf); });
        }
        Tensor qkv = conv(w.qkv, xn, nullptr, 1, 0, 0, nullptr, 0, nullptr);
        tfree(xn);
        Tensor a = talloc(x.B, x.H, x.W, 128);
        if (!w.g2) {
            // Residual(PreNorm(dim, Attention(dim))): full softmax attention, to_out without LayerNorm, + x
            const float* q = qkv.p;
            float* o = a.p;
            const int B = x.B;
            push_other(OP_ATTN, [=](hipStream_t s) { launch_full_attention(q, o, B, N, s); });
            tfree(qkv);
            Tensor y = conv(w.out, a, nullptr, 1, 0, 0, nullptr, 0, &x);
            tfree(a);
            return y;
        }
        {
            AttnWorkspace ws;
            ws.nch = attn_num_chunks(N);
            ws.pmax = pl->alloc((size_t)x.B * ws.nch * 128, false);
            ws.pctx = pl->alloc((size_t)x.B * 4 * ws.nch * 1024, false);
            ws.psum = pl->alloc((size_t)x.B * 4 * ws.nch * 32, false);
            ws.ctx = pl->alloc((size_t)x.B * 4 * 1024, false);
            const float* q = qkv.p;
            float* o = a.p;
            const int B = x.B;
            const bool bf = x.bf16;
            push_other(OP_ATTN, [=](hipStream_t s) { launch_linear_attention(q, o, B, N, ws, s, bf); });
        }
        tfree(qkv);
        if (!naive && (x.C == 64 || x.C == 128) && !(e->cfg.flags & IRSDE_FLAG_NO_FUSED_LN)) {
            // to_out conv + LayerNorm + residual in one kernel: the conv tile holds the whole channel row
            fused_ln_g = w.g2;
            Tensor y = conv(w.out, a, nullptr, 1, 0, 0, nullptr, 0, &x);
            fused_ln_g = nullptr;
            tfree(a);
            return y;
        }
        Tensor o = conv(w.out, a, nullptr, 1, 0, 0, nullptr, 0, nullptr);
        tfree(a);
        Tensor y = talloc(x.B, x.H, x.W, x.C);
        {
            const float *op = o.p, *g = w.g2, *r = x.p;
            float* yp = y.p;
            const int C = x.C;
            const bool bf = x.bf16;
            push_other(OP_LN, [=](hipStream_t s) { launch_layernorm(op, g, r, yp, M, C, 1e-5f, s, bf); });
        }
        tfree(o);
        return y;
    }

    void push_other(OpKind k, std::function<void(hipStream_t)> fn) {
        Op op;
        op.kind = k;
        op.desc = k == OP_LN ? "layernorm" : (k == OP_ATTN ? "linear_attention" : "other");
        op.fn = std::move(fn);
        pl->net_ops.push_back(std::move(op));
    }
};

// ConditionalNAFNet.forward — DenoisingNAFNet_arch.py:149-187
void build_naf_plan(irsde_engine* e, Plan* pl, Builder& b, int P) {
    const int B = pl->B;
    Tensor x;
    {   // intro 3x3 (+bias) as 3 row taps over the zero-bordered NHWC input (border 3: first tap row/col = +2)
        ConvParams p;
        p.in0 = pl->x0; p.C0 = e->naf_intro.Cin; p.pix0 = P;
        p.Hin = pl->Hp + 6; p.Win = pl->Wp + 6;
        p.w = e->naf_intro.w; p.Cout = e->naf_intro.Cout; p.KH = 3; p.KW = 1; p.stride = 1; p.pad_y = -2; p.pad_x = -2;
        p.B = B; p.Ho = pl->Hp; p.Wo = pl->Wp;
        p.bias = e->naf_intro.bias;
        x = b.talloc(B, pl->Hp, pl->Wp, p.Cout);
        p.out = x.p; p.out_stride = p.Cout;
        b.push_conv(p);
        const double real = 2.0 * (double)B * pl->Hp * pl->Wp * p.Cout * 9.0 * (2.0 * e->cfg.in_nc);
        pl->conv_flops += real - pl->net_ops.back().flops;
        pl->net_ops.back().flops = real;
    }
    b.tap("intro", x);
    // latent variant (latent-dehazing/models/modules/DenoisingNAFNet_arch.py:162-176): ending(x + intro output)
    const bool intro_skip = (e->cfg.flags & IRSDE_FLAG_NAF_INTRO_SKIP) != 0;
    const Tensor intro = x;
    std::vector<Tensor> encs;
    for (size_t i = 0; i < e->naf_enc.size(); ++i) {
        for (auto& blk : e->naf_enc[i]) {
            Tensor y = b.nafblock(blk, x);
            if (!(intro_skip && x.p == intro.p)) b.tfree(x);
            x = y;
        }
        b.tap("encoders." + std::to_string(i), x);
        encs.push_back(x);
        Builder::ConvOpts od;
        od.stride = 2;
        x = b.conv_naf(e->naf_downs[i], x, od);  // Conv2d(chan, 2 chan, 2, 2)
        b.tap("downs." + std::to_string(i), x);
    }
    for (auto& blk : e->naf_mid) {
        Tensor y = b.nafblock(blk, x);
        b.tfree(x);
        x = y;
    }
    b.tap("middle", x);
    for (size_t i = 0; i < e->naf_dec.size(); ++i) {
        Tensor skip = encs[encs.size() - 1 - i];
        Builder::ConvOpts ou;
        ou.shuffle = 1; ou.res = &skip;  // Conv2d(chan, 2 chan, 1) -> PixelShuffle(2) -> + enc_skip
        Tensor y = b.conv_naf(e->naf_ups[i], x, ou);
        b.tfree(x);
        b.tfree(skip);
        x = y;
        b.tap("ups." + std::to_string(i), x);
        for (auto& blk : e->naf_dec[i]) {
            Tensor z = b.nafblock(blk, x);
            b.tfree(x);
            x = z;
        }
        b.tap("decoders." + std::to_string(i), x);
    }
    if (intro_skip) {
        Tensor y = b.talloc(x.B, x.H, x.W, x.C);
        const float *xa = x.p, *xb = intro.p;
        float* yo = y.p;
        const size_t n = x.numel();
        b.push_other(OP_OTHER, [=](hipStream_t s) { launch_add(xa, xb, yo, n, s); });
        b.tfree(x);
        b.tfree(intro);
        x = y;
    }
    Builder::ConvOpts oe;
    oe.pad = 1; oe.out_stride = pl->pred_stride;
    Tensor pr = b.conv_naf(e->naf_ending, x, oe);
    b.tfree(x);
    pl->pred = pr.p;
}

Plan* get_plan(irsde_engine* e, int B, int H, int W, bool per_sample_film) {
    for (auto& p : e->plans)
        if (p->B == B && p->H == H && p->W == W && p->per_sample_film == per_sample_film) {
            p->last_use = ++e->use_counter;
            return p.get();
        }
    if (e->plans.size() >= 4) {  // LRU eviction
        size_t lru = 0;
        for (size_t i = 1; i < e->plans.size(); ++i)
            if (e->plans[i]->last_use < e->plans[lru]->last_use) lru = i;
        IRSDE_HIP_CHECK(hipDeviceSynchronize());
        e->plans.erase(e->plans.begin() + lru);
    }
    ensure_film_cur(e, per_sample_film ? B : 1);
    if (naf_lens(e)) {
        if (e->cam_set < B) throw HipError("latent-bokeh ConditionalNAFNet: irsde_set_lens_info must cover the batch first");
        if (e->cam_rows < B) throw HipError("internal: lens table smaller than the batch");
    }

    const int depth = e->cfg.depth, nf = e->cfg.nf, in_nc = e->cfg.in_nc;
    const int sdiv = 1 << depth;
    std::unique_ptr<Plan> plan(new Plan());
    Plan* pl = plan.get();
    pl->B = B; pl->H = H; pl->W = W;
    pl->Hp = (H + sdiv - 1) / sdiv * sdiv;
    pl->Wp = (W + sdiv - 1) / sdiv * sdiv;
    pl->per_sample_film = per_sample_film;
    pl->pred_stride = (e->cfg.out_nc + 3) & ~3;
    pl->last_use = ++e->use_counter;
    // F.pad 'reflect' needs pad < dim (DenoisingUNet_arch.py:82)
    if (e->arch != 1 && (pl->Hp - H >= H || pl->Wp - W >= W)) throw HipError("image too small for reflect padding");

    const size_t img = (size_t)B * in_nc * H * W;
    pl->xin = pl->alloc(img, false);
    pl->cin = pl->alloc(img, false);
    const bool uncond = e->arch == 0 && (e->cfg.flags & IRSDE_FLAG_UNCOND_FULLATTN) != 0;
    const int P = ((uncond ? 1 : 2) * in_nc + 3) & ~3;
    const size_t x0n = (size_t)B * (pl->Hp + 6) * (pl->Wp + 6) * P + 64;
    pl->x0 = pl->alloc(x0n, false);
    IRSDE_HIP_CHECK(hipMemset(pl->x0, 0, x0n * sizeof(float)));

    Builder b{e, pl, (e->cfg.flags & IRSDE_FLAG_KEEP_ACTIVATIONS) == 0, (e->cfg.flags & IRSDE_FLAG_NAIVE_CONV) != 0,
              per_sample_film ? e->film_row : 0};
    {
        const float *xi = pl->xin, *ci = uncond ? nullptr : pl->cin;
        float* x0 = pl->x0;
        const int Hp = pl->Hp, Wp = pl->Wp;
        const int reflect = e->arch == 1 ? 0 : 1;  // NAFNet zero-pads (DenoisingNAFNet_arch.py:189-194)
        b.push_other(OP_OTHER, [=](hipStream_t s) { launch_prep_input(xi, ci, x0, B, in_nc, H, W, Hp, Wp, s, reflect); });
    }
    if (e->arch == 1) {
        build_naf_plan(e, pl, b, P);
        e->plans.push_back(std::move(plan));
        return pl;
    }
    // init_conv 7x7 (DenoisingUNet_arch.py:96) as 7 row taps over the zero-bordered input
    Tensor x;
    {
        ConvParams p;
        p.in0 = pl->x0; p.C0 = e->init_conv.Cin; p.pix0 = P;
        p.Hin = pl->Hp + 6; p.Win = pl->Wp + 6;
        p.w = e->init_conv.w; p.Cout = nf; p.KH = 7; p.KW = 1; p.stride = 1; p.pad_y = 0; p.pad_x = 0;
        p.B = B; p.Ho = pl->Hp; p.Wo = pl->Wp;
        x = b.talloc(B, pl->Hp, pl->Wp, nf);
        p.out = x.p; p.out_stride = nf; p.out_bf16 = x.bf16;  // the prepped input x0 stays fp32
        b.push_conv(p);
        // algorithmic accounting: 7x7 x (2*in_nc) real MACs, not the padded 7 x 64
        const double real = 2.0 * (double)B * pl->Hp * pl->Wp * nf * 49.0 * ((uncond ? 1.0 : 2.0) * in_nc);
        pl->conv_flops += real - pl->net_ops.back().flops;
        pl->net_ops.back().flops = real;  // (exec_flops keeps the padded 7 x 64 K that is actually issued)
    }
    b.tap("init_conv", x);
    Tensor x_init = x;
    std::vector<Tensor> hs;
    for (int i = 0; i < depth; ++i) {
        const std::string d = "downs." + std::to_string(i) + ".";
        Tensor a = b.resblock(e->down_res[2 * i], x, nullptr);
        if (x.p != x_init.p) b.tfree(x);
        b.tap(d + "0", a);
        hs.push_back(a);
        Tensor c = b.resblock(e->down_res[2 * i + 1], a, nullptr);
        b.tap(d + "1", c);
        Tensor g = b.attn(e->down_attn[i], c);
        b.tfree(c);
        b.tap(d + "2", g);
        hs.push_back(g);
        if (i != depth - 1)
            x = b.conv(e->down_conv[i], g, nullptr, 2, 1, 0, nullptr, 0, nullptr);  // Downsample 4x4 s2 p1
        else
            x = b.conv(e->down_conv[i], g, nullptr, 1, 1, 0, nullptr, 0, nullptr);
        b.tap(d + "3", x);
    }
    {
        Tensor a = b.resblock(e->mid1, x, nullptr);
        b.tfree(x);
        b.tap("mid_block1", a);
        Tensor g = b.attn(e->mid_attn, a);
        b.tfree(a);
        b.tap("mid_attn", g);
        x = b.resblock(e->mid2, g, nullptr);
        b.tfree(g);
        b.tap("mid_block2", x);
    }
    for (int j = 0; j < depth; ++j) {
        const std::string u = "ups." + std::to_string(j) + ".";
        Tensor s1 = hs.back(); hs.pop_back();
        Tensor a = b.resblock(e->up_res[2 * j], x, &s1);
        b.tfree(x); b.tfree(s1);
        b.tap(u + "0", a);
        Tensor s2 = hs.back(); hs.pop_back();
        Tensor c = b.resblock(e->up_res[2 * j + 1], a, &s2);
        b.tfree(a); b.tfree(s2);
        b.tap(u + "1", c);
        Tensor g = b.attn(e->up_attn[j], c);
        b.tfree(c);
        b.tap(u + "2", g);
        if (j != depth - 1)
            x = b.conv(e->up_conv[j], g, nullptr, 1, 1, 1, nullptr, 0, nullptr);  // nearest x2 fused into the 3x3
        else
            x = b.conv(e->up_conv[j], g, nullptr, 1, 1, 0, nullptr, 0, nullptr);
        b.tfree(g);
        b.tap(u + "3", x);
    }
    {
        Tensor f = b.resblock(e->final_res, x, &x_init);
        b.tfree(x); b.tfree(x_init);
        b.tap("final_res_block", f);
        Tensor pr = b.conv(e->final_conv, f, nullptr, 1, 1, 0, nullptr, 0, nullptr, pl->pred_stride);
        b.tfree(f);
        pl->pred = pr.p;
    }
    e->plans.push_back(std::move(plan));
    return pl;
}

// UNet.encode / UNet.decode — latent-dehazing/models/modules/UNet_arch.py:59-91
LatentPlan* get_latent_plan(irsde_engine* e, int B, int H, int W, bool decode) {
    for (auto& lp : e->lat_plans)
        if (lp->decode == decode && lp->plan->B == B && lp->plan->H == H && lp->plan->W == W) {
            lp->plan->last_use = ++e->use_counter;
            return lp.get();
        }
    if (e->lat_plans.size() >= 4) {
        size_t lru = 0;
        for (size_t i = 1; i < e->lat_plans.size(); ++i)
            if (e->lat_plans[i]->plan->last_use < e->lat_plans[lru]->plan->last_use) lru = i;
        IRSDE_HIP_CHECK(hipDeviceSynchronize());
        e->lat_plans.erase(e->lat_plans.begin() + lru);
    }
    const int depth = (int)e->lat_mult.size(), ch = e->lat_ch;
    auto dim = [&](int i) { return i == 0 ? ch : ch * e->lat_mult[i - 1]; };
    const int sdiv = 1 << depth;  // check_image_size pads to 2^depth although only depth-1 levels downsample (:52-57)
    std::unique_ptr<LatentPlan> lp(new LatentPlan());
    lp->decode = decode;
    lp->plan.reset(new Plan());
    Plan* pl = lp->plan.get();
    pl->B = B; pl->H = H; pl->W = W;
    pl->Hp = (H + sdiv - 1) / sdiv * sdiv;
    pl->Wp = (W + sdiv - 1) / sdiv * sdiv;
    pl->last_use = ++e->use_counter;
    if (pl->Hp - H >= H || pl->Wp - W >= W) throw HipError("image too small for reflect padding");
    // skips never alias anything else: they cross the encode/decode boundary (decode: they are inputs)
    Builder b{e, pl, false, (e->cfg.flags & IRSDE_FLAG_NAIVE_CONV) != 0, 0};
    // hidden list geometry: h[0] = init_conv output, then two entries per level
    std::vector<std::pair<int, int>> hgeo;  // (level, logical channels)
    hgeo.push_back({0, ch});
    for (int i = 0; i < depth; ++i) {
        hgeo.push_back({i, dim(i)});
        hgeo.push_back({i, dim(i)});
    }
    const int hl = pl->Hp >> (depth - 1), wl = pl->Wp >> (depth - 1);
    if (!decode) {
        lp->image = b.talloc(B, pl->Hp, pl->Wp, rup32(e->lat_in));
        Tensor x = b.conv(e->lat_init, lp->image, nullptr, 1, 1, 0, nullptr, 0, nullptr);
        lp->hidden.push_back(x);
        for (int i = 0; i < depth; ++i) {
            Tensor a = b.resblock(e->lat_enc_res[2 * i], x, nullptr);
            lp->hidden.push_back(a);
            Tensor c = b.resblock(e->lat_enc_res[2 * i + 1], a, nullptr);
            Tensor g = i == depth - 1 ? b.attn(e->lat_enc_attn, c) : c;
            lp->hidden.push_back(g);
            x = i != depth - 1 ? b.conv(e->lat_down[i], g, nullptr, 2, 1, 0, nullptr, 0, nullptr)   // Downsample 4x4 s2 p1
                               : b.conv(e->lat_down[i], g, nullptr, 1, 1, 0, nullptr, 0, nullptr);  // default_conv 3x3
        }
        lp->latent = b.conv(e->lat_latent, x, nullptr, 1, 0, 0, nullptr, 0, nullptr);
    } else {
        lp->latent = b.talloc(B, hl, wl, rup32(e->lat_embed));
        for (auto& g : hgeo) lp->hidden.push_back(b.talloc(B, pl->Hp >> g.first, pl->Wp >> g.first, rup32(g.second)));
        Tensor x = b.conv(e->lat_post, lp->latent, nullptr, 1, 0, 0, nullptr, 0, nullptr);
        const int nh = (int)lp->hidden.size();
        for (int j = 0; j < depth; ++j) {
            Tensor a = b.resblock(e->lat_dec_res[2 * j], x, &lp->hidden[nh - (2 * j + 1)]);
            Tensor c = b.resblock(e->lat_dec_res[2 * j + 1], a, &lp->hidden[nh - (2 * j + 2)]);
            Tensor g = j == 0 ? b.attn(e->lat_dec_attn, c) : c;
            x = j != depth - 1 ? b.conv(e->lat_up[j], g, nullptr, 1, 1, 1, nullptr, 0, nullptr)    // nearest x2 + 3x3 (+bias)
                               : b.conv(e->lat_up[j], g, nullptr, 1, 1, 0, nullptr, 0, nullptr);   // default_conv 3x3
        }
        Tensor y = b.talloc(x.B, x.H, x.W, x.C);
        {
            const float *xa = x.p, *xb = lp->hidden[0].p;
            float* yo = y.p;
            const size_t n = x.numel();
            b.push_other(OP_OTHER, [=](hipStream_t s) { launch_add(xa, xb, yo, n, s); });
        }
        lp->image = b.conv(e->lat_final, y, nullptr, 1, 1, 0, nullptr, 0, nullptr, 4);
    }
    for (auto& g : hgeo) lp->hidden_c.push_back(g.second);
    e->lat_plans.push_back(std::move(lp));
    return e->lat_plans.back().get();
}

hipEvent_t get_event(irsde_engine* e, size_t i) {
    while (e->ev_pool.size() <= i) {
        hipEvent_t ev;
        IRSDE_HIP_CHECK(hipEventCreate(&ev));
        e->ev_pool.push_back(ev);
    }
    return e->ev_pool[i];
}

void run_net(Plan* pl, hipStream_t s) {
    for (auto& op : pl->net_ops) op.fn(s);
}

UpdateParams make_update(irsde_engine* e, Plan* pl) {
    UpdateParams u{};
    u.x = pl->xin; u.mu = pl->cin; u.pred = pl->pred;
    const int64_t ps = pl->pred_stride;
    u.sb = (int64_t)pl->Hp * pl->Wp * ps; u.sc = 1; u.sy = (int64_t)pl->Wp * ps; u.sx = ps;
    u.st = e->step; u.ctl = e->ctl;
    u.B = pl->B; u.C = e->cfg.in_nc; u.H = pl->H; u.W = pl->W;
    return u;
}

void one_step(irsde_engine* e, Plan* pl, hipStream_t s) {
    launch_step_begin(e->step, e->film_table, e->film_row, e->film_cur, e->coef_table, s);
    run_net(pl, s);
    launch_sde_update(make_update(e, pl), s);
}

int guard(const std::function<void()>& f) {
    try {
        f();
        return IRSDE_OK;
    } catch (const HipError& ex) {
        g_last_error = ex.what();
        const std::string m = ex.what();
        if (m.find("weight") != std::string::npos) return IRSDE_ERR_WEIGHT;
        if (m.find(" failed: ") != std::string::npos) return IRSDE_ERR_HIP;
        return IRSDE_ERR_INVALID;
    } catch (const std::exception& ex) {
        g_last_error = ex.what();
        return IRSDE_ERR_INVALID;
    }
}

}  // namespace
}  // namespace irsde

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* irsde_last_error(void) { return g_last_error.c_str(); }
int irsde_version(void) { return 100; }

int irsde_create(const irsde_config* cfg, irsde_engine** out) {
    return guard([&] {
        if (!cfg || !out) throw HipError("null argument");
        if (cfg->nf % 32 || cfg->nf < 32) throw HipError("nf must be a positive multiple of 32");
        if (cfg->depth < 1 || cfg->depth > 6) throw HipError("depth out of range");
        if (cfg->in_nc < 1 || cfg->in_nc > 4 || cfg->out_nc < 1 || cfg->out_nc > 4)
            throw HipError("in_nc/out_nc must be in 1..4");
        if (cfg->in_nc != cfg->out_nc) throw HipError("sampler needs in_nc == out_nc");
        if ((cfg->nf << cfg->depth) > 2048) throw HipError("nf * 2^depth must be <= 2048");
        auto* e = new irsde_engine();
        e->cfg = *cfg;
        if (cfg->flags & IRSDE_FLAG_BF16_ACT) {
            if (cfg->flags & (IRSDE_FLAG_UNCOND_FULLATTN | IRSDE_FLAG_NAIVE_CONV)) {
                delete e;
                throw HipError("IRSDE_FLAG_BF16_ACT: only the conditional UNet on the MFMA kernels stores bf16 activations");
            }
            e->cfg.flags |= IRSDE_FLAG_BF16;
        }
        e->time_dim = cfg->nf * 4;
        build_inventory(e);
        *out = e;
    });
}

int irsde_create_nafnet(const irsde_nafnet_config* cfg, irsde_engine** out) {
    return guard([&] {
        if (!cfg || !out) throw HipError("null argument");
        if (cfg->width % 32 || cfg->width < 32) throw HipError("width must be a positive multiple of 32");
        if (cfg->img_channel < 1 || cfg->img_channel > 8) throw HipError("img_channel must be in 1..8");
        if (cfg->n_enc < 1 || cfg->n_enc > 6 || cfg->n_dec != cfg->n_enc) throw HipError("need 1..6 encoder stages and as many decoder stages");
        if ((cfg->width << cfg->n_enc) > 2048) throw HipError("width * 2^stages must be <= 2048");
        if (cfg->flags & IRSDE_FLAG_BF16_ACT) throw HipError("IRSDE_FLAG_BF16_ACT: conditional UNet only");
        auto* e = new irsde_engine();
        e->arch = 1;
        e->cfg.in_nc = e->cfg.out_nc = cfg->img_channel;
        e->cfg.nf = cfg->width;
        e->cfg.depth = cfg->n_enc;  // pad multiple 2^stages (padder_size, DenoisingNAFNet_arch.py:147)
        e->cfg.device = cfg->device;
        e->cfg.flags = cfg->flags;
        e->time_dim = cfg->width * 4;
        for (int i = 0; i < cfg->n_enc; ++i) {
            if (cfg->enc_blk_nums[i] < 0 || cfg->dec_blk_nums[i] < 0) throw HipError("negative block count");
            e->naf_enc_nums.push_back(cfg->enc_blk_nums[i]);
            e->naf_dec_nums.push_back(cfg->dec_blk_nums[i]);
        }
        e->naf_mid_num = cfg->middle_blk_num;
        build_inventory_naf(e);
        *out = e;
    });
}

void irsde_destroy(irsde_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    e->plans.clear();
    e->lat_plans.clear();
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->ev_out) (void)hipEventDestroy(e->ev_out);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    for (float* p : e->dev_allocs) (void)hipFree(p);
    for (auto& kv : e->bf16_copies) (void)hipFree(kv.second);
    if (e->coef_table) (void)hipFree(e->coef_table);
    if (e->film_table) (void)hipFree(e->film_table);
    delete e;
}

int irsde_num_weights(const irsde_engine* e) { return e ? (int)e->names.size() : 0; }
const char* irsde_weight_name(const irsde_engine* e, int i) {
    if (!e || i < 0 || i >= (int)e->names.size()) return nullptr;
    return e->names[i].c_str();
}
int irsde_weight_shape(const irsde_engine* e, int i, int64_t shape[4], int* ndim) {
    if (!e || i < 0 || i >= (int)e->names.size()) return IRSDE_ERR_INVALID;
    const auto& t = e->host.at(e->names[i]);
    *ndim = (int)t.shape.size();
    for (size_t k = 0; k < t.shape.size(); ++k) shape[k] = t.shape[k];
    return IRSDE_OK;
}

int irsde_load_weight(irsde_engine* e, const char* name, const float* data, const int64_t* shape, int ndim) {
    return guard([&] {
        if (!e || !name || !data) throw HipError("null argument");
        if (e->finalized) throw HipError("weights already finalized: create a new engine to reload weights");
        auto it = e->host.find(name);
        if (it == e->host.end()) throw HipError(std::string("unknown weight name: ") + name);
        HostTensor& t = it->second;
        size_t n = 1;
        bool same = ndim == (int)t.shape.size();
        for (int k = 0; k < ndim; ++k) {
            n *= (size_t)shape[k];
            if (same && shape[k] != t.shape[k]) same = false;
        }
        if (!same) throw HipError(std::string("weight shape mismatch for ") + name);
        t.data.assign(data, data + n);
        t.loaded = true;
    });
}

int irsde_finalize_weights(irsde_engine* e) {
    return guard([&] {
        if (!e) throw HipError("null engine");
        if (e->finalized) return;
        finalize(e);
    });
}

int irsde_set_schedule(irsde_engine* e, int T, const float* coef) {
    return guard([&] {
        if (!e || !coef || T < 1) throw HipError("bad schedule arguments");
        if (!e->finalized) throw HipError("set_schedule: weights not finalized");
        IRSDE_HIP_CHECK(hipSetDevice(e->cfg.device));
        IRSDE_HIP_CHECK(hipDeviceSynchronize());
        std::lock_guard<std::mutex> lk(e->mu);
        // captured graphs bake the table pointers into their kernel nodes: drop them with the old tables
        for (auto& pl : e->plans) {
            if (pl->graph_exec) (void)hipGraphExecDestroy(pl->graph_exec);
            if (pl->graph) (void)hipGraphDestroy(pl->graph);
            pl->graph_exec = nullptr;
            pl->graph = nullptr;
        }
        if (e->coef_table) (void)hipFree(e->coef_table);
        if (e->film_table) (void)hipFree(e->film_table);
        e->coef_table = e->film_table = nullptr;
        e->T = T;
        IRSDE_HIP_CHECK(hipMalloc(&e->coef_table, (size_t)(T + 1) * IRSDE_COEF_STRIDE * 4));
        IRSDE_HIP_CHECK(hipMemcpy(e->coef_table, coef, (size_t)(T + 1) * IRSDE_COEF_STRIDE * 4, hipMemcpyHostToDevice));
        IRSDE_HIP_CHECK(hipMalloc(&e->film_table, (size_t)(T + 1) * e->film_row * 4));
        std::vector<float> tv(T + 1);
        for (int t = 0; t <= T; ++t) tv[t] = (float)t;
        float* dtv = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&dtv, (T + 1) * sizeof(float)));
        IRSDE_HIP_CHECK(hipMemcpy(dtv, tv.data(), (T + 1) * sizeof(float), hipMemcpyHostToDevice));
        compute_film_rows(e, dtv, T + 1, e->film_table, e->stream);
        (void)hipFree(dtv);
    });
}

int irsde_unet_forward(irsde_engine* e, const float* xt, const float* cond, const int64_t* t_host, int nt, int B, int H,
                       int W, float* out, void* stream) {
    return guard([&] {
        const bool uncond_e = e && e->arch == 0 && (e->cfg.flags & IRSDE_FLAG_UNCOND_FULLATTN);
        if (!e || !xt || (!cond && !uncond_e) || !t_host || !out) throw HipError("null argument");
        if (!e->finalized) throw HipError("unet_forward: weights not finalized");
        if (nt != 1 && nt != B) throw HipError("unet_forward: need 1 or B timesteps");
        if (B < 1 || H < 2 || W < 2) throw HipError("unet_forward: bad shape");
        std::lock_guard<std::mutex> lk(e->mu);
        IRSDE_HIP_CHECK(hipSetDevice(e->cfg.device));
        hipStream_t user = reinterpret_cast<hipStream_t>(stream);
        const bool per_sample = nt > 1;
        Plan* pl = get_plan(e, B, H, W, per_sample);
        hipStream_t s = e->stream;
        IRSDE_HIP_CHECK(hipEventRecord(e->ev_in, user));
        IRSDE_HIP_CHECK(hipStreamWaitEvent(s, e->ev_in, 0));
        const size_t img = (size_t)B * e->cfg.in_nc * H * W * sizeof(float);
        IRSDE_HIP_CHECK(hipMemcpyAsync(pl->xin, xt, img, hipMemcpyDeviceToDevice, s));
        if (cond) IRSDE_HIP_CHECK(hipMemcpyAsync(pl->cin, cond, img, hipMemcpyDeviceToDevice, s));
        const bool in_table = nt == 1 && e->film_table && t_host[0] >= 0 && t_host[0] <= e->T;
        if (in_table) {
            IRSDE_HIP_CHECK(hipMemcpyAsync(e->film_cur, e->film_table + (size_t)t_host[0] * e->film_row,
                                           (size_t)e->film_row * 4, hipMemcpyDeviceToDevice, s));
        } else {
            std::vector<float> tv(nt);
            for (int i = 0; i < nt; ++i) tv[i] = (float)t_host[i];
            float* dtv = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&dtv, nt * sizeof(float)));
            IRSDE_HIP_CHECK(hipMemcpy(dtv, tv.data(), nt * sizeof(float), hipMemcpyHostToDevice));
            compute_film_rows(e, dtv, nt, e->film_cur, s);
            (void)hipFree(dtv);
        }
        run_net(pl, s);
        launch_unpack_pred(pl->pred, out, B, e->cfg.out_nc, H, W, pl->Hp, pl->Wp, pl->pred_stride, s);
        IRSDE_HIP_CHECK(hipEventRecord(e->ev_out, s));
        IRSDE_HIP_CHECK(hipStreamWaitEvent(user, e->ev_out, 0));
    });
}

int irsde_sample(irsde_engine* e, int mode, const float* xT, const float* mu, const float* noise, uint64_t seed,
                 uint64_t image_offset, int B, int H, int W, int T, int t_stop, float* out, void* stream,
                 int flags) {
    return guard([&] {
        if (!e || !xT || !out) throw HipError("null argument");
        if (!e->finalized || !e->film_table) throw HipError("sample: weights/schedule not set");
        if (mode < 0 || mode > 4) throw HipError("sample: bad mode");
        const bool uncond_e = e->arch == 0 && (e->cfg.flags & IRSDE_FLAG_UNCOND_FULLATTN);
        if ((mode >= 3) != uncond_e) throw HipError("sample: DenoisingSDE modes (3,4) go with the unconditional network and vice versa");
        if (!mu && !uncond_e) throw HipError("null argument");
        if (T <= 0) T = e->T;
        if (T > e->T) throw HipError("sample: T exceeds the schedule length");
        if (t_stop < 0 || t_stop >= T) throw HipError("sample: t_stop must be in [0, T)");
        const int nsteps = T - t_stop;
        if (B < 1 || H < 2 || W < 2) throw HipError("sample: bad shape");
        std::lock_guard<std::mutex> lk(e->mu);
        IRSDE_HIP_CHECK(hipSetDevice(e->cfg.device));
        hipStream_t user = reinterpret_cast<hipStream_t>(stream);
        Plan* pl = get_plan(e, B, H, W, false);
        hipStream_t s = e->stream;
        const bool profile = (flags & IRSDE_SAMPLE_PROFILE) != 0;
        const bool graph = (flags & IRSDE_SAMPLE_GRAPH) != 0 && !profile;
        IRSDE_HIP_CHECK(hipEventRecord(e->ev_in, user));
        IRSDE_HIP_CHECK(hipStreamWaitEvent(s, e->ev_in, 0));
        const size_t img = (size_t)B * e->cfg.in_nc * H * W;
        IRSDE_HIP_CHECK(hipMemcpyAsync(pl->xin, xT, img * 4, hipMemcpyDeviceToDevice, s));
        if (mu) IRSDE_HIP_CHECK(hipMemcpyAsync(pl->cin, mu, img * 4, hipMemcpyDeviceToDevice, s));
        launch_set_ctl(e->ctl, mode, noise, (long long)img, seed, image_offset, s);
        launch_set_step(e->step, T, s);

        if (graph) {
            if (!pl->graph_exec) {
                IRSDE_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                try {
                    one_step(e, pl, s);
                } catch (...) {
                    hipGraph_t g = nullptr;
                    (void)hipStreamEndCapture(s, &g);
                    if (g) (void)hipGraphDestroy(g);
                    throw;
                }
                IRSDE_HIP_CHECK(hipStreamEndCapture(s, &pl->graph));
                IRSDE_HIP_CHECK(hipGraphInstantiate(&pl->graph_exec, pl->graph, nullptr, nullptr, 0));
            }
            for (int i = 0; i < nsteps; ++i) IRSDE_HIP_CHECK(hipGraphLaunch(pl->graph_exec, s));
        } else if (!profile) {
            for (int i = 0; i < nsteps; ++i) one_step(e, pl, s);
        } else {
            // eager with an event before every kernel group; interval k..k+1 belongs to group k
            std::vector<int> kinds;
            size_t ei = 0;
            auto mark = [&](int kind) {
                IRSDE_HIP_CHECK(hipEventRecord(get_event(e, ei++), s));
                kinds.push_back(kind);
            };
            for (int i = 0; i < nsteps; ++i) {
                mark(OP_OTHER);
                launch_step_begin(e->step, e->film_table, e->film_row, e->film_cur, e->coef_table, s);
                for (auto& op : pl->net_ops) {
                    mark(op.kind);
                    op.fn(s);
                }
                mark(OP_OTHER);
                launch_sde_update(make_update(e, pl), s);
            }
            IRSDE_HIP_CHECK(hipEventRecord(get_event(e, ei++), s));
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            double ms[OP_NKINDS] = {0, 0, 0, 0, 0};
            const size_t per_step = pl->net_ops.size() + 2;
            e->op_ms.assign(pl->net_ops.size(), 0.0);
            e->op_desc.clear();
            for (auto& op : pl->net_ops) e->op_desc.push_back(op.desc);
            e->op_steps = nsteps;
            for (size_t k = 0; k < kinds.size(); ++k) {
                float t;
                IRSDE_HIP_CHECK(hipEventElapsedTime(&t, e->ev_pool[k], e->ev_pool[k + 1]));
                ms[kinds[k]] += t;
                const size_t in_step = k % per_step;
                if (in_step >= 1 && in_step <= pl->net_ops.size()) e->op_ms[in_step - 1] += t;
            }
            hipEvent_t e0 = e->ev_pool[0], e1 = e->ev_pool[kinds.size()];
            float wall;
            IRSDE_HIP_CHECK(hipEventElapsedTime(&wall, e0, e1));
            size_t nconv = 0;
            for (auto& op : pl->net_ops) nconv += op.kind == OP_CONV;
            e->profile[0] = ms[OP_CONV];
            e->profile[1] = pl->conv_flops * nsteps;
            e->profile[2] = (double)nconv * nsteps;
            e->profile[3] = pl->conv_bytes * nsteps;
            e->profile[4] = ms[OP_LN];
            e->profile[5] = ms[OP_ATTN];
            e->profile[6] = ms[OP_OTHER];
            e->profile[7] = wall;
            e->profile[8] = nsteps;
            e->profile[9] = ms[OP_WINO];
            e->profile[10] = pl->conv_exec_flops * nsteps;
            e->profile[11] = 0;
        }
        IRSDE_HIP_CHECK(hipMemcpyAsync(out, pl->xin, img * 4, hipMemcpyDeviceToDevice, s));
        IRSDE_HIP_CHECK(hipEventRecord(e->ev_out, s));
        IRSDE_HIP_CHECK(hipStreamWaitEvent(user, e->ev_out, 0));
    });
}

int irsde_sde_step(int mode, int t, const float* coef_row, float* x, const float* mu, const float* eps_hat,
                   const float* noise_t, uint64_t seed, uint64_t image_offset, int B, int C, int H, int W,
                   void* stream) {
    return guard([&] {
        if (!coef_row || !x || !eps_hat || (!mu && mode < 3)) throw HipError("null argument");
        if (!mu) mu = x;  // DenoisingSDE has no mu term; the kernel still reads the pointer
        if (mode < 0 || mode > 4) throw HipError("sde_step: bad mode");
        UpdateParams u{};
        u.x = x; u.mu = mu; u.pred = eps_hat;
        u.sb = (int64_t)C * H * W; u.sc = (int64_t)H * W; u.sy = W; u.sx = 1;
        // noise_t is the draw for this step: index it with tstride 0
        u.noise = noise_t; u.noise_tstride = 0;
        u.st = nullptr; u.ctl = nullptr; u.t_imm = t;
        for (int i = 0; i < IRSDE_COEF_STRIDE; ++i) u.coef_imm[i] = coef_row[i];
        u.mode = mode; u.B = B; u.C = C; u.H = H; u.W = W; u.seed = seed; u.image_offset = image_offset;
        launch_sde_update(u, reinterpret_cast<hipStream_t>(stream));
    });
}

int irsde_philox_normal(float* out, int B, int CHW, int t, uint64_t seed, uint64_t image_offset, void* stream) {
    return guard([&] {
        if (!out) throw HipError("null argument");
        launch_philox_normal(out, B, CHW, t, seed, image_offset, reinterpret_cast<hipStream_t>(stream));
    });
}

int irsde_get_profile(const irsde_engine* e, double out[12]) {
    if (!e || !out) return IRSDE_ERR_INVALID;
    for (int i = 0; i < 12; ++i) out[i] = e->profile[i];
    return IRSDE_OK;
}

int irsde_debug_tap(irsde_engine* e, const char* name, float* dst, int64_t dims[4]) {
    return guard([&] {
        if (!e || !name || !dims) throw HipError("null argument");
        if (!(e->cfg.flags & IRSDE_FLAG_KEEP_ACTIVATIONS)) throw HipError("debug_tap needs IRSDE_FLAG_KEEP_ACTIVATIONS");
        Plan* pl = nullptr;
        for (auto& p : e->plans)
            if (!pl || p->last_use > pl->last_use) pl = p.get();
        if (!pl) throw HipError("debug_tap: no forward has run");
        auto it = pl->taps.find(name);
        if (it == pl->taps.end()) throw HipError(std::string("debug_tap: unknown tap ") + name);
        const Tensor& t = it->second;
        dims[0] = t.B; dims[1] = t.C; dims[2] = t.H; dims[3] = t.W;
        if (!dst) return;
        IRSDE_HIP_CHECK(hipSetDevice(e->cfg.device));
        IRSDE_HIP_CHECK(hipDeviceSynchronize());
        float* tmp = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&tmp, t.numel() * 4));
        launch_nhwc_to_nchw(t.p, tmp, t.B, t.C, t.H, t.W, e->stream, t.bf16);
        IRSDE_HIP_CHECK(hipStreamSynchronize(e->stream));
        IRSDE_HIP_CHECK(hipMemcpy(dst, tmp, t.numel() * 4, hipMemcpyDeviceToHost));
        (void)hipFree(tmp);
    });
}

int irsde_op_profile(irsde_engine* e, char* buf, int buflen) {
    return guard([&] {
        if (!e || !buf || buflen < 1) throw HipError("null argument");
        std::string out;
        char line[512];
        for (size_t i = 0; i < e->op_ms.size(); ++i) {
            snprintf(line, sizeof line, "%9.4f ms  %s\n", e->op_ms[i] / std::max(e->op_steps, 1), e->op_desc[i].c_str());
            out += line;
        }
        strncpy(buf, out.c_str(), buflen - 1);
        buf[buflen - 1] = 0;
    });
}

int irsde_plan_describe(irsde_engine* e, int B, int H, int W, char* buf, int buflen) {
    return guard([&] {
        if (!e || !buf || buflen < 1) throw HipError("null argument");
        if (!e->finalized) throw HipError("plan_describe: weights not finalized");
        std::lock_guard<std::mutex> lk(e->mu);
        IRSDE_HIP_CHECK(hipSetDevice(e->cfg.device));
        Plan* pl = get_plan(e, B, H, W, false);
        std::string out;
        for (auto& op : pl->net_ops) out += op.desc + "\n";
        strncpy(buf, out.c_str(), buflen - 1);
        buf[buflen - 1] = 0;
    });
}

int irsde_work_model(irsde_engine* e, int B, int H, int W, double out[2]) {
    return guard([&] {
        if (!e || !out) throw HipError("null argument");
        if (!e->finalized) throw HipError("work_model: weights not finalized");
        std::lock_guard<std::mutex> lk(e->mu);
        IRSDE_HIP_CHECK(hipSetDevice(e->cfg.device));
        Plan* pl = get_plan(e, B, H, W, false);
        out[0] = pl->conv_flops;
        out[1] = pl->conv_bytes;
    });
}

int irsde_debug_conv(const float* in0, int C0, const float* in1, int C1, int B, int Hin, int Win, int in_shift,
                     const float* w_oihw, int Cout, int KH, int KW, int stride, int pad, const float* bias,
                     const float* film, int film_bstride, int silu, const float* res, float* out, int naive,
                     int splits, void* stream) {
    return guard([&] {
        if (!in0 || !w_oihw || !out) throw HipError("null argument");
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        conv_global_init();
        const int Cin = C0 + C1;
        std::vector<float> pk((size_t)Cout * KH * KW * Cin);
        for (int o = 0; o < Cout; ++o)
            for (int i = 0; i < Cin; ++i)
                for (int ky = 0; ky < KH; ++ky)
                    for (int kx = 0; kx < KW; ++kx)
                        pk[(((size_t)o * KH + ky) * KW + kx) * Cin + i] = w_oihw[(((size_t)o * Cin + i) * KH + ky) * KW + kx];
        float *dw = nullptr, *db = nullptr, *dp = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&dw, pk.size() * 4));
        IRSDE_HIP_CHECK(hipMemcpy(dw, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        if (bias) {
            IRSDE_HIP_CHECK(hipMalloc(&db, Cout * 4));
            IRSDE_HIP_CHECK(hipMemcpy(db, bias, Cout * 4, hipMemcpyHostToDevice));
        }
        ConvParams p;
        p.in0 = in0; p.C0 = C0; p.pix0 = C0; p.in1 = in1; p.C1 = C1; p.pix1 = C1;
        p.Hin = Hin; p.Win = Win; p.in_shift = in_shift;
        p.w = dw; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad_y = pad; p.pad_x = pad;
        p.B = B;
        p.Ho = ((Hin << in_shift) + 2 * pad - KH) / stride + 1;
        p.Wo = ((Win << in_shift) + 2 * pad - KW) / stride + 1;
        p.out = out; p.out_stride = Cout; p.bias = db; p.film = film; p.film_bstride = film_bstride; p.silu = silu;
        p.res = res; p.res_stride = Cout;
        float* dz = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&dz, 1024));
        IRSDE_HIP_CHECK(hipMemset(dz, 0, 1024));
        p.zeros = dz;
        const int wino_tile = (naive == 2 || naive == 12 || naive == 22) ? 2 : (naive == 3 || naive == 13 || naive == 23) ? 4 : 0;
        if (splits > 1 && naive != 1 && !wino_tile) {
            p.splits = splits;
            IRSDE_HIP_CHECK(hipMalloc(&dp, (size_t)splits * B * p.Ho * p.Wo * Cout * 4));
            p.partial = dp;
        }
        if (wino_tile) {  // naive / 10: 0 = production dispatch, 1 / 2 = force the batch-loop GEMM kernel (all / 2 components per block)
            const int tile = wino_tile, ncomp = (tile + 2) * (tile + 2);
            if (!wino_shape_ok(p, tile)) throw HipError("debug_conv: shape not eligible for Winograd");
            std::vector<float> U((size_t)ncomp * Cout * Cin);
            wino_transform_weights(pk.data(), Cout, Cin, U.data(), tile);
            const long long T = (long long)B * (p.Ho / tile) * (p.Wo / tile);
            float *dU = nullptr, *dV = nullptr, *dM = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&dU, U.size() * 4));
            IRSDE_HIP_CHECK(hipMemcpy(dU, U.data(), U.size() * 4, hipMemcpyHostToDevice));
            IRSDE_HIP_CHECK(hipMalloc(&dV, (size_t)ncomp * T * Cin * 4));
            IRSDE_HIP_CHECK(hipMalloc(&dM, (size_t)ncomp * T * Cout * 4));
            const WinoPlan wp = make_wino(p, dU, dV, dM, tile);
            launch_wino_input(wp.in, s);
            conv_set_variant(naive >= 20 ? 72 : naive >= 10 ? 71 : 0);
            launch_conv(wp.gemm, s);
            conv_set_variant(0);
            launch_wino_output(wp.out, s);
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            (void)hipFree(dU); (void)hipFree(dV); (void)hipFree(dM);
        } else if (naive == 1) {
            launch_conv_naive(p, s);
        } else {
            unsigned short *dbf = nullptr, *a0 = nullptr, *a1 = nullptr, *ar = nullptr, *ao = nullptr;
            const bool act = naive == 204 || naive == 260 || naive == 261;  // + bf16 activation storage (IRSDE_FLAG_BF16_ACT)
            if (naive == 4 || naive == 160 || naive == 161 || act) {  // bf16-MFMA mode (variants 60 / 61: force the 256 / 128 tile)
                IRSDE_HIP_CHECK(hipMalloc(&dbf, pk.size() * 2));
                launch_f32_to_bf16(dw, dbf, pk.size(), s);
                p.w_bf = dbf;
            }
            const size_t npix_in = (size_t)B * Hin * Win, nout = (size_t)B * p.Ho * p.Wo * Cout;
            if (act) {  // the caller's fp32 tensors are rounded into bf16 copies; the bf16 result is widened back
                auto to_bf = [&](const float* src, size_t n) {
                    unsigned short* d = nullptr;
                    IRSDE_HIP_CHECK(hipMalloc(&d, n * 2 + 64));
                    launch_f32_to_bf16(src, d, n, s);
                    return d;
                };
                a0 = to_bf(in0, npix_in * C0);
                p.in0 = reinterpret_cast<const float*>(a0);
                if (in1) { a1 = to_bf(in1, npix_in * C1); p.in1 = reinterpret_cast<const float*>(a1); }
                if (res) { ar = to_bf(res, nout); p.res = reinterpret_cast<const float*>(ar); }
                IRSDE_HIP_CHECK(hipMalloc(&ao, nout * 2 + 64));
                p.out = reinterpret_cast<float*>(ao);
                p.in_bf16 = p.out_bf16 = 1;
            }
            conv_set_variant(act ? (naive == 204 ? 0 : naive - 200) : (naive >= 100 ? naive - 100 : 0));  // tile variants
            launch_conv(p, s);
            conv_set_variant(0);
            if (act) launch_bf16_to_f32(ao, out, nout, s);
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            for (unsigned short* q : {dbf, a0, a1, ar, ao})
                if (q) (void)hipFree(q);
        }
        IRSDE_HIP_CHECK(hipStreamSynchronize(s));
        (void)hipFree(dw);
        (void)hipFree(dz);
        if (db) (void)hipFree(db);
        if (dp) (void)hipFree(dp);
    });
}

int irsde_bench_conv(int variant, int B, int H, int W, int Cin, int Cout, int K, int stride, int up, int epi, int iters,
                     double* ms_out) {
    return guard([&] {
        if (!ms_out || iters < 1) throw HipError("bad argument");
        conv_global_init();
        hipStream_t s = nullptr;
        IRSDE_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        ConvParams p;
        const int pad = K / 2 - (stride == 2 ? 1 : 0) + (K == 4 ? 0 : 0);
        p.B = B; p.Hin = H; p.Win = W; p.in_shift = up; p.C0 = Cin; p.pix0 = Cin;
        p.Cout = Cout; p.KH = K; p.KW = K; p.stride = stride; p.pad_y = p.pad_x = (K == 4 ? 1 : K / 2);
        (void)pad;
        p.Ho = ((H << up) + 2 * p.pad_y - K) / stride + 1;
        p.Wo = ((W << up) + 2 * p.pad_x - K) / stride + 1;
        const size_t nin = (size_t)B * H * W * Cin, nw = (size_t)Cout * K * K * Cin, nout = (size_t)B * p.Ho * p.Wo * Cout;
        float *din = nullptr, *dw = nullptr, *dout = nullptr, *dres = nullptr, *dfilm = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&din, nin * 4));
        IRSDE_HIP_CHECK(hipMalloc(&dw, nw * 4));
        IRSDE_HIP_CHECK(hipMalloc(&dout, nout * 4));
        IRSDE_HIP_CHECK(hipMalloc(&dres, nout * 4));
        IRSDE_HIP_CHECK(hipMalloc(&dfilm, (size_t)2 * Cout * 4 + 1024));
        float* dz = dfilm + 2 * Cout;
        IRSDE_HIP_CHECK(hipMemset(dz, 0, 1024));
        p.zeros = dz;
        launch_fill_random(din, nin, 1, 1.0f, s);
        launch_fill_random(dw, nw, 2, 1.0f / sqrtf((float)(K * K * Cin)), s);
        launch_fill_random(dres, nout, 3, 1.0f, s);
        launch_fill_random(dfilm, (size_t)2 * Cout, 4, 0.3f, s);
        p.in0 = din; p.w = dw; p.out = dout; p.out_stride = Cout;
        unsigned short* dbf = nullptr;
        if (variant >= 60 && variant <= 62) {  // bf16-MFMA mode: 60 = 256x256 tile, 61 = 128x128, 62 = automatic
            IRSDE_HIP_CHECK(hipMalloc(&dbf, nw * 2));
            launch_f32_to_bf16(dw, dbf, nw, s);
            p.w_bf = dbf;
        }
        if (epi == 1) { p.film = dfilm; p.silu = 1; }
        if (epi == 2) { p.silu = 1; p.res = dres; p.res_stride = Cout; }
        conv_set_variant(variant);
        hipEvent_t e0, e1;
        IRSDE_HIP_CHECK(hipEventCreate(&e0));
        IRSDE_HIP_CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 2; ++i) launch_conv(p, s);
        IRSDE_HIP_CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) launch_conv(p, s);
        IRSDE_HIP_CHECK(hipEventRecord(e1, s));
        IRSDE_HIP_CHECK(hipStreamSynchronize(s));
        float ms = 0;
        IRSDE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = ms / iters;
        conv_set_variant(0);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        (void)hipFree(din); (void)hipFree(dw); (void)hipFree(dout); (void)hipFree(dres); (void)hipFree(dfilm);
        if (dbf) (void)hipFree(dbf);
        (void)hipStreamDestroy(s);
    });
}

int irsde_create_latent_unet(const irsde_latent_unet_config* cfg, irsde_engine** out) {
    return guard([&] {
        if (!cfg || !out) throw HipError("null argument");
        if (cfg->in_ch < 1 || cfg->in_ch > 32 || cfg->out_ch < 1 || cfg->out_ch > 4) throw HipError("in_ch must be in 1..32 and out_ch in 1..4");
        if (cfg->ch < 1 || cfg->n_mult < 1 || cfg->n_mult > 6) throw HipError("ch / ch_mult out of range");
        if (cfg->embed_dim < 1 || cfg->embed_dim > 32) throw HipError("embed_dim must be in 1..32");
        if (cfg->flags & IRSDE_FLAG_BF16_ACT) throw HipError("IRSDE_FLAG_BF16_ACT: conditional UNet only");
        auto* e = new irsde_engine();
        e->arch = 2;
        e->cfg.in_nc = cfg->in_ch; e->cfg.out_nc = cfg->out_ch; e->cfg.nf = cfg->ch; e->cfg.depth = cfg->n_mult;
        e->cfg.device = cfg->device; e->cfg.flags = cfg->flags;
        e->lat_in = cfg->in_ch; e->lat_out = cfg->out_ch; e->lat_ch = cfg->ch; e->lat_embed = cfg->embed_dim;
        for (int i = 0; i < cfg->n_mult; ++i) {
            if (cfg->ch_mult[i] < 1 || cfg->ch * cfg->ch_mult[i] > 2048) throw HipError("ch * ch_mult out of range");
            e->lat_mult.push_back(cfg->ch_mult[i]);
        }
        build_inventory_latent(e);
        *out = e;
    });
}

int irsde_latent_shapes(irsde_engine* e, int H, int W, int64_t latent_chw[3], int64_t* hidden_chw, int* n_hidden) {
    return guard([&] {
        if (!e || e->arch != 2 || !latent_chw || !n_hidden) throw HipError("latent_shapes: not a latent UNet engine / null argument");
        const int depth = (int)e->lat_mult.size(), sdiv = 1 << depth;
        const int Hp = (H + sdiv - 1) / sdiv * sdiv, Wp = (W + sdiv - 1) / sdiv * sdiv;
        latent_chw[0] = e->lat_embed; latent_chw[1] = Hp >> (depth - 1); latent_chw[2] = Wp >> (depth - 1);
        *n_hidden = 2 * depth + 1;
        if (hidden_chw) {
            auto dim = [&](int i) { return i == 0 ? e->lat_ch : e->lat_ch * e->lat_mult[i - 1]; };
            for (int k = 0; k < 2 * depth + 1; ++k) {
                const int lvl = k == 0 ? 0 : (k - 1) / 2;
                hidden_chw[3 * k] = dim(lvl); hidden_chw[3 * k + 1] = Hp >> lvl; hidden_chw[3 * k + 2] = Wp >> lvl;
            }
        }
    });
}

int irsde_latent_encode(irsde_engine* e, const float* x, int B, int H, int W, float* latent, float* const* hidden,
                        void* stream) {
    return guard([&] {
        if (!e || e->arch != 2 || !x || !latent || !hidden) throw HipError("latent_encode: not a latent UNet engine / null argument");
        if (!e->finalized) throw HipError("latent_encode: weights not finalized");
        if (B < 1 || H < 2 || W < 2) throw HipError("latent_encode: bad shape");
        std::lock_guard<std::mutex> lk(e->mu);
        IRSDE_HIP_CHECK(hipSetDevice(e->cfg.device));
        hipStream_t user = reinterpret_cast<hipStream_t>(stream), s = e->stream;
        LatentPlan* lp = get_latent_plan(e, B, H, W, false);
        Plan* pl = lp->plan.get();
        IRSDE_HIP_CHECK(hipEventRecord(e->ev_in, user));
        IRSDE_HIP_CHECK(hipStreamWaitEvent(s, e->ev_in, 0));
        launch_nchw_to_nhwc_pad(x, lp->image.p, B, e->lat_in, H, W, pl->Hp, pl->Wp, lp->image.C, 1, s);  // F.pad 'reflect'
        run_net(pl, s);
        const Tensor& L = lp->latent;
        launch_unpack_pred(L.p, latent, B, e->lat_embed, L.H, L.W, L.H, L.W, L.C, s);
        for (size_t k = 0; k < lp->hidden.size(); ++k) {
            const Tensor& h = lp->hidden[k];
            if (!hidden[k]) throw HipError("latent_encode: null hidden pointer");
            launch_unpack_pred(h.p, hidden[k], B, lp->hidden_c[k], h.H, h.W, h.H, h.W, h.C, s);
        }
        IRSDE_HIP_CHECK(hipEventRecord(e->ev_out, s));
        IRSDE_HIP_CHECK(hipStreamWaitEvent(user, e->ev_out, 0));
    });
}

int irsde_latent_decode(irsde_engine* e, const float* latent, const float* const* hidden, int B, int H, int W, float* out,
                        void* stream) {
    return guard([&] {
        if (!e || e->arch != 2 || !latent || !hidden || !out) throw HipError("latent_decode: not a latent UNet engine / null argument");
        if (!e->finalized) throw HipError("latent_decode: weights not finalized");
        std::lock_guard<std::mutex> lk(e->mu);
        IRSDE_HIP_CHECK(hipSetDevice(e->cfg.device));
        hipStream_t user = reinterpret_cast<hipStream_t>(stream), s = e->stream;
        LatentPlan* lp = get_latent_plan(e, B, H, W, true);
        Plan* pl = lp->plan.get();
        IRSDE_HIP_CHECK(hipEventRecord(e->ev_in, user));
        IRSDE_HIP_CHECK(hipStreamWaitEvent(s, e->ev_in, 0));
        const Tensor& L = lp->latent;
        launch_nchw_to_nhwc_pad(latent, L.p, B, e->lat_embed, L.H, L.W, L.H, L.W, L.C, 0, s);
        for (size_t k = 0; k < lp->hidden.size(); ++k) {
            const Tensor& h = lp->hidden[k];
            if (!hidden[k]) throw HipError("latent_decode: null hidden pointer");
            launch_nchw_to_nhwc_pad(hidden[k], h.p, B, lp->hidden_c[k], h.H, h.W, h.H, h.W, h.C, 0, s);
        }
        run_net(pl, s);
        launch_unpack_pred(lp->image.p, out, B, e->lat_out, H, W, pl->Hp, pl->Wp, 4, s);  // x[..., :H, :W]
        IRSDE_HIP_CHECK(hipEventRecord(e->ev_out, s));
        IRSDE_HIP_CHECK(hipStreamWaitEvent(user, e->ev_out, 0));
    });
}

int irsde_set_lens_info(irsde_engine* e, const float* info, int B) {
    return guard([&] {
        if (!e || !info || B < 1) throw HipError("null argument");
        if (e->arch != 1 || !naf_lens(e)) throw HipError("set_lens_info: not a latent-bokeh ConditionalNAFNet engine");
        if (!e->finalized) throw HipError("set_lens_info: weights not finalized");
        std::lock_guard<std::mutex> lk(e->mu);
        IRSDE_HIP_CHECK(hipSetDevice(e->cfg.device));
        hipStream_t s = e->stream;
        if (e->cam_rows < B) {  // plans bake the table pointer: drop them when it moves
            IRSDE_HIP_CHECK(hipDeviceSynchronize());
            e->plans.clear();
            e->cam_cur = e->dmalloc((size_t)B * e->cam_row);
            e->cam_rows = B;
        }
        const int width = e->cfg.nf, td = e->time_dim;
        // cam_embed = cam_mlp(cat_i SinusoidalPosEmb(lens_info_i))  (:172-173); rows b*3+i of the sinusoid table are row b
        float *dv = nullptr, *emb = nullptr, *w1 = nullptr, *g1 = nullptr, *h2 = nullptr, *g2 = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&dv, (size_t)B * 3 * 4));
        IRSDE_HIP_CHECK(hipMemcpy(dv, info, (size_t)B * 3 * 4, hipMemcpyHostToDevice));
        IRSDE_HIP_CHECK(hipMalloc(&emb, (size_t)B * 3 * width * 4));
        IRSDE_HIP_CHECK(hipMalloc(&w1, (size_t)B * 2 * td * 4));
        IRSDE_HIP_CHECK(hipMalloc(&g1, (size_t)B * td * 4));
        IRSDE_HIP_CHECK(hipMalloc(&h2, (size_t)B * td * 4));
        IRSDE_HIP_CHECK(hipMalloc(&g2, (size_t)B * (td / 2) * 4));
        launch_sinusoid(dv, e->freqs, emb, B * 3, width / 2, s);
        launch_row_linear(emb, 3 * width, e->cm_w1, e->cm_b1, w1, 2 * td, B, 3 * width, 2 * td, ACT_NONE, ACT_NONE, s);
        launch_row_gate(w1, g1, B, td, s);
        launch_row_linear(g1, td, e->cm_w3, e->cm_b3, h2, td, B, td, td, ACT_NONE, ACT_NONE, s);
        launch_row_gate(h2, g2, B, td / 2, s);  // the block's cam_mlp starts with SimpleGate (:22-24)
        for (NafBlockW* b : e->naf_all)
            launch_row_linear(g2, td / 2, b->cam_w, b->cam_b, e->cam_cur + b->cam_off, e->cam_row, B, td / 2, 2 * b->c, ACT_NONE,
                              ACT_NONE, s);
        IRSDE_HIP_CHECK(hipStreamSynchronize(s));
        for (float* q : {dv, emb, w1, g1, h2, g2}) (void)hipFree(q);
        e->cam_set = B;
    });
}

int irsde_eval_metrics(const float* out, const float* gt, int B, int C, int H, int W, int crop_border, double* metrics,
                       void* stream) {
    return guard([&] {
        if (!out || !gt || !metrics || B < 1) throw HipError("null argument");
        std::vector<double> sums((size_t)B * 4);
        eval_metrics(out, gt, B, C, H, W, crop_border, sums.data(), reinterpret_cast<hipStream_t>(stream));
        const double Hc = H - 2 * crop_border, Wc = W - 2 * crop_border;
        const double n_rgb = Hc * Wc * C, n_y = Hc * Wc, v_rgb = (Hc - 10) * (Wc - 10) * C, v_y = (Hc - 10) * (Wc - 10);
        auto psnr = [](double sse, double n) {
            const double mse = sse / n;
            return mse == 0.0 ? INFINITY : 20.0 * log10(255.0 / sqrt(mse));
        };
        for (int b = 0; b < B; ++b) {
            metrics[b * 4 + 0] = psnr(sums[b * 4 + 0], n_rgb);
            metrics[b * 4 + 1] = sums[b * 4 + 1] / v_rgb;
            metrics[b * 4 + 2] = C == 3 ? psnr(sums[b * 4 + 2], n_y) : NAN;
            metrics[b * 4 + 3] = C == 3 ? sums[b * 4 + 3] / v_y : NAN;
        }
    });
}

int irsde_tensor2img(const float* in, unsigned char* out, int B, int C, int H, int W, void* stream) {
    return guard([&] {
        if (!in || !out) throw HipError("null argument");
        tensor2img_u8(in, out, B, C, H, W, reinterpret_cast<hipStream_t>(stream));
    });
}

}  // extern "C"
