// Weight side of the engine: inventory under the reference's state_dict names, repacking into the kernel layouts
// ([Cout][KH][KW][Cin], Winograd U = G g G^T, interleaved SimpleGate rows, ...), FiLM / time-embedding rows.
#include "engine.h"
#include <cmath>

using namespace irsde;

namespace irsde {

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
            if (tile == 4 && (e->cfg.flags & (IRSDE_FLAG_SPLIT_BF16X2 | IRSDE_FLAG_SPLIT_F16X2)) && I >= split_min_cin()) {
                unsigned short* up = nullptr;   // hi / lo pairs of U, made on the device once
                IRSDE_HIP_CHECK(hipMalloc(&up, U.size() * 4));
                e->dev_allocs.push_back(reinterpret_cast<float*>(up));
                const bool f16 = (e->cfg.flags & IRSDE_FLAG_SPLIT_F16X2) != 0;
                if (f16) {   // power-of-two scale that brings max |U| to (256, 512]: exact, undone by the GEMM
                    float mx = 0.f;
                    for (float v : U) mx = std::max(mx, std::fabs(v));
                    c.wino_up_scale = mx > 0.f ? std::exp2(std::floor(std::log2(512.0f / mx))) : 1.f;
                }
                launch_split_pairs(c.wino_u4, up, (size_t)36 * O, I, e->stream, f16, c.wino_up_scale);
                IRSDE_HIP_CHECK(hipStreamSynchronize(e->stream));
                c.wino_up = up;
            }
            if (tile == 4 && !(e->cfg.flags & IRSDE_FLAG_NO_WINOGRAD_FUSED) && O % 32 == 0 && I % 16 == 0 && I <= kWinoFusedMaxCin &&
                O <= kWinoFusedMaxCout) {
                std::vector<float> Uf(U.size());
                wino_fused_pack_weights(U.data(), O, I, Uf.data());
                c.wino_uf = e->upload(Uf);
            }
            if (tile == 4 && !(e->cfg.flags & IRSDE_FLAG_NO_WINOGRAD_FUSED) && wino_fused64_enabled() && O % 64 == 0 && I % 64 == 0 &&
                I <= wino_fused64_max_cin() && O <= wino_fused64_max_cout()) {
                std::vector<float> Uf(U.size());
                wino_fused64_pack_weights(U.data(), O, I, Uf.data());
                c.wino_uf64 = e->upload(Uf);
                if ((e->cfg.flags & IRSDE_FLAG_SPLIT_F16X2) && wino_fused64_pair_enabled()) {   // the fp16-pair twin of the kernel: same fragment order
                    float mx = 0.f;
                    for (float v : U) mx = std::max(mx, std::fabs(v));
                    c.wino_uf64p_scale = mx > 0.f ? std::exp2(std::floor(std::log2(512.0f / mx))) : 1.f;
                    unsigned short* up = nullptr;
                    IRSDE_HIP_CHECK(hipMalloc(&up, Uf.size() * 4));
                    e->dev_allocs.push_back(reinterpret_cast<float*>(up));
                    launch_wino_fused64_split_weights(c.wino_uf64, up, Uf.size(), c.wino_uf64p_scale, e->stream);
                    IRSDE_HIP_CHECK(hipStreamSynchronize(e->stream));
                    c.wino_uf64p = up;
                }
            }
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


// The fp16 weight streams + fp32 vectors of naf_chain_kernel (csrc/naf_chain.hip) for the blocks `prefixes` (consecutive 512-channel NAFBlocks).
// Stream of wave w, block i: [conv1: 4 passes x 16 k steps x (lo tile, hi tile)] [sca.1: 16 k steps x 4 tiles] [conv3: 2 passes x 16 x 2 tiles] [conv4: as conv1]
// [conv5: as conv3]; a fragment = 64 lanes x 8 halves: lane l holds W[tile channel base + (l & 15)][32 ks + 8 (l >> 4) + 0 .. 7].
NafChainW pack_naf_chain(irsde_engine* e, const std::vector<std::string>& prefixes, const NafBlockW& first) {
    constexpr int C = 512, FR = 448;
    const int nb = (int)prefixes.size();
    std::vector<unsigned short> w(naf_chain_weight_halves(nb));
    std::vector<float> vecs(naf_chain_vec_floats(nb));
    const size_t NV = vecs.size() / nb;
    auto h16 = [](float v) {
        const _Float16 h = (_Float16)v;   // round to nearest even
        unsigned short u;
        memcpy(&u, &h, 2);
        return u;
    };
    for (int i = 0; i < nb; ++i) {
        const std::string& p = prefixes[i];
        const float* w1 = need(e, p + "conv1.weight").data.data();
        const float* ws = need(e, p + "sca.1.weight").data.data();
        const float* w3 = need(e, p + "conv3.weight").data.data();
        const float* w4 = need(e, p + "conv4.weight").data.data();
        const float* w5 = need(e, p + "conv5.weight").data.data();
        for (int wave = 0; wave < 8; ++wave) {
            unsigned short* dst = w.data() + ((size_t)wave * nb + i) * FR * 512;
            auto frag = [&](const float* W, int chbase, int ks) {
                for (int l = 0; l < 64; ++l)
                    for (int k = 0; k < 8; ++k) dst[l * 8 + k] = h16(W[(size_t)(chbase + (l & 15)) * C + 32 * ks + 8 * (l >> 4) + k]);
                dst += 512;
            };
            auto gated = [&](const float* W) {   // 1024 outputs: gate pairs (j, j + 512)
                for (int ps = 0; ps < 4; ++ps)
                    for (int ks = 0; ks < 16; ++ks) {
                        frag(W, 64 * wave + 16 * ps, ks);
                        frag(W, C + 64 * wave + 16 * ps, ks);
                    }
            };
            auto plain = [&](const float* W) {   // 512 outputs
                for (int ps = 0; ps < 2; ++ps)
                    for (int ks = 0; ks < 16; ++ks) {
                        frag(W, 64 * wave + 32 * ps, ks);
                        frag(W, 64 * wave + 32 * ps + 16, ks);
                    }
            };
            auto sca = [&](const float* W) {   // r06: [k step][tile of the wave]: four quarters of 4 k steps, each accumulated on its own (naf_chain.hip)
                for (int ks = 0; ks < 16; ++ks)
                    for (int t = 0; t < 4; ++t) frag(W, 64 * wave + 16 * t, ks);
            };
            gated(w1); sca(ws); plain(w3); gated(w4); plain(w5);
            if (dst != w.data() + ((size_t)wave * nb + i + 1) * FR * 512) throw HipError("pack_naf_chain: stream length mismatch");
        }
        float* v = vecs.data() + (size_t)i * NV;
        auto put = [&](int off, const std::string& name, size_t nexp) {
            const HostTensor& t = need(e, p + name);
            if (t.data.size() != nexp) throw HipError("pack_naf_chain: unexpected size of " + p + name);
            std::copy(t.data.begin(), t.data.end(), v + off);
        };
        // offsets: naf_chain.hip NV_*
        put(0, "norm1.g", C); put(512, "norm2.g", C); put(1024, "conv1.bias", 2 * C); put(2048, "conv2.bias", 2 * C);
        {   // NV_TAP = 3072: conv2.weight [2c][1][3][3] as fp16 rows of 16 halves per channel (taps 0 .. 8, then zeros)
            const HostTensor& t = need(e, p + "conv2.weight");
            if (t.data.size() != (size_t)18 * C) throw HipError("pack_naf_chain: unexpected conv2.weight size");
            unsigned short* hd = reinterpret_cast<unsigned short*>(v + 3072);
            for (int ch = 0; ch < 2 * C; ++ch)
                for (int k = 0; k < 16; ++k) hd[ch * 16 + k] = k < 9 ? h16(t.data[(size_t)ch * 9 + k]) : (unsigned short)0;
        }
        put(11264, "sca.1.bias", C); put(11776, "conv3.bias", C); put(12288, "beta", C); put(12800, "conv4.bias", 2 * C); put(13824, "conv5.bias", C);
        put(14336, "gamma", C);
    }
    NafChainW cw;
    unsigned short* dw = reinterpret_cast<unsigned short*>(e->dmalloc((w.size() + 1) / 2));
    IRSDE_HIP_CHECK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    cw.w = dw;
    cw.vecs = e->upload(vecs);
    cw.nblocks = nb;
    cw.film_off = first.film_off;
    cw.cam_off = first.cam_off;
    return cw;
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
    // fused NAFBlock chains (fp16 mode): every group of blocks with 512 channels; the plan uses one where the feature map is 8 x 8
    e->naf_chain_enc.assign(e->naf_enc.size(), NafChainW());
    e->naf_chain_dec.assign(e->naf_dec.size(), NafChainW());
    e->naf_chain_mid = NafChainW();
    if ((e->cfg.flags & IRSDE_FLAG_FP16) && !(e->cfg.flags & (IRSDE_FLAG_NO_NAF_CHAIN | IRSDE_FLAG_NAIVE_CONV))) {
        auto names = [](const std::string& base, size_t n) {
            std::vector<std::string> v;
            for (size_t j = 0; j < n; ++j) v.push_back(base + std::to_string(j) + ".");
            return v;
        };
        for (size_t i = 0; i < e->naf_enc.size(); ++i)
            if (!e->naf_enc[i].empty() && e->naf_enc[i][0].c == 512)
                e->naf_chain_enc[i] = pack_naf_chain(e, names("encoders." + std::to_string(i) + ".", e->naf_enc[i].size()), e->naf_enc[i][0]);
        if (!e->naf_mid.empty() && e->naf_mid[0].c == 512) e->naf_chain_mid = pack_naf_chain(e, names("middle_blks.", e->naf_mid.size()), e->naf_mid[0]);
        for (size_t i = 0; i < e->naf_dec.size(); ++i)
            if (!e->naf_dec[i].empty() && e->naf_dec[i][0].c == 512)
                e->naf_chain_dec[i] = pack_naf_chain(e, names("decoders." + std::to_string(i) + ".", e->naf_dec[i].size()), e->naf_dec[i][0]);
    }
}

// ---------------------------------------------------------------------------------------------
// Latent UNet (arch == 2): inventory / packing — latent-dehazing/models/modules/UNet_arch.py:17-57.
// Channel counts there are small and irregular (8, 40, 96 ...); every NHWC tensor is stored with its channel count
// rounded up to 32 and the packed weights carry zero rows / zero K columns for the padding, so the padding channels
// hold exact zeros everywhere and the implicit-GEMM kernel (32-channel K chunks) needs no special case.
// ---------------------------------------------------------------------------------------------
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
    DeviceScope dev_scope(e->cfg.device);
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
    // plans bake the film_cur pointer: drop them — after everything queued on the engine stream (a previous asynchronous
    // irsde_sample / forward may still be replaying their graphs and reading their arenas) has finished
    if (e->stream) IRSDE_HIP_CHECK(hipStreamSynchronize(e->stream));
    e->plans.clear();
}

}  // namespace irsde
