// Plan construction: one network evaluation as a static launch list over a static activation arena
// (ConditionalUNet / ConditionalNAFNet forward, latent UNet encode / decode).
#include "engine.h"
#include <atomic>

using namespace irsde;

namespace irsde {

// ---------------------------------------------------------------------------------------------
// Plan construction: one network evaluation as a static launch list over a static arena
// ---------------------------------------------------------------------------------------------
struct Builder {
    irsde_engine* e;
    Plan* pl;
    bool reuse;
    bool naive;
    int film_bstride;
    int b0 = 0;   // first image of this plan within the call's batch (sub-batch plans): base of the per-image tables
    const float* film_base() const { return pl->own_film ? pl->own_film : e->film_cur + (size_t)b0 * film_bstride; }
    const float* cam_base() const { return e->cam_cur + (size_t)b0 * e->cam_row; }
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
        if (!naive && (e->cfg.flags & IRSDE_FLAG_BF16)) {
            p.w_bf = e->bf16_copy(p.w, (size_t)p.Cout * p.KH * p.KW * (p.C0 + p.C1));
            p.f16 = (e->cfg.flags & IRSDE_FLAG_FP16) ? 1 : 0;
        } else if (!naive && (e->cfg.flags & (IRSDE_FLAG_SPLIT_BF16X2 | IRSDE_FLAG_SPLIT_F16X2)) && p.Cout >= 64 && p.nz == 1 && !p.ln_g &&
                   p.KH * p.KW * (p.C0 + p.C1) >= split_direct_min_k()) {
            // (!ln_g: the fused LayerNorm epilogue normalises over the tile's BN columns and needs BN == Cout, which the PAIR tile
            //  choice does not guarantee — to_out layers of the IRSDE_FLAG_NO_FUSED_ATTN plan stay on the f32 kernel)
            // split-operand arithmetic for the direct layers too (PAIR kernels): fp32 storage, 16-bit hi + lo operand pairs
            const size_t nw = (size_t)p.Cout * p.KH * p.KW * (p.C0 + p.C1);
            const auto pc = e->pair_copy_il(p.w, (size_t)p.Cout, p.KH * p.KW * (p.C0 + p.C1));
            (void)nw;
            p.w_pair = pc.p; p.pair_scale = pc.inv_scale;
            p.f16 = (e->cfg.flags & IRSDE_FLAG_SPLIT_F16X2) ? 1 : 0;
        }
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
                     p.w_pair ? (p.f16 ? "(split f16x2)" : "(split bf16x2)") : p.w_bf ? (p.f16 ? "(fp16)" : "(bf16)") : "", M, p.Cout, p.C0 + p.C1, p.KH, p.KW, p.stride, p.in_shift, splits, blocks, op.flops);
            op.desc = buf;
            // r06: which 3x3 kernel a 16-bit layer runs on (launch_conv's own choice, conv_igemm.hip) — the plan tests pin the benchmarked plan's kernel mix with it
            if (!naive && p.w_bf && conv_halo_eligible(p)) op.desc += conv_halo2_wanted(p, 0) ? " kernel=halo512" : " kernel=halo256";
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
            if (w.wino_up && wino_shape_ok(p, 4) && push_wino(p, w.wino_u4, 4, w.wino_up, w.wino_up_scale)) return out;
            if (w.wino_uf64p && push_wino_fused(p, reinterpret_cast<const float*>(w.wino_uf64p), true, w.wino_uf64p_scale)) return out;
            if (w.wino_uf64 && push_wino_fused(p, w.wino_uf64, true)) return out;
            if (w.wino_uf && push_wino_fused(p, w.wino_uf, false)) return out;
            if (w.wino_u4 && wino_shape_ok(p, 4) && push_wino(p, w.wino_u4, 4)) return out;
            if (w.wino_u2 && wino_shape_ok(p, 2) && push_wino(p, w.wino_u2, 2)) return out;
        }
        push_conv(p);
        return out;
    }

    // Winograd F(4x4,3x3) with both transforms inside the GEMM kernel (wino_fused.hip): the big feature maps
    // (k64: the r03 kernel, 16 tiles x 64 couts per block; else 32 tiles x 32 couts)
    // pair_uscale != 0: Uf holds fp16 hi / lo halves of pair_uscale * U (the PAIR instance of the 64-cout kernel)
    bool push_wino_fused(const ConvParams& d, const float* Uf, bool k64, float pair_uscale = 0.f) {
        ConvParams dd = d;
        dd.zeros = e->zeros;
        if (pair_uscale != 0.f) dd.pair_scale = 1.0f / (kWinoFused64PairVScale * pair_uscale);
        auto eligible = [&](const ConvParams& q) { return k64 ? wino_fused64_eligible(q) : wino_fused_eligible(q); };
        // r05: the kernels address their tensors through 32-bit buffer offsets; a batch whose tensor passes 2 GiB (16 x 512^2 x 128 channels) used to
        // fall back to the three-launch path / the 32-cout kernel (16 x 512^2 ran 7 % slower per image than 8 x 512^2).  Same kernel on 2 / 4 slices
        // of the batch instead: every slice is an independent launch with its own base pointers.
        int parts = 1;
        if (!eligible(dd)) {
            for (parts = 2; parts <= 4; parts *= 2) {
                if (d.B % parts) continue;
                ConvParams q = dd;
                q.B = d.B / parts;
                if (eligible(q)) break;
            }
            if (parts > 4) return false;
            dd.B = d.B / parts;
        }
        const int Ctot = d.C0 + d.C1;
        const long long T = (long long)dd.B * (d.Ho / 4) * (d.Wo / 4);   // per launch
        const long long blocks = k64 ? wino_fused64_num_blocks(dd) : (long long)dd.B * ((d.Ho / 4 + 3) / 4) * ((d.Wo / 4 + 7) / 8) * (d.Cout / 32);
        if (T < (k64 ? wino_fused64_min_tiles() : wino_fused_min_tiles()) || blocks < 256) return false;
        // r06: wino4_fused64t_kernel (32 tiles x 64 couts per work item, every weight fragment feeds two tile groups; bit-identical results) where it measured
        // faster than the 16-tile kernel on the B = 16 256^2 plan (profiles/r06_o_wino_t_sweep.txt): enough K-loop work per resident block to amortise its
        // longer prologue / epilogue (work items per block x 16-channel chunks >= 64; >= 32 for a layer without residual whose blocks walk >= 8 items), not the
        // fused-upsample layers (their gathers are the cheaper ones: 4 tiles share a source pixel).
        bool use_t = false;
        if (k64 && pair_uscale == 0.f && wino_fused64t_mode() && wino_fused64t_eligible(dd)) {
            const double per_block = (double)wino_fused64t_num_items(dd) / std::max(1, device_cu_count());
            const double work = per_block * (Ctot / 16);
            use_t = wino_fused64t_mode() >= 2 || (!d.in_shift && (work >= 64.0 || (work >= 32.0 && per_block >= 8.0 && !d.res)));
        }
        Op op;
        op.kind = OP_CONV;
        op.flops = conv_flops(d);
        op.exec_flops = 36 * 2.0 * (double)T * parts * Ctot * d.Cout;
        op.bytes = 4.0 * (double)d.B * d.Hin * d.Win * Ctot + 4.0 * (double)d.B * d.Ho * d.Wo * d.Cout + 4.0 * 9.0 * (double)d.Cout * Ctot;
        pl->conv_flops += op.flops;
        pl->conv_exec_flops += op.exec_flops;
        pl->conv_bytes += op.bytes;
        char buf[256];
        const bool pair = pair_uscale != 0.f;
        snprintf(buf, sizeof buf, "conv(%swinograd F4 fused) %s T=%lld Cout=%d Cin=%d up=%d blocks=%lld flops=%.4g exec=%.4g%s", pair ? "split f16x2 " : "",
                 use_t ? "32x64" : k64 ? "16x64" : "32x32", T * parts, d.Cout, Ctot, d.in_shift, blocks * parts, op.flops, op.exec_flops,
                 parts > 1 ? (parts == 2 ? " (2 batch slices)" : " (4 batch slices)") : "");
        op.desc = buf;
        std::vector<ConvParams> slices;
        for (int i = 0; i < parts; ++i) {
            ConvParams q = dd;   // (dd.B is the slice's batch)
            const size_t b0 = (size_t)i * dd.B;
            q.in0 = dd.in0 + b0 * dd.Hin * dd.Win * dd.pix0;
            if (dd.in1) q.in1 = dd.in1 + b0 * dd.Hin * dd.Win * dd.pix1;
            q.out = dd.out + b0 * dd.Ho * dd.Wo * dd.out_stride;
            if (dd.res) q.res = dd.res + b0 * dd.Ho * dd.Wo * dd.res_stride;
            if (dd.film && dd.film_bstride) q.film = dd.film + b0 * dd.film_bstride;
            slices.push_back(q);
        }
        if (use_t)
            op.fn = [slices, Uf](hipStream_t s) { for (const auto& q : slices) launch_wino_fused64t(q, Uf, s, 0); };
        else if (k64)
            op.fn = [slices, Uf, pair](hipStream_t s) { for (const auto& q : slices) launch_wino_fused64(q, Uf, s, pair ? 4 : 0); };
        else
            op.fn = [slices, Uf](hipStream_t s) { for (const auto& q : slices) launch_wino_fused(q, Uf, s); };
        pl->net_ops.push_back(std::move(op));
        return true;
    }

    // Winograd F(2x2,3x3): input transform -> 16 batched GEMMs on the MFMA kernel -> output transform + epilogue
    // (Up: IRSDE_FLAG_SPLIT_BF16X2 — the component GEMMs on bf16 hi / lo pairs, gemm_split.hip)
    bool push_wino(const ConvParams& d, const float* U, int tile, const unsigned short* Up = nullptr, float up_scale = 1.f) {
        const int Ctot = d.C0 + d.C1;
        const int ncomp = (tile + 2) * (tile + 2);
        const long long T = (long long)d.B * (d.Ho / tile) * (d.Wo / tile);
        const long long gemm_blocks = ncomp * ((T + 127) / 128) * ((d.Cout + 127) / 128);
        if (gemm_blocks < 256) return false;  // tiny layers: direct conv + split-K
        ConvParams dd = d;
        dd.zeros = e->zeros;
        // pair GEMM: 256 x 256 tiles (fewer than 256 output channels leave half of every tile empty: the fused f32 kernel is faster
        // there — 256 -> 128 @ 256^2: 1.88 vs 1.56 ms), 32-bit offsets per component
        if (Up && (T < 256 || d.Cout < 256 || (unsigned long long)T * Ctot * 4ull >= 0xffffffffull)) return false;
        float* V = pl->alloc((size_t)ncomp * T * Ctot, true);   // (pairs: 2 planes x 2 bytes = the same size)
        float* Mb = pl->alloc((size_t)ncomp * T * d.Cout, true);
        WinoPlan wp = make_wino(dd, U, V, Mb, tile);
        WinoSplitPlan sp;
        if (Up) {
            sp = make_wino_pairs(dd, Up, reinterpret_cast<unsigned short*>(V), Mb, (e->cfg.flags & IRSDE_FLAG_SPLIT_F16X2) != 0, up_scale);
            wp.in = sp.in;
        }
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
            snprintf(buf, sizeof buf, "conv(%s F%d gemm x%d) T=%lld Cout=%d Cin=%d flops=%.4g exec=%.4g", Up ? (sp.f16 ? "split f16x2 winograd" : "split bf16x2 winograd") : "winograd", tile, ncomp, T,
                     d.Cout, Ctot, op.flops, op.exec_flops);
            op.desc = buf;
            if (Up) {
                const SplitGemmArgs sg = sp.gemm;
                const bool f16 = sp.f16;
                op.fn = [sg, f16](hipStream_t s) { launch_gemm_split_pairs(sg, 36, s, 0, f16); };
            } else {
                const ConvParams g = wp.gemm;
                op.fn = [g](hipStream_t s) { launch_conv(g, s); };
            }
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

    // r06: norm + FiLM + 1x1 convolution (+ SimpleGate) as one launch (naf_lnconv_kernel, kernels_misc.hip): fp16 operand mode, c = 64 / 128 / 256
    bool lnconv_ok(const ConvW& cw, const Tensor& x) const {
        return !naive && (e->cfg.flags & IRSDE_FLAG_FP16) && naf_lnconv_enabled() && cw.KH == 1 && cw.KW == 1 &&
               naf_lnconv_ok(x.C, cw.Cout, (long long)x.B * x.H * x.W);
    }
    Tensor lnconv(const ConvW& cw, const Tensor& x, const float* g, const float* fscale, const float* fshift, int gate, const float* gate_film) {
        const int64_t M = (int64_t)x.B * x.H * x.W, ppi = (int64_t)x.H * x.W;
        const int c = x.C, Cout = cw.Cout, fb = film_bstride, gfb = gate_film ? e->cam_row : 0;
        Tensor out = talloc(x.B, x.H, x.W, gate ? Cout / 2 : Cout);
        const unsigned short* w16 = e->bf16_copy(cw.w, (size_t)Cout * c);   // (IRSDE_FLAG_FP16: the copy holds IEEE fp16)
        const float *xp = x.p, *bias = cw.bias;
        float* op = out.p;
        Op o;
        o.kind = OP_CONV;
        o.flops = 2.0 * (double)M * c * Cout;
        o.exec_flops = o.flops;
        o.bytes = 4.0 * (double)M * c + 4.0 * (double)M * (gate ? Cout / 2 : Cout) + 2.0 * (double)Cout * c;
        pl->conv_flops += o.flops;
        pl->conv_exec_flops += o.exec_flops;
        pl->conv_bytes += o.bytes;
        char buf[200];
        snprintf(buf, sizeof buf, "conv(fp16, LayerNorm + FiLM fused%s) M=%lld Cout=%d Cin=%d k=1x1 blocks=%lld flops=%.4g", gate ? ", gate" : "", (long long)M, Cout, c,
                 (long long)((M + 63) / 64) * (Cout / 64), o.flops);
        o.desc = buf;
        o.fn = [=](hipStream_t s) { launch_naf_lnconv(xp, g, fscale, fshift, fb, ppi, w16, bias, op, M, c, Cout, gate, gate_film, gfb, s); };
        pl->net_ops.push_back(std::move(o));
        return out;
    }

    // conv3 / conv5 on the same kernel (no LayerNorm): out = res + (W (in * in_scale) + bias) * ch_scale
    Tensor pwconv(const ConvW& cw, const Tensor& in, const float* in_scale, const float* ch_scale, const Tensor& res) {
        const int64_t M = (int64_t)in.B * in.H * in.W, ppi = (int64_t)in.H * in.W;
        const int c = in.C, Cout = cw.Cout;
        Tensor out = talloc(in.B, in.H, in.W, Cout);
        const unsigned short* w16 = e->bf16_copy(cw.w, (size_t)Cout * c);
        const float *xp = in.p, *bias = cw.bias, *rp = res.p;
        float* op = out.p;
        Op o;
        o.kind = OP_CONV;
        o.flops = 2.0 * (double)M * c * Cout;
        o.exec_flops = o.flops;
        o.bytes = 4.0 * (double)M * c + 8.0 * (double)M * Cout + 2.0 * (double)Cout * c;
        pl->conv_flops += o.flops;
        pl->conv_exec_flops += o.exec_flops;
        pl->conv_bytes += o.bytes;
        char buf[200];
        snprintf(buf, sizeof buf, "conv(fp16, one-piece K%s) M=%lld Cout=%d Cin=%d k=1x1 blocks=%lld flops=%.4g", in_scale ? ", SCA scale" : "", (long long)M, Cout, c,
                 (long long)((M + 63) / 64) * (Cout / 64), o.flops);
        o.desc = buf;
        o.fn = [=](hipStream_t s) { launch_naf_pwconv(xp, in_scale, ppi, w16, bias, ch_scale, rp, op, M, c, Cout, s); };
        pl->net_ops.push_back(std::move(o));
        return out;
    }

    // NAFBlock.forward — DenoisingNAFNet_arch.py:56-82
    Tensor nafblock(const NafBlockW& w, const Tensor& x) {
        const int64_t M = (int64_t)x.B * x.H * x.W;
        const int64_t ppi = (int64_t)x.H * x.W;
        const int c = w.c;
        const float* film = film_base() + w.film_off;  // [shift_att | scale_att | shift_ffn | scale_ffn]
        const int fb = film_bstride;
        Tensor u;
        if (lnconv_ok(w.conv1, x)) {
            u = lnconv(w.conv1, x, w.g1, film + c, film, 0, nullptr);
        } else {
            Tensor t1 = talloc(x.B, x.H, x.W, c);
            {
                const float *xp = x.p, *g = w.g1;
                float* o = t1.p;
                push_other(OP_LN, [=](hipStream_t s) { launch_layernorm_film(xp, g, film + c, film, fb, ppi, o, M, c, 1e-5f, s); });
            }
            u = conv_naf(w.conv1, t1, ConvOpts());
            tfree(t1);
        }
        Tensor gt = talloc(x.B, x.H, x.W, c);
        const int nt = dwgate_tiles(x.H, x.W, c);
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
        Tensor y;
        if (lnconv_ok(w.conv3, gt) && x.C == w.conv3.Cout) {
            y = pwconv(w.conv3, gt, sca, w.beta, x);
        } else {
            ConvOpts o3;
            o3.in_scale = sca; o3.ch_scale = w.beta; o3.res = &x;
            y = conv_naf(w.conv3, gt, o3);
        }
        tfree(gt);
        pl->release(partial);
        pl->release(sca);
        pl->release(mean);
        Tensor v;
        if (lnconv_ok(w.conv4, y)) {
            v = lnconv(w.conv4, y, w.g2, film + 3 * c, film + 2 * c, 1, naf_lens(e) ? cam_base() + w.cam_off : nullptr);
        } else {
            Tensor t2 = talloc(x.B, x.H, x.W, c);
            {
                const float *yp = y.p, *g = w.g2;
                float* o = t2.p;
                push_other(OP_LN, [=](hipStream_t s) { launch_layernorm_film(yp, g, film + 3 * c, film + 2 * c, fb, ppi, o, M, c, 1e-5f, s); });
            }
            ConvOpts o4;
            o4.gate = 1;
            if (naf_lens(e)) o4.gate_film = cam_base() + w.cam_off;  // x * (cam_scale + 1) + cam_shift after the gate (:82-83)
            v = conv_naf(w.conv4, t2, o4);
            tfree(t2);
        }
        Tensor out;
        if (lnconv_ok(w.conv5, v) && y.C == w.conv5.Cout) {
            out = pwconv(w.conv5, v, nullptr, w.gamma, y);
        } else {
            ConvOpts o5;
            o5.ch_scale = w.gamma; o5.res = &y;
            out = conv_naf(w.conv5, v, o5);
        }
        tfree(v);
        tfree(y);
        return out;
    }

    // a run of NAFBlocks as one launch (naf_chain.hip): one work-group per image walks the whole run
    bool naf_chain_ok(const NafChainW& cw, const Tensor& x) const {
        return cw.nblocks > 0 && !naive && !x.bf16 && naf_chain_shape_ok(x.H, x.W, x.C);
    }
    // r06: G work-groups per image where they all fit the compute units at the same time (a spinning group holds its CU): the batch's groups next to
    // those of the call's other concurrent sub-batches
    int chain_groups(int B) const {
        const int mode = forced_chain_groups() ? forced_chain_groups() : naf_chain_split_mode();
        if (mode == 1) return 1;
        const int budget = device_cu_count() / std::max(1, pl->slot > 0 ? e->plan_parts : 1);
        for (int G = 4; G >= 2; G >>= 1)
            if ((mode == 0 || mode == G) && naf_chain_split_groups(B, G) <= budget) return G;
        return 1;
    }
    Tensor nafchain(NafChainW& cw, const Tensor& x) {
        Tensor out = talloc(x.B, x.H, x.W, x.C);
        const float *xp = x.p, *film = film_base(), *cam = naf_lens(e) ? cam_base() : nullptr;
        float* op = out.p;
        const int B = x.B, fb = film_bstride, cb = e->cam_row;
        const int G = chain_groups(B);
        if (G > 1 && !cw.wsplit[G]) {
            cw.wsplit[G] = reinterpret_cast<unsigned short*>(e->dmalloc((naf_chain_weight_halves(cw.nblocks) + 1) / 2));
            naf_chain_build_split_weights(cw.w, cw.wsplit[G], cw.nblocks, G, e->stream);
        }
        const NafChainW c = cw;
        Op o;
        o.kind = OP_CONV;
        // 1x1 convolutions of the run: conv1 + conv4 (512 -> 1024), conv3 + conv5 + sca.1 on the pooled vector (512 -> 512)
        o.flops = 2.0 * (double)c.nblocks * B * ((double)x.H * x.W * 512.0 * (1024.0 + 512.0 + 1024.0 + 512.0) + 512.0 * 512.0);
        o.exec_flops = o.flops;
        o.bytes = (double)c.nblocks * (3.5 * 1024 * 1024 + 4.0 * (double)naf_chain_vec_floats(1)) + 2.0 * 4.0 * (double)x.numel();
        pl->conv_flops += o.flops;
        pl->conv_exec_flops += o.exec_flops;
        pl->conv_bytes += o.bytes;
        char buf[200];
        snprintf(buf, sizeof buf, "naf_chain(fp16) blocks=%d B=%d c=%d hw=%dx%d groups=%d flops=%.4g", c.nblocks, B, x.C, x.H, x.W, G, o.flops);
        o.desc = buf;
        if (G > 1) {
            void* scratch = pl->alloc((naf_chain_split_scratch_bytes(B) + 3) / 4, false);
            IRSDE_HIP_CHECK(hipMemset(scratch, 0, naf_chain_split_scratch_bytes(B)));   // (the error flag is read by irsde_sample even if this plan never ran)
            pl->chain_err.push_back(naf_chain_split_error_flag(scratch, B));
            pl->chain_scratch.push_back({scratch, B});
            o.fn = [=](hipStream_t s) { launch_naf_chain_split(xp, op, c.wsplit[G], c.vecs, c.nblocks, B, film, fb, c.film_off, cam, cb, c.cam_off, G, scratch, s); };
        } else {
            o.fn = [=](hipStream_t s) { launch_naf_chain(xp, op, c.w, c.vecs, c.nblocks, B, film, fb, c.film_off, cam, cb, c.cam_off, s); };
        }
        pl->net_ops.push_back(std::move(o));
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
        Tensor h1 = conv(w.b1, in0, in1, 1, 1, 0, w.mlp_w ? film_base() + w.film_off : nullptr, 1, nullptr);
        Tensor out = conv(w.b2, h1, nullptr, 1, 1, 0, nullptr, 1, &R);
        tfree(h1);
        if (w.has_res) tfree(R);
        return out;
    }

    // Residual(PreNorm(dim, LinearAttention(dim))) — module_util.py:20-26,82-90,150-178
    Tensor attn(const AttnW& w, const Tensor& x) {
        const int64_t M = (int64_t)x.B * x.H * x.W;
        const int N = x.H * x.W;
        const bool fused_kv = w.g2 && !naive && !x.bf16 && !(e->cfg.flags & (IRSDE_FLAG_BF16 | IRSDE_FLAG_NO_FUSED_ATTN)) &&
                              x.C % 32 == 0 && x.C <= 256;
        const bool fused_all = fused_kv && (x.C == 64 || x.C == 128 || x.C == 256) && !(e->cfg.flags & IRSDE_FLAG_NO_FUSED_LN);
        // r05: the 16-bit operand modes (IRSDE_FLAG_BF16 [+ _ACT] / IRSDE_FLAG_FP16) take the same two fused kernels with ONE bf16 / fp16 plane per
        // projection operand (fp32 LayerNorm, softmax, context, accumulation); with IRSDE_FLAG_BF16_ACT they read and write the bf16 tensors directly
        const bool fused_16 = w.g2 && !naive && (e->cfg.flags & IRSDE_FLAG_BF16) && !(e->cfg.flags & (IRSDE_FLAG_NO_FUSED_ATTN | IRSDE_FLAG_NO_FUSED_LN)) &&
                              (x.C == 64 || x.C == 128 || x.C == 256) && x.bf16 == act_bf16();
        if (fused_16) {
            AttnWorkspace ws;
            ws.nch = attn_num_chunks(N, x.B);
            ws.pmax = pl->alloc((size_t)x.B * ws.nch * 128, false);
            ws.pctx = pl->alloc((size_t)x.B * 4 * ws.nch * 1024, false);
            ws.psum = pl->alloc((size_t)x.B * 4 * ws.nch * 32, false);
            ws.ctx = pl->alloc((size_t)x.B * 4 * 1024, false);
            Tensor y = talloc(x.B, x.H, x.W, x.C);
            if (y.bf16 != x.bf16) throw HipError("attention: mixed activation storage types");
            const float *xp = x.p, *wqkv = w.qkv.w, *wo = w.out.w, *bo = w.out.bias, *g1 = w.g1, *g2 = w.g2;
            float* yp = y.p;
            const int B = x.B, C = x.C;
            if (!bo) throw HipError("attention: to_out.0.bias missing");
            const unsigned short* wqkv16 = e->bf16_copy(wqkv, (size_t)384 * C);   // bf16, or IEEE fp16 under IRSDE_FLAG_FP16 (rows: q | k | v)
            const unsigned short* wo16 = e->bf16_copy(wo, (size_t)C * 128);
            const unsigned short* wkv16 = wqkv16 + (size_t)128 * C;
            const int op16 = (e->cfg.flags & IRSDE_FLAG_FP16) ? 3 : 2;
            const bool abf = x.bf16;
            const char* tag = op16 == 3 ? " (fp16 operands)" : abf ? " (bf16 operands + storage)" : " (bf16 operands)";
            push_other(OP_ATTN, [=](hipStream_t s) {
                launch_attention_kv_context(xp, wqkv + (size_t)128 * C, B, N, C, ws, s, g1, 1e-5f, wkv16, 1.f, op16, abf);
            });
            pl->net_ops.back().desc = std::string("linear_attention LayerNorm + k,v projection") + tag + " + context (fused) C=" + std::to_string(C);
            push_other(OP_ATTN, [=](hipStream_t s) {
                launch_attention_q_out_fused(xp, xp, wqkv, wo, bo, g2, yp, B, N, C, 1e-5f, ws, s, g1, wqkv16, wo16, 1.f, 1.f, op16, abf);
            });
            pl->net_ops.back().desc = std::string("linear_attention LayerNorm + q projection") + tag +
                                      " + softmax + context + to_out + LayerNorm + residual (fused) C=" + std::to_string(C);
            return y;
        }
        if (fused_all) {
            // whole Residual(PreNorm(LinearAttention)) block in two kernels + the context merge: PreNorm's LayerNorm runs on
            // the tiles both kernels stage, so no normalised copy of x, no q / k / v, no attention output reach HBM
            AttnWorkspace ws;
            ws.nch = attn_num_chunks(N, x.B);
            ws.pmax = pl->alloc((size_t)x.B * ws.nch * 128, false);
            ws.pctx = pl->alloc((size_t)x.B * 4 * ws.nch * 1024, false);
            ws.psum = pl->alloc((size_t)x.B * 4 * ws.nch * 32, false);
            ws.ctx = pl->alloc((size_t)x.B * 4 * 1024, false);
            Tensor y = talloc(x.B, x.H, x.W, x.C);
            const float *xp = x.p, *wq = w.qkv.w, *wkv = w.qkv.w + (size_t)128 * x.C, *wo = w.out.w, *bo = w.out.bias, *g1 = w.g1, *g2 = w.g2;
            float* yp = y.p;
            const int B = x.B, C = x.C;
            if (!bo) throw HipError("attention: to_out.0.bias missing");
            const unsigned short* wkv_pair = nullptr;
            float kv_inv = 1.f;
            if (e->cfg.flags & IRSDE_FLAG_SPLIT_F16X2) {   // the k / v projection on fp16 hi + lo operand pairs
                const auto pc = e->pair_copy(wkv, (size_t)256 * C);
                wkv_pair = pc.p; kv_inv = pc.inv_scale;
            }
            push_other(OP_ATTN, [=](hipStream_t s) { launch_attention_kv_context(xp, wkv, B, N, C, ws, s, g1, 1e-5f, wkv_pair, kv_inv); });
            pl->net_ops.back().desc = std::string("linear_attention LayerNorm + k,v projection") + (wkv_pair ? " (split f16x2)" : "") + " + context (fused) C=" + std::to_string(C);
            const unsigned short *wq_pair = nullptr, *wo_pair = nullptr;
            float wq_inv = 1.f, wo_inv = 1.f;
            if (e->cfg.flags & IRSDE_FLAG_SPLIT_F16X2) {   // the q and to_out projections on fp16 hi + lo operand pairs
                const auto pq = e->pair_copy(wq, (size_t)128 * C), po = e->pair_copy(wo, (size_t)C * 128);
                wq_pair = pq.p; wq_inv = pq.inv_scale; wo_pair = po.p; wo_inv = po.inv_scale;
            }
            push_other(OP_ATTN, [=](hipStream_t s) {
                launch_attention_q_out_fused(xp, xp, wq, wo, bo, g2, yp, B, N, C, 1e-5f, ws, s, g1, wq_pair, wo_pair, wq_inv, wo_inv);
            });
            pl->net_ops.back().desc = std::string("linear_attention LayerNorm + q projection") + (wq_pair ? " (split f16x2)" : "") +
                                      " + softmax + context + to_out + LayerNorm + residual (fused) C=" + std::to_string(C);
            return y;
        }
        Tensor xn = talloc(x.B, x.H, x.W, x.C);
        {
            const float *xp = x.p, *g = w.g1;
            float* o = xn.p;
            const int C = x.C;
            const bool bf = x.bf16;
            push_other(OP_LN, [=](hipStream_t s) { launch_layernorm(xp, g, nullptr, o, M, C, 1e-5f, s, bf); });
        }
        Tensor a = talloc(x.B, x.H, x.W, 128);
        if (fused_kv) {
            // fp32 fused form: the k and v thirds of to_qkv never reach HBM — their projection, the softmax over the pixels and
            // the context run in one kernel on the LayerNorm output; only q (rows 0..127 of to_qkv.weight) is a convolution
            AttnWorkspace ws;
            ws.nch = attn_num_chunks(N, x.B);
            ws.pmax = pl->alloc((size_t)x.B * ws.nch * 128, false);
            ws.pctx = pl->alloc((size_t)x.B * 4 * ws.nch * 1024, false);
            ws.psum = pl->alloc((size_t)x.B * 4 * ws.nch * 32, false);
            ws.ctx = pl->alloc((size_t)x.B * 4 * 1024, false);
            {
                const float *xp = xn.p, *wkv = w.qkv.w + (size_t)128 * x.C;
                const int B = x.B, C = x.C;
                push_other(OP_ATTN, [=](hipStream_t s) { launch_attention_kv_context(xp, wkv, B, N, C, ws, s); });
                pl->net_ops.back().desc = "linear_attention k,v projection + context (fused) C=" + std::to_string(C);
            }
            ConvW wq = w.qkv;  // rows 0..127 = q
            wq.Cout = 128;
            Tensor q = conv(wq, xn, nullptr, 1, 0, 0, nullptr, 0, nullptr);
            tfree(xn);
            {
                const float* qp = q.p;
                float* o = a.p;
                const int B = x.B;
                push_other(OP_ATTN, [=](hipStream_t s) { launch_attention_q_out(qp, o, B, N, ws, s); });
                pl->net_ops.back().desc = "linear_attention softmax(q) . context";
            }
            tfree(q);
            return attn_tail(w, x, a, M);
        }
        Tensor qkv = conv(w.qkv, xn, nullptr, 1, 0, 0, nullptr, 0, nullptr);
        tfree(xn);
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
            ws.nch = attn_num_chunks(N, x.B);
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
        return attn_tail(w, x, a, M);
    }

    // to_out (1x1 conv + bias) -> LayerNorm -> + x   (module_util.py:158-161, Residual)
    Tensor attn_tail(const AttnW& w, const Tensor& x, const Tensor& a, int64_t M) {
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
        if (b.naf_chain_ok(e->naf_chain_enc[i], x)) {
            Tensor y = b.nafchain(e->naf_chain_enc[i], x);
            if (!(intro_skip && x.p == intro.p)) b.tfree(x);
            x = y;
        } else
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
    if (b.naf_chain_ok(e->naf_chain_mid, x)) {
        Tensor y = b.nafchain(e->naf_chain_mid, x);
        b.tfree(x);
        x = y;
    } else
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
        if (b.naf_chain_ok(e->naf_chain_dec[i], x)) {
            Tensor z = b.nafchain(e->naf_chain_dec[i], x);
            b.tfree(x);
            x = z;
        } else
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

// Sub-batches of a NAFNet sampler step (engine_api.hip: sample_split).  Why: on small latents most of a NAFNet evaluation is per-image latency, not
// throughput — naf_chain_kernel keeps ONE CU per image busy for ~45 % of the step (64 of 256 CUs at BASELINE configs[4]'s batch of 64) while the other
// levels' kernels are bandwidth-bound on all CUs: independent sub-batches on concurrent streams let one part's chain run under the other parts' levels.
// Only where a level actually runs as a chain, and only as two parts of >= 32 images (measured: smaller or more parts lose what the overlap wins).
// irsde_debug_force_subbatches (test / measurement hook): 0 = the heuristic below.  Process-wide (every engine sees it) and read under each engine's own
// mutex: atomic, and not to be toggled while another thread is sampling — the split decides which plans a call builds.
static std::atomic<int> g_force_chain_groups{0};   // irsde_debug_force_chain_groups: 0 = the rule (Builder::chain_groups)
void set_force_chain_groups(int g) { g_force_chain_groups.store(g, std::memory_order_relaxed); }
int forced_chain_groups() { return g_force_chain_groups.load(std::memory_order_relaxed); }
static std::atomic<int> g_force_subbatches{0};
void set_force_subbatches(int n) { g_force_subbatches.store(n, std::memory_order_relaxed); }
int naf_subbatches(const irsde_engine* e, int B, int H, int W) {
    if (e->arch == 2 || (e->cfg.flags & (IRSDE_FLAG_NAIVE_CONV | IRSDE_FLAG_KEEP_ACTIVATIONS))) return 1;
    int n = g_force_subbatches.load(std::memory_order_relaxed);
    if (n <= 0) {
        static const int env = tuning_env_int("IRSDE_SUBBATCHES", 0);
        n = env;
    }
    if (n <= 0 && e->arch != 1) n = 1;   // (the UNets split only when forced: measurement hook for the small strong-scaling shards)
    if (n <= 0) {
        bool chain = false;
        const int nlev = (int)e->naf_enc.size(), ps = 1 << nlev;
        const int Hp = (H + ps - 1) / ps * ps, Wp = (W + ps - 1) / ps * ps;
        if ((e->cfg.flags & IRSDE_FLAG_FP16) && !(e->cfg.flags & IRSDE_FLAG_NO_NAF_CHAIN))
            for (int i = 0; i < nlev; ++i)
                if (e->naf_chain_enc[i].nblocks > 0 && naf_chain_shape_ok(Hp >> i, Wp >> i, e->naf_intro.Cout << i)) chain = true;
        // r05 (profiles/r05_notes.md section 4): 64 images as 2 x 32: +3 %, smaller parts a loss — with the chain on one CU per image.  r06: the chain runs on 4 groups per
        // image and everything else at these batches is latency-bound on an under-filled GPU: two parts pay from 8 images on (8 / 16 / 24 / 32 / 48 / 64 images:
        // +1.9 / +2.7 / +2.6 / +4.1 / +1.7 / +2 %, profiles/r06_aq_subbatch_ab.txt); 4 parts lose at every batch (two of four streams run, two wait)
        n = chain && B >= 8 ? 2 : 1;
    }
    n = std::min(n, (int)irsde_engine::kMaxSub);
    while (n > 1 && B % n) --n;
    return std::max(n, 1);
}

// The least recently used plan leaves the cache (with the other parts of its split batch: they are used together).  Returns false when the cache is empty.
static bool evict_lru_plan(irsde_engine* e) {
    if (e->plans.empty()) return false;
    size_t lru = 0;
    for (size_t i = 1; i < e->plans.size(); ++i)
        if (e->plans[i]->last_use < e->plans[lru]->last_use) lru = i;
    IRSDE_HIP_CHECK(hipDeviceSynchronize());
    const int vb = e->plans[lru]->B, vh = e->plans[lru]->H, vw = e->plans[lru]->W;
    const bool split = e->plans[lru]->slot > 0;
    for (size_t i = e->plans.size(); i-- > 0;)
        if (i == lru || (split && e->plans[i]->slot > 0 && e->plans[i]->B == vb && e->plans[i]->H == vh && e->plans[i]->W == vw))
            e->plans.erase(e->plans.begin() + i);
    return true;
}

static Plan* build_plan(irsde_engine* e, int B, int H, int W, bool per_sample_film, int slot, int b0);

// Cached plans are bounded by count (8: a split batch holds up to four sub-batch plans) AND by device memory: every plan owns its activation arena
// (16 x 512^2: several GiB), so a build that runs out of memory drops the least recently used plans and tries again instead of failing the sampler call
// (ADVICE r05).
Plan* get_plan(irsde_engine* e, int B, int H, int W, bool per_sample_film, int slot, int b0) {
    for (auto& p : e->plans)
        if (p->B == B && p->H == H && p->W == W && p->per_sample_film == per_sample_film && p->slot == slot && p->b0 == b0) {
            p->last_use = ++e->use_counter;
            return p.get();
        }
    if (e->plans.size() >= 8) evict_lru_plan(e);
    for (;;) {
        try {
            return build_plan(e, B, H, W, per_sample_film, slot, b0);
        } catch (const HipOutOfMemory&) {
            // (the parts of the split batch being built are the most recently used entries: they go last)
            if (!evict_lru_plan(e)) throw;
        }
    }
}

static Plan* build_plan(irsde_engine* e, int B, int H, int W, bool per_sample_film, int slot, int b0) {
    ensure_film_cur(e, per_sample_film ? b0 + B : 1);
    if (naf_lens(e)) {
        if (e->cam_set < b0 + B) throw HipError("latent-bokeh ConditionalNAFNet: irsde_set_lens_info must cover the batch first");
        if (e->cam_rows < b0 + B) throw HipError("internal: lens table smaller than the batch");
    }

    const int depth = e->cfg.depth, nf = e->cfg.nf, in_nc = e->cfg.in_nc;
    const int sdiv = 1 << depth;
    std::unique_ptr<Plan> plan(new Plan());
    Plan* pl = plan.get();
    pl->B = B; pl->H = H; pl->W = W;
    pl->Hp = (H + sdiv - 1) / sdiv * sdiv;
    pl->Wp = (W + sdiv - 1) / sdiv * sdiv;
    pl->per_sample_film = per_sample_film;
    pl->slot = slot; pl->b0 = b0;
    pl->pred_stride = (e->cfg.out_nc + 3) & ~3;
    pl->last_use = ++e->use_counter;
    // F.pad 'reflect' needs pad < dim (DenoisingUNet_arch.py:82)
    if (e->arch != 1 && (pl->Hp - H >= H || pl->Wp - W >= W)) throw HipError("image too small for reflect padding");

    if (slot > 0) {
        if (per_sample_film) throw HipError("internal: sub-batch plans are sampler plans (one time step for the whole part)");
        pl->own_film = pl->alloc((size_t)e->film_row + 64, false);
        pl->own_step = reinterpret_cast<StepState*>(pl->alloc(sizeof(StepState) / 4 + 4, false));
        IRSDE_HIP_CHECK(hipMemset(pl->own_step, 0, sizeof(StepState)));
    }
    const size_t img = (size_t)B * in_nc * H * W;
    pl->xin = pl->alloc(img, false);
    pl->cin = pl->alloc(img, false);
    const bool uncond = e->arch == 0 && (e->cfg.flags & IRSDE_FLAG_UNCOND_FULLATTN) != 0;
    const int P = ((uncond ? 1 : 2) * in_nc + 3) & ~3;
    const size_t x0n = (size_t)B * (pl->Hp + 6) * (pl->Wp + 6) * P + 64;
    pl->x0 = pl->alloc(x0n, false);
    IRSDE_HIP_CHECK(hipMemset(pl->x0, 0, x0n * sizeof(float)));

    Builder b{e, pl, (e->cfg.flags & IRSDE_FLAG_KEEP_ACTIVATIONS) == 0, (e->cfg.flags & IRSDE_FLAG_NAIVE_CONV) != 0,
              per_sample_film ? e->film_row : 0, b0};
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
    // r04 (ABI 104): a decode plan reads its skips where the encode plan of the same shape leaves them (no NCHW round trip when the caller
    // passes hidden == NULL), so the two plans of a shape live and die together
    LatentPlan* enc = decode ? get_latent_plan(e, B, H, W, false) : nullptr;
    // the LRU counts SHAPES (an encode / decode pair is one entry): four shapes in rotation never rebuild
    size_t n_shapes = 0;
    bool shape_known = false;
    for (size_t i = 0; i < e->lat_plans.size(); ++i) {
        const Plan* q = e->lat_plans[i]->plan.get();
        bool first = true;
        for (size_t j = 0; j < i; ++j) {
            const Plan* r = e->lat_plans[j]->plan.get();
            if (r->B == q->B && r->H == q->H && r->W == q->W) { first = false; break; }
        }
        n_shapes += first ? 1 : 0;
        if (q->B == B && q->H == H && q->W == W) shape_known = true;
    }
    if (!shape_known && n_shapes >= 4) {
        size_t lru = e->lat_plans.size();
        for (size_t i = 0; i < e->lat_plans.size(); ++i) {
            if (e->lat_plans[i].get() == enc) continue;
            if (lru == e->lat_plans.size() || e->lat_plans[i]->plan->last_use < e->lat_plans[lru]->plan->last_use) lru = i;
        }
        IRSDE_HIP_CHECK(hipDeviceSynchronize());
        const int vb = e->lat_plans[lru]->plan->B, vh = e->lat_plans[lru]->plan->H, vw = e->lat_plans[lru]->plan->W;
        for (size_t i = e->lat_plans.size(); i-- > 0;)
            if (e->lat_plans[i].get() != enc && e->lat_plans[i]->plan->B == vb && e->lat_plans[i]->plan->H == vh && e->lat_plans[i]->plan->W == vw)
                e->lat_plans.erase(e->lat_plans.begin() + i);
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
        lp->hidden = enc->hidden;   // shared storage (same NHWC geometry: the skips are conv outputs of 64 .. 256 channels)
        for (size_t k = 0; k < hgeo.size(); ++k)
            if (lp->hidden[k].H != (pl->Hp >> hgeo[k].first) || lp->hidden[k].W != (pl->Wp >> hgeo[k].first) || lp->hidden[k].C != rup32(hgeo[k].second))
                throw HipError("latent plan: encode / decode skip geometry mismatch");
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

}  // namespace irsde
