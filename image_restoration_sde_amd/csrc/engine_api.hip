// libirsde_hip.so — the C ABI (include/irsde_hip.h) and the sampler loop.
//
// Host-side structure (all C++; PyTorch never appears here):
//   Engine      weights in kernel layout, FiLM/time table, coefficient table, plans
//   Plan        per (B,H,W): static activation arena + the launch list of ONE network evaluation
//               (ConditionalUNet.forward, DenoisingUNet_arch.py:85-134) built once, replayed T times
//   sample()    the reverse loop (sde_utils.py:252-299): [step_begin, prep, net, update] per t, either
//               eager or as one captured hipGraph replayed T times; the step index lives in device
//               memory (StepState) so the graph is t-invariant.
#include "engine.h"
#include <cmath>
#include "../../include/irsde_hip_debug.h"

using namespace irsde;

namespace irsde {
namespace {

thread_local std::string g_last_error;

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
    u.st = pl->own_step ? pl->own_step : e->step; u.ctl = e->ctl;
    u.B = pl->B; u.C = e->cfg.in_nc; u.H = pl->H; u.W = pl->W;
    u.batch0 = pl->b0;
    return u;
}

void one_step(irsde_engine* e, Plan* pl, hipStream_t s) {
    launch_step_begin(e->step, e->film_table, e->film_row, e->film_cur, e->coef_table, s);
    run_net(pl, s);
    launch_sde_update(make_update(e, pl), s);
}

// r05: a batch split into concurrent sub-batches (plans sp[0 .. n), images [b0, b0 + B / n) each).  Every part is a complete, independent sampler: its
// own step counter / coefficient row / FiLM row (Plan::own_step, own_film), its own captured step graph, its own stream (part 0: the engine stream,
// part i > 0: sub_stream[i - 1]) — the parts fork once behind the call's inputs (ev_fork) and join once in front of its outputs (ev_join); there is
// no per-step dependency between them, so the hardware queues overlap one part's per-image latency-bound kernels (naf_chain_kernel: one CU per image)
// with the other parts' bandwidth-bound ones.  (First version: fork / join INSIDE one captured step graph — the branches of a hipGraph replay did not
// overlap on ROCm 7.2: 156.9 -> 161.6 images/s on BASELINE configs[4], gpurun_out r05c.)  Nothing in a part depends on another part (no cross-batch op
// in the score network): the result is the un-split one up to the tilings the smaller plans choose.
void ensure_sub_streams(irsde_engine* e, int n) {
    if (!e->ev_fork) IRSDE_HIP_CHECK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    for (int i = 0; i + 1 < n; ++i) {
        if (!e->sub_stream[i]) IRSDE_HIP_CHECK(hipStreamCreateWithFlags(&e->sub_stream[i], hipStreamNonBlocking));
        if (!e->ev_join[i]) IRSDE_HIP_CHECK(hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming));
    }
}
void one_step_part(irsde_engine* e, Plan* pl, hipStream_t s) {
    launch_step_begin(pl->own_step, e->film_table, e->film_row, pl->own_film, e->coef_table, s);
    run_net(pl, s);
    launch_sde_update(make_update(e, pl), s);
}

// irsde_debug_conv / irsde_bench_conv only: selects a kernel variant for the launches of ONE call and always returns to
// the production dispatch (also when a launch throws).
struct VariantScope {
    explicit VariantScope(int v) { conv_set_variant(v); }
    ~VariantScope() { conv_set_variant(0); }
};

// IRSDE_FLAG_FP16 = the 16-bit operand mode (IRSDE_FLAG_BF16's kernels, plans and weight copies) with IEEE fp16 rounding
void apply_fp16_flag(irsde_engine* e) {
    if (!(e->cfg.flags & IRSDE_FLAG_FP16)) return;
    if (e->cfg.flags & IRSDE_FLAG_BF16_ACT) {
        delete e;
        throw HipError("IRSDE_FLAG_FP16 keeps fp32 activation storage: it cannot be combined with IRSDE_FLAG_BF16_ACT");
    }
    e->cfg.flags |= IRSDE_FLAG_BF16;
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
int irsde_version(void) { return 107; }  // changelog: include/irsde_hip.h

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
        apply_fp16_flag(e);
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
        apply_fp16_flag(e);
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
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    e->plans.clear();
    e->lat_plans.clear();
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->ev_out) (void)hipEventDestroy(e->ev_out);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    for (auto ev : e->ev_join) if (ev) (void)hipEventDestroy(ev);
    for (auto st : e->sub_stream) if (st) (void)hipStreamDestroy(st);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    for (float* p : e->dev_allocs) (void)hipFree(p);
    for (auto& kv : e->bf16_copies) (void)hipFree(kv.second);
    if (e->coef_table) (void)hipFree(e->coef_table);
    if (e->film_table) (void)hipFree(e->film_table);
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);  // destroy runs from a Python finaliser: leave the current device alone
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
        DeviceScope dev_scope(e->cfg.device);
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
        DeviceScope dev_scope(e->cfg.device);
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
        if (T < 0) T = e->T;
        if (T > e->T) throw HipError("sample: T exceeds the schedule length");
        if (B < 1 || H < 2 || W < 2) throw HipError("sample: bad shape");
        if (t_stop < 0 || (T > 0 && t_stop >= T) || (T == 0 && t_stop != 0)) throw HipError("sample: t_stop must be in [0, T)");
        const int nsteps = T - t_stop;
        std::lock_guard<std::mutex> lk(e->mu);
        DeviceScope dev_scope(e->cfg.device);
        hipStream_t user = reinterpret_cast<hipStream_t>(stream);
        if (T == 0) {
            // the reference's `for t in reversed(range(1, T + 1))` runs zero steps and returns the clone of xt
            // (e.g. reverse_ode(x, T=sde.get_optimal_timestep(sigma)) when the argmin is index 0)
            IRSDE_HIP_CHECK(hipMemcpyAsync(out, xT, (size_t)B * e->cfg.in_nc * H * W * 4, hipMemcpyDeviceToDevice, user));
            return;
        }
        hipStream_t s = e->stream;
        const bool profile = (flags & IRSDE_SAMPLE_PROFILE) != 0;
        const bool graph = (flags & IRSDE_SAMPLE_GRAPH) != 0 && !profile;
        const size_t img = (size_t)B * e->cfg.in_nc * H * W;
        // r06: a split NAFBlock-chain launch (naf_chain.hip, G work-groups per image) whose groups were not co-resident raised its error flag instead of
        // hanging the GPU: the results of that call were garbage — fail loudly at the next one (the call itself is asynchronous)
        for (auto& p : e->plans)
            for (const unsigned* f : p->chain_err) {
                unsigned v = 0;
                IRSDE_HIP_CHECK(hipMemcpy(&v, f, 4, hipMemcpyDeviceToHost));
                if (v) {
                    for (auto& sc : p->chain_scratch) naf_chain_split_reset(sc.first, sc.second);
                    char code[32];
                    snprintf(code, sizeof code, " [flag 0x%x]", v);
                    throw HipError(std::string("sample: the previous call's split NAFBlock-chain kernel timed out waiting for a work-group that was not resident (its results were invalid); "
                                               "set IRSDE_TUNING=1 IRSDE_NAF_CHAIN_SPLIT=1 if other work shares this GPU") + code);
                }
            }
        const int nsub = profile ? 1 : naf_subbatches(e, B, H, W);   // (the event-instrumented pass times the un-split plan: its kernels are the same)
        if (nsub > 1) {
            const int Bs = B / nsub;
            const size_t simg = img / nsub;
            std::vector<Plan*> sp(nsub);
            struct PartsScope { irsde_engine* e; PartsScope(irsde_engine* e_, int n) : e(e_) { e->plan_parts = n; } ~PartsScope() { e->plan_parts = 1; } } parts_scope(e, nsub);
            for (int i = 0; i < nsub; ++i) sp[i] = get_plan(e, Bs, H, W, false, i + 1, i * Bs);
            for (int i = 0; i < nsub; ++i) (void)get_plan(e, Bs, H, W, false, i + 1, i * Bs);   // (all parts most recently used: none of them is the next eviction victim)
            ensure_sub_streams(e, nsub);
            IRSDE_HIP_CHECK(hipEventRecord(e->ev_in, user));
            IRSDE_HIP_CHECK(hipStreamWaitEvent(s, e->ev_in, 0));
            launch_set_ctl(e->ctl, mode, noise, (long long)img, seed, image_offset, s);   // call-level arguments, shared by the parts (read-only during the call)
            IRSDE_HIP_CHECK(hipEventRecord(e->ev_fork, s));
            for (int i = 0; i < nsub; ++i) {
                hipStream_t t = i == 0 ? s : e->sub_stream[i - 1];
                Plan* pl = sp[i];
                if (i > 0) IRSDE_HIP_CHECK(hipStreamWaitEvent(t, e->ev_fork, 0));
                IRSDE_HIP_CHECK(hipMemcpyAsync(pl->xin, xT + i * simg, simg * 4, hipMemcpyDeviceToDevice, t));
                if (mu) IRSDE_HIP_CHECK(hipMemcpyAsync(pl->cin, mu + i * simg, simg * 4, hipMemcpyDeviceToDevice, t));
                launch_set_step(pl->own_step, T, t);
                if (graph && !pl->graph_exec) {
                    IRSDE_HIP_CHECK(hipStreamBeginCapture(t, hipStreamCaptureModeThreadLocal));
                    try {
                        one_step_part(e, pl, t);
                    } catch (...) {
                        hipGraph_t g = nullptr;
                        (void)hipStreamEndCapture(t, &g);
                        if (g) (void)hipGraphDestroy(g);
                        throw;
                    }
                    IRSDE_HIP_CHECK(hipStreamEndCapture(t, &pl->graph));
                    IRSDE_HIP_CHECK(hipGraphInstantiate(&pl->graph_exec, pl->graph, nullptr, nullptr, 0));
                }
            }
            // step-major launch order: the host feeds every part's queue in turn (a part-major order would enqueue T steps of part 0 before part 1 starts)
            for (int k = 0; k < nsteps; ++k)
                for (int i = 0; i < nsub; ++i) {
                    hipStream_t t = i == 0 ? s : e->sub_stream[i - 1];
                    if (graph) IRSDE_HIP_CHECK(hipGraphLaunch(sp[i]->graph_exec, t));
                    else one_step_part(e, sp[i], t);
                }
            for (int i = 0; i < nsub; ++i) {
                hipStream_t t = i == 0 ? s : e->sub_stream[i - 1];
                IRSDE_HIP_CHECK(hipMemcpyAsync(out + i * simg, sp[i]->xin, simg * 4, hipMemcpyDeviceToDevice, t));
                if (i > 0) {
                    IRSDE_HIP_CHECK(hipEventRecord(e->ev_join[i - 1], t));
                    IRSDE_HIP_CHECK(hipStreamWaitEvent(s, e->ev_join[i - 1], 0));
                }
            }
            IRSDE_HIP_CHECK(hipEventRecord(e->ev_out, s));
            IRSDE_HIP_CHECK(hipStreamWaitEvent(user, e->ev_out, 0));
            return;
        }
        Plan* pl = get_plan(e, B, H, W, false);
        IRSDE_HIP_CHECK(hipEventRecord(e->ev_in, user));
        IRSDE_HIP_CHECK(hipStreamWaitEvent(s, e->ev_in, 0));
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
            e->op_split = naf_subbatches(e, B, H, W);
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
        DeviceScope dev_scope(e->cfg.device);
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
        if (e->op_split > 1) {   // (label what was timed — parsers skip lines without the "ms" column)
            snprintf(line, sizeof line, "note: the un-split plan; irsde_sample runs this batch as %d concurrent sub-batches (own plans, streams, step graphs)\n", e->op_split);
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
        DeviceScope dev_scope(e->cfg.device);
        Plan* pl = get_plan(e, B, H, W, false);
        std::string out;
        for (auto& op : pl->net_ops) out += op.desc + "\n";
        const int nsub = naf_subbatches(e, B, H, W);
        if (nsub > 1) out += "note: the un-split plan; irsde_sample runs this batch as " + std::to_string(nsub) + " concurrent sub-batches of " + std::to_string(B / nsub) + " images (own plans)\n";
        strncpy(buf, out.c_str(), buflen - 1);
        buf[buflen - 1] = 0;
    });
}

int irsde_work_model(irsde_engine* e, int B, int H, int W, double out[2]) {
    return guard([&] {
        if (!e || !out) throw HipError("null argument");
        if (!e->finalized) throw HipError("work_model: weights not finalized");
        std::lock_guard<std::mutex> lk(e->mu);
        DeviceScope dev_scope(e->cfg.device);
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
        if (naive == 44 || naive == 45) {  // three-launch Winograd F(4x4,3x3) with the engine's pair GEMM: 44 fp16 pairs, 45 bf16 pairs
            const bool f16 = naive == 44;
            if (!wino_shape_ok(p, 4) || Cin % 32) throw HipError("debug_conv: shape not eligible for the pair GEMM");
            std::vector<float> U((size_t)36 * Cout * Cin);
            wino_transform_weights(pk.data(), Cout, Cin, U.data(), 4);
            float mx = 0.f;
            for (float v : U) mx = std::max(mx, std::fabs(v));
            const float us = f16 && mx > 0.f ? std::exp2(std::floor(std::log2(512.0f / mx))) : 1.f;
            const long long T = (long long)B * (p.Ho / 4) * (p.Wo / 4);
            float *dU = nullptr, *dM = nullptr;
            unsigned short *dUp = nullptr, *dVp = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&dU, U.size() * 4));
            IRSDE_HIP_CHECK(hipMemcpy(dU, U.data(), U.size() * 4, hipMemcpyHostToDevice));
            IRSDE_HIP_CHECK(hipMalloc(&dUp, U.size() * 4));
            IRSDE_HIP_CHECK(hipMalloc(&dVp, (size_t)36 * T * Cin * 4));
            IRSDE_HIP_CHECK(hipMalloc(&dM, (size_t)36 * T * Cout * 4));
            launch_split_pairs(dU, dUp, (size_t)36 * Cout, Cin, s, f16, us);
            const WinoSplitPlan sp = make_wino_pairs(p, dUp, dVp, dM, f16, us);
            launch_wino_input(sp.in, s);
            launch_gemm_split_pairs(sp.gemm, 36, s, 0, f16);
            launch_wino_output(sp.out, s);
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            (void)hipFree(dU); (void)hipFree(dUp); (void)hipFree(dVp); (void)hipFree(dM);
        } else if (naive == 42 || naive == 43) {  // three-launch Winograd F(4x4,3x3) with split-operand component GEMMs: 2 / 3 bf16 planes
            const int npl = naive - 40;
            if (!wino_shape_ok(p, 4)) throw HipError("debug_conv: shape not eligible for Winograd");
            std::vector<float> U((size_t)36 * Cout * Cin);
            wino_transform_weights(pk.data(), Cout, Cin, U.data(), 4);
            const long long T = (long long)B * (p.Ho / 4) * (p.Wo / 4);
            float *dU = nullptr, *dM = nullptr;
            unsigned short *dUs = nullptr, *dVs = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&dU, U.size() * 4));
            IRSDE_HIP_CHECK(hipMemcpy(dU, U.data(), U.size() * 4, hipMemcpyHostToDevice));
            IRSDE_HIP_CHECK(hipMalloc(&dUs, U.size() * 2 * npl));
            IRSDE_HIP_CHECK(hipMalloc(&dVs, (size_t)36 * T * Cin * 2 * npl));
            IRSDE_HIP_CHECK(hipMalloc(&dM, (size_t)36 * T * Cout * 4));
            launch_split_planes(dU, dUs, U.size(), U.size(), npl, s);
            const WinoSplitPlan sp = make_wino_split(p, dUs, dVs, dM, npl);
            launch_wino_input(sp.in, s);
            launch_gemm_split(sp.gemm, npl, 36, s);
            launch_wino_output(sp.out, s);
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            (void)hipFree(dU); (void)hipFree(dUs); (void)hipFree(dVs); (void)hipFree(dM);
        } else if (naive == 35 || naive == 37 || naive == 39 || naive == 56 || naive == 58 || naive == 61) {  // the 64-cout fused Winograd kernel on fp16 hi + lo operand pairs (IRSDE_FLAG_SPLIT_F16X2's big-feature-map path)
            if (!wino_fused64_eligible(p)) throw HipError("debug_conv: shape not eligible for the fused Winograd kernel");
            std::vector<float> U((size_t)36 * Cout * Cin), Uf((size_t)36 * Cout * Cin);
            wino_transform_weights(pk.data(), Cout, Cin, U.data(), 4);
            wino_fused64_pack_weights(U.data(), Cout, Cin, Uf.data());
            float mx = 0.f;
            for (float v : U) mx = std::max(mx, std::fabs(v));
            const float usc = mx > 0.f ? std::exp2(std::floor(std::log2(512.0f / mx))) : 1.f;
            float* dUf = nullptr;
            unsigned short* dUp = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&dUf, Uf.size() * 4));
            IRSDE_HIP_CHECK(hipMalloc(&dUp, Uf.size() * 4));
            IRSDE_HIP_CHECK(hipMemcpy(dUf, Uf.data(), Uf.size() * 4, hipMemcpyHostToDevice));
            launch_wino_fused64_split_weights(dUf, dUp, Uf.size(), usc, s);
            p.pair_scale = 1.0f / (kWinoFused64PairVScale * usc);
            launch_wino_fused64(p, reinterpret_cast<const float*>(dUp), s, naive == 61 ? 52 : naive == 58 ? 44 : naive == 56 ? 24 : naive == 39 ? 9 : naive == 37 ? 4 + 64 : 4);   // 37: + cout block by XCD where legal; 39: r03's one-block-per-tile-group kernel
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            (void)hipFree(dUf); (void)hipFree(dUp);
        } else if (naive == 33 || naive == 34 || naive == 36 || naive == 38 || (naive >= 50 && naive <= 55) || naive == 57 || naive == 60 || naive == 62 || naive == 63) {  // fused Winograd F(4x4,3x3) kernels (wino_fused.hip): 33 = 32 couts per block, 34 = 64
            if (naive == 33 ? !wino_fused_eligible(p) : !wino_fused64_eligible(p)) throw HipError("debug_conv: shape not eligible for the fused Winograd kernel");
            std::vector<float> U((size_t)36 * Cout * Cin), Uf((size_t)36 * Cout * Cin);
            wino_transform_weights(pk.data(), Cout, Cin, U.data(), 4);
            if (naive == 33) wino_fused_pack_weights(U.data(), Cout, Cin, Uf.data());
            else wino_fused64_pack_weights(U.data(), Cout, Cin, Uf.data());
            float* dUf = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&dUf, Uf.size() * 4));
            IRSDE_HIP_CHECK(hipMemcpy(dUf, Uf.data(), Uf.size() * 4, hipMemcpyHostToDevice));
            if (naive == 62 || naive == 63) {   // r06: the two-tile-group kernel (wino_fused_t.hip; 63: + cout block by XCD where legal)
                if (!wino_fused64t_eligible(p)) throw HipError("debug_conv: shape not eligible for the two-tile-group fused Winograd kernel");
                launch_wino_fused64t(p, dUf, s, naive == 63 ? 64 : 0);
            } else
            if (naive == 33) launch_wino_fused(p, dUf, s);
            else if (naive == 60) launch_wino_fused64(p, dUf, s, 48);   // r04's single-stream kernel (61: its fp16-pair twin)
            else if (naive == 55 || naive == 57) launch_wino_fused64(p, dUf, s, naive == 55 ? 20 : 40);   // the register-patch persistent kernel at fixed grid sizes (it IS production: launch_wino_fused64 defaults to persist = 1)
            else if (naive >= 50) {   // r04 tuning twins of the persistent kernel: 50 .. 54 = OPT 15 / 1 / 2 / 4 / 8
                static const int opts[5] = {15, 1, 2, 4, 8};
                wino_fused64_set_opt(opts[naive - 50]);
                launch_wino_fused64(p, dUf, s, 26);
            } else launch_wino_fused64(p, dUf, s, naive == 38 ? 10 : naive == 36 ? 64 : 0);   // 36: + cout block by XCD where legal; 38: r03's one-block-per-tile-group kernel
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            (void)hipFree(dUf);
        } else if (wino_tile) {  // naive / 10: 0 = production dispatch, 1 / 2 = force the batch-loop GEMM kernel (all / 2 components per block)
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
            {
                VariantScope vs(naive >= 20 ? 72 : naive >= 10 ? 71 : 0);
                launch_conv(wp.gemm, s);
            }
            launch_wino_output(wp.out, s);
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            (void)hipFree(dU); (void)hipFree(dV); (void)hipFree(dM);
        } else if (naive == 46 || naive == 47) {   // direct implicit GEMM on the PAIR kernels: 46 fp16 hi + lo pieces, 47 bf16
            const bool f16 = naive == 46;
            float mx = 0.f;
            for (float v : pk) mx = std::max(mx, std::fabs(v));
            const float sc = f16 && mx > 0.f ? std::exp2(std::floor(std::log2(512.0f / mx))) : 1.f;
            unsigned short* dwp = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&dwp, pk.size() * 4));
            launch_split_pairs(dw, dwp, (size_t)Cout, KH * KW * (C0 + C1), s, f16, sc);
            p.w_pair = dwp; p.pair_scale = 1.0f / sc; p.f16 = f16 ? 1 : 0;
            launch_conv(p, s);
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            (void)hipFree(dwp);
        } else if (naive == 1) {
            launch_conv_naive(p, s);
        } else {
            unsigned short *dbf = nullptr, *a0 = nullptr, *a1 = nullptr, *ar = nullptr, *ao = nullptr;
            const bool act = naive == 204 || (naive >= 260 && naive <= 263);  // + bf16 activation storage (IRSDE_FLAG_BF16_ACT)
            if (naive == 4 || (naive >= 160 && naive <= 163) || act) {  // bf16-MFMA mode (variants 60 / 61: force the 256 / 128 tile; 64 / 65: the 512- / 256-pixel halo kernel)
                IRSDE_HIP_CHECK(hipMalloc(&dbf, pk.size() * 2));
                launch_f32_to_bf16(dw, dbf, pk.size(), s);
                p.w_bf = dbf;
            }
            const bool f16 = naive == 5 || (naive >= 165 && naive <= 167);  // fp16-MFMA mode: production dispatch / generic 128-row tile / 512- / 256-pixel halo kernel
            if (f16) {
                IRSDE_HIP_CHECK(hipMalloc(&dbf, pk.size() * 2));
                launch_f32_to_f16(dw, dbf, pk.size(), s);
                p.w_bf = dbf;
                p.f16 = 1;
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
            {
                const int halo_v = (naive == 162 || naive == 262 || naive == 166) ? 64 : (naive == 163 || naive == 263 || naive == 167) ? 65 : 0;   // 64 / 65: force the 512- / 256-pixel halo kernel
                VariantScope vs(halo_v ? halo_v : f16 ? (naive == 165 ? 61 : 0) : act ? (naive == 204 ? 0 : naive - 200) : (naive >= 100 ? naive - 100 : 0));  // tile variants
                launch_conv(p, s);
            }
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

int irsde_debug_split_gemm(const float* A, const float* Bm, float* C, int M, int N, int K, int ncomp, int nplanes, void* stream) {
    return guard([&] {
        // nplanes 2 / 3: the 128 x 128 plane-major prototype kernel; 42: the engine's pair-interleaved two-plane kernel
        if (nplanes == 42 || nplanes == 44) {   // the pair-interleaved two-plane kernel (LDS-DMA, 256 x 256 tiles): 42 bf16 pieces, 44 fp16 pieces
            const bool f16 = nplanes == 44;
            const float sa = f16 ? 1.0f / 16.0f : 1.f, sb = f16 ? 64.0f : 1.f;   // (any powers of two: the hook exercises the scaling)
            hipStream_t s2 = reinterpret_cast<hipStream_t>(stream);
            conv_global_init();
            unsigned short *pa = nullptr, *pb = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&pa, (size_t)ncomp * M * K * 4));
            IRSDE_HIP_CHECK(hipMalloc(&pb, (size_t)ncomp * N * K * 4));
            launch_split_pairs(A, pa, (size_t)ncomp * M, K, s2, f16, sa);
            launch_split_pairs(Bm, pb, (size_t)ncomp * N, K, s2, f16, sb);
            SplitGemmArgs gp;
            gp.a = pa; gp.b = pb; gp.out = C;
            gp.pA = (long long)M * K; gp.pB = (long long)N * K; gp.pO = (long long)M * N;
            gp.M = M; gp.N = N; gp.K = K; gp.lda = K; gp.ldc = N;
            gp.out_scale = 1.0f / (sa * sb);
            launch_gemm_split_pairs(gp, ncomp, s2, 0, f16);
            IRSDE_HIP_CHECK(hipStreamSynchronize(s2));
            (void)hipFree(pa); (void)hipFree(pb);
            return;
        }
        if (nplanes != 2 && nplanes != 3) throw HipError("debug_split_gemm: nplanes must be 2, 3, 42 or 44");
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        conv_global_init();
        const size_t na = (size_t)ncomp * M * K, nb = (size_t)ncomp * N * K;
        unsigned short *da = nullptr, *db = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&da, na * 2 * nplanes));
        IRSDE_HIP_CHECK(hipMalloc(&db, nb * 2 * nplanes));
        launch_split_planes(A, da, na, na, nplanes, s);
        launch_split_planes(Bm, db, nb, nb, nplanes, s);
        SplitGemmArgs g;
        g.a = da; g.b = db; g.out = C;
        g.plA = (long long)na; g.plB = (long long)nb;
        g.pA = (long long)M * K; g.pB = (long long)N * K; g.pO = (long long)M * N;
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N;
        g.n_inner = gemm_split_inner(M, N, ncomp);
        launch_gemm_split(g, nplanes, ncomp, s);
        IRSDE_HIP_CHECK(hipStreamSynchronize(s));
        (void)hipFree(da); (void)hipFree(db);
    });
}

int irsde_debug_force_chain_groups(int g) {
    set_force_chain_groups(g == 1 || g == 2 || g == 4 ? g : 0);
    return IRSDE_OK;
}

int irsde_debug_force_subbatches(int n) {
    set_force_subbatches(n < 0 ? 0 : n);
    return IRSDE_OK;
}

int irsde_bench_naf_chain(int variant, int nblocks, int B, int iters, double* ms_out) {
    return guard([&] {
        if (!ms_out || nblocks < 1 || nblocks > 64 || B < 1 || iters < 1) throw HipError("bench_naf_chain: bad argument");
        const int G = variant == 22 ? 2 : (variant == 24 || variant == 25 || variant == 26) ? 4 : 1;   // (26, PROBES build: 24 with one group per image missing — must report the spin timeout)   // (25, PROBES build: 24 + its cycle stamps)   // r06: 22 / 24 = the kernel with 2 / 4 work-groups per image
#ifdef IRSDE_PROBES
        if (variant != 0 && variant != 1 && variant != 2 && variant != 11 && G == 1) throw HipError("bench_naf_chain: bad variant");
#else
        if (variant != 0 && variant != 1 && G == 1) throw HipError("bench_naf_chain: variants 2 / 11 are measurement twins (make PROBES=1)");   // (before anything is allocated)
#endif
        conv_global_init();
        hipStream_t s;
        IRSDE_HIP_CHECK(hipStreamCreate(&s));
        const size_t nx = (size_t)B * 64 * 512, nw = naf_chain_weight_halves(nblocks), nv = naf_chain_vec_floats(nblocks);
        float *dx = nullptr, *dout = nullptr, *dwf = nullptr, *dvec = nullptr, *dfilm = nullptr;
        unsigned short* dw = nullptr;
        IRSDE_HIP_CHECK(hipMalloc(&dx, nx * 4));
        IRSDE_HIP_CHECK(hipMalloc(&dout, nx * 4));
        IRSDE_HIP_CHECK(hipMalloc(&dvec, nv * 4));
        IRSDE_HIP_CHECK(hipMalloc(&dfilm, (size_t)nblocks * 2048 * 4));
        IRSDE_HIP_CHECK(hipMalloc(&dw, nw * 2));
        const size_t chunk = (size_t)64 << 20;   // f32 staging of the random weights, converted to fp16 piecewise
        IRSDE_HIP_CHECK(hipMalloc(&dwf, chunk * 4));
        launch_fill_random(dx, nx, 1, 1.0f, s);
        launch_fill_random(dvec, nv, 2, 0.1f, s);
        launch_fill_random(dfilm, (size_t)nblocks * 2048, 3, 0.1f, s);
        for (size_t o = 0; o < nw; o += chunk) {
            const size_t n = std::min(chunk, nw - o);
            launch_fill_random(dwf, n, 4 + (unsigned)(o / chunk), 0.04f, s);
            launch_f32_to_f16(dwf, dw + o, n, s);
        }
        unsigned short* dwg = nullptr;
        void* dscratch = nullptr;
        if (G > 1) {
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            IRSDE_HIP_CHECK(hipMalloc(&dwg, nw * 2));
            naf_chain_build_split_weights(dw, dwg, nblocks, G, s);
            IRSDE_HIP_CHECK(hipMalloc(&dscratch, naf_chain_split_scratch_bytes(B)));
            IRSDE_HIP_CHECK(hipMemset(dscratch, 0, naf_chain_split_scratch_bytes(B)));
        }
        auto run = [&]() {
            if (G > 1) launch_naf_chain_split(dx, dout, dwg, dvec, nblocks, B, dfilm, 0, 0, nullptr, 0, 0, G, dscratch, s);
            else launch_naf_chain(dx, dout, dw, dvec, nblocks, B, dfilm, 0, 0, nullptr, 0, 0, s, variant == 11 ? 1 : variant);
        };
        if (variant == 26) {
#ifdef IRSDE_PROBES
            naf_chain_set_sabotage(1);
#else
            throw HipError("bench_naf_chain: variant 26 is a PROBES-build test");
#endif
        }
        struct SabotageOff { ~SabotageOff() { naf_chain_set_sabotage(0); } } sabotage_off;
        run();   // warm
        if (variant == 25) {
#ifdef IRSDE_PROBES
            const int ng = naf_chain_split_groups(B, 4);
            unsigned long long* dd = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&dd, (size_t)ng * 8 * 16 * 8));
            IRSDE_HIP_CHECK(hipMemsetAsync(dd, 0, (size_t)ng * 8 * 16 * 8, s));
            naf_chain_set_debug(dd);
            run();
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            naf_chain_set_debug(nullptr);
            std::vector<unsigned long long> hd((size_t)ng * 8 * 16);
            IRSDE_HIP_CHECK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
            double acc[16] = {0};
            double nwv = 0;
            for (size_t w = 0; w < (size_t)ng * 8; ++w) {
                if (!hd[w * 16 + 15]) continue;   // (a group of an image slot past the batch)
                for (int k = 0; k < 16; ++k) acc[k] += (double)hd[w * 16 + k];
                nwv += 1;
            }
            static const char* names[12] = {"norm1 (incl. residual-stream fetch)", "conv1 GEMM passes", "depthwise 3x3 + gate + pool", "gated fetch behind barrier (pool)", "sca.1 GEMM", "conv3 GEMM + residual", "norm2 (incl. residual-stream fetch)", "conv4 GEMM + gate", "conv5 GEMM + residual", "scale-vector / gated fetch (sca, conv4)", "group barriers (6 per block)", "publishing (gated slice, residual slice)"};
            printf("naf_chain stamps, 4 groups per image: %d blocks, B=%d; shader cycles per wave and block (mean over %.0f waves)\n", nblocks, B, nwv);
            for (int k = 0; k < 12; ++k) printf("  %-42s %9.0f\n", names[k], acc[k] / nwv / nblocks);
            printf("  %-42s %9.0f\n", "whole kernel / blocks", acc[15] / nwv / nblocks);
            fflush(stdout);
            (void)hipFree(dd);
#else
            throw HipError("bench_naf_chain: variant 25 is a measurement twin (make PROBES=1)");
#endif
        }
        if (variant == 11) {   // the stamp twin once: per-phase cycle budget per block, averaged over all waves
            unsigned long long* dd = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&dd, (size_t)B * 8 * 16 * 8));
            IRSDE_HIP_CHECK(hipMemsetAsync(dd, 0, (size_t)B * 8 * 16 * 8, s));
            naf_chain_set_debug(dd);
            launch_naf_chain(dx, dout, dw, dvec, nblocks, B, dfilm, 0, 0, nullptr, 0, 0, s, 11);
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            naf_chain_set_debug(nullptr);
            std::vector<unsigned long long> hd((size_t)B * 8 * 16);
            IRSDE_HIP_CHECK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
            double acc[16] = {0};
            for (size_t i = 0; i < hd.size(); ++i) acc[i % 16] += (double)hd[i];
            const double nwv = (double)B * 8, nb = nblocks;
            static const char* names[10] = {"norm1", "conv1 GEMM passes", "depthwise 3x3 + gate + pool", "barrier (pool)", "sca.1 GEMM", "conv3 GEMM + residual", "norm2", "conv4 GEMM + gate", "conv5 GEMM + residual", "barriers (sca, conv4)"};
            printf("naf_chain stamps: %d blocks, B=%d; shader cycles per wave and block (mean over %d waves); MFMA floor per wave: conv1 / conv4 512 x 16, conv3 / conv5 256 x 16, sca 64 x 16\n", nblocks, B, B * 8);
            for (int k = 0; k < 10; ++k) printf("  %-30s %9.0f\n", names[k], acc[k] / nwv / nb);
            printf("  %-30s %9.0f\n", "whole kernel / blocks", acc[15] / nwv / nb);
            fflush(stdout);
            (void)hipFree(dd);
        }
        hipEvent_t e0, e1;
        IRSDE_HIP_CHECK(hipEventCreate(&e0));
        IRSDE_HIP_CHECK(hipEventCreate(&e1));
        IRSDE_HIP_CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) run();
        IRSDE_HIP_CHECK(hipEventRecord(e1, s));
        IRSDE_HIP_CHECK(hipStreamSynchronize(s));
        float ms = 0.f;
        IRSDE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = ms / iters;
        if (G > 1) {
            unsigned flag = 0;
            IRSDE_HIP_CHECK(hipMemcpy(&flag, naf_chain_split_error_flag(dscratch, B), 4, hipMemcpyDeviceToHost));
            (void)hipFree(dwg); (void)hipFree(dscratch);
            if (flag) throw HipError("bench_naf_chain: the split kernel's groups were not co-resident (spin timeout)");
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        (void)hipFree(dx); (void)hipFree(dout); (void)hipFree(dwf); (void)hipFree(dvec); (void)hipFree(dfilm); (void)hipFree(dw);
        (void)hipStreamDestroy(s);
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
        const int halo_force = (variant == 64 || variant == 66) ? 64 : (variant == 65 || variant == 67) ? 65 : 0;   // r05: 64 / 65 = variant 62 with the 512- / 256-pixel halo kernel forced, 66 / 67 = the same on variant 63
        if (halo_force) variant = variant <= 65 ? 62 : 63;
        if (variant >= 60 && variant <= 63) {  // bf16-MFMA mode: 60 = 256x256 tile, 61 = 128x128, 62 = automatic, 63 = automatic + bf16 activations
            IRSDE_HIP_CHECK(hipMalloc(&dbf, nw * 2));
            launch_f32_to_bf16(dw, dbf, nw, s);
            p.w_bf = dbf;
        }
        if (epi == 1) { p.film = dfilm; p.silu = 1; }
        if (epi == 2) { p.silu = 1; p.res = dres; p.res_stride = Cout; }
        unsigned short* dabf = nullptr;
        if (variant == 63) {  // bf16 activation storage (the output / residual buffers are simply twice the size needed)
            IRSDE_HIP_CHECK(hipMalloc(&dabf, nin * 2));
            launch_f32_to_bf16(din, dabf, nin, s);
            launch_f32_to_bf16(dout, reinterpret_cast<unsigned short*>(dres), nout / 2, s);
            p.in0 = reinterpret_cast<const float*>(dabf);
            p.in_bf16 = 1; p.out_bf16 = 1;
        }
        // 80: fused Winograd F(4x4,3x3) kernel; 81: the three-launch Winograd F(4x4,3x3) path (random U: timing only)
        float *dU = nullptr, *dV = nullptr, *dM = nullptr;
        WinoPlan wp{};
        unsigned short* dwp = nullptr;
        if (variant == 480 || variant == 481 || variant == 482) {   // direct convolution on the PAIR kernels: 480 fp16 pieces, 481 bf16 pieces, 482 = 480 without the 256 x 256 tile
            IRSDE_HIP_CHECK(hipMalloc(&dwp, nw * 4));
            launch_split_pairs(dw, dwp, (size_t)Cout, K * K * Cin, s, variant != 481, 64.0f);
            p.w_pair = dwp; p.pair_scale = 1.0f / 64.0f; p.f16 = variant != 481 ? 1 : 0;
            variant = variant == 482 ? 61 : 0;
        }
        if (variant >= 472 && variant <= 476) {   // the pair-interleaved two-plane component GEMMs alone (v3 kernel): 472 full, 473 no loads, 475 no MFMAs, 476 no output stores
            if (K != 3 || stride != 1 || !wino_shape_ok(p, 4)) throw HipError("bench_conv: Winograd variants need an eligible 3x3 stride-1 layer");
            const long long T = (long long)B * (p.Ho / 4) * (p.Wo / 4);
            float *vf = nullptr, *uf = nullptr, *mo = nullptr;
            unsigned short *vp = nullptr, *up = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&vf, (size_t)36 * T * Cin * 4));
            IRSDE_HIP_CHECK(hipMalloc(&uf, (size_t)36 * Cout * Cin * 4));
            IRSDE_HIP_CHECK(hipMalloc(&mo, (size_t)36 * T * Cout * 4));
            IRSDE_HIP_CHECK(hipMalloc(&vp, (size_t)36 * T * Cin * 4));
            IRSDE_HIP_CHECK(hipMalloc(&up, (size_t)36 * Cout * Cin * 4));
            launch_fill_random(vf, (size_t)36 * T * Cin, 7, 1.0f, s);
            launch_fill_random(uf, (size_t)36 * Cout * Cin, 8, 0.05f, s);
            launch_split_pairs(vf, vp, (size_t)36 * T, Cin, s);
            launch_split_pairs(uf, up, (size_t)36 * Cout, Cin, s);
            SplitGemmArgs gp;
            gp.a = vp; gp.b = up; gp.out = mo;
            gp.pA = T * Cin; gp.pB = (long long)Cout * Cin; gp.pO = T * Cout;
            gp.M = (int)T; gp.N = Cout; gp.K = Cin; gp.lda = Cin; gp.ldc = Cout;
            const int abl = variant - 472;
            hipEvent_t e0, e1;
            IRSDE_HIP_CHECK(hipEventCreate(&e0));
            IRSDE_HIP_CHECK(hipEventCreate(&e1));
            for (int i = 0; i < 2; ++i) launch_gemm_split_pairs(gp, 36, s, abl);
            IRSDE_HIP_CHECK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) launch_gemm_split_pairs(gp, 36, s, abl);
            IRSDE_HIP_CHECK(hipEventRecord(e1, s));
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            float ms = 0;
            IRSDE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            *ms_out = ms / iters;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            (void)hipFree(vf); (void)hipFree(uf); (void)hipFree(mo); (void)hipFree(vp); (void)hipFree(up);
            (void)hipFree(din); (void)hipFree(dw); (void)hipFree(dout); (void)hipFree(dres); (void)hipFree(dfilm);
            (void)hipStreamDestroy(s);
            return;
        }
        const bool split_v = variant == 412 || variant == 413 || variant == 422 || variant == 423;  // split-operand GEMMs: 41x whole three-launch layer, 42x the GEMM alone; x = planes
        unsigned short *dUs = nullptr, *dVs = nullptr;
        WinoSplitPlan sp{};
        if (variant == 80 || variant == 81 || variant == 421 || split_v || (variant >= 83 && variant <= 82 + 255) || (variant >= 400 && variant <= 410) || (variant >= 430 && variant <= 435) || (variant >= 440 && variant <= 454) || (variant >= 460 && variant <= 469) || (variant >= 4610 && variant <= 4613) || (variant >= 4650 && variant <= 4653) || (variant >= 4700 && variant < 4800) || (variant >= 1000 && variant < 1064) || (variant >= 2000 && variant <= 2004) || (variant >= 2010 && variant <= 2012) || variant == 2020) {
            if (K != 3 || stride != 1) throw HipError("bench_conv: Winograd variants need a 3x3 stride-1 layer");
            IRSDE_HIP_CHECK(hipMalloc(&dU, (size_t)36 * nw / 9 * 4));
            launch_fill_random(dU, (size_t)36 * nw / 9, 5, 1.0f / sqrtf((float)(9 * Cin)), s);
            if (variant != 81 && variant != 421 && !split_v && !wino_fused_eligible(p)) throw HipError("bench_conv: shape not eligible for the fused Winograd kernel");
            if (variant >= 400 && !wino_fused64_eligible(p)) throw HipError("bench_conv: shape not eligible for the 64-cout fused Winograd kernel");
            if (variant == 404 || variant == 405 || variant == 434 || variant == 444 || variant == 452) {  // the fp16-pair twin: the random weights as hi / lo halves
                float* dUp = nullptr;
                IRSDE_HIP_CHECK(hipMalloc(&dUp, (size_t)36 * nw / 9 * 4));
                launch_wino_fused64_split_weights(dU, reinterpret_cast<unsigned short*>(dUp), (size_t)36 * nw / 9, 256.0f, s);
                IRSDE_HIP_CHECK(hipStreamSynchronize(s));
                (void)hipFree(dU);
                dU = dUp;
                p.pair_scale = 1.0f / (kWinoFused64PairVScale * 256.0f);
            }
            if (variant == 81 || variant == 421) {
                if (!wino_shape_ok(p, 4)) throw HipError("bench_conv: shape not eligible for Winograd F(4x4,3x3)");
                const long long T = (long long)B * (p.Ho / 4) * (p.Wo / 4);
                IRSDE_HIP_CHECK(hipMalloc(&dV, (size_t)36 * T * Cin * 4));
                IRSDE_HIP_CHECK(hipMalloc(&dM, (size_t)36 * T * Cout * 4));
                wp = make_wino(p, dU, dV, dM, 4);
                if (variant == 421) launch_wino_input(wp.in, s);
            }
            if (split_v) {
                if (!wino_shape_ok(p, 4)) throw HipError("bench_conv: shape not eligible for Winograd F(4x4,3x3)");
                const int npl = variant % 10;
                const long long T = (long long)B * (p.Ho / 4) * (p.Wo / 4);
                const size_t nu = (size_t)36 * Cout * Cin;
                IRSDE_HIP_CHECK(hipMalloc(&dUs, nu * 2 * npl));
                IRSDE_HIP_CHECK(hipMalloc(&dVs, (size_t)36 * T * Cin * 2 * npl));
                IRSDE_HIP_CHECK(hipMalloc(&dM, (size_t)36 * T * Cout * 4));
                launch_split_planes(dU, dUs, nu, nu, npl, s);
                sp = make_wino_split(p, dUs, dVs, dM, npl);
                launch_wino_input(sp.in, s);
            }
        }
        if (variant == 82) {  // fused Winograd kernel once, with per-wave phase stamps: prints the averaged timeline
            IRSDE_HIP_CHECK(hipMalloc(&dU, (size_t)36 * nw / 9 * 4));
            launch_fill_random(dU, (size_t)36 * nw / 9, 5, 1.0f / sqrtf((float)(9 * Cin)), s);
            const int nb = wino_fused_num_blocks(p);
            unsigned long long* dd = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&dd, (size_t)nb * 128 * 8));
            launch_wino_fused(p, dU, s);  // warm
            IRSDE_HIP_CHECK(hipMemsetAsync(dd, 0, (size_t)nb * 128 * 8, s));
            launch_wino_fused(p, dU, s, dd);
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            std::vector<unsigned long long> hd((size_t)nb * 128);
            IRSDE_HIP_CHECK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
            double ph[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            unsigned long long rt_min = ~0ull, rt_max = 0;
            for (int bi = 0; bi < nb; ++bi)
                for (int w = 0; w < 8; ++w) {
                    const unsigned long long* t = &hd[((size_t)bi * 8 + w) * 16];
                    for (int k = 0; k < 4; ++k) ph[w >= 4][k] += (double)(t[k + 1] - t[k]) / ((double)nb * 4);
                    rt_min = std::min(rt_min, t[7]);
                    rt_max = std::max(rt_max, t[7]);
                }
            // block start times (100 MHz realtime counter) -> how long the launch kept dispatching new blocks
            printf("wino_fused timeline B=%d %dx%d Cin=%d Cout=%d: %d blocks, block starts span %.1f us\n", B, p.Ho, p.Wo, Cin, Cout, nb,
                   (double)(rt_max - rt_min) / 100.0);
            printf("  MFMA waves     (shader cycles): start->V[0] ready %.0f | K loop %.0f | acc->LDS+barrier %.0f | epilogue %.0f\n", ph[0][0],
                   ph[0][1], ph[0][2], ph[0][3]);
            printf("  producer waves (shader cycles): start->chunk 0 done %.0f | K loop rest %.0f | wait MFMA+acc %.0f | epilogue %.0f\n",
                   ph[1][0], ph[1][1], ph[1][2], ph[1][3]);
            fflush(stdout);
            (void)hipFree(dd);
            (void)hipFree(dU);
            dU = nullptr;
            *ms_out = 0.0;
            (void)hipFree(din); (void)hipFree(dw); (void)hipFree(dout); (void)hipFree(dres); (void)hipFree(dfilm);
            (void)hipStreamDestroy(s);
            return;
        }
        if (variant == 465 || (variant >= 4650 && variant <= 4653)) {   // r06: the two-tile-group kernel once with per-wave cycle stamps (4650 / 4651 / 4652: no patch traffic / output stores dropped / residual loads dropped)
            if (!wino_fused64t_eligible(p)) throw HipError("bench_conv: shape not eligible for the two-tile-group fused Winograd kernel");
            const int nbp = 256;
            unsigned long long* dd = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&dd, (size_t)nbp * 64 * 8));
            launch_wino_fused64t(p, dU, s, 0);  // warm
            IRSDE_HIP_CHECK(hipMemsetAsync(dd, 0, (size_t)nbp * 64 * 8, s));
            wino_fused64t_set_debug(dd);
            launch_wino_fused64t(p, dU, s, variant == 465 ? 5 : 12 + (variant - 4650));
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            wino_fused64t_set_debug(nullptr);
            std::vector<unsigned long long> hd((size_t)nbp * 64);
            IRSDE_HIP_CHECK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
            double a5[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int nwv = 0;
            for (int bi = 0; bi < nbp; ++bi)
                for (int w = 0; w < 8; ++w) {
                    const unsigned long long* t = &hd[((size_t)bi * 8 + w) * 8];
                    if (!t[3]) continue;
                    for (int k = 0; k < 8; ++k) a5[k] += (double)t[k];
                    nwv++;
                }
            const int nst = Cin / 16;
            const double items = a5[4] / std::max(nwv, 1), chunks = items * nst;
            printf("wino4_fused64t stamps B=%d %dx%d Cin=%d Cout=%d: %.1f items x %d chunks of 16 channels per block; shader cycles per wave (mean over %d waves)\n", B, p.Ho, p.Wo, Cin,
                   Cout, items, nst, nwv);
            printf("  kernel %.0f = K loops incl. transform slices %.0f (%.0f per chunk; MFMA floor per SIMD 9216) + chunk barrier waits %.0f (%.0f per chunk) + epilogue, exchange, first-chunk transform %.0f (%.0f per item)\n",
                   a5[3] / nwv, a5[0] / nwv, a5[0] / nwv / chunks, a5[1] / nwv, a5[1] / nwv / chunks, (a5[2] + a5[5] + a5[6] + a5[7]) / nwv, (a5[2] + a5[5] + a5[6] + a5[7]) / nwv / items);
            printf("  per item: first stage + exchange writes + ring %.0f | exchange barriers + reads %.0f | second stage, stores, gathers %.0f | zero, first-chunk transform, barrier %.0f\n",
                   a5[5] / nwv / items, a5[6] / nwv / items, a5[7] / nwv / items, a5[2] / nwv / items);
            fflush(stdout);
            (void)hipFree(dd);
            (void)hipFree(dU);
            dU = nullptr;
            *ms_out = 0.0;
            (void)hipFree(din); (void)hipFree(dw); (void)hipFree(dout); (void)hipFree(dres); (void)hipFree(dfilm);
            (void)hipStreamDestroy(s);
            return;
        }
        if (variant == 435 || (variant >= 2000 && variant <= 2004) || (variant >= 2010 && variant <= 2012) || variant == 2020) {  // the persistent fused Winograd kernel once with per-wave cycle stamps: prints the averaged budget
            // 2000 + k: the stamp twins (k = 0 every r04 OPT bit, 1 no weight traffic, 2 no patch traffic, 3 patches from an L2-resident window, 4 = 435)
            // 2010 + k: the halo kernel's stamp twins (k = 0 production, 1 no weight traffic, 2 no halo traffic)
            const bool halo = variant >= 2010 && variant <= 2012;
            const bool single = variant == 2020;   // the single-stream kernel: 4 waves per block, every wave both roles
            const int stamp_variant = single ? 53 : halo ? 45 + (variant - 2010) : variant == 435 || variant == 2004 ? 25 : 27 + (variant - 2000);
            const int nbp = 256;
            unsigned long long* dd = nullptr;
            IRSDE_HIP_CHECK(hipMalloc(&dd, (size_t)nbp * 64 * 8));
            launch_wino_fused64(p, dU, s, 20);  // warm
            IRSDE_HIP_CHECK(hipMemsetAsync(dd, 0, (size_t)nbp * 64 * 8, s));
            wino_fused64_set_debug(dd);
            launch_wino_fused64(p, dU, s, stamp_variant);
            IRSDE_HIP_CHECK(hipStreamSynchronize(s));
            wino_fused64_set_debug(nullptr);
            std::vector<unsigned long long> hd((size_t)nbp * 64);
            IRSDE_HIP_CHECK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
            double acc[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
            int nw[2] = {0, 0};
            for (int bi = 0; bi < nbp; ++bi)
                for (int w = 0; w < 8; ++w) {
                    const unsigned long long* t = &hd[((size_t)bi * 8 + w) * 8];
                    if (!t[3]) continue;
                    for (int k = 0; k < 5; ++k) acc[w >= 4][k] += (double)t[k];
                    nw[w >= 4]++;
                }
            if (single) {
                const int nst = Cin / 32;
                const double items = acc[0][4] / std::max(nw[0], 1);
                const double chunks = items * nst;
                printf("wino4_fused64s stamps B=%d %dx%d Cin=%d Cout=%d: %.1f items x %d chunks per block; shader cycles per wave (mean over %d waves)\n", B, p.Ho, p.Wo, Cin, Cout, items, nst, nw[0]);
                printf("  kernel %.0f = K loop incl. transform slices %.0f (%.0f per chunk; MFMA floor 9216) + barrier wait %.0f (%.0f per chunk) + epilogue %.0f (%.0f per item)\n",
                       acc[0][3] / nw[0], acc[0][0] / nw[0], acc[0][0] / nw[0] / chunks, acc[0][1] / nw[0], acc[0][1] / nw[0] / chunks, acc[0][2] / nw[0], acc[0][2] / nw[0] / items);
            } else
            if (halo) {   // per 16-channel step; producer columns: DMA issue, transform, barrier wait (both kinds of step), halo wait (off step)
                double pa[6] = {0, 0, 0, 0, 0, 0};
                int pn = 0;
                for (int bi = 0; bi < nbp; ++bi)
                    for (int w = 4; w < 8; ++w) {
                        const unsigned long long* t = &hd[((size_t)bi * 8 + w) * 8];
                        if (!t[3]) continue;
                        for (int k = 0; k < 6; ++k) pa[k] += (double)t[k];
                        pn++;
                    }
                const int nst = Cin / 16;
                const double items = acc[0][4] / std::max(nw[0], 1);
                const double steps = items * nst;
                printf("wino4_fused64h stamps B=%d %dx%d Cin=%d Cout=%d: %.1f items x %d steps of 16 channels per block; shader cycles per wave (mean over %d + %d waves)\n", B, p.Ho,
                       p.Wo, Cin, Cout, items, nst, nw[0], pn);
                printf("  MFMA waves    : kernel %.0f = K-loop compute %.0f (%.0f per step; MFMA floor 4608) + barrier wait %.0f (%.0f per step) + epilogue %.0f (%.0f per item)\n",
                       acc[0][3] / nw[0], acc[0][0] / nw[0], acc[0][0] / nw[0] / steps, acc[0][1] / nw[0], acc[0][1] / nw[0] / steps, acc[0][2] / nw[0], acc[0][2] / nw[0] / items);
                printf("  producer waves: kernel %.0f = halo DMA issue %.0f (%.0f per DMA step) + transform %.0f (%.0f per own step) + halo wait in the off step %.0f (%.0f per off step) + barrier wait %.0f (%.0f per step)\n",
                       pa[3] / pn, pa[0] / pn, pa[0] / pn / (steps / 2), pa[1] / pn, pa[1] / pn / (steps / 2), pa[5] / pn, pa[5] / pn / (steps / 2), pa[2] / pn, pa[2] / pn / steps);
                fflush(stdout);
                (void)hipFree(dd);
                (void)hipFree(dU);
                dU = nullptr;
                *ms_out = 0.0;
                (void)hipFree(din); (void)hipFree(dw); (void)hipFree(dout); (void)hipFree(dres); (void)hipFree(dfilm);
                (void)hipStreamDestroy(s);
                return;
            }
            const int nchk = Cin / 32;
            const double items = acc[0][4] / std::max(nw[0], 1), chunks = acc[1][4] / std::max(nw[1], 1);
            printf("wino4_fused64p stamps B=%d %dx%d Cin=%d Cout=%d: %.1f items x %d chunks per block; shader cycles per wave (mean over %d + %d waves)\n", B, p.Ho, p.Wo,
                   Cin, Cout, items, nchk, nw[0], nw[1]);
            printf("  MFMA waves    : kernel %.0f = K-loop compute %.0f (%.0f per chunk; MFMA floor 9216) + barrier wait %.0f (%.0f per chunk) + epilogue %.0f (%.0f per item)\n",
                   acc[0][3] / nw[0], acc[0][0] / nw[0], acc[0][0] / nw[0] / (items * nchk), acc[0][1] / nw[0], acc[0][1] / nw[0] / (items * nchk), acc[0][2] / nw[0],
                   acc[0][2] / nw[0] / items);
            printf("  producer waves: kernel %.0f = load issue %.0f (%.0f per chunk) + data wait, transform, LDS writes %.0f (%.0f per chunk) + barrier wait %.0f (%.0f per chunk)\n",
                   acc[1][3] / nw[1], acc[1][0] / nw[1], acc[1][0] / nw[1] / chunks, acc[1][1] / nw[1], acc[1][1] / nw[1] / chunks, acc[1][2] / nw[1], acc[1][2] / nw[1] / chunks);
            fflush(stdout);
            (void)hipFree(dd);
            (void)hipFree(dU);
            dU = nullptr;
            *ms_out = 0.0;
            (void)hipFree(din); (void)hipFree(dw); (void)hipFree(dout); (void)hipFree(dres); (void)hipFree(dfilm);
            (void)hipStreamDestroy(s);
            return;
        }
        auto run = [&] {
            if (variant == 80) {
                launch_wino_fused(p, dU, s);
            } else if (variant >= 400 && variant <= 410) {  // 406 / 407 / 408: non-temporal epilogue traffic / + patch loads / no hint at all
                 // 64-cout fused Winograd kernel: 400 production, 401 no weight traffic, 402 no patch traffic, 403 short U ring, 404 / 405 fp16 pairs (ring 12 / 18)
                launch_wino_fused64(p, dU, s, variant - 400);
            } else if (variant >= 4700 && variant < 4800) {  // r06 tuning: the two-tile-group kernel with a start skew of (variant - 4700) x 1000 cycles per phase class
                wino_fused64t_set_skew((variant - 4700) * 1000);
                launch_wino_fused64t(p, dU, s, 0);
                wino_fused64t_set_skew(0);
            } else if (variant == 4613) {   // the residual tile gathered into registers instead of through LDS (residual layers)
                launch_wino_fused64t(p, dU, s, 16);
            } else if (variant >= 4610 && variant <= 4612) {  // 8-byte twins of the two-tile-group kernel: 4610 residual loads, 4611 output stores, 4612 weight units
                launch_wino_fused64t(p, dU, s, variant - 4601);
            } else if (variant >= 466 && variant <= 469) {  // 466 no non-temporal hint; 467 / 468 / 469 measurement twins: no transform arithmetic / + no gathers / no gathers only
                launch_wino_fused64t(p, dU, s, variant == 466 ? 4 : variant - 461);
            } else if (variant >= 460 && variant <= 463) {  // r06 two-tile-group kernel: 460 production, 461 / 462 weight fragments / patch gathers read zeros, 463 three weight units in flight
                launch_wino_fused64t(p, dU, s, variant - 460);
            } else if (variant >= 448 && variant <= 454) {  // r04 single-stream kernel: 448 f32, 450 patch loads read zeros, 452 fp16 pairs; 449 / 451 / 454 measurement twins
                launch_wino_fused64(p, dU, s, variant - 400);
            } else if (variant >= 440 && variant <= 444) {  // r04 halo kernel: 440 production, 441 / 442 weight fragments / halo fetches read zeros, 444 fp16 pairs
                launch_wino_fused64(p, dU, s, variant - 400);
            } else if (variant >= 1000 && variant < 1064) {  // r04 tuning twins of the persistent kernel: OPT = variant - 1000 (see wino4_fused64p_kernel)
                wino_fused64_set_opt(variant - 1000);
                launch_wino_fused64(p, dU, s, 26);
            } else if (variant >= 430 && variant <= 434) {  // r04 persistent kernel: 430 production, 431 / 432 weight fragments / patch loads read zeros, 433 no nt hint, 434 fp16 pairs
                launch_wino_fused64(p, dU, s, variant - 410);
            } else if (variant >= 83 && variant <= 82 + 255) {  // tuning aids: dflags = variant - 82 (1 no patch traffic, 2 no weight traffic, 4 / 8 producer / MFMA waves at s_setprio 2)
                launch_wino_fused(p, dU, s, nullptr, variant - 82);
            } else if (variant == 81) {
                launch_wino_input(wp.in, s);
                launch_conv(wp.gemm, s);
                launch_wino_output(wp.out, s);
            } else if (variant == 421) {   // the f32 component GEMMs alone
                launch_conv(wp.gemm, s);
            } else if (variant == 412 || variant == 413) {
                launch_wino_input(sp.in, s);
                launch_gemm_split(sp.gemm, sp.nplanes, 36, s);
                launch_wino_output(sp.out, s);
            } else if (variant == 422 || variant == 423) {   // the split-operand component GEMMs alone
                launch_gemm_split(sp.gemm, sp.nplanes, 36, s);
            } else {
                launch_conv(p, s);
            }
        };
        VariantScope vs(halo_force ? halo_force : variant >= 80 || variant == 63 || variant == 62 ? 0 : variant);
        hipEvent_t e0, e1;
        IRSDE_HIP_CHECK(hipEventCreate(&e0));
        IRSDE_HIP_CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 2; ++i) run();
        IRSDE_HIP_CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) run();
        IRSDE_HIP_CHECK(hipEventRecord(e1, s));
        IRSDE_HIP_CHECK(hipStreamSynchronize(s));
        float ms = 0;
        IRSDE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = ms / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        (void)hipFree(din); (void)hipFree(dw); (void)hipFree(dout); (void)hipFree(dres); (void)hipFree(dfilm);
        if (dbf) (void)hipFree(dbf);
        if (dabf) (void)hipFree(dabf);
        if (dwp) (void)hipFree(dwp);
        for (float* q : {dU, dV, dM})
            if (q) (void)hipFree(q);
        if (dUs) (void)hipFree(dUs);
        if (dVs) (void)hipFree(dVs);
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
        apply_fp16_flag(e);
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
        if (!e || e->arch != 2 || !x || !latent) throw HipError("latent_encode: not a latent UNet engine / null argument");
        if (!e->finalized) throw HipError("latent_encode: weights not finalized");
        if (B < 1 || H < 2 || W < 2) throw HipError("latent_encode: bad shape");
        std::lock_guard<std::mutex> lk(e->mu);
        DeviceScope dev_scope(e->cfg.device);
        hipStream_t user = reinterpret_cast<hipStream_t>(stream), s = e->stream;
        LatentPlan* lp = get_latent_plan(e, B, H, W, false);
        Plan* pl = lp->plan.get();
        IRSDE_HIP_CHECK(hipEventRecord(e->ev_in, user));
        IRSDE_HIP_CHECK(hipStreamWaitEvent(s, e->ev_in, 0));
        launch_nchw_to_nhwc_pad(x, lp->image.p, B, e->lat_in, H, W, pl->Hp, pl->Wp, lp->image.C, 1, s);  // F.pad 'reflect'
        run_net(pl, s);
        const Tensor& L = lp->latent;
        launch_unpack_pred(L.p, latent, B, e->lat_embed, L.H, L.W, L.H, L.W, L.C, s);
        lp->resident = true;   // the skips stay in place for irsde_latent_decode(hidden = NULL) / irsde_latent_hidden
        for (size_t k = 0; hidden && k < lp->hidden.size(); ++k) {
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
        if (!e || e->arch != 2 || !latent || !out) throw HipError("latent_decode: not a latent UNet engine / null argument");
        if (!e->finalized) throw HipError("latent_decode: weights not finalized");
        std::lock_guard<std::mutex> lk(e->mu);
        DeviceScope dev_scope(e->cfg.device);
        hipStream_t user = reinterpret_cast<hipStream_t>(stream), s = e->stream;
        if (!hidden) {   // residency first: a call that is going to be refused must not build (or evict) any plan
            bool res = false;
            for (auto& q : e->lat_plans)
                if (!q->decode && q->plan->B == B && q->plan->H == H && q->plan->W == W) res = q->resident;
            if (!res) throw HipError("latent_decode: hidden == NULL needs a preceding irsde_latent_encode of the same B x H x W on this engine");
        }
        LatentPlan* lp = get_latent_plan(e, B, H, W, true);
        Plan* pl = lp->plan.get();
        LatentPlan* enc = get_latent_plan(e, B, H, W, false);   // (exists: the decode plan was built on it)
        if (hidden) enc->resident = false;   // the caller's skips overwrite the shared storage
        IRSDE_HIP_CHECK(hipEventRecord(e->ev_in, user));
        IRSDE_HIP_CHECK(hipStreamWaitEvent(s, e->ev_in, 0));
        const Tensor& L = lp->latent;
        launch_nchw_to_nhwc_pad(latent, L.p, B, e->lat_embed, L.H, L.W, L.H, L.W, L.C, 0, s);
        for (size_t k = 0; hidden && k < lp->hidden.size(); ++k) {
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

int irsde_latent_hidden(irsde_engine* e, int B, int H, int W, int k, float* out, void* stream) {
    return guard([&] {
        if (!e || e->arch != 2 || !out) throw HipError("latent_hidden: not a latent UNet engine / null argument");
        std::lock_guard<std::mutex> lk(e->mu);
        DeviceScope dev_scope(e->cfg.device);
        hipStream_t user = reinterpret_cast<hipStream_t>(stream), s = e->stream;
        LatentPlan* enc = nullptr;
        for (auto& lp : e->lat_plans)
            if (!lp->decode && lp->plan->B == B && lp->plan->H == H && lp->plan->W == W) enc = lp.get();
        if (!enc || !enc->resident) throw HipError("latent_hidden: no resident skips of an irsde_latent_encode with this B x H x W");
        if (k < 0 || k >= (int)enc->hidden.size()) throw HipError("latent_hidden: bad skip index");
        IRSDE_HIP_CHECK(hipEventRecord(e->ev_in, user));
        IRSDE_HIP_CHECK(hipStreamWaitEvent(s, e->ev_in, 0));
        const Tensor& h = enc->hidden[k];
        launch_unpack_pred(h.p, out, B, enc->hidden_c[k], h.H, h.W, h.H, h.W, h.C, s);
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
        DeviceScope dev_scope(e->cfg.device);
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
