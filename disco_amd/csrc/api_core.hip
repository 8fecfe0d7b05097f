// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): context lifetime, tuning, stage timers, device-memory helpers
#include "host.h"

using namespace disco;
using namespace disco_host;

static char g_create_err[512] = "";

// (tests/hipemu compiles these sources with g++ for logic tests and defines HIPEMU: the string must not claim a GPU there)
#ifdef HIPEMU
extern "C" const char* disco_version(void) { return "disco_hip 0.3.0 (hipemu host TEST build, not a product)"; }
#else
extern "C" const char* disco_version(void) { return "disco_hip 0.3.0 (gfx950)"; }
#endif

extern "C" const char* disco_last_error(const disco_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

extern "C" int disco_create(disco_ctx** out, const disco_cfg* cfg) {
    if (!out || !cfg) {
        snprintf(g_create_err, sizeof(g_create_err), "disco_create: null argument");
        return DISCO_E_ARG;
    }
    *out = nullptr;
    if (cfg->rooms < 1 || cfg->nodes < 1 || cfg->mics < 1 || cfg->length < 1 || cfg->mask_pow < 0 ||
        cfg->ref_mic < 0 || cfg->ref_mic >= cfg->mics) {
        snprintf(g_create_err, sizeof(g_create_err), "disco_create: rooms/nodes/mics/length/ref_mic out of range");
        return DISCO_E_ARG;
    }
    if (cfg->n_fft != 512 && cfg->n_fft != 1024) {
        snprintf(g_create_err, sizeof(g_create_err), "disco_create: n_fft must be 512 or 1024");
        return DISCO_E_UNSUPPORTED;
    }
    if (cfg->hop * 2 != cfg->n_fft) {
        snprintf(g_create_err, sizeof(g_create_err), "disco_create: hop must equal n_fft/2");
        return DISCO_E_UNSUPPORTED;
    }
    if (cfg->pad_mode == DISCO_PAD_REFLECT && cfg->length <= cfg->n_fft / 2) {
        snprintf(g_create_err, sizeof(g_create_err), "disco_create: reflect padding needs length > n_fft/2");
        return DISCO_E_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || cfg->device < 0 || cfg->device >= ndev) {
        snprintf(g_create_err, sizeof(g_create_err), "disco_create: no HIP device %d (found %d)", cfg->device, ndev);
        return DISCO_E_HIP_BASE;
    }
    disco_ctx* ctx = new (std::nothrow) disco_ctx();
    if (!ctx) return DISCO_E_ARG;
    ctx->cfg = *cfg;
    ctx->T = 1 + cfg->length / cfg->hop;
    ctx->F = cfg->n_fft / 2 + 1;
    ctx->d_win = nullptr;
    ctx->d_tw = nullptr;
    ctx->own_ws = nullptr;
    ctx->own_ws_bytes = 0;
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    ctx->pending_chunks = 0;
    ctx->pending_P = 0;
    ctx->k0 = 0;
    ctx->Kl = cfg->nodes;
    ctx->zblk = cfg->nodes;
    ctx->geom_rooms = cfg->rooms;
    ctx->half[0] = ctx->half[1] = nullptr;
    ctx->parent = nullptr;
    ctx->side_stream = nullptr;
    ctx->ev_fork = ctx->ev_join = nullptr;
    ctx->tune_runw = ctx->tune_cov_chunks = ctx->tune_step2_chunks = ctx->tune_pairs = 0;
    // per-context options (disco_set_option); the environment may preset them, and is read HERE only -- never inside a compute call
    for (int i = 0; i < DISCO_N_OPTIONS; ++i) {
        const char* e = getenv(disco_host::option_table()[i].env);
        ctx->opt[i] = e ? atoi(e) : disco_host::option_table()[i].def;
    }
    ctx->stage_on = false;
    ctx->scratch2 = nullptr;
    ctx->scratch2_bytes = 0;
    ctx->loc_chunks = 0;
    ctx->loc_M = 0;
    ctx->loc_X = ctx->loc_mask = nullptr;
    ctx->pending_skiploc = 0;
    ctx->ref_ws = nullptr;
    ctx->ref_y = ctx->ref_s = ctx->ref_n = nullptr;
    ctx->d_tw_conv = nullptr;
    ctx->conv_ws = nullptr;
    ctx->conv_ws_bytes = 0;
    ctx->err[0] = 0;
    const int N = cfg->n_fft;
    std::vector<float> win(N);
    std::vector<c32> tw(N);
    const double two_pi = 6.283185307179586476925286766559;
    for (int i = 0; i < N; ++i) {
        win[i] = (float)(0.5 - 0.5 * std::cos(two_pi * i / N));          // scipy get_window('hann', N, fftbins=True)
        tw[i].x = (float)std::cos(two_pi * i / N);
        tw[i].y = (float)(-std::sin(two_pi * i / N));
    }
    DevGuard dev_guard_(cfg->device);                     // the caller's current device is restored on return
    hipError_t e = dev_guard_.ok ? hipSuccess : hipErrorInvalidValue;
    ctx->n_cu = 0;
    if (e == hipSuccess) e = hipDeviceGetAttribute(&ctx->n_cu, hipDeviceAttributeMultiprocessorCount, cfg->device);
    if (ctx->n_cu < 1) ctx->n_cu = 1;
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_win, N * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_tw, N * sizeof(c32));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_win, win.data(), N * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_tw, tw.data(), N * sizeof(c32), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        snprintf(g_create_err, sizeof(g_create_err), "disco_create: HIP error %s", hipGetErrorString(e));
        disco_destroy(ctx);
        return DISCO_E_HIP_BASE - (int)e;
    }
    // the partial-sum blocks are sized here, so that no compute call allocates (or implicitly synchronises) on first use
    if (!(cfg->flags & DISCO_FLAG_LAZY_SCRATCH)) {
        const int rc = reserve_scratch(ctx);
        if (rc) {
            snprintf(g_create_err, sizeof(g_create_err), "disco_create: %.480s", ctx->err);
            disco_destroy(ctx);
            return rc;
        }
    }
    // large batches: the two half-batch children of the overlapped whole-path calls (their partial-sum blocks are sized now, too)
    if (!(cfg->flags & DISCO_FLAG_NO_CHILDREN) && ctx->opt[DISCO_OPT_OVERLAP_SOLVES]) {
        const int rc = disco_host::ensure_halves(ctx);
        if (rc) {
            snprintf(g_create_err, sizeof(g_create_err), "disco_create: %.480s", ctx->err);
            disco_destroy(ctx);
            return rc;
        }
    }
    *out = ctx;
    return 0;
}

namespace disco_host {
// The overlapped form applies to batches that still fill the chip when halved (rooms x nodes >= 1024; option values 2 / 3 force it
// for any batch of >= 2 rooms: tests), when every node of a room is here.
bool overlap_applies(const disco_ctx* ctx) {
    const int o = ctx->opt[DISCO_OPT_OVERLAP_SOLVES];
    if (!o || ctx->parent || ctx->cfg.rooms < 2 || sharded(ctx)) return false;
    return o >= 2 || (long long)ctx->cfg.rooms * ctx->cfg.nodes >= 1024;
}

int ensure_halves(disco_ctx* ctx) {
    if (!overlap_applies(ctx) || ctx->half[0]) return 0;
    const int ra = ctx->cfg.rooms / 2, rb = ctx->cfg.rooms - ra;
    // both children or none: a failure part-way (out of memory for the second child's blocks) must not leave half[0] set and
    // half[1] NULL -- the next call would return early on half[0] and the overlapped route would dereference the missing one
    auto drop = [ctx](int rc, const char* msg) {
        char keep[sizeof(ctx->err)];
        snprintf(keep, sizeof(keep), "%s", msg);
        for (int h = 0; h < 2; ++h) {
            if (ctx->half[h]) disco_destroy(ctx->half[h]);
            ctx->half[h] = nullptr;
        }
        return fail(ctx, rc, keep);
    };
    for (int h = 0; h < 2; ++h) {
        disco_cfg c = ctx->cfg;
        c.rooms = h ? rb : ra;
        c.flags |= DISCO_FLAG_NO_CHILDREN | DISCO_FLAG_LAZY_SCRATCH;
        disco_ctx* ch = nullptr;
        const int rc = disco_create(&ch, &c);
        if (rc) return drop(rc, disco_last_error(nullptr));
        ch->parent = ctx;
        ch->geom_rooms = ctx->cfg.rooms;                   // the launch geometry of the whole batch
        ch->tune_runw = ctx->tune_runw;
        ch->tune_cov_chunks = ctx->tune_cov_chunks;
        ch->tune_step2_chunks = ctx->tune_step2_chunks;
        ch->tune_pairs = ctx->tune_pairs;
        for (int i = 0; i < DISCO_N_OPTIONS; ++i) ch->opt[i] = ctx->opt[i];
        ch->opt[DISCO_OPT_OVERLAP_SOLVES] = 0;
        ch->stage_on = ctx->stage_on;
        ctx->half[h] = ch;
        if (!(ctx->cfg.flags & DISCO_FLAG_LAZY_SCRATCH)) {
            const int rs = reserve_scratch(ch);
            if (rs) return drop(rs, ch->err);
        }
    }
    if (!ctx->side_stream) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    return 0;
}
}  // namespace disco_host

extern "C" void disco_destroy(disco_ctx* ctx) {
    if (!ctx) return;
    DevGuard dev_guard_(ctx->cfg.device);
    for (int h = 0; h < 2; ++h) disco_destroy(ctx->half[h]);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    stage_clear(ctx);
    if (ctx->d_win) (void)hipFree(ctx->d_win);
    if (ctx->d_tw) (void)hipFree(ctx->d_tw);
    if (ctx->own_ws) (void)hipFree(ctx->own_ws);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->scratch2) (void)hipFree(ctx->scratch2);
    if (ctx->d_tw_conv && ctx->d_tw_conv != ctx->d_tw) (void)hipFree(ctx->d_tw_conv);
    if (ctx->conv_ws) (void)hipFree(ctx->conv_ws);
    delete ctx;
}

extern "C" int disco_set_node_shard(disco_ctx* ctx, int first_node, int node_count) {
    DISCO_ENTER(ctx);
    if (first_node < 0 || node_count < 1 || first_node + node_count > ctx->cfg.nodes)
        return fail(ctx, DISCO_E_ARG, "disco_set_node_shard: shard outside [0, nodes)");
    ctx->k0 = first_node;
    ctx->Kl = node_count;
    ctx->pending_chunks = 0;
    return 0;
}

extern "C" int disco_stage_timing(disco_ctx* ctx, int enable) {
    DISCO_ENTER(ctx);
    stage_clear(ctx);
    ctx->stage_on = enable != 0;
    for (int h = 0; h < 2; ++h)
        if (ctx->half[h]) {
            stage_clear(ctx->half[h]);
            ctx->half[h]->stage_on = ctx->stage_on;
        }
    return 0;
}

extern "C" int disco_stage_report(disco_ctx* ctx, char* names, float* total_ms, int* launches, int64_t* rooms, int max_stages) {
    DISCO_ENTER(ctx);
    if (max_stages < 0 || (max_stages > 0 && (!names || !total_ms || !launches)))
        return fail(ctx, DISCO_E_ARG, "disco_stage_report: bad argument");
    int n = 0;
    // the context's own records, then those of its half-batch children merged in by name
    disco_ctx* src[3] = {ctx, ctx->half[0], ctx->half[1]};
    for (disco_ctx* c : src) {
        if (!c) continue;
        for (auto& st : c->stages) {
            float ms = 0.f;
            for (auto& e : st.evs) {
                float d = 0.f;
                HIPCHK(ctx, hipEventSynchronize(e.second));
                HIPCHK(ctx, hipEventElapsedTime(&d, e.first, e.second));
                ms += d;
            }
            int at = -1;
            for (int i = 0; i < n; ++i)
                if (!strncmp(names + 32 * i, st.name, 32)) at = i;
            if (at < 0) {
                if (n >= max_stages) continue;
                at = n++;
                snprintf(names + 32 * at, 32, "%s", st.name);
                total_ms[at] = 0.f;
                launches[at] = 0;
                if (rooms) rooms[at] = 0;
            }
            total_ms[at] += ms;
            launches[at] += (int)st.evs.size();
            if (rooms) rooms[at] += (int64_t)st.evs.size() * c->cfg.rooms;
        }
    }
    return n;
}

extern "C" int disco_set_tuning(disco_ctx* ctx, int stft_frames_per_wave, int cov_chunks, int step2_chunks, int istft_pairs) {
    DISCO_ENTER(ctx);
    if (stft_frames_per_wave < 0 || stft_frames_per_wave > 1024 || cov_chunks < 0 || step2_chunks < 0 || istft_pairs < 0 ||
        istft_pairs == 1 || istft_pairs > 4096)
        return fail(ctx, DISCO_E_ARG, "disco_set_tuning: need 0 <= stft_frames_per_wave <= 1024, chunks >= 0, istft_pairs 0 or 2..4096");
    ctx->tune_runw = stft_frames_per_wave;
    ctx->tune_cov_chunks = cov_chunks;
    ctx->tune_step2_chunks = step2_chunks;
    ctx->tune_pairs = istft_pairs;
    ctx->pending_chunks = 0;           // partial sums of another geometry must not be re-used
    ctx->loc_M = 0;
    for (int h = 0; h < 2; ++h)
        if (ctx->half[h]) {
            const int rc = disco_set_tuning(ctx->half[h], stft_frames_per_wave, cov_chunks, step2_chunks, istft_pairs);
            if (rc) return fail(ctx, rc, ctx->half[h]->err);
            if (!(ctx->cfg.flags & DISCO_FLAG_LAZY_SCRATCH)) {
                const int rs = reserve_scratch(ctx->half[h]);
                if (rs) return fail(ctx, rs, ctx->half[h]->err);
            }
        }
    if (!(ctx->cfg.flags & DISCO_FLAG_LAZY_SCRATCH)) return reserve_scratch(ctx);     // the new geometry may need larger blocks
    return 0;
}

namespace disco_host {
const OptionInfo* option_table() {
    static const OptionInfo t[DISCO_N_OPTIONS] = {
        {"room_cov", "DISCO_ROOM_COV", 1},
        {"overlap_solves", "DISCO_OVERLAP_SOLVES", 1},
        {"solve_dpp", "DISCO_SOLVE_DPP", 1},
        {"solve_thread", "DISCO_SOLVE_THREAD", 1},
        {"fuse_wide_istft", "DISCO_FUSE_WIDE_ISTFT", 1},
        {"online_sq32", "DISCO_ONLINE_SQ32", 1},
    };
    return t;
}
}  // namespace disco_host

static int option_index(const char* key) {
    if (!key) return -1;
    for (int i = 0; i < DISCO_N_OPTIONS; ++i)
        if (!strcmp(key, disco_host::option_table()[i].key)) return i;
    return -1;
}

extern "C" int disco_set_option(disco_ctx* ctx, const char* key, int value) {
    DISCO_ENTER(ctx);
    const int i = option_index(key);
    if (i < 0) return fail(ctx, DISCO_E_ARG, "disco_set_option: unknown key");
    ctx->opt[i] = value;
    if (i == DISCO_OPT_OVERLAP_SOLVES) return disco_host::ensure_halves(ctx);     // (may allocate: this is not a compute call)
    for (int h = 0; h < 2; ++h)
        if (ctx->half[h]) ctx->half[h]->opt[i] = value;
    // a route may ask for larger partial-sum blocks than the one sized so far (e.g. "room_dma" 0: chunked sums): size them HERE, so
    // that the next compute call still allocates nothing (this is not a compute call)
    if (!(ctx->cfg.flags & DISCO_FLAG_LAZY_SCRATCH)) {
        int rc = disco_host::reserve_scratch(ctx);
        for (int h = 0; h < 2 && !rc; ++h)
            if (ctx->half[h] && (rc = disco_host::reserve_scratch(ctx->half[h]))) return fail(ctx, rc, ctx->half[h]->err);
        return rc;
    }
    return 0;
}

extern "C" int disco_get_option(const disco_ctx* ctx, const char* key, int* value) {
    if (!ctx || !value) return DISCO_E_ARG;
    const int i = option_index(key);
    if (i < 0) return DISCO_E_ARG;
    *value = ctx->opt[i];
    return 0;
}

extern "C" int disco_set_z_blocks(disco_ctx* ctx, int nodes_per_block) {
    DISCO_ENTER(ctx);
    if (nodes_per_block < 1 || ctx->cfg.nodes % nodes_per_block)
        return fail(ctx, DISCO_E_ARG, "disco_set_z_blocks: nodes_per_block must divide cfg.nodes");
    ctx->zblk = nodes_per_block;
    return 0;
}

extern "C" int disco_n_frames(const disco_ctx* ctx) { return ctx ? ctx->T : DISCO_E_ARG; }
extern "C" int disco_n_freq(const disco_ctx* ctx) { return ctx ? ctx->F : DISCO_E_ARG; }

extern "C" int disco_dev_alloc(disco_ctx* ctx, size_t bytes, void** dptr) {
    DISCO_ENTER(ctx);
    if (!dptr) return fail(ctx, DISCO_E_ARG, "disco_dev_alloc: null argument");
    HIPCHK(ctx, hipMalloc(dptr, bytes ? bytes : 1));
    return 0;
}
extern "C" int disco_dev_free(disco_ctx* ctx, void* dptr) {
    DISCO_ENTER(ctx);
    HIPCHK(ctx, hipFree(dptr));
    return 0;
}
extern "C" int disco_h2d(disco_ctx* ctx, void* dst, const void* src, size_t bytes, disco_stream s) {
    DISCO_ENTER(ctx);
    if ((!dst && bytes) || (!src && bytes)) return fail(ctx, DISCO_E_ARG, "disco_h2d: null argument");
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
    return 0;
}
extern "C" int disco_d2h(disco_ctx* ctx, void* dst, const void* src, size_t bytes, disco_stream s) {
    DISCO_ENTER(ctx);
    if ((!dst && bytes) || (!src && bytes)) return fail(ctx, DISCO_E_ARG, "disco_d2h: null argument");
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
    return 0;
}
extern "C" int disco_sync(disco_ctx* ctx, disco_stream s) {
    DISCO_ENTER(ctx);
    HIPCHK(ctx, hipStreamSynchronize((hipStream_t)s));
    return 0;
}
