// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/disco_hip.h"
#include "k_apply.h"
#include "k_conv.h"
#include "k_cov.h"
#include "k_fused.h"
#include "k_ism.h"
#include "k_solve.h"
#include "k_metrics.h"
#include "k_online.h"
#include "k_room.h"
#include "k_stft.h"
#include "k_vad.h"

using namespace disco;

// Step 2 of the whole-path entry point (512-point STFT, <= 4 mics, <= 4 nodes): 1 = the filter + iSTFT pass re-transforms the
// samples (k_step2_stft_apply_istft) instead of reading the stored spectra back; the environment variable
// DISCO_STEP2_FROM_SAMPLES overrides it per context (A/B runs, tests of both paths).  Measured on C3: 5.9 ms from the
// samples against 4.5-4.8 ms from the spectra (two more forward transforms per node-frame cost more LDS-write time than
// the 10 GB of HBM reads they save) -> default 0.
#ifndef DISCO_STEP2_FROM_SAMPLES_DEFAULT
#define DISCO_STEP2_FROM_SAMPLES_DEFAULT 0
#endif

struct disco_ctx {
    disco_cfg cfg;
    int T, F;
    float* d_win;
    c32* d_tw;
    void* own_ws;
    size_t own_ws_bytes;
    void* scratch;            // covariance chunk partials (grown on demand)
    size_t scratch_bytes;
    int pending_chunks, pending_P;   // geometry of the partials currently in `scratch` (0 = none)
    void* scratch2;                  // step-2 partials when the step-1 ones in `scratch` are re-used (SKIPLOC)
    size_t scratch2_bytes;
    int loc_chunks, loc_M;           // geometry of the step-1 partials kept in `scratch` for that re-use
    const void *loc_X, *loc_mask;    // the STFT / mask arrays those step-1 partials were computed from (identity check of the re-use)
    int pending_skiploc;             // the pending step-2 partials (scratch2) lack their leading loc_M x loc_M block
    c32* d_tw_conv;                  // 1024-point twiddles of disco_rir_convolve (== d_tw when n_fft is 1024), lazy
    void* conv_ws;                   // its spectra workspace, lazy
    size_t conv_ws_bytes;
    int k0, Kl;                      // node shard: this context holds nodes [k0, k0 + Kl) of every room (default 0, K)
    int zblk;                        // layout of the exchanged-signal arguments Zs / Zn / Z (disco_set_z_blocks; default K = plain)
    int tune_runw, tune_cov_chunks, tune_step2_chunks, tune_pairs;   // disco_set_tuning overrides (0 = batch-size heuristic)
    int from_samples;                // whole-path step-2 kernels re-transform the samples instead of reading X back (env DISCO_STEP2_FROM_SAMPLES)
    // per-stage hipEvent timers of the whole-path entry points (disco_stage_timing / disco_stage_report)
    struct StageRec {
        char name[32];
        std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
    };
    bool stage_on;
    std::vector<StageRec> stages;
    char err[512];
};

static char g_create_err[512] = "";

#define HIPCHK(ctx, call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s", #call, hipGetErrorString(e_)); \
            return DISCO_E_HIP_BASE - (int)e_;                                                  \
        }                                                                                       \
    } while (0)

static int fail(disco_ctx* ctx, int code, const char* msg) {
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s", msg);
    return code;
}

static int check_launch(disco_ctx* ctx, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(ctx->err, sizeof(ctx->err), "launch of %s failed: %s", what, hipGetErrorString(e));
        return DISCO_E_HIP_BASE - (int)e;
    }
    return 0;
}

// Every entry point runs on the context's own device, whatever the calling thread's current device is, and leaves the
// caller's current device as it found it (two contexts on two GPUs in one process; a host such as torch switching devices).
struct DevGuard {
    int prev = -1;
    bool ok = true;
    explicit DevGuard(int device) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur != device) {
            ok = hipSetDevice(device) == hipSuccess;
            prev = cur;
        }
    }
    ~DevGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
};
#define DISCO_ENTER(ctx)                                                                             \
    if (!(ctx)) return DISCO_E_ARG;                                                                  \
    DevGuard dev_guard_((ctx)->cfg.device);                                                          \
    if (!dev_guard_.ok) return fail((ctx), DISCO_E_HIP_BASE, "hipSetDevice(cfg.device) failed")

// ---- per-stage timers ---------------------------------------------------------------------------------------------------
// STAGE(ctx, s, "name", call): when disco_stage_timing(ctx, 1) is in force, brackets `call` (one or more launches on stream
// s) with two hipEvents recorded on that stream; otherwise just evaluates it.  Nothing is synchronised here.
static void stage_clear(disco_ctx* ctx) {
    for (auto& st : ctx->stages)
        for (auto& e : st.evs) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    ctx->stages.clear();
}
struct StageScope {
    disco_ctx* ctx;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const char* name;
    StageScope(disco_ctx* c, disco_stream s, const char* n) : ctx(c), st((hipStream_t)s), name(n) {
        if (!ctx->stage_on) return;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            e0 = e1 = nullptr;
            return;
        }
        (void)hipEventRecord(e0, st);
    }
    ~StageScope() {
        if (!e0) return;
        (void)hipEventRecord(e1, st);
        for (auto& r : ctx->stages)
            if (!strncmp(r.name, name, sizeof(r.name))) {
                r.evs.emplace_back(e0, e1);
                return;
            }
        disco_ctx::StageRec r;
        snprintf(r.name, sizeof(r.name), "%s", name);
        r.evs.emplace_back(e0, e1);
        ctx->stages.push_back(std::move(r));
    }
};
#define STAGE(ctx, s, name, call) ([&]() { StageScope stage_scope_((ctx), (s), (name)); return (call); }())

// (tests/hipemu compiles these sources with g++ for logic tests and defines HIPEMU: the string must not claim a GPU there)
#ifdef HIPEMU
extern "C" const char* disco_version(void) { return "disco_hip 0.3.0 (hipemu host TEST build, not a product)"; }
#else
extern "C" const char* disco_version(void) { return "disco_hip 0.3.0 (gfx950)"; }
#endif

extern "C" const char* disco_last_error(const disco_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

static int reserve_scratch(disco_ctx* ctx);

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
    ctx->tune_runw = ctx->tune_cov_chunks = ctx->tune_step2_chunks = ctx->tune_pairs = 0;
    {
        const char* e = getenv("DISCO_STEP2_FROM_SAMPLES");
        ctx->from_samples = e ? atoi(e) : DISCO_STEP2_FROM_SAMPLES_DEFAULT;
    }
    ctx->stage_on = false;
    ctx->scratch2 = nullptr;
    ctx->scratch2_bytes = 0;
    ctx->loc_chunks = 0;
    ctx->loc_M = 0;
    ctx->loc_X = ctx->loc_mask = nullptr;
    ctx->pending_skiploc = 0;
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
            snprintf(g_create_err, sizeof(g_create_err), "disco_create: %s", ctx->err);
            disco_destroy(ctx);
            return rc;
        }
    }
    *out = ctx;
    return 0;
}

extern "C" void disco_destroy(disco_ctx* ctx) {
    if (!ctx) return;
    DevGuard dev_guard_(ctx->cfg.device);
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
    return 0;
}

extern "C" int disco_stage_report(disco_ctx* ctx, char* names, float* total_ms, int* launches, int max_stages) {
    DISCO_ENTER(ctx);
    if (max_stages < 0 || (max_stages > 0 && (!names || !total_ms || !launches)))
        return fail(ctx, DISCO_E_ARG, "disco_stage_report: bad argument");
    int n = 0;
    for (auto& st : ctx->stages) {
        if (n >= max_stages) break;
        float ms = 0.f;
        for (auto& e : st.evs) {
            float d = 0.f;
            HIPCHK(ctx, hipEventSynchronize(e.second));
            HIPCHK(ctx, hipEventElapsedTime(&d, e.first, e.second));
            ms += d;
        }
        snprintf(names + 32 * n, 32, "%s", st.name);
        total_ms[n] = ms;
        launches[n] = (int)st.evs.size();
        ++n;
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
    if (!(ctx->cfg.flags & DISCO_FLAG_LAZY_SCRATCH)) return reserve_scratch(ctx);     // the new geometry may need larger blocks
    return 0;
}

extern "C" int disco_set_z_blocks(disco_ctx* ctx, int nodes_per_block) {
    DISCO_ENTER(ctx);
    if (nodes_per_block < 1 || ctx->cfg.nodes % nodes_per_block)
        return fail(ctx, DISCO_E_ARG, "disco_set_z_blocks: nodes_per_block must divide cfg.nodes");
    ctx->zblk = nodes_per_block;
    return 0;
}

// whole-path entry points work on all nodes of a room and on their own plain [R][K] exchanged-signal arrays
static inline bool sharded(const disco_ctx* ctx) { return ctx->Kl != ctx->cfg.nodes || ctx->zblk != ctx->cfg.nodes; }

extern "C" int disco_n_frames(const disco_ctx* ctx) { return ctx ? ctx->T : DISCO_E_ARG; }
extern "C" int disco_n_freq(const disco_ctx* ctx) { return ctx ? ctx->F : DISCO_E_ARG; }

extern "C" int disco_dev_alloc(disco_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return DISCO_E_ARG;
    HIPCHK(ctx, hipMalloc(dptr, bytes ? bytes : 1));
    return 0;
}
extern "C" int disco_dev_free(disco_ctx* ctx, void* dptr) {
    DISCO_ENTER(ctx);
    HIPCHK(ctx, hipFree(dptr));
    return 0;
}
extern "C" int disco_h2d(disco_ctx* ctx, void* dst, const void* src, size_t bytes, disco_stream s) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return DISCO_E_ARG;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
    return 0;
}
extern "C" int disco_d2h(disco_ctx* ctx, void* dst, const void* src, size_t bytes, disco_stream s) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return DISCO_E_ARG;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
    return 0;
}
extern "C" int disco_sync(disco_ctx* ctx, disco_stream s) {
    DISCO_ENTER(ctx);
    HIPCHK(ctx, hipStreamSynchronize((hipStream_t)s));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// STFT family
// ---------------------------------------------------------------------------------------------------------
static inline int stft_runs(int T) { return (T + STFT_RUN - 1) / STFT_RUN; }
static inline long long stft_blocks(long long n_witems) { return (n_witems + STFT_WAVES - 1) / STFT_WAVES; }

template <int N>
static bool launch_stft(int chp, dim3 grid, hipStream_t st, const float* x, c32* X, const float* win, const c32* tw, int chans,
                        int L, int T, int pad_mode, int runs, long long n_items) {
    const dim3 block(64 * STFT_WAVES);
    // one channel pair per wave (k_stft_pairs) where k_stft's all-pairs-in-registers form drops to one wave per SIMD (measured:
    // N = 1024 from 2 pairs on, N = 512 from 3); the grid is (group, run) then, not waves
    if (chp >= (N == 1024 ? 2 : 3) && chp <= STFT_WAVES && n_items <= 0x7fffffffLL) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_pairs<N>), dim3((unsigned)n_items), block, 0, st, x, X, win, tw, chans, L, T, pad_mode,
                           runs);
        return true;
    }
    switch (chp) {
#define C_(P_)                                                                                                          \
    case P_:                                                                                                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft<N, P_>), grid, block, 0, st, x, X, win, tw, chans, L, T, pad_mode, runs, n_items); \
        return true;
        C_(1) C_(2)
#undef C_
    }
    return false;
}

extern "C" int disco_stft(disco_ctx* ctx, const float* x, int64_t n_sig, int chans, disco_c32* X, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!x || !X || n_sig < 1 || chans < 1) return fail(ctx, DISCO_E_ARG, "disco_stft: bad argument");
    if (chans > 8) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_stft: more than 8 channels per signal group");
    const int runs = stft_runs(ctx->T);
    const long long n_items = (long long)n_sig * runs;
    if (stft_blocks(n_items) > 0x7fffffffLL)
        return fail(ctx, DISCO_E_UNSUPPORTED, "disco_stft: batch too large for one launch");
    const disco_cfg& c = ctx->cfg;
    const dim3 grid((unsigned)stft_blocks(n_items));
    const int chp = (chans + 1) / 2;
    const bool ok = c.n_fft == 512
        ? launch_stft<512>(chp, grid, (hipStream_t)s, x, (c32*)X, ctx->d_win, ctx->d_tw, chans, c.length, ctx->T, c.pad_mode, runs, n_items)
        : launch_stft<1024>(chp, grid, (hipStream_t)s, x, (c32*)X, ctx->d_win, ctx->d_tw, chans, c.length, ctx->T, c.pad_mode, runs, n_items);
    if (!ok) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_stft: unsupported channel count");
    return check_launch(ctx, "k_stft");
}

extern "C" int disco_mask_oracle(disco_ctx* ctx, const float* s_ref, const float* n_ref, int64_t n_sig, float* mask,
                                 disco_stream s) {
    DISCO_ENTER(ctx);
    if (!s_ref || !n_ref || !mask || n_sig < 1) return fail(ctx, DISCO_E_ARG, "disco_mask_oracle: bad argument");
    const disco_cfg& c = ctx->cfg;
    if (c.mask_type < DISCO_MASK_IRM || c.mask_type > DISCO_MASK_IAM)
        return fail(ctx, DISCO_E_ARG, "disco_mask_oracle: unknown mask type");
    const int runs = stft_runs(ctx->T);
    const long long n_items = (long long)n_sig * runs;
    if (stft_blocks(n_items) > 0x7fffffffLL)
        return fail(ctx, DISCO_E_UNSUPPORTED, "disco_mask_oracle: batch too large for one launch");
    const float thr = powf(10.f, c.mask_bin_thr_db / 10.f);                 // math_utils.py db2lin (power)
    const dim3 grid((unsigned)stft_blocks(n_items));
    StageScope stage_scope_(ctx, s, "mask_oracle");
    if (c.n_fft == 512)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mask_oracle<512>), grid, dim3(64 * STFT_WAVES), 0,
                           (hipStream_t)s, s_ref, n_ref, mask, ctx->d_win, ctx->d_tw, c.length, ctx->T, c.pad_mode,
                           c.mask_type, c.mask_pow, thr, runs, n_items);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mask_oracle<1024>), grid, dim3(64 * STFT_WAVES), 0,
                           (hipStream_t)s, s_ref, n_ref, mask, ctx->d_win, ctx->d_tw, c.length, ctx->T, c.pad_mode,
                           c.mask_type, c.mask_pow, thr, runs, n_items);
    return check_launch(ctx, "k_mask_oracle");
}

extern "C" int disco_tf_mask(disco_ctx* ctx, const disco_c32* S, const disco_c32* N, int64_t n_elem, int mask_type,
                             int mask_pow, float bin_thr_db, float* mask, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!S || !N || !mask || n_elem < 1 || mask_pow < 0) return fail(ctx, DISCO_E_ARG, "disco_tf_mask: bad argument");
    if (mask_type < DISCO_MASK_IRM || mask_type > DISCO_MASK_IAM) return fail(ctx, DISCO_E_ARG, "disco_tf_mask: unknown mask type");
    const unsigned grid = (unsigned)std::min<long long>((n_elem + 255) / 256, 8192);
    hipLaunchKernelGGL(k_tf_mask, dim3(grid), dim3(256), 0, (hipStream_t)s, (const c32*)S, (const c32*)N, mask,
                       (long long)n_elem, mask_type, mask_pow, powf(10.f, bin_thr_db / 10.f));
    return check_launch(ctx, "k_tf_mask");
}

extern "C" int disco_istft(disco_ctx* ctx, const disco_c32* Z, int64_t n_sig, float* out, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!Z || !out || n_sig < 1) return fail(ctx, DISCO_E_ARG, "disco_istft: bad argument");
    const disco_cfg& c = ctx->cfg;
    const int n_seg = (c.length + c.hop - 1) / c.hop;
    const int bps = (n_seg + ISTFT_SEGS - 1) / ISTFT_SEGS;
    const long long grid = (long long)n_sig * bps;
    if (grid > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_istft: batch too large for one launch");
    if (c.n_fft == 512)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_istft<512>), dim3((unsigned)grid), dim3(64 * STFT_WAVES), 0, (hipStream_t)s,
                           (const c32*)Z, out, ctx->d_win, ctx->d_tw, c.length, ctx->T, bps);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_istft<1024>), dim3((unsigned)grid), dim3(64 * STFT_WAVES), 0, (hipStream_t)s,
                           (const c32*)Z, out, ctx->d_win, ctx->d_tw, c.length, ctx->T, bps);
    return check_launch(ctx, "k_istft");
}

// ---------------------------------------------------------------------------------------------------------
// covariance / solve / apply
// ---------------------------------------------------------------------------------------------------------
// (M, KR) instantiation table: every split of P = M + KR <= 8 channels.
#define DISCO_FOR_MKR(X_) \
    X_(1, 0) X_(1, 1) X_(1, 2) X_(1, 3) X_(1, 4) X_(1, 5) X_(1, 6) X_(1, 7) \
    X_(2, 0) X_(2, 1) X_(2, 2) X_(2, 3) X_(2, 4) X_(2, 5) X_(2, 6)          \
    X_(3, 0) X_(3, 1) X_(3, 2) X_(3, 3) X_(3, 4) X_(3, 5)                   \
    X_(4, 0) X_(4, 1) X_(4, 2) X_(4, 3) X_(4, 4)                            \
    X_(5, 0) X_(5, 1) X_(5, 2) X_(5, 3)                                     \
    X_(6, 0) X_(6, 1) X_(6, 2)                                              \
    X_(7, 0) X_(7, 1)                                                       \
    X_(8, 0)

static int cov_chunks(const disco_ctx* ctx) {
    const long long g = (long long)ctx->cfg.rooms * ctx->cfg.nodes;
    long long c = (2048 + g - 1) / g;
    if (c > 8) c = 8;
    if (ctx->tune_cov_chunks > 0) c = ctx->tune_cov_chunks;
    if (c > ctx->T) c = ctx->T;
    if (c < 1) c = 1;
    return (int)c;
}

static int ensure_scratch(disco_ctx* ctx, size_t bytes) {
    if (ctx->scratch_bytes >= bytes) return 0;
    if (ctx->scratch) {
        HIPCHK(ctx, hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    HIPCHK(ctx, hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return 0;
}

static int ensure_scratch2(disco_ctx* ctx, size_t bytes) {
    if (ctx->scratch2_bytes >= bytes) return 0;
    if (ctx->scratch2) {
        HIPCHK(ctx, hipFree(ctx->scratch2));
        ctx->scratch2 = nullptr;
        ctx->scratch2_bytes = 0;
    }
    HIPCHK(ctx, hipMalloc(&ctx->scratch2, bytes));
    ctx->scratch2_bytes = bytes;
    return 0;
}

static int cov_finalize(disco_ctx* ctx, int chunks, int P, disco_c32* Rss, disco_c32* Rnn, disco_stream s) {
    const long long n_gf = (long long)ctx->cfg.rooms * ctx->Kl * ctx->F;
    hipLaunchKernelGGL(k_cov_finalize, dim3((unsigned)std::min<long long>((n_gf + 127) / 128, 65535)), dim3(128), 0,
                       (hipStream_t)s, (const float4*)ctx->scratch, (c32*)Rss, (c32*)Rnn, n_gf, ctx->F, chunks, P,
                       1.0f / (float)ctx->T);
    return check_launch(ctx, "k_cov_finalize");
}

// chunk partials of the masked covariances into ctx->scratch ([R*K][chunks][F][NP] float4)
// (M, KR) shapes for which the block-partitioned kernel k_cov_split is instantiated: 9 <= M + KR <= 16, and the step-1
// shapes (KR = 0) whose 2 * M(M+1)/2 complex accumulators no longer fit one thread without spilling (M >= 7)
#define DISCO_FOR_SPLIT(X_)                                                      \
    X_(7, 0) X_(8, 0)                                                            \
    X_(8, 1) X_(8, 2) X_(8, 3) X_(8, 4) X_(8, 5) X_(8, 6) X_(8, 7) X_(8, 8)      \
    X_(4, 5) X_(4, 6) X_(4, 7) X_(4, 8) X_(4, 9) X_(4, 10) X_(4, 11) X_(4, 12)   \
    X_(2, 7) X_(2, 8) X_(2, 9) X_(2, 10) X_(2, 11) X_(2, 12) X_(2, 13) X_(2, 14)

template <int M, int KR>
static void launch_cov_split(bool skiploc, unsigned nblk, hipStream_t st, const CovArgs& a) {
    if constexpr (KR > 0) {
        // shapes with remote rows (all have an even M, and F - 1 is a multiple of 64 for both FFT sizes): frames staged through
        // LDS once per workgroup (k_cov.h; 7.3 ms per C5 launch, the per-wave fetches of k_cov_split: 9.6 ms)
        static_assert(M % 2 == 0, "k_cov_split_lds fetches X in 16-byte granules");
        const unsigned nb = DISCO_COV_XCD ? (nblk + DISCO_COV_XCD - 1) / DISCO_COV_XCD * DISCO_COV_XCD : nblk;      // see the kernel's id -> item map
        if (skiploc)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov_split_lds<M, KR, true>), dim3(nb), dim3(64 * cov_split_waves<KR, true>()), 0, st, a);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov_split_lds<M, KR, false>), dim3(nb), dim3(64 * cov_split_waves<KR, false>()), 0, st, a);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov_split<M, KR, false>), dim3(nblk), dim3(64 * cov_split_waves<KR, false>()), 0, st, a);
    }
}

static bool cov_split_shape(int M, int KR) {
#define X_(M_, KR_) if (M == M_ && KR == KR_) return true;
    DISCO_FOR_SPLIT(X_)
#undef X_
    return false;
}

// skiploc (step 2 only, internal): the caller guarantees that `scratch` holds the step-1 partial sums of THIS X with THIS
// mask (ctx->loc_M == M): the leading M x M block is then neither accumulated nor written, the partial sums go to `scratch2`
// and the solver assembles the pencil from both.  Honoured only by k_cov_split; the return value of *skiploc_used says so.
static int cov_partials(disco_ctx* ctx, const disco_c32* X, const float* mask, const disco_c32* Zs, const disco_c32* Zn,
                        int mask_remote, int P, int* chunks_out, disco_stream s, bool skiploc = false) {
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, KR = P - M;
    if (!X || !mask) return fail(ctx, DISCO_E_ARG, "disco_cov_masked: null argument");
    if (KR != 0 && KR != c.nodes - 1) return fail(ctx, DISCO_E_ARG, "disco_cov_masked: P must be M or M + K - 1");
    if (KR > 0 && (!Zs || !Zn)) return fail(ctx, DISCO_E_ARG, "disco_cov_masked: Zs/Zn required when P > M");
    if (P > CB_PMAX || M > 8) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_cov_masked: P > 16 or M > 8");
    const int chunks = cov_chunks(ctx);
    const long long G = (long long)c.rooms * ctx->Kl;
    const int NP = P * (P + 1) / 2;
    const size_t need = (size_t)G * chunks * ctx->F * NP * sizeof(float4);
    const bool same = (Zs == Zn);
    const bool split = (KR == 0 || (P > 8 && same && mask_remote && (ctx->F - 1) % 64 == 0)) && cov_split_shape(M, KR);
    skiploc = skiploc && split && KR > 0 && ctx->loc_M == M && ctx->loc_X == X && ctx->loc_mask == mask;
    int rc = 0;
    rc = skiploc ? ensure_scratch2(ctx, need) : ensure_scratch(ctx, need);
    if (rc) return rc;
    CovArgs a;
    a.X = (const c32*)X;
    a.mask = mask;
    a.Zs = (const c32*)Zs;
    a.Zn = (const c32*)Zn;
    a.part = (float4*)(skiploc ? ctx->scratch2 : ctx->scratch);
    a.K = c.nodes;
    a.T = ctx->T;
    a.F = ctx->F;
    a.chunks = chunks;
    a.mask_remote = mask_remote;
    a.Kl = ctx->Kl;
    a.k0 = ctx->k0;
    a.zblk = ctx->zblk;
    a.R = c.rooms;
    const dim3 grid((unsigned)(G * chunks)), block((unsigned)(ctx->F - 1 + 64));
    bool launched = false;
    if (split) {                // 9 <= P <= 16, one vector for both statistics: one block of pairs per wave
        const int tiles = (ctx->F - 1 + 63) / 64;
        const long long nblk = G * (tiles + 1) * chunks;
        if (nblk > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_cov_masked: batch too large");
#define X_(M_, KR_)                                                                                                  \
    if (!launched && M == M_ && KR == KR_) {                                                                         \
        launch_cov_split<M_, KR_>(skiploc, (unsigned)nblk, (hipStream_t)s, a);                                       \
        launched = true;                                                                                             \
    }
        DISCO_FOR_SPLIT(X_)
#undef X_
    }
#define X_(M_, KR_)                                                                                                  \
    if (!launched && M == M_ && KR == KR_) {                                                                         \
        if (c.n_fft == 512) {                                                                                        \
            if (KR_ == 0 || same)                                                                                    \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov<M_, KR_, true, 320>), grid, block, 0, (hipStream_t)s, a);   \
            else                                                                                                     \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov<M_, KR_, false, 320>), grid, block, 0, (hipStream_t)s, a);  \
        } else {                                                                                                     \
            if (KR_ == 0 || same)                                                                                    \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov<M_, KR_, true, 576>), grid, block, 0, (hipStream_t)s, a);   \
            else                                                                                                     \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov<M_, KR_, false, 576>), grid, block, 0, (hipStream_t)s, a);  \
        }                                                                                                            \
        launched = true;                                                                                             \
    }
    DISCO_FOR_MKR(X_)
#undef X_
    if (!launched) {            // 9 <= P <= 16: pairs split over the waves of a workgroup
        const int tiles = (ctx->F - 1 + 63) / 64;
        const long long nblk = G * (tiles + 1) * chunks;
        if (nblk > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_cov_masked: batch too large");
        if (KR == 0 || same)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov_big<true>), dim3((unsigned)nblk), dim3(64 * CB_S), 0, (hipStream_t)s, a, M, KR);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov_big<false>), dim3((unsigned)nblk), dim3(64 * CB_S), 0, (hipStream_t)s, a, M, KR);
    }
    *chunks_out = chunks;
    ctx->pending_chunks = chunks;
    ctx->pending_P = P;
    ctx->pending_skiploc = skiploc ? 1 : 0;
    if (!skiploc) {
        // `scratch` now holds THIS call's partial sums: step-1 ones (P == M, all nodes here) can be re-used by a step 2
        // on the same mask, anything else invalidates what k_stft_cov / an earlier step-1 call left
        ctx->loc_M = (KR == 0 && !sharded(ctx)) ? M : 0;
        ctx->loc_chunks = chunks;
        ctx->loc_X = X;
        ctx->loc_mask = mask;
    }
    return check_launch(ctx, "k_cov");
}

// Wide shapes (P = M + K - 1 > 8), all nodes of a room on this GPU, mask_for_z = 'local', step-1 partial sums of THIS X with
// THIS mask still in `scratch`: z of every node AND the step-2 partial sums of every node from ONE pass over X (k_room.h),
// instead of disco_apply + cov_partials; room_cov_ok says whether the shape and the context's state qualify.
#define DISCO_FOR_ROOM(X_) X_(8, 8) X_(8, 6) X_(8, 4) X_(8, 2) X_(4, 8) X_(4, 6)
#ifndef DISCO_ROOM_COV
#define DISCO_ROOM_COV 1
#endif
#ifndef DISCO_ROOM_DMA
#define DISCO_ROOM_DMA 1
#endif
static bool room_cov_ok(const disco_ctx* ctx, const disco_c32* X, const float* mask) {
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, K = c.nodes;
    bool shape = false;
#define X_(M_, K_) if (M == M_ && K == K_) shape = true;
    DISCO_FOR_ROOM(X_)
#undef X_
    const char* env = getenv("DISCO_ROOM_COV");
    const bool want = env ? atoi(env) != 0 : DISCO_ROOM_COV != 0;
    if (!want || !shape || M + K - 1 <= 8 || sharded(ctx) || !X || !mask) return false;
    if (!(ctx->loc_M == M && ctx->loc_X == X && ctx->loc_mask == mask)) return false;       // the leading M x M block must be step 1's
    return (long long)K * ctx->T * ctx->F * M <= 0x0fffffffLL;                               // 32-bit BYTE offsets inside a room (8 B per element)
}

static int room_cov_partials(disco_ctx* ctx, const disco_c32* X, const float* mask, const disco_c32* w_loc, disco_c32* z,
                             int* chunks_out, disco_stream s) {
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, K = c.nodes, P = M + K - 1;
    if (!w_loc || !z || !room_cov_ok(ctx, X, mask)) return fail(ctx, DISCO_E_ARG, "room covariance: shape / state does not qualify");
    const int chunks = cov_chunks(ctx);
    const long long G = (long long)c.rooms * K;
    const int NP = P * (P + 1) / 2;
    int rc = ensure_scratch2(ctx, (size_t)G * chunks * ctx->F * NP * sizeof(float4));
    if (rc) return rc;
    RoomArgs a;
    a.X = (const c32*)X;
    a.mask = mask;
    a.w = (const c32*)w_loc;
    a.z = (c32*)z;
    a.part = (float4*)ctx->scratch2;
    a.T = ctx->T;
    a.F = ctx->F;
    a.chunks = chunks;
    a.tiles = (ctx->F + 31) / 32;
    a.R = c.rooms;
    const long long nblk = (long long)c.rooms * a.tiles * chunks;
    if (nblk > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "room covariance: batch too large for one launch");
    // frames through an LDS-DMA ring three ahead (default), or staged through registers one ahead (DISCO_ROOM_DMA=0; k_room.h)
    const char* env_dma = getenv("DISCO_ROOM_DMA");
    const bool dma = env_dma ? atoi(env_dma) != 0 : DISCO_ROOM_DMA != 0;
#define X_(M_, K_)                                                                                                       \
    if (M == M_ && K == K_) {                                                                                            \
        if (dma)                                                                                                         \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_room_cov_dma<M_, K_>), dim3((unsigned)nblk), dim3(RoomGeom<M_, K_>::NT), 0, (hipStream_t)s, a); \
        else                                                                                                             \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_room_cov<M_, K_>), dim3((unsigned)nblk), dim3(RoomGeom<M_, K_>::NT), 0, (hipStream_t)s, a);     \
    }
    DISCO_FOR_ROOM(X_)
#undef X_
    *chunks_out = chunks;
    ctx->pending_chunks = chunks;
    ctx->pending_P = P;
    ctx->pending_skiploc = 1;
    return check_launch(ctx, "k_room_cov");
}

extern "C" int disco_cov_masked(disco_ctx* ctx, const disco_c32* X, const float* mask, const disco_c32* Zs,
                                const disco_c32* Zn, int mask_remote, int P, disco_c32* Rss, disco_c32* Rnn,
                                disco_stream s) {
    DISCO_ENTER(ctx);
    if ((Rss == nullptr) != (Rnn == nullptr)) return fail(ctx, DISCO_E_ARG, "disco_cov_masked: Rss and Rnn must both be given or both be NULL");
    int chunks = 1;
    int rc = cov_partials(ctx, X, mask, Zs, Zn, mask_remote, P, &chunks, s);
    if (rc || !Rss) return rc;
    return cov_finalize(ctx, chunks, P, Rss, Rnn, s);
}

template <int P>
static void launch_solve(const SolveSrc& src, long long n_prob, double mu, c32* w, c32* t1, hipStream_t s) {
    if constexpr (P <= 4) {             // one thread per pencil (k_solve_small.h)
        const long long grid = (n_prob + SOLVE_SMALL_THREADS - 1) / SOLVE_SMALL_THREADS;
        if (src.part)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gevd_mwf_r1_thread<P, true>), dim3((unsigned)grid), dim3(SOLVE_SMALL_THREADS), 0, s, src, n_prob,
                               mu, w, t1);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gevd_mwf_r1_thread<P, false>), dim3((unsigned)grid), dim3(SOLVE_SMALL_THREADS), 0, s, src, n_prob,
                               mu, w, t1);
        return;
    }
    const int probs = SolveGeom<P>::PROBS;
    const long long grid = (n_prob + probs - 1) / probs;
    if (src.part)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gevd_mwf_r1<P, true>), dim3((unsigned)grid), dim3(SolveGeom<P>::THREADS), 0, s, src,
                           n_prob, mu, w, t1);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gevd_mwf_r1<P, false>), dim3((unsigned)grid), dim3(SolveGeom<P>::THREADS), 0, s, src,
                           n_prob, mu, w, t1);
}

static int solve_dispatch(disco_ctx* ctx, const SolveSrc& src, int64_t n_prob, int P, float mu, disco_c32* w, disco_c32* t1,
                          disco_stream s) {
    if (P < 1 || P > 16) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_gevd_mwf_r1: P must be in 1..16");
    if (n_prob / 4 > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_gevd_mwf_r1: batch too large");
    hipStream_t st = (hipStream_t)s;
    switch (P) {
#define C_(P_) case P_: launch_solve<P_>(src, n_prob, (double)mu, (c32*)w, (c32*)t1, st); break;
        C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
#undef C_
    }
    return check_launch(ctx, "k_gevd_mwf_r1");
}

extern "C" int disco_gevd_mwf_r1(disco_ctx* ctx, const disco_c32* Rss, const disco_c32* Rnn, int64_t n_prob, int P,
                                 float mu, disco_c32* w, disco_c32* t1, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!Rss || !Rnn || !w || n_prob < 1) return fail(ctx, DISCO_E_ARG, "disco_gevd_mwf_r1: bad argument");
    SolveSrc src;
    src.Rss = (const c32*)Rss;
    src.Rnn = (const c32*)Rnn;
    src.part = nullptr;
    src.F = 1;
    src.chunks = 1;
    src.inv_T = 1.f;
    src.part_loc = nullptr;
    src.chunks_loc = 0;
    src.M_loc = 0;
    return solve_dispatch(ctx, src, n_prob, P, mu, w, t1, s);
}

// The two branches of intern_filter the hot path never takes (see k_mwf_variants)
extern "C" int disco_mwf_filter(disco_ctx* ctx, const disco_c32* Rxx, const disco_c32* Rnn, int64_t n_prob, int P, float mu,
                                int type, disco_c32* w, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!Rxx || !Rnn || !w || n_prob < 1) return fail(ctx, DISCO_E_ARG, "disco_mwf_filter: bad argument");
    if (type != DISCO_FILTER_R1_MWF && type != DISCO_FILTER_MWF) return fail(ctx, DISCO_E_ARG, "disco_mwf_filter: unknown filter type");
    if (P < 1 || P > 16) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_mwf_filter: P must be in 1..16");
    if (n_prob / 4 > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_mwf_filter: batch too large");
    hipStream_t st = (hipStream_t)s;
    switch (P) {
#define C_(P_)                                                                                                          \
    case P_: {                                                                                                          \
        const dim3 grid((unsigned)((n_prob + SolveGeom<P_>::PROBS - 1) / SolveGeom<P_>::PROBS)), block(SolveGeom<P_>::THREADS);  \
        if (type == DISCO_FILTER_MWF)                                                                                   \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mwf_variants<P_, FILTER_MWF>), grid, block, 0, st, (const c32*)Rxx, (const c32*)Rnn, \
                               (long long)n_prob, (double)mu, (c32*)w);                                                 \
        else                                                                                                            \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mwf_variants<P_, FILTER_R1_MWF>), grid, block, 0, st, (const c32*)Rxx, (const c32*)Rnn, \
                               (long long)n_prob, (double)mu, (c32*)w);                                                 \
    } break;
        C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
#undef C_
    }
    return check_launch(ctx, "k_mwf_variants");
}

extern "C" int disco_gevd_mwf_r1_pending(disco_ctx* ctx, float mu, disco_c32* w, disco_c32* t1, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!w) return fail(ctx, DISCO_E_ARG, "disco_gevd_mwf_r1_pending: null argument");
    if (ctx->pending_chunks < 1 || !ctx->scratch)
        return fail(ctx, DISCO_E_ARG, "disco_gevd_mwf_r1_pending: no covariance call has left partial sums in this context");
    SolveSrc src;
    src.Rss = nullptr;
    src.Rnn = nullptr;
    src.part = (const float4*)(ctx->pending_skiploc ? ctx->scratch2 : ctx->scratch);
    src.F = ctx->F;
    src.chunks = ctx->pending_chunks;
    src.inv_T = 1.0f / (float)ctx->T;
    src.part_loc = ctx->pending_skiploc ? (const float4*)ctx->scratch : nullptr;
    src.chunks_loc = ctx->pending_skiploc ? ctx->loc_chunks : 0;
    src.M_loc = ctx->pending_skiploc ? ctx->loc_M : 0;
    return solve_dispatch(ctx, src, (int64_t)ctx->cfg.rooms * ctx->Kl * ctx->F, ctx->pending_P, mu, w, t1, s);
}

extern "C" int disco_apply(disco_ctx* ctx, const disco_c32* X, const disco_c32* Z, const disco_c32* w, int P, int conj_w,
                           disco_c32* out, disco_stream s) {
    DISCO_ENTER(ctx);
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, KR = P - M;
    if (!X || !w || !out) return fail(ctx, DISCO_E_ARG, "disco_apply: null argument");
    if (KR != 0 && KR != c.nodes - 1) return fail(ctx, DISCO_E_ARG, "disco_apply: P must be M or M + K - 1");
    if (KR > 0 && !Z) return fail(ctx, DISCO_E_ARG, "disco_apply: Z required when P > M");
    if (P > CB_PMAX) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_apply: P > 16 not supported");
    const long long G = (long long)c.rooms * ctx->Kl;
    const long long TF = (long long)ctx->T * ctx->F;
    int bpn = (int)std::min<long long>((TF + 255) / 256, 64);
    while ((long long)bpn * G > 0x7fffffffLL && bpn > 1) bpn >>= 1;
    const dim3 grid((unsigned)(G * bpn)), block(256);
    bool launched = false;
#define X_(M_, KR_)                                                                                                  \
    if (!launched && M == M_ && KR == KR_) {                                                                         \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_apply<M_, KR_>), grid, block, 0, (hipStream_t)s, (const c32*)X,         \
                           (const c32*)Z, (const c32*)w, (c32*)out, c.nodes, ctx->T, ctx->F, conj_w, bpn, ctx->Kl,   \
                           ctx->k0, ctx->zblk, (long long)c.rooms);                                                                                 \
        launched = true;                                                                                             \
    }
    DISCO_FOR_MKR(X_)
#undef X_
    if (!launched) {
        const int tiles = (ctx->F + 63) / 64;
        int t_chunks = (int)std::min<long long>(std::max<long long>(1, (8192 + G * tiles - 1) / (G * tiles)), std::max(1, ctx->T / 8));
        while (G * tiles * t_chunks > 0x7ffffff0LL && t_chunks > 1) t_chunks >>= 1;
        const long long items_m = G * tiles * t_chunks;
        const dim3 grid_m((unsigned)((items_m + DISCO_APPLY_XCD - 1) / DISCO_APPLY_XCD * DISCO_APPLY_XCD));      // ids are dealt over the XCDs
        switch (M) {
#define C_(M_)                                                                                                          \
    case M_:                                                                                                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_apply_m<M_>), grid_m, dim3(64), 0, (hipStream_t)s, (const c32*)X, (const c32*)Z, \
                           (const c32*)w, (c32*)out, KR, c.nodes, ctx->T, ctx->F, conj_w, tiles, t_chunks, ctx->Kl, ctx->k0, ctx->zblk, \
                           (long long)c.rooms);                                                                    \
        break;
            C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8)
#undef C_
            default: return fail(ctx, DISCO_E_UNSUPPORTED, "disco_apply: more than 8 mics per node");
        }
    }
    return check_launch(ctx, "k_apply");
}

extern "C" int disco_noise_residual(disco_ctx* ctx, const disco_c32* X, const disco_c32* z, disco_c32* zn, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!X || !z || !zn) return fail(ctx, DISCO_E_ARG, "disco_noise_residual: null argument");
    const disco_cfg& c = ctx->cfg;
    const long long n = (long long)c.rooms * ctx->Kl * ctx->T * ctx->F;
    hipLaunchKernelGGL(k_noise_residual, dim3((unsigned)std::min<long long>((n + 255) / 256, 16384)), dim3(256), 0,
                       (hipStream_t)s, (const c32*)X, (const c32*)z, (c32*)zn, n, c.mics, c.ref_mic);
    return check_launch(ctx, "k_noise_residual");
}

// ---------------------------------------------------------------------------------------------------------
// STFT + step-1 covariance in one pass
// ---------------------------------------------------------------------------------------------------------
template <int N, bool STORE = true>
static bool launch_stft_cov(int M, dim3 grid, hipStream_t st, const float* y, const float* mask, c32* X, float4* part,
                            const float* win, const c32* tw, int L, int T, int pad_mode, int chunks, int runw) {
    const dim3 block(64 * STFT_WAVES);
    switch (M) {
#define C_(M_)                                                                                                          \
    case M_:                                                                                                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_cov<N, M_, STORE>), grid, block, 0, st, y, mask, X, part, win, tw, L, T, pad_mode, \
                           chunks, runw);                                                                               \
        return true;
        C_(1) C_(2) C_(3) C_(4) C_(5) C_(6)
#undef C_
    }
    if constexpr (N == 512) {          // the 1024-point spectrum tile of 7-8 mics does not fit the 160 KiB LDS
        switch (M) {
#define C_(M_)                                                                                                          \
    case M_:                                                                                                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_cov<N, M_, STORE>), grid, block, 0, st, y, mask, X, part, win, tw, L, T, pad_mode, \
                           chunks, runw);                                                                               \
        return true;
            C_(7) C_(8)
#undef C_
        }
    }
    return false;
}

// frame chunks (= workgroups per node) of the fused STFT + covariance pass and the frames each of its waves streams: runs as long
// as possible (<= 80 frames) while leaving >= ~2048 workgroups for the chip
static int stft_cov_chunks(const disco_ctx* ctx, int* runw_out) {
    const long long G = (long long)ctx->cfg.rooms * ctx->cfg.nodes;
    const long long chunks_wanted = std::max<long long>(1, (2048 + G - 1) / G);
    int runw = (int)((ctx->T + STFT_WAVES * chunks_wanted - 1) / (STFT_WAVES * chunks_wanted));
    runw = std::min(80, std::max(8, runw));
    if (ctx->tune_runw > 0) runw = ctx->tune_runw;
    if (runw_out) *runw_out = runw;
    return (ctx->T + STFT_WAVES * runw - 1) / (STFT_WAVES * runw);
}

// store = false (internal, single-node path): the spectra are not written (X may be NULL); only for shapes the fused kernel takes
static int stft_cov_partials(disco_ctx* ctx, const float* y, const float* mask_z, disco_c32* X, int* chunks_out, disco_stream s,
                             bool store = true) {
    if (!y || !mask_z || (store && !X)) return fail(ctx, DISCO_E_ARG, "disco_stft_cov_fused: null argument");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "fused kernels need every node of a room on this GPU (node shard active)");
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics;
    if (M > 8) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_stft_cov_fused: more than 8 mics per node");
    if (c.n_fft == 1024 && M > 6) {        // staged form of the same two operations
        if (!store) return fail(ctx, DISCO_E_UNSUPPORTED, "stft_cov without store: shape needs the staged kernels");
        int rc0 = STAGE(ctx, s, "stft", disco_stft(ctx, y, (int64_t)c.rooms * c.nodes, M, X, s));
        if (rc0) return rc0;
        return STAGE(ctx, s, "cov1", cov_partials(ctx, X, mask_z, nullptr, nullptr, 0, M, chunks_out, s));
    }
    const long long G = (long long)c.rooms * c.nodes;
    int runw = 0;
    const int chunks = stft_cov_chunks(ctx, &runw);
    const int NP = M * (M + 1) / 2;
    int rc = ensure_scratch(ctx, (size_t)G * chunks * ctx->F * NP * sizeof(float4));
    if (rc) return rc;
    if (G * chunks > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_stft_cov_fused: batch too large");
    const dim3 grid((unsigned)(G * chunks));
    const bool ok = !store
        ? STAGE(ctx, s, "stft_cov1_nostore", c.n_fft == 512
            ? (launch_stft_cov<512, false>(M, grid, (hipStream_t)s, y, mask_z, nullptr, (float4*)ctx->scratch, ctx->d_win, ctx->d_tw, c.length,
                                           ctx->T, c.pad_mode, chunks, runw))
            : (launch_stft_cov<1024, false>(M, grid, (hipStream_t)s, y, mask_z, nullptr, (float4*)ctx->scratch, ctx->d_win, ctx->d_tw, c.length,
                                            ctx->T, c.pad_mode, chunks, runw)))
        : STAGE(ctx, s, "stft_cov1", c.n_fft == 512
        ? launch_stft_cov<512>(M, grid, (hipStream_t)s, y, mask_z, (c32*)X, (float4*)ctx->scratch, ctx->d_win, ctx->d_tw, c.length,
                               ctx->T, c.pad_mode, chunks, runw)
        : launch_stft_cov<1024>(M, grid, (hipStream_t)s, y, mask_z, (c32*)X, (float4*)ctx->scratch, ctx->d_win, ctx->d_tw, c.length,
                                ctx->T, c.pad_mode, chunks, runw));
    if (!ok) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_stft_cov_fused: unsupported mic count");
    *chunks_out = chunks;
    ctx->pending_chunks = chunks;
    ctx->pending_P = M;
    ctx->pending_skiploc = 0;
    ctx->loc_chunks = chunks;          // kept for a possible re-use by step 2 of the same disco_tango_enhance call
    ctx->loc_M = M;
    ctx->loc_X = X;
    ctx->loc_mask = mask_z;
    if (!store) ctx->loc_M = 0;       // nothing to pair these partial sums with later
    return check_launch(ctx, "k_stft_cov");
}

extern "C" int disco_stft_cov_fused(disco_ctx* ctx, const float* y, const float* mask_z, disco_c32* X, disco_c32* Rss,
                                    disco_c32* Rnn, disco_stream s) {
    DISCO_ENTER(ctx);
    if ((Rss == nullptr) != (Rnn == nullptr)) return fail(ctx, DISCO_E_ARG, "disco_stft_cov_fused: Rss and Rnn must both be given or both be NULL");
    int chunks = 1;
    int rc = stft_cov_partials(ctx, y, mask_z, X, &chunks, s);
    if (rc || !Rss) return rc;
    return cov_finalize(ctx, chunks, ctx->cfg.mics, Rss, Rnn, s);
}

// ---------------------------------------------------------------------------------------------------------
// step 2 with the in-register z exchange
// ---------------------------------------------------------------------------------------------------------
static int step2_chunks(const disco_ctx* ctx, int tiles_plus_1) {
    const long long base = (long long)ctx->cfg.rooms * tiles_plus_1;
    long long c = (4096 + base - 1) / base;
    if (c > 8) c = 8;
    if (ctx->tune_step2_chunks > 0) c = ctx->tune_step2_chunks;
    if (c > ctx->T) c = ctx->T;
    if (c < 1) c = 1;
    return (int)c;
}

// skiploc: the caller guarantees that `scratch` still holds the step-1 partial sums of THIS X with THIS mask (only
// disco_tango_enhance can know); the leading M x M block is then neither accumulated nor written and the step-2 partials
// go to `scratch2`.
static int step2_cov_partials(disco_ctx* ctx, const disco_c32* X, const float* mask_w, const disco_c32* w_loc,
                              disco_c32* z_out, int* chunks_out, disco_stream s, bool skiploc = false) {
    if (!X || !mask_w || !w_loc) return fail(ctx, DISCO_E_ARG, "disco_step2_cov_fused: null argument");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "fused kernels need every node of a room on this GPU (node shard active)");
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, K = c.nodes, P = M + K - 1;
    if (P > 8) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_cov_fused: M + K - 1 > 8 not supported yet");
    const int tiles = (ctx->F - 1) / 64;
    const int chunks = step2_chunks(ctx, tiles + 1);
    const long long G = (long long)c.rooms * K;
    const int NP = P * (P + 1) / 2;
    const size_t need = (size_t)G * chunks * ctx->F * NP * sizeof(float4);
    int rc = 0;
    rc = skiploc ? ensure_scratch2(ctx, need) : ensure_scratch(ctx, need);
    if (rc) return rc;
    Step2Args a;
    a.X = (const c32*)X;
    a.mask = mask_w;
    a.w_loc = (const c32*)w_loc;
    a.w_glo = nullptr;
    a.z_out = (c32*)z_out;
    a.yf = nullptr;
    a.part = (float4*)(skiploc ? ctx->scratch2 : ctx->scratch);
    a.K = K;
    a.T = ctx->T;
    a.F = ctx->F;
    a.chunks = chunks;
    const long long nblk = (long long)c.rooms * (tiles + 1) * chunks;
    if (nblk > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_cov_fused: batch too large");
    bool launched = false;
#define X_(M_, KR_)                                                                                                  \
    if (!launched && M == M_ && K == KR_ + 1) {                                                                      \
        if (skiploc)                                                                                                 \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step2_cov_fused<M_, KR_ + 1, true>), dim3((unsigned)nblk),          \
                               dim3(64 * (KR_ + 1)), 0, (hipStream_t)s, a);                                          \
        else                                                                                                         \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step2_cov_fused<M_, KR_ + 1, false>), dim3((unsigned)nblk),         \
                               dim3(64 * (KR_ + 1)), 0, (hipStream_t)s, a);                                          \
        launched = true;                                                                                             \
    }
    DISCO_FOR_MKR(X_)
#undef X_
    if (!launched) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_cov_fused: unsupported (M, K) combination");
    *chunks_out = chunks;
    ctx->pending_chunks = chunks;
    ctx->pending_P = P;
    ctx->pending_skiploc = skiploc ? 1 : 0;
    if (!skiploc) ctx->loc_M = 0;
    return check_launch(ctx, "k_step2_cov_fused");
}

extern "C" int disco_step2_cov_fused_reuse(disco_ctx* ctx, const disco_c32* X, const float* mask_w, const disco_c32* w_loc,
                                           disco_c32* z_out, disco_stream s) {
    DISCO_ENTER(ctx);
    if (ctx->loc_M != ctx->cfg.mics || ctx->cfg.nodes < 2 || ctx->Kl != ctx->cfg.nodes)
        return fail(ctx, DISCO_E_ARG, "disco_step2_cov_fused_reuse: no step-1 partial sums of disco_stft_cov_fused are held by this context");
    if (ctx->loc_X != X || ctx->loc_mask != mask_w)
        return fail(ctx, DISCO_E_ARG, "disco_step2_cov_fused_reuse: X / mask_w are not the arrays the held step-1 partial sums were computed from");
    int chunks = 1;
    return step2_cov_partials(ctx, X, mask_w, w_loc, z_out, &chunks, s, true);
}
extern "C" int disco_step2_cov_fused(disco_ctx* ctx, const disco_c32* X, const float* mask_w, const disco_c32* w_loc,
                                     disco_c32* z_out, disco_c32* Rss, disco_c32* Rnn, disco_stream s) {
    DISCO_ENTER(ctx);
    if ((Rss == nullptr) != (Rnn == nullptr)) return fail(ctx, DISCO_E_ARG, "disco_step2_cov_fused: Rss and Rnn must both be given or both be NULL");
    int chunks = 1;
    int rc = step2_cov_partials(ctx, X, mask_w, w_loc, z_out, &chunks, s);
    if (rc || !Rss) return rc;
    return cov_finalize(ctx, chunks, ctx->cfg.mics + ctx->cfg.nodes - 1, Rss, Rnn, s);
}

extern "C" int disco_step2_apply_fused(disco_ctx* ctx, const disco_c32* X, const disco_c32* w_loc, const disco_c32* w_glo,
                                       disco_c32* z_out, disco_c32* yf, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!X || !w_loc || !w_glo || !yf) return fail(ctx, DISCO_E_ARG, "disco_step2_apply_fused: null argument");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "fused kernels need every node of a room on this GPU (node shard active)");
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, K = c.nodes, P = M + K - 1;
    if (P > 8) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_apply_fused: M + K - 1 > 8 not supported yet");
    Step2Args a;
    a.X = (const c32*)X;
    a.mask = nullptr;
    a.w_loc = (const c32*)w_loc;
    a.w_glo = (const c32*)w_glo;
    a.z_out = (c32*)z_out;
    a.yf = (c32*)yf;
    a.part = nullptr;
    a.K = K;
    a.T = ctx->T;
    a.F = ctx->F;
    const int tiles = (ctx->F - 1) / 64;
    a.chunks = step2_chunks(ctx, tiles + 1);
    const long long nblk = (long long)c.rooms * (tiles + 1) * a.chunks;
    if (nblk > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_apply_fused: batch too large");
    bool launched = false;
#define X_(M_, KR_)                                                                                                  \
    if (!launched && M == M_ && K == KR_ + 1) {                                                                      \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step2_apply_fused<M_, KR_ + 1>), dim3((unsigned)nblk), dim3(64 * (KR_ + 1)), \
                           0, (hipStream_t)s, a);                                                                    \
        launched = true;                                                                                             \
    }
    DISCO_FOR_MKR(X_)
#undef X_
    if (!launched) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_apply_fused: unsupported (M, K) combination");
    return check_launch(ctx, "k_step2_apply_fused");
}

template <int M, int K>
static bool launch_apply_istft(const Step2Args& a, float* out, const float* win, const c32* tw, int L, int bpr, int pairs, dim3 grid,
                               hipStream_t st) {
    if constexpr (sizeof(ApplyIstftShared<512, M, K>) <= 160 * 1024) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step2_apply_istft<512, M, K>), grid, dim3(64 * K), 0, st, a, out, win, tw, L, bpr, pairs);
        return true;
    } else {
        return false;
    }
}

extern "C" int disco_step2_apply_istft_fused(disco_ctx* ctx, const disco_c32* X, const disco_c32* w_loc,
                                             const disco_c32* w_glo, float* out, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!X || !w_loc || !w_glo || !out) return fail(ctx, DISCO_E_ARG, "disco_step2_apply_istft_fused: null argument");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "fused kernels need every node of a room on this GPU (node shard active)");
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, K = c.nodes, P = M + K - 1;
    if (c.n_fft != 512 || P > 8) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_apply_istft_fused: needs n_fft = 512 and M + K - 1 <= 8");
    Step2Args a;
    a.X = (const c32*)X;
    a.mask = nullptr;
    a.w_loc = (const c32*)w_loc;
    a.w_glo = (const c32*)w_glo;
    a.z_out = nullptr;
    a.yf = nullptr;
    a.part = nullptr;
    a.K = K;
    a.T = ctx->T;
    a.F = ctx->F;
    a.chunks = 1;
    const int n_seg = (c.length + c.hop - 1) / c.hop;
    // frame pairs per workgroup: as many as possible (<= 64) while leaving >= ~8192 waves (a workgroup has K of them)
    const long long units = (long long)c.rooms * K;
    const long long bpr_wanted = std::max<long long>(1, (8192 + units - 1) / units);
    int pairs = (int)(((n_seg + bpr_wanted - 1) / bpr_wanted + 2) / 2);
    pairs = std::min(64, std::max(4, pairs));
    if (ctx->tune_pairs > 0) pairs = ctx->tune_pairs;
    const int bpr = (n_seg + 2 * pairs - 2) / (2 * pairs - 1);
    const long long nblk = (long long)c.rooms * bpr;
    if (nblk > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_apply_istft_fused: batch too large");
    bool launched = false, tried = false;
#define X_(M_, KR_)                                                                                                  \
    if (!tried && M == M_ && K == KR_ + 1) {                                                                         \
        tried = true;                                                                                                \
        launched = launch_apply_istft<M_, KR_ + 1>(a, out, ctx->d_win, ctx->d_tw, c.length, bpr, pairs, dim3((unsigned)nblk), \
                                                   (hipStream_t)s);                                                  \
    }
    DISCO_FOR_MKR(X_)
#undef X_
    if (!launched) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_apply_istft_fused: shape does not fit the LDS budget");
    return check_launch(ctx, "k_step2_apply_istft");
}

// the same pass from the samples (k_step2_stft_apply_istft): 512-point STFT, M <= 4, 2 <= K <= 4
static bool from_samples_shape(const disco_cfg& c) { return c.n_fft == 512 && c.mics <= 4 && c.nodes >= 2 && c.nodes <= 4; }

static int step2_stft_apply_istft(disco_ctx* ctx, const float* y, const disco_c32* w_loc, const disco_c32* w_glo, float* out,
                                  disco_stream s) {
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, K = c.nodes;
    if (!from_samples_shape(c)) return DISCO_E_UNSUPPORTED;
    const int n_seg = (c.length + c.hop - 1) / c.hop;
    const long long units = (long long)c.rooms * K;
    const long long bpr_wanted = std::max<long long>(1, (8192 + units - 1) / units);
    int pairs = (int)(((n_seg + bpr_wanted - 1) / bpr_wanted + 2) / 2);
    pairs = std::min(64, std::max(4, pairs));
    if (ctx->tune_pairs > 0) pairs = ctx->tune_pairs;
    const int bpr = (n_seg + 2 * pairs - 2) / (2 * pairs - 1);
    const long long nblk = (long long)c.rooms * bpr;
    if (nblk > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_tango_enhance: batch too large for one launch");
    bool launched = false;
#define X_(M_, K_)                                                                                                        \
    if (!launched && M == M_ && K == K_) {                                                                                \
        launched = true;                                                                                                  \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step2_stft_apply_istft<512, M_, K_>), dim3((unsigned)nblk), dim3(64 * K_), 0,   \
                           (hipStream_t)s, y, (const c32*)w_loc, (const c32*)w_glo, out, ctx->d_win, ctx->d_tw, c.length, ctx->T, \
                           c.pad_mode, bpr, pairs);                                                                       \
    }
    X_(1, 2) X_(1, 3) X_(1, 4) X_(2, 2) X_(2, 3) X_(2, 4) X_(3, 2) X_(3, 3) X_(3, 4) X_(4, 2) X_(4, 3) X_(4, 4)
#undef X_
    if (!launched) return DISCO_E_UNSUPPORTED;
    return check_launch(ctx, "k_step2_stft_apply_istft");
}

// ---------------------------------------------------------------------------------------------------------
// whole path
// ---------------------------------------------------------------------------------------------------------
namespace {
struct WsLayout {
    size_t X, z, yf, Rss, Rnn, w, w2, total;
};
inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
WsLayout ws_layout(const disco_ctx* ctx) {
    const disco_cfg& c = ctx->cfg;
    const size_t G = (size_t)c.rooms * c.nodes, TF = (size_t)ctx->T * ctx->F;
    const size_t Pmax = (size_t)c.mics + c.nodes - 1;
    WsLayout l;
    size_t o = 0;
    l.X = o;   o = align_up(o + G * TF * c.mics * sizeof(c32));
    l.z = o;   o = align_up(o + G * TF * sizeof(c32));
    l.yf = o;  o = align_up(o + G * TF * sizeof(c32));
    l.Rss = o; o = align_up(o + G * ctx->F * Pmax * Pmax * sizeof(c32));
    l.Rnn = o; o = align_up(o + G * ctx->F * Pmax * Pmax * sizeof(c32));
    l.w = o;   o = align_up(o + G * ctx->F * Pmax * sizeof(c32));
    l.w2 = o;  o = align_up(o + G * ctx->F * Pmax * sizeof(c32));
    l.total = o;
    return l;
}
}  // namespace

extern "C" size_t disco_workspace_bytes(const disco_ctx* ctx) { return ctx ? ws_layout(ctx).total : 0; }

static int tango_enhance_fused(disco_ctx* ctx, const float* y, const float* mask_z, const float* mask_w, float* out,
                               disco_c32* z_y, disco_c32* yf, char* ws, const WsLayout& l, disco_stream s);

static int solve_from_partials(disco_ctx* ctx, int chunks, int P, disco_c32* w, disco_stream s) {
    (void)chunks;
    (void)P;                     // geometry is the pending state the covariance call just recorded
    return disco_gevd_mwf_r1_pending(ctx, ctx->cfg.mu, w, nullptr, s);
}

// caller's workspace if given (size-checked), else the context's own (grown on demand)
static int acquire_ws(disco_ctx* ctx, void* workspace, size_t workspace_bytes, const WsLayout& l, char** ws_out, const char* who) {
    char* ws = (char*)workspace;
    if (ws) {
        if (workspace_bytes < l.total) {
            std::string m = std::string(who) + ": workspace too small";
            return fail(ctx, DISCO_E_ARG, m.c_str());
        }
    } else {
        if (ctx->own_ws_bytes < l.total) {
            if (ctx->own_ws) {
                HIPCHK(ctx, hipFree(ctx->own_ws));
                ctx->own_ws = nullptr;
                ctx->own_ws_bytes = 0;
            }
            HIPCHK(ctx, hipMalloc(&ctx->own_ws, l.total));
            ctx->own_ws_bytes = l.total;
        }
        ws = (char*)ctx->own_ws;
    }
    *ws_out = ws;
    return 0;
}

// Single node, enhanced output only: iSTFT(w^H STFT(y)) straight from the samples (k_stft_apply_istft), 512-point STFT, M <= 4
static int stft_apply_istft(disco_ctx* ctx, const float* y, const disco_c32* w, float* out, disco_stream s) {
    const disco_cfg& c = ctx->cfg;
    const long long G = (long long)c.rooms * c.nodes;
    const int n_seg = (c.length + c.hop - 1) / c.hop;
    const long long runs_wanted = std::max<long long>(1, (8192 + G - 1) / G);          // >= ~8192 waves
    int pairs = (int)(((n_seg + runs_wanted - 1) / runs_wanted + 2) / 2);
    pairs = std::min(64, std::max(4, pairs));
    if (ctx->tune_pairs > 0) pairs = ctx->tune_pairs;
    const int runs = (n_seg + 2 * pairs - 2) / (2 * pairs - 1);
    const long long items = G * runs;
    if (stft_blocks(items) > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_tango_enhance: batch too large for one launch");
    const dim3 grid((unsigned)stft_blocks(items)), block(64 * STFT_WAVES);
    switch (c.mics) {
#define C_(M_)                                                                                                          \
    case M_:                                                                                                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_apply_istft<512, M_>), grid, block, 0, (hipStream_t)s, y, (const c32*)w, out, ctx->d_win, \
                           ctx->d_tw, c.length, ctx->T, c.pad_mode, runs, pairs, items);                                \
        break;
        C_(1) C_(2) C_(3) C_(4)
#undef C_
        default: return DISCO_E_UNSUPPORTED;
    }
    return check_launch(ctx, "k_stft_apply_istft");
}

// Both partial-sum blocks at the largest size any covariance call of this context can ask for with the present geometry:
// [R * Kl][chunks][F][P (P + 1) / 2] float4 with P = M + K - 1 and the largest of the three chunk counts.
static int reserve_scratch(disco_ctx* ctx) {
    const disco_cfg& c = ctx->cfg;
    const size_t G = (size_t)c.rooms * ctx->Kl;
    const size_t P = (size_t)std::min(c.mics + c.nodes - 1, 16);
    const size_t NP = P * (P + 1) / 2;
    int chunks = std::max(cov_chunks(ctx), step2_chunks(ctx, (ctx->F - 1) / 64 + 1));
    if (c.mics <= 8) chunks = std::max(chunks, stft_cov_chunks(ctx, nullptr));
    const size_t need = G * (size_t)chunks * ctx->F * NP * sizeof(float4);
    int rc = ensure_scratch(ctx, need);
    if (!rc) rc = ensure_scratch2(ctx, need);
    return rc;
}

extern "C" int disco_tango_enhance(disco_ctx* ctx, const float* y, const float* mask_z, const float* mask_w, float* out,
                                   disco_c32* z_y, disco_c32* yf, void* workspace, size_t workspace_bytes, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!y || !mask_z || !mask_w || !out) return fail(ctx, DISCO_E_ARG, "disco_tango_enhance: null argument");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_tango_enhance: node shard active, drive the staged calls around an all-gather of z");
    const disco_cfg& c = ctx->cfg;
    const WsLayout l = ws_layout(ctx);
    char* ws = nullptr;
    int rcw = acquire_ws(ctx, workspace, workspace_bytes, l, &ws, "disco_tango_enhance");
    if (rcw) return rcw;
    if (c.nodes > 1 && c.mics + c.nodes - 1 <= 8 && !(c.flags & DISCO_FLAG_STAGED_STEP2))
        return tango_enhance_fused(ctx, y, mask_z, mask_w, out, z_y, yf, ws, l, s);
    disco_c32* X = (disco_c32*)(ws + l.X);
    disco_c32* z = z_y ? z_y : (disco_c32*)(ws + l.z);
    disco_c32* yo = yf ? yf : (disco_c32*)(ws + l.yf);
    disco_c32* w = (disco_c32*)(ws + l.w);
    const int64_t G = (int64_t)c.rooms * c.nodes;
    const int M = c.mics, P2 = c.mics + c.nodes - 1;
    int rc;
    int chunks1 = 1;
    if (c.nodes == 1 && mask_w == mask_z && !z_y && !yf && c.n_fft == 512 && M <= 4) {
        // single node, enhanced output only (config C2): nothing is materialised -- one pass over the samples for the
        // statistics, one for filter + iSTFT with the spectra recomputed (get_z_signals.py:274-315 + tango.py:528)
        if ((rc = stft_cov_partials(ctx, y, mask_z, nullptr, &chunks1, s, false))) return rc;
        if ((rc = STAGE(ctx, s, "solve1", solve_from_partials(ctx, chunks1, M, w, s)))) return rc;
        return STAGE(ctx, s, "stft_apply_istft", stft_apply_istft(ctx, y, w, out, s));
    }
    // step 1 (tango.py:326-376): STFT + covariance in one pass, solve straight from the partial sums
    if ((rc = stft_cov_partials(ctx, y, mask_z, X, &chunks1, s))) return rc;
    if ((rc = STAGE(ctx, s, "solve1", solve_from_partials(ctx, chunks1, M, w, s)))) return rc;
    if (c.nodes == 1 && mask_w == mask_z && !z_y && !yf && c.n_fft == 512) {
        // single node, enhanced output only (config C2): filter + iSTFT in one pass over X, z never reaches HBM
        rc = STAGE(ctx, s, "step2_apply_istft", disco_step2_apply_istft_fused(ctx, X, w, w, out, s));
        if (rc != DISCO_E_UNSUPPORTED) return rc;
    }
    const bool room = c.nodes > 1 && mask_w == mask_z && room_cov_ok(ctx, X, mask_w);       // wide shapes: z + step-2 statistics in one pass
    if (!room && (rc = STAGE(ctx, s, "apply1", disco_apply(ctx, X, nullptr, w, M, 1, z, s)))) return rc;
    if (c.nodes == 1 && mask_w == mask_z) {
        // single node, same mask: step 2 would rebuild the very same statistics from the very same inputs
        // (P = M, nothing to append), so w_glo == w_loc and yf == z_y bit for bit (config C2).
        if (yf) HIPCHK(ctx, hipMemcpyAsync(yf, z, (size_t)G * ctx->T * ctx->F * sizeof(c32), hipMemcpyDeviceToDevice, (hipStream_t)s));
        return STAGE(ctx, s, "istft", disco_istft(ctx, z, G, out, s));
    }
    // exchange + step 2 (tango.py:378-450), mask_for_z = 'local'
    int chunks2 = 1;              // partial sums stay pending; the local M x M block is step 1's when the mask is the same
    if (room) {
        if ((rc = STAGE(ctx, s, "room_cov2", room_cov_partials(ctx, X, mask_w, w, z, &chunks2, s)))) return rc;
    } else if ((rc = STAGE(ctx, s, "cov2", cov_partials(ctx, X, mask_w, z, z, 1, P2, &chunks2, s, mask_w == mask_z)))) return rc;
    if ((rc = STAGE(ctx, s, "solve2", disco_gevd_mwf_r1_pending(ctx, c.mu, w, nullptr, s)))) return rc;
    if ((rc = STAGE(ctx, s, "apply2", disco_apply(ctx, X, z, w, P2, 1, yo, s)))) return rc;
    return STAGE(ctx, s, "istft", disco_istft(ctx, yo, G, out, s));
}

// The same path with step 2 on the in-register z exchange (default whenever all nodes of a room share the GPU).
static int tango_enhance_fused(disco_ctx* ctx, const float* y, const float* mask_z, const float* mask_w, float* out,
                               disco_c32* z_y, disco_c32* yf, char* ws, const WsLayout& l, disco_stream s) {
    const disco_cfg& c = ctx->cfg;
    disco_c32* X = (disco_c32*)(ws + l.X);
    disco_c32* yo = yf ? yf : (disco_c32*)(ws + l.yf);
    disco_c32* w_loc = (disco_c32*)(ws + l.w);
    disco_c32* w_glo = (disco_c32*)(ws + l.w2);
    const int64_t G = (int64_t)c.rooms * c.nodes;
    const int M = c.mics, P2 = c.mics + c.nodes - 1;
    int rc;
    int chunks = 1;
    if ((rc = stft_cov_partials(ctx, y, mask_z, X, &chunks, s))) return rc;
    if ((rc = STAGE(ctx, s, "solve1", solve_from_partials(ctx, chunks, M, w_loc, s)))) return rc;
    // same mask array in both steps (oracle masks; a DNN mask re-used, tango.py:388-389): the leading M x M block of the
    // step-2 covariances IS the step-1 covariance still held as partial sums -> not recomputed
    if (mask_w == mask_z && ctx->loc_M == M && c.nodes > 1)
        rc = STAGE(ctx, s, "step2_cov", disco_step2_cov_fused_reuse(ctx, X, mask_w, w_loc, z_y, s));
    else
        rc = STAGE(ctx, s, "step2_cov", step2_cov_partials(ctx, X, mask_w, w_loc, z_y, &chunks, s));
    if (rc) return rc;
    if ((rc = STAGE(ctx, s, "solve2", solve_from_partials(ctx, chunks, P2, w_glo, s)))) return rc;
    if (!yf && c.n_fft == 512) {           // yf not asked for: filter + iSTFT in one pass, yf stays on chip
        if (ctx->from_samples && from_samples_shape(c)) {       // ... and the spectra are re-transformed, not read back
            rc = STAGE(ctx, s, "step2_stft_apply_istft", step2_stft_apply_istft(ctx, y, w_loc, w_glo, out, s));
            if (rc != DISCO_E_UNSUPPORTED) return rc;
        }
        rc = STAGE(ctx, s, "step2_apply_istft", disco_step2_apply_istft_fused(ctx, X, w_loc, w_glo, out, s));
        if (rc != DISCO_E_UNSUPPORTED) return rc;
    }
    if ((rc = STAGE(ctx, s, "step2_apply", disco_step2_apply_fused(ctx, X, w_loc, w_glo, nullptr, yo, s)))) return rc;
    return STAGE(ctx, s, "istft", disco_istft(ctx, yo, G, out, s));
}

// ---- reference outputs: all nine returns of offline_tango, device resident ------------------------------------------------
namespace {
struct RefLayout {
    size_t Xy, Xs, Xn, zy, zs, zn_, znres, rows_s, rows_n, mz, mw, mc, Rss, Rnn, Rtmp, w_loc, w_glo, total;
};
RefLayout ref_layout(const disco_ctx* ctx) {
    const disco_cfg& c = ctx->cfg;
    const size_t G = (size_t)c.rooms * c.nodes, TF = (size_t)ctx->T * ctx->F;
    const size_t Pmax = (size_t)c.mics + c.nodes - 1;
    RefLayout l;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = align_up(o + bytes); return at; };
    l.Xy = take(G * TF * c.mics * sizeof(c32));
    l.Xs = take(G * TF * c.mics * sizeof(c32));
    l.Xn = take(G * TF * c.mics * sizeof(c32));
    l.zy = take(G * TF * sizeof(c32));
    l.zs = take(G * TF * sizeof(c32));
    l.zn_ = take(G * TF * sizeof(c32));
    l.znres = take(G * TF * sizeof(c32));
    l.rows_s = take(G * TF * sizeof(c32));
    l.rows_n = take(G * TF * sizeof(c32));
    l.mz = take(G * TF * sizeof(float));
    l.mw = take(G * TF * sizeof(float));
    l.mc = take(G * TF * sizeof(float));
    l.Rss = take(G * ctx->F * c.mics * c.mics * sizeof(c32));
    l.Rnn = take(G * ctx->F * c.mics * c.mics * sizeof(c32));
    l.Rtmp = take(G * ctx->F * c.mics * c.mics * sizeof(c32));
    l.w_loc = take(G * ctx->F * c.mics * sizeof(c32));
    l.w_glo = take(G * ctx->F * Pmax * sizeof(c32));
    l.total = o;
    return l;
}
inline unsigned ew_grid(long long n) { return (unsigned)std::min<long long>((n + 255) / 256, 16384); }
}  // namespace

extern "C" size_t disco_reference_workspace_bytes(const disco_ctx* ctx) { return ctx ? ref_layout(ctx).total : 0; }

extern "C" size_t disco_owned_bytes(const disco_ctx* ctx) {
    return ctx ? ctx->scratch_bytes + ctx->scratch2_bytes + ctx->own_ws_bytes + ctx->conv_ws_bytes : 0;
}

extern "C" int disco_reserve(disco_ctx* ctx, int own_workspace) {
    DISCO_ENTER(ctx);
    if (own_workspace < 0 || own_workspace > 2) return fail(ctx, DISCO_E_ARG, "disco_reserve: own_workspace must be 0, 1 or 2");
    int rc = reserve_scratch(ctx);
    if (rc || !own_workspace) return rc;
    size_t need = ws_layout(ctx).total;
    if (own_workspace == 2) need = std::max(need, ref_layout(ctx).total);
    if (ctx->own_ws_bytes < need) {
        if (ctx->own_ws) {
            HIPCHK(ctx, hipFree(ctx->own_ws));
            ctx->own_ws = nullptr;
            ctx->own_ws_bytes = 0;
        }
        HIPCHK(ctx, hipMalloc(&ctx->own_ws, need));
        ctx->own_ws_bytes = need;
    }
    return 0;
}

extern "C" int disco_tango_reference(disco_ctx* ctx, const float* y, const float* s_img, const float* n_img, const float* mask_z_in,
                                     const float* mask_w_in, int mask_for_z, int steps, const disco_ref_outputs* out,
                                     void* workspace, size_t workspace_bytes, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!y || !s_img || !n_img || !out) return fail(ctx, DISCO_E_ARG, "disco_tango_reference: null argument");
    if (steps < 1 || steps > 3) return fail(ctx, DISCO_E_ARG, "disco_tango_reference: steps must be 1, 2 or 3");
    if (mask_for_z < DISCO_MZ_LOCAL || mask_for_z > DISCO_MZ_PREVIOUS) return fail(ctx, DISCO_E_ARG, "disco_tango_reference: unknown mask_for_z");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_tango_reference: node shard active");
    const disco_cfg& c = ctx->cfg;
    if (c.mask_type < DISCO_MASK_IRM || c.mask_type > DISCO_MASK_IAM) return fail(ctx, DISCO_E_ARG, "disco_tango_reference: unknown mask type");
    const int M = c.mics, K = c.nodes, P2 = M + K - 1;
    if (P2 > CB_PMAX || M > 8) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_tango_reference: M + K - 1 > 16 or M > 8");
    // the workspace: the caller's, or a context-owned one kept in `own_ws` (shared with the enhanced-output entry points)
    const RefLayout l = ref_layout(ctx);
    char* ws = (char*)workspace;
    if (ws) {
        if (workspace_bytes < l.total) return fail(ctx, DISCO_E_ARG, "disco_tango_reference: workspace too small");
    } else {
        if (ctx->own_ws_bytes < l.total) {
            if (steps == 2) return fail(ctx, DISCO_E_ARG, "disco_tango_reference: steps = 2 needs the workspace of the preceding steps = 1 call");
            if (ctx->own_ws) {
                HIPCHK(ctx, hipFree(ctx->own_ws));
                ctx->own_ws = nullptr;
                ctx->own_ws_bytes = 0;
            }
            HIPCHK(ctx, hipMalloc(&ctx->own_ws, l.total));
            ctx->own_ws_bytes = l.total;
        }
        ws = (char*)ctx->own_ws;
    }
    hipStream_t st = (hipStream_t)s;
    const int64_t G = (int64_t)c.rooms * K;
    const long long nTF = (long long)G * ctx->T * ctx->F;
    const size_t plane_b = (size_t)nTF * sizeof(c32), mask_b = (size_t)nTF * sizeof(float);
    disco_c32 *Xy = (disco_c32*)(ws + l.Xy), *Xs = (disco_c32*)(ws + l.Xs), *Xn = (disco_c32*)(ws + l.Xn);
    disco_c32 *zy = (disco_c32*)(ws + l.zy), *zs = (disco_c32*)(ws + l.zs), *zn_ = (disco_c32*)(ws + l.zn_), *znres = (disco_c32*)(ws + l.znres);
    disco_c32 *rows_s = (disco_c32*)(ws + l.rows_s), *rows_n = (disco_c32*)(ws + l.rows_n);
    float *mz = (float*)(ws + l.mz), *mw = (float*)(ws + l.mw), *mc = (float*)(ws + l.mc);
    disco_c32 *Rss = (disco_c32*)(ws + l.Rss), *Rnn = (disco_c32*)(ws + l.Rnn), *Rtmp = (disco_c32*)(ws + l.Rtmp), *w_loc = (disco_c32*)(ws + l.w_loc), *w_glo = (disco_c32*)(ws + l.w_glo);
    const float thr = powf(10.f, c.mask_bin_thr_db / 10.f);
    const bool oracle_sigs = mask_for_z == DISCO_MZ_ORACLE_REFS || mask_for_z == DISCO_MZ_ORACLE_ZS;     // tango.py:343
    auto give = [&](void* dst, const void* src, size_t bytes) -> int {
        if (dst && dst != src) HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
        return 0;
    };
    int rc;
    if (steps & 1) {
        // ---- STFTs of the mixture and of both images (tango.py:335-337)
        if ((rc = disco_stft(ctx, y, G, M, Xy, s))) return rc;
        if ((rc = disco_stft(ctx, s_img, G, M, Xs, s))) return rc;
        if ((rc = disco_stft(ctx, n_img, G, M, Xn, s))) return rc;
        // ---- step-1 mask at the reference microphone (tango.py:338-342)
        if (mask_z_in) {
            HIPCHK(ctx, hipMemcpyAsync(mz, mask_z_in, mask_b, hipMemcpyDeviceToDevice, st));
        } else {
            hipLaunchKernelGGL(k_tf_mask_channel, dim3(ew_grid(nTF)), dim3(256), 0, st, (const c32*)Xs, (const c32*)Xn, mz, nTF, M, c.ref_mic,
                               c.mask_type, c.mask_pow, thr);
            if ((rc = check_launch(ctx, "k_tf_mask_channel"))) return rc;
        }
        // ---- step 1: local statistics, filter, compressed signals (tango.py:343-376)
        if (oracle_sigs) {            // s_hat = S, n_hat = N: Rss from the target image alone, Rnn from the noise image alone
            hipLaunchKernelGGL(k_fill_f32, dim3(ew_grid(nTF)), dim3(256), 0, st, mc, 1.f, nTF);
            if ((rc = disco_cov_masked(ctx, Xs, mc, nullptr, nullptr, 0, M, Rss, Rtmp, s))) return rc;      // mask 1: Rss = <S S^H>  (Rtmp = 0)
            hipLaunchKernelGGL(k_fill_f32, dim3(ew_grid(nTF)), dim3(256), 0, st, mc, 0.f, nTF);
            if ((rc = disco_cov_masked(ctx, Xn, mc, nullptr, nullptr, 0, M, Rtmp, Rnn, s))) return rc;      // mask 0: Rnn = <N N^H>  (Rtmp = 0)
            if ((rc = disco_gevd_mwf_r1(ctx, Rss, Rnn, G * ctx->F, M, c.mu, w_loc, nullptr, s))) return rc;
        } else {
            int chunks = 1;
            if ((rc = cov_partials(ctx, Xy, mz, nullptr, nullptr, 0, M, &chunks, s))) return rc;
            if ((rc = disco_gevd_mwf_r1_pending(ctx, c.mu, w_loc, nullptr, s))) return rc;
        }
        if ((rc = disco_apply(ctx, Xy, nullptr, w_loc, M, 1, zy, s))) return rc;
        if ((rc = disco_apply(ctx, Xs, nullptr, w_loc, M, 1, zs, s))) return rc;
        if ((rc = disco_apply(ctx, Xn, nullptr, w_loc, M, 1, zn_, s))) return rc;
        if ((rc = disco_noise_residual(ctx, Xy, zy, znres, s))) return rc;                                  // tango.py:376
        if ((rc = give(out->z_y, zy, plane_b)) || (rc = give(out->z_s, zs, plane_b)) || (rc = give(out->z_n, zn_, plane_b)) ||
            (rc = give(out->zn, znres, plane_b)) || (rc = give(out->masks_z, mz, mask_b)))
            return rc;
    }
    if (!(steps & 2)) return 0;
    // ---- step-2 mask at channel 0 (tango.py:388-394)
    if (mask_w_in) {
        HIPCHK(ctx, hipMemcpyAsync(mw, mask_w_in, mask_b, hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL(k_tf_mask_channel, dim3(ew_grid(nTF)), dim3(256), 0, st, (const c32*)Xs, (const c32*)Xn, mw, nTF, M, 0,
                           c.mask_type, c.mask_pow, thr);
        if ((rc = check_launch(ctx, "k_tf_mask_channel"))) return rc;
    }
    if ((rc = give(out->mask_w, mw, mask_b))) return rc;
    // ---- the exchanged rows (tango.py:396-429) and the global statistics (433-440)
    const disco_c32 *Zs_rows = zy, *Zn_rows = zy;
    int mask_remote = 0;
    if (K > 1) {
        switch (mask_for_z) {
            case DISCO_MZ_LOCAL: mask_remote = 1; break;
            case DISCO_MZ_NONE: Zn_rows = znres; break;
            case DISCO_MZ_DISTANT:
                hipLaunchKernelGGL(k_mask_rows, dim3(ew_grid(nTF)), dim3(256), 0, st, (const c32*)zy, (const float*)mw, (c32*)rows_s, (c32*)rows_n, nTF);
                Zs_rows = rows_s;
                Zn_rows = rows_n;
                break;
            case DISCO_MZ_COMPRESSED: {           // the sender's mask from ITS compressed target / noise (get_mask(z_s, z_n), tango.py:402-403)
                hipLaunchKernelGGL(k_tf_mask, dim3(ew_grid(nTF)), dim3(256), 0, st, (const c32*)zs, (const c32*)zn_, mc, nTF, c.mask_type,
                                   c.mask_pow, thr);
                hipLaunchKernelGGL(k_mask_rows, dim3(ew_grid(nTF)), dim3(256), 0, st, (const c32*)zy, (const float*)mc, (c32*)rows_s, (c32*)rows_n, nTF);
                Zs_rows = rows_s;
                Zn_rows = rows_n;
            } break;
            case DISCO_MZ_ORACLE_REFS:
                hipLaunchKernelGGL(k_pick_channel, dim3(ew_grid(nTF)), dim3(256), 0, st, (const c32*)Xs, (c32*)rows_s, nTF, M, c.ref_mic);
                hipLaunchKernelGGL(k_pick_channel, dim3(ew_grid(nTF)), dim3(256), 0, st, (const c32*)Xn, (c32*)rows_n, nTF, M, c.ref_mic);
                Zs_rows = rows_s;
                Zn_rows = rows_n;
                break;
            case DISCO_MZ_ORACLE_ZS: Zs_rows = zs; Zn_rows = zn_; break;
            default: break;                   // 'previous': unmasked z_y in both
        }
        if ((rc = check_launch(ctx, "reference rows"))) return rc;
    }
    int chunks2 = 1;
    if ((rc = cov_partials(ctx, Xy, mw, K > 1 ? Zs_rows : nullptr, K > 1 ? Zn_rows : nullptr, mask_remote, P2, &chunks2, s))) return rc;
    if ((rc = disco_gevd_mwf_r1_pending(ctx, c.mu, w_glo, nullptr, s))) return rc;
    // ---- the global filter on the mixture and on both images (tango.py:445-450); outputs straight into the caller's arrays
    if (out->yf && (rc = disco_apply(ctx, Xy, K > 1 ? zy : nullptr, w_glo, P2, 1, out->yf, s))) return rc;
    if (out->sf && (rc = disco_apply(ctx, Xs, K > 1 ? zs : nullptr, w_glo, P2, 1, out->sf, s))) return rc;
    if (out->nf && (rc = disco_apply(ctx, Xn, K > 1 ? zn_ : nullptr, w_glo, P2, 1, out->nf, s))) return rc;
    return 0;
}

// ---- helper of the mask-estimation DNN (disco_amd/dnn/crnn.py): the pointwise half of a GRU step -----------------------------
extern "C" int disco_gru_gates(disco_ctx* ctx, const float* gi, int64_t gi_stride, const float* gh, const float* gh_bias,
                               const float* h_prev, float* h_out, int64_t n, int H, disco_stream s) {
    // ctx may be NULL (the DNN owns no context): the launch then goes to the calling thread's current device
    if (!gi || !h_out || (!gh && !gh_bias) || n < 1 || H < 1 || gi_stride < 3 * (int64_t)H) return ctx ? fail(ctx, DISCO_E_ARG, "disco_gru_gates: bad argument") : DISCO_E_ARG;
    DevGuard dev_guard_(ctx ? ctx->cfg.device : [] { int d = 0; (void)hipGetDevice(&d); return d; }());
    const long long total = (long long)n * H;
    hipLaunchKernelGGL(k_gru_gates, dim3((unsigned)std::min<long long>((total + 255) / 256, 65536)), dim3(256), 0, (hipStream_t)s, gi,
                       (long long)gi_stride, gh, gh_bias, h_prev, h_out, (long long)n, H);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return DISCO_E_HIP_BASE - (int)e;
    return 0;
}

extern "C" int disco_maxpool_last4(disco_ctx* ctx, const float* x, const float* bias, int64_t n_rows, int row_len, int rows_per_channel,
                                   int channels, float* out, disco_stream s) {
    if (!x || !out || n_rows < 1 || row_len < 4 || (bias && (rows_per_channel < 1 || channels < 1))) return ctx ? fail(ctx, DISCO_E_ARG, "disco_maxpool_last4: bad argument") : DISCO_E_ARG;
    DevGuard dev_guard_(ctx ? ctx->cfg.device : [] { int d = 0; (void)hipGetDevice(&d); return d; }());
    const long long total = (long long)n_rows * (row_len / 4);
    hipLaunchKernelGGL(k_maxpool_last4, dim3((unsigned)std::min<long long>((total + 255) / 256, 1 << 20)), dim3(256), 0, (hipStream_t)s, x, bias, out,
                       (long long)n_rows, row_len, rows_per_channel > 0 ? rows_per_channel : 1, channels > 0 ? channels : 1);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : DISCO_E_HIP_BASE - (int)e;
}

extern "C" int disco_selftest_pk(disco_ctx* ctx, const disco_c32* a, const disco_c32* b, const disco_c32* c, int64_t n,
                                 disco_c32* out_hw, disco_c32* out_ref, disco_stream s) {
    static_assert(PK_SELFTEST_OPS == DISCO_PK_SELFTEST_OPS, "header and kernel disagree");
    if (!a || !b || !c || !out_hw || !out_ref || n < 1) return ctx ? fail(ctx, DISCO_E_ARG, "disco_selftest_pk: bad argument") : DISCO_E_ARG;
    DevGuard dev_guard_(ctx ? ctx->cfg.device : [] { int d = 0; (void)hipGetDevice(&d); return d; }());
    hipLaunchKernelGGL(k_pk_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, (const c32*)a, (const c32*)b, (const c32*)c,
                       (long long)n, (c32*)out_hw, (c32*)out_ref);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : DISCO_E_HIP_BASE - (int)e;
}

extern "C" int disco_crnn_windows(disco_ctx* ctx, const float* feat, int64_t B, int C, int Tp, int T, int W, int n_keep, float* out,
                                  disco_stream s) {
    if (!feat || !out || B < 1 || C < 1 || T < 1 || W < 1 || Tp < T + W - 1 || n_keep < 4 || n_keep % 4 || n_keep > C * W * 4 ||
        ((uintptr_t)feat & 15) || ((uintptr_t)out & 15))
        return ctx ? fail(ctx, DISCO_E_ARG, "disco_crnn_windows: bad argument") : DISCO_E_ARG;
    DevGuard dev_guard_(ctx ? ctx->cfg.device : [] { int d = 0; (void)hipGetDevice(&d); return d; }());
    const long long total = (long long)B * T * (n_keep / 4);
    hipLaunchKernelGGL(k_crnn_windows, dim3((unsigned)std::min<long long>((total + 255) / 256, 1 << 20)), dim3(256), 0, (hipStream_t)s,
                       (const float4*)feat, (float4*)out, (long long)B, C, Tp, T, W, n_keep / 4);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : DISCO_E_HIP_BASE - (int)e;
}

// ---- online / adaptive mode ------------------------------------------------------------------------------

extern "C" int disco_online_mwf(disco_ctx* ctx, const disco_c32* X, const disco_c32* Z, const float* mask, int P,
                                float lambda_cor, float mu, int update_every, float init_diag, disco_c32* out,
                                disco_c32* w_last, disco_stream s) {
    DISCO_ENTER(ctx);
    const disco_cfg& c = ctx->cfg;
    if (!X || !mask || !out) return fail(ctx, DISCO_E_ARG, "disco_online_mwf: null argument");
    if (P != c.mics && P != c.mics + c.nodes - 1) return fail(ctx, DISCO_E_ARG, "disco_online_mwf: P must be mics or mics + nodes - 1");
    if (P > c.mics && !Z) return fail(ctx, DISCO_E_ARG, "disco_online_mwf: P > mics needs the exchanged z");
    if (!(lambda_cor >= 0.f && lambda_cor < 1.f) || update_every < 1 || !(init_diag > 0.f))
        return fail(ctx, DISCO_E_ARG, "disco_online_mwf: need 0 <= lambda < 1, update_every >= 1, init_diag > 0");
    if (P > 16) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_online_mwf: P must be <= 16");
    OnlineArgs a;
    a.X = (const c32*)X;
    a.Z = P > c.mics ? (const c32*)Z : nullptr;
    a.mask = mask;
    a.out = (c32*)out;
    a.w_last = (c32*)w_last;
    a.K = c.nodes;
    a.Kl = ctx->Kl;
    a.k0 = ctx->k0;
    a.T = ctx->T;
    a.F = ctx->F;
    a.M = c.mics;
    a.update_every = update_every;
    a.lambda_cor = lambda_cor;
    a.init_diag = init_diag;
    a.mu = (double)mu;
    a.n_prob = (long long)c.rooms * ctx->Kl * ctx->F;
    a.zblk = ctx->zblk;
    a.R = c.rooms;
    hipStream_t st = (hipStream_t)s;
    switch (P) {
#define C_(P_)                                                                                                          \
    case P_: {                          /* one thread per (room, node, bin) */                                          \
        const long long grid = (a.n_prob + SOLVE_SMALL_THREADS - 1) / SOLVE_SMALL_THREADS;                              \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_online_mwf_thread<P_>), dim3((unsigned)grid), dim3(SOLVE_SMALL_THREADS), 0, st, a); \
    } break;
        C_(1) C_(2) C_(3) C_(4)
#undef C_
#define C_(P_)                                                                                                          \
    case P_: {                          /* a group of 8 / 16 lanes per (room, node, bin) */                             \
        const long long grid = (a.n_prob + SolveGeom<P_>::PROBS - 1) / SolveGeom<P_>::PROBS;                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_online_mwf<P_>), dim3((unsigned)grid), dim3(SolveGeom<P_>::THREADS), 0, st, a); \
    } break;
        C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
#undef C_
        default: return fail(ctx, DISCO_E_UNSUPPORTED, "disco_online_mwf: no kernel for this P");
    }
    return check_launch(ctx, "k_online_mwf");
}

extern "C" int disco_tango_online(disco_ctx* ctx, const float* y, const float* mask_z, const float* mask_w, float lambda_cor,
                                  int update_every, float init_diag, float* out, disco_c32* z_y, disco_c32* yf,
                                  void* workspace, size_t workspace_bytes, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!y || !mask_z || !mask_w || !out) return fail(ctx, DISCO_E_ARG, "disco_tango_online: null argument");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_tango_online: node shard active, drive disco_online_mwf around an all-gather of z");
    const disco_cfg& c = ctx->cfg;
    const WsLayout l = ws_layout(ctx);
    char* ws = nullptr;
    int rc = acquire_ws(ctx, workspace, workspace_bytes, l, &ws, "disco_tango_online");
    if (rc) return rc;
    disco_c32* X = (disco_c32*)(ws + l.X);
    disco_c32* z = z_y ? z_y : (disco_c32*)(ws + l.z);
    disco_c32* yo = yf ? yf : (disco_c32*)(ws + l.yf);
    const int64_t G = (int64_t)c.rooms * c.nodes;
    if ((rc = STAGE(ctx, s, "stft", disco_stft(ctx, y, G, c.mics, X, s)))) return rc;
    if ((rc = STAGE(ctx, s, "online1", disco_online_mwf(ctx, X, nullptr, mask_z, c.mics, lambda_cor, c.mu, update_every, init_diag, z, nullptr, s)))) return rc;
    if (c.nodes == 1 && mask_w == mask_z) {             // nothing to append: step 2 would repeat step 1
        if (yf) HIPCHK(ctx, hipMemcpyAsync(yf, z, (size_t)G * ctx->T * ctx->F * sizeof(c32), hipMemcpyDeviceToDevice, (hipStream_t)s));
        return disco_istft(ctx, z, G, out, s);
    }
    if ((rc = STAGE(ctx, s, "online2", disco_online_mwf(ctx, X, z, mask_w, c.mics + c.nodes - 1, lambda_cor, c.mu, update_every, init_diag, yo, nullptr, s)))) return rc;
    return STAGE(ctx, s, "istft", disco_istft(ctx, yo, G, out, s));
}

// ---- evaluation metrics ----------------------------------------------------------------------------------

extern "C" int disco_pair_stats(disco_ctx* ctx, const float* a, const float* b, int64_t n_sig, int64_t len, int start, int stop,
                                double* stats, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!a || !b || !stats || n_sig < 1 || len < 1) return fail(ctx, DISCO_E_ARG, "disco_pair_stats: bad argument");
    if (start < 0 || stop > len || stop < start) return fail(ctx, DISCO_E_ARG, "disco_pair_stats: need 0 <= start <= stop <= len");
    if (n_sig > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_pair_stats: batch too large");
    hipLaunchKernelGGL(k_pair_stats, dim3((unsigned)n_sig), dim3(METRIC_THREADS), 0, (hipStream_t)s, a, b, (long long)len, start, stop, stats);
    return check_launch(ctx, "k_pair_stats");
}

extern "C" int disco_band_stats(disco_ctx* ctx, const float* x, int64_t n_sig, int64_t len, int start, int stop,
                                const double* b, const double* a, int n_bands, double* stats, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!x || !b || !a || !stats || n_sig < 1 || len < 1) return fail(ctx, DISCO_E_ARG, "disco_band_stats: bad argument");
    if (start < 0 || stop > len || stop < start) return fail(ctx, DISCO_E_ARG, "disco_band_stats: need 0 <= start <= stop <= len");
    if (n_bands < 1 || n_bands > METRIC_THREADS) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_band_stats: 1 <= n_bands <= 256");
    const int spb = std::min(IIR_MAX_SPB, METRIC_THREADS / n_bands);
    const long long grid = (n_sig + spb - 1) / spb;
    if (grid > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_band_stats: batch too large");
    hipLaunchKernelGGL(k_band_stats, dim3((unsigned)grid), dim3(METRIC_THREADS), 0, (hipStream_t)s, x, (long long)n_sig, (long long)len,
                       start, stop, b, a, n_bands, spb, stats);
    return check_launch(ctx, "k_band_stats");
}

// ---- 'ivad' mask (tango.py:217-221 + sigproc_utils.py:12-55) ---------------------------------------------------------------

extern "C" int disco_mask_ivad(disco_ctx* ctx, const float* s_ref, int64_t n_sig, float* mask, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!s_ref || !mask || n_sig < 1) return fail(ctx, DISCO_E_ARG, "disco_mask_ivad: bad argument");
    const disco_cfg& c = ctx->cfg;
    if ((c.length + c.hop - 1) / c.hop > VAD_MAX_SEG) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_mask_ivad: signal longer than 4096 hops");
    if (n_sig > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_mask_ivad: batch too large");
    hipLaunchKernelGGL(k_vad_mask, dim3((unsigned)n_sig), dim3(VAD_THREADS), 0, (hipStream_t)s, s_ref, mask, c.length, ctx->T, ctx->F,
                       c.n_fft, c.hop, 0.001f, 0.99f, 2);
    return check_launch(ctx, "k_vad_mask");
}

// ---- RIR convolution, the step before the path -----------------------------------------------------------

extern "C" int disco_rir_convolve(disco_ctx* ctx, const float* dry, const float* rir, int64_t n_sig, int n_ch, int dry_len,
                                  int rir_len, float* out, int out_len, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!dry || !rir || !out || n_sig < 1 || n_ch < 1 || dry_len < 1 || rir_len < 1 || out_len < 1)
        return fail(ctx, DISCO_E_ARG, "disco_rir_convolve: bad argument");
    const int P = (rir_len + CV_B - 1) / CV_B;
    if (P > 16) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_rir_convolve: impulse responses longer than 8192 taps");
    const int nb = (out_len + CV_B - 1) / CV_B;
    if (n_sig * n_ch > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_rir_convolve: batch too large");
    hipStream_t st = (hipStream_t)s;
    if (!ctx->d_tw_conv) {
        if (ctx->cfg.n_fft == CV_N) {
            ctx->d_tw_conv = ctx->d_tw;
        } else {
            std::vector<c32> tw(CV_N);
            const double two_pi = 6.283185307179586476925286766559;
            for (int i = 0; i < CV_N; ++i) {
                tw[i].x = (float)std::cos(two_pi * i / CV_N);
                tw[i].y = (float)(-std::sin(two_pi * i / CV_N));
            }
            HIPCHK(ctx, hipMalloc((void**)&ctx->d_tw_conv, CV_N * sizeof(c32)));
            HIPCHK(ctx, hipMemcpy(ctx->d_tw_conv, tw.data(), CV_N * sizeof(c32), hipMemcpyHostToDevice));
        }
    }
    const size_t x_bytes = (size_t)n_sig * nb * CV_F * sizeof(c32);
    const size_t h_bytes = (size_t)n_sig * n_ch * P * CV_F * sizeof(c32);
    const size_t need = align_up(x_bytes) + h_bytes;
    if (ctx->conv_ws_bytes < need) {
        if (ctx->conv_ws) HIPCHK(ctx, hipFree(ctx->conv_ws));
        ctx->conv_ws = nullptr;
        ctx->conv_ws_bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->conv_ws, need));
        ctx->conv_ws_bytes = need;
    }
    c32* X = (c32*)ctx->conv_ws;
    c32* H = (c32*)((char*)ctx->conv_ws + align_up(x_bytes));
    const long long nx = (long long)n_sig * nb, nh = (long long)n_sig * n_ch * P;
    auto grid_of = [](long long items) { return dim3((unsigned)std::min<long long>((items + CV_WAVES - 1) / CV_WAVES, 1 << 20)); };
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_spectra<true>), grid_of(nx), dim3(64 * CV_WAVES), 0, st, dry, (long long)dry_len, nb, nx, X,
                       ctx->d_tw_conv);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_spectra<false>), grid_of(nh), dim3(64 * CV_WAVES), 0, st, rir, (long long)rir_len, P, nh, H,
                       ctx->d_tw_conv);
    const dim3 grid((unsigned)(n_sig * n_ch));
    if (P <= 8)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_mac_ifft<8>), grid, dim3(64 * CV_WAVES), 0, st, X, H, out, ctx->d_tw_conv, n_ch, nb, P, out_len);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_mac_ifft<16>), grid, dim3(64 * CV_WAVES), 0, st, X, H, out, ctx->d_tw_conv, n_ch, nb, P, out_len);
    return check_launch(ctx, "k_conv_mac_ifft");
}

// ---- iterated (DANSE-style) continuation -------------------------------------------------------------------------------

extern "C" int disco_tango_enhance_iterated(disco_ctx* ctx, const float* y, const float* mask_z, const float* mask_w, int iters,
                                            float* out, disco_c32* z_y, disco_c32* yf, void* workspace, size_t workspace_bytes,
                                            disco_stream s) {
    DISCO_ENTER(ctx);
    if (!y || !mask_z || !mask_w || !out || iters < 1) return fail(ctx, DISCO_E_ARG, "disco_tango_enhance_iterated: bad argument");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_tango_enhance_iterated: node shard active");
    const disco_cfg& c = ctx->cfg;
    const WsLayout l = ws_layout(ctx);
    char* ws = nullptr;
    int rc = acquire_ws(ctx, workspace, workspace_bytes, l, &ws, "disco_tango_enhance_iterated");
    if (rc) return rc;
    disco_c32* X = (disco_c32*)(ws + l.X);
    disco_c32* z = z_y ? z_y : (disco_c32*)(ws + l.z);
    disco_c32* yo = yf ? yf : (disco_c32*)(ws + l.yf);
    disco_c32* w_loc = (disco_c32*)(ws + l.w);
    disco_c32* w_glo = (disco_c32*)(ws + l.w2);
    const int64_t G = (int64_t)c.rooms * c.nodes;
    const int M = c.mics, P2 = c.mics + c.nodes - 1;
    int chunks = 1;
    if ((rc = stft_cov_partials(ctx, y, mask_z, X, &chunks, s))) return rc;
    if ((rc = STAGE(ctx, s, "solve1", disco_gevd_mwf_r1_pending(ctx, c.mu, w_loc, nullptr, s)))) return rc;
    for (int it = 0; it < iters; ++it) {
        // compression with the current w_loc (step 1's, then the local part of the previous iteration's filter) and the step-2
        // statistics: ONE pass over X for every node of a room where the shape allows it (k_room_cov), else the filter pass
        // followed by the covariance pass that reads X again and the K - 1 remote z's
        int chunks2 = 1;
        if (mask_w == mask_z && room_cov_ok(ctx, X, mask_w)) {
            if ((rc = STAGE(ctx, s, "room_cov2", room_cov_partials(ctx, X, mask_w, w_loc, z, &chunks2, s)))) return rc;
        } else {
            if ((rc = STAGE(ctx, s, "apply1", disco_apply(ctx, X, nullptr, w_loc, M, 1, z, s)))) return rc;
            if ((rc = STAGE(ctx, s, "cov2", cov_partials(ctx, X, mask_w, c.nodes > 1 ? z : nullptr, c.nodes > 1 ? z : nullptr, 1, P2,
                                                         &chunks2, s, mask_w == mask_z && c.nodes > 1)))) return rc;
        }
        if ((rc = STAGE(ctx, s, "solve2", disco_gevd_mwf_r1_pending(ctx, c.mu, w_glo, nullptr, s)))) return rc;
        if (it + 1 < iters) {
            // yf of this iteration is not needed; the next one re-compresses with the local part of this iteration's filter
            const long long nb = (long long)G * ctx->F;
            hipLaunchKernelGGL(k_filter_head, dim3((unsigned)std::min<long long>((nb * M + 255) / 256, 65535)), dim3(256), 0, (hipStream_t)s,
                               (const c32*)w_glo, (c32*)w_loc, nb, M, P2);
            if ((rc = check_launch(ctx, "k_filter_head"))) return rc;
        }
    }
    if ((rc = STAGE(ctx, s, "apply2", disco_apply(ctx, X, c.nodes > 1 ? z : nullptr, w_glo, P2, 1, yo, s)))) return rc;
    return STAGE(ctx, s, "istft", disco_istft(ctx, yo, G, out, s));
}

// Local part of a P-entry filter (the iterated scheme's re-compression filter); honours the node shard, so the node-sharded
// driver can run the DANSE-style iterations with one all-gather of z per iteration.
extern "C" int disco_filter_head(disco_ctx* ctx, const disco_c32* w_glo, int P, disco_c32* w_loc, disco_stream s) {
    DISCO_ENTER(ctx);
    const disco_cfg& c = ctx->cfg;
    if (!w_glo || !w_loc) return fail(ctx, DISCO_E_ARG, "disco_filter_head: null argument");
    if (P < c.mics) return fail(ctx, DISCO_E_ARG, "disco_filter_head: P < M");
    const long long nb = (long long)c.rooms * ctx->Kl * ctx->F;
    hipLaunchKernelGGL(k_filter_head, dim3((unsigned)std::min<long long>((nb * c.mics + 255) / 256, 65535)), dim3(256), 0, (hipStream_t)s,
                       (const c32*)w_glo, (c32*)w_loc, nb, c.mics, P);
    return check_launch(ctx, "k_filter_head");
}

// ---- image-source RIR generator --------------------------------------------------------------------------

extern "C" int disco_ism_rir(disco_ctx* ctx, const float* room_dims, const float* absorption, const float* src, const float* mic,
                             int64_t n_room, int n_src, int n_mic, int max_order, float fs, float c_sound, float* rir, int rir_len,
                             disco_stream s) {
    DISCO_ENTER(ctx);
    if (!room_dims || !absorption || !src || !mic || !rir || n_room < 1 || n_src < 1 || n_mic < 1 || max_order < 0 || !(fs > 0.f) ||
        !(c_sound > 0.f) || rir_len < 1)
        return fail(ctx, DISCO_E_ARG, "disco_ism_rir: bad argument");
    if (rir_len > ISM_MAX_LEN) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_ism_rir: responses longer than 8192 taps");
    if (max_order > 64) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_ism_rir: max_order > 64");
    const long long n = (long long)n_room * n_src * n_mic;
    if (n > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_ism_rir: batch too large");
    hipLaunchKernelGGL(k_ism_rir, dim3((unsigned)n), dim3(ISM_THREADS), 0, (hipStream_t)s, room_dims, absorption, src, mic, n_src, n_mic,
                       max_order, fs, c_sound, rir, rir_len);
    return check_launch(ctx, "k_ism_rir");
}
