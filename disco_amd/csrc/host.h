// Host-side internals of libdisco_hip.so shared by its translation units (api_*.hip): the context, the error / device-guard /
// stage-timer helpers, and the stage functions the whole-path entry points chain.  Not part of the C ABI (include/disco_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/disco_hip.h"
#include "common.h"


using disco::c32;

// per-context options (disco_set_option / disco_get_option); the environment variable of the same meaning presets the value at
// disco_create and is never looked at again
enum {
    DISCO_OPT_ROOM_COV = 0,               // "room_cov": one-pass room kernel for the wide shapes (0: apply + staged covariance)
    DISCO_OPT_OVERLAP_SOLVES,           // "overlap_solves": whole-path calls run the batch as two halves on two streams
    DISCO_OPT_SOLVE_DPP,                // "solve_dpp": 9 <= P <= 16 solved in registers with DPP row broadcasts (k_solve_dpp.h; 0: the LDS group solver)
    DISCO_OPT_SOLVE_THREAD,             // "solve_thread": 5 <= P <= 8 solved one THREAD per pencil (k_solve_small.h at one wave per SIMD, AGPRs as the second register file) instead of the LDS group solver
    DISCO_OPT_FUSE_WIDE_ISTFT,          // "fuse_wide_istft": whole-path calls of the wide shapes (P > 8) end in ONE filter + iSTFT pass (k_apply_istft_wide) instead of disco_apply + disco_istft
    DISCO_OPT_ONLINE_SQ32,              // "online_sq32": the online mode's thread solves (P <= 7) square in packed float32 (k_solve_small.h); 0: float64 throughout
    DISCO_N_OPTIONS
};
namespace disco_host {
struct OptionInfo {
    const char *key, *env;
    int def;
};
const OptionInfo* option_table();
}  // namespace disco_host

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
    const void* ref_ws;              // workspace in which a disco_tango_reference(steps = 1) call left its state for a steps = 2 call (else NULL)
    const void *ref_y, *ref_s, *ref_n;   // ... and the inputs that state was computed from
    c32* d_tw_conv;                  // 1024-point twiddles of disco_rir_convolve (== d_tw when n_fft is 1024), lazy
    void* conv_ws;                   // its spectra workspace, lazy
    size_t conv_ws_bytes;
    int k0, Kl;                      // node shard: this context holds nodes [k0, k0 + Kl) of every room (default 0, K)
    int zblk;                        // layout of the exchanged-signal arguments Zs / Zn / Z (disco_set_z_blocks; default K = plain)
    int tune_runw, tune_cov_chunks, tune_step2_chunks, tune_pairs;   // disco_set_tuning overrides (0 = batch-size heuristic)
    int opt[DISCO_N_OPTIONS];        // disco_set_option values (DISCO_OPT_*)
    int n_cu;                        // compute units of cfg.device (the persistent kernels start one workgroup per CU)
    int geom_rooms;                  // batch size the launch-geometry heuristics look at (cfg.rooms; a half-batch child: its parent's)
    // two half-batch children (rooms split [0, R/2) and [R/2, R)) + the second stream / events of the overlapped whole-path calls
    // (DISCO_OPT_OVERLAP_SOLVES): one half's solves run beside the other half's streaming kernels.  NULL when not in use.
    disco_ctx* half[2];
    disco_ctx* parent;               // set in a child
    hipStream_t side_stream;
    hipEvent_t ev_fork, ev_join;
    // per-stage hipEvent timers of the whole-path entry points (disco_stage_timing / disco_stage_report)
    struct StageRec {
        char name[32];
        std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
    };
    bool stage_on;
    std::vector<StageRec> stages;
    char err[512];
};

#define HIPCHK(ctx, call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s", #call, hipGetErrorString(e_)); \
            return DISCO_E_HIP_BASE - (int)e_;                                                  \
        }                                                                                       \
    } while (0)

static inline int fail(disco_ctx* ctx, int code, const char* msg) {
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s", msg);
    return code;
}

static inline int check_launch(disco_ctx* ctx, const char* what) {
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
static inline void stage_clear(disco_ctx* ctx) {
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

// whole-path entry points work on all nodes of a room and on their own plain [R][K] exchanged-signal arrays
static inline bool sharded(const disco_ctx* ctx) { return ctx->Kl != ctx->cfg.nodes || ctx->zblk != ctx->cfg.nodes; }

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

namespace disco_host {
using disco::c32;
struct WsLayout {
    size_t X, z, yf, Rss, Rnn, w, w2, total;
};
struct RefLayout {
    size_t Xy, Xs, Xn, zy, zs, zn_, znres, rows_s, rows_n, mz, mw, mc, Rss, Rnn, Rtmp, w_loc, w_glo, total;
};
inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
inline unsigned ew_grid(long long n) { return (unsigned)std::min<long long>((n + 255) / 256, 16384); }
WsLayout ws_layout(const disco_ctx* ctx);
RefLayout ref_layout(const disco_ctx* ctx);

// launch geometry (batch-size heuristics / disco_set_tuning)
int cov_chunks(const disco_ctx* ctx);
int room_chunks(const disco_ctx* ctx);
int cov1_f64_chunks(const disco_ctx* ctx);
int step2_chunks(const disco_ctx* ctx, int tiles_plus_1);
int stft_cov_chunks(const disco_ctx* ctx, int* runw_out);

// library-owned device memory
int ensure_scratch(disco_ctx* ctx, size_t bytes);
int ensure_scratch2(disco_ctx* ctx, size_t bytes);
int reserve_scratch(disco_ctx* ctx);
// half-batch children of the overlapped whole-path calls: created when the batch is large enough (or the option forces it)
int ensure_halves(disco_ctx* ctx);
bool overlap_applies(const disco_ctx* ctx);
int acquire_ws(disco_ctx* ctx, void* workspace, size_t workspace_bytes, const WsLayout& l, char** ws_out, const char* who);

// transforms of signals of any length with the context's window / FFT size / padding (api_stft.hip)
int stft_any(disco_ctx* ctx, const float* x, int64_t n_sig, int chans, disco_c32* X, int L, int T, disco_stream s);
int istft_any(disco_ctx* ctx, const disco_c32* Z, int64_t n_sig, float* out, int L, int T, disco_stream s, bool solo);

// stages (each leaves its partial sums pending in the context; see the definitions)
int cov_finalize(disco_ctx* ctx, int chunks, int P, disco_c32* Rss, disco_c32* Rnn, disco_stream s);
int cov_partials(disco_ctx* ctx, const disco_c32* X, const float* mask, const disco_c32* Zs, const disco_c32* Zn, int mask_remote, int P,
                 int* chunks_out, disco_stream s, bool skiploc = false);
bool room_cov_ok(const disco_ctx* ctx, const disco_c32* X, const float* mask);
int room_cov_partials(disco_ctx* ctx, const disco_c32* X, const float* mask, const disco_c32* w_loc, disco_c32* z, int* chunks_out,
                      disco_stream s, bool store_z = true);
int stft_cov_partials(disco_ctx* ctx, const float* y, const float* mask_z, disco_c32* X, int* chunks_out, disco_stream s,
                      bool store = true);
int step2_cov_partials(disco_ctx* ctx, const disco_c32* X, const float* mask_w, const disco_c32* w_loc, disco_c32* z_out, int* chunks_out,
                       disco_stream s, bool skiploc = false);
int stft_apply_istft(disco_ctx* ctx, const float* y, const disco_c32* w, float* out, disco_stream s);
bool step2_apply_istft_ok(const disco_ctx* ctx);
bool apply_istft_wide_ok(const disco_ctx* ctx);
int apply_istft_wide(disco_ctx* ctx, const disco_c32* X, const disco_c32* Z, const disco_c32* w, disco_c32* yf, float* out, disco_stream s);
}  // namespace disco_host
