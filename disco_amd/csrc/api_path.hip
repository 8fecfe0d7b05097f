// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): whole-path entry points
#include "host.h"

#include <functional>
#include "k_apply.h"
#include "k_cov.h"
#include "k_stft.h"

using namespace disco;
using namespace disco_host;

// ---------------------------------------------------------------------------------------------------------
// whole path
// ---------------------------------------------------------------------------------------------------------
namespace disco_host {
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
    // an overlapped call hands each half-batch child its own slice: the two slices must fit
    if (ctx->half[0] && ctx->half[1]) l.total = std::max(l.total, align_up(ws_layout(ctx->half[0]).total) + ws_layout(ctx->half[1]).total);
    return l;
}
}  // namespace disco_host

// ---- a whole-path call as a list of steps -----------------------------------------------------------------------------------
// Every whole-path entry point builds the list of its stages (one kernel family each) and runs it: in order on the caller's stream,
// or -- the overlapped forms, DISCO_OPT_OVERLAP_SOLVES -- over the two half-batch children:
//   1 / 2 (default / forced)  child A entirely on the caller's stream, child B entirely on the side stream: the two sequences drift
//         against each other, so a solve -- a compute kernel that moves almost no data -- mostly meets a streaming kernel of the other
//         half, and each half's kernel tails are filled by the other half.  Measured on C3: 19.67 -> 19.14 ms.
// (Round 3 also built a software-pipelined order -- streaming kernels on the caller's stream, only the solves on the side stream: every solve
// hidden, but two half-batch launches of a streaming kernel cost more than one whole-batch launch, C3 18.99 -> 19.29 ms; removed in round 5.)
// Nothing synchronises with the host, and the side stream is forked from / joined to the caller's, so the sequence can still be captured
// into one hipGraph.
namespace {
struct Step {
    const char* name;                          // stage name for the timers; nullptr: the callee brackets its own stages
    bool side;                                 // a solve: runs on the side stream of an overlapped call
    std::function<int(disco_stream)> run;
};
using Steps = std::vector<Step>;

int run_step(disco_ctx* ctx, Step& x, disco_stream st) { return x.name ? STAGE(ctx, st, x.name, x.run(st)) : x.run(st); }

int run_steps(disco_ctx* ctx, Steps& steps, disco_stream s) {
    for (auto& x : steps) {
        const int rc = run_step(ctx, x, s);
        if (rc) return rc;
    }
    return 0;
}

int run_pipelined(disco_ctx* ctx, Steps (&steps)[2], disco_stream s) {
    hipStream_t s0 = (hipStream_t)s, s1 = ctx->side_stream;
    const size_t n = steps[0].size();
    if (steps[1].size() != n) return fail(ctx, DISCO_E_ARG, "overlapped call: step lists do not match");
    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, s0));
    HIPCHK(ctx, hipStreamWaitEvent(s1, ctx->ev_fork, 0));
    int rc = 0;
    // one child per stream.  A failure inside the loop is collected, not returned: the join below must be reached whatever happened, or the
    // caller's stream stays forked (and a capture unjoined)
    for (int h = 0; h < 2 && !rc; ++h)
        for (size_t i = 0; i < n && !rc; ++i) {
            rc = run_step(ctx->half[h], steps[h][i], (disco_stream)(h ? s1 : s0));
            if (rc) snprintf(ctx->err, sizeof(ctx->err), "%.500s", ctx->half[h]->err);
        }
    const hipError_t e1 = hipEventRecord(ctx->ev_join, s1), e2 = hipStreamWaitEvent(s0, ctx->ev_join, 0);       // the join, whatever happened above
    if (!rc && (e1 != hipSuccess || e2 != hipSuccess)) {
        snprintf(ctx->err, sizeof(ctx->err), "joining the side stream failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        rc = DISCO_E_HIP_BASE - (int)(e1 != hipSuccess ? e1 : e2);
    }
    return rc;
}

// arguments of one (half-)batch of a whole-path call
struct PathArgs {
    const float *y, *mask_z, *mask_w;
    float* out;
    disco_c32 *z_y, *yf;
    char* ws;
};
// the slice of the caller's arrays / workspace that child h (rooms [r0, r0 + rooms_h)) works on
PathArgs child_args(const disco_ctx* ctx, const PathArgs& a, int h) {
    const disco_cfg& c = ctx->cfg;
    const size_t r0 = h ? (size_t)ctx->half[0]->cfg.rooms : 0;
    const size_t K = c.nodes, TF = (size_t)ctx->T * ctx->F, sy = K * c.mics * c.length, sm = K * TF, so = K * c.length;
    PathArgs b;
    b.y = a.y + r0 * sy;
    b.mask_z = a.mask_z + r0 * sm;
    b.mask_w = a.mask_w == a.mask_z ? b.mask_z : a.mask_w + r0 * sm;
    b.out = a.out + r0 * so;
    b.z_y = a.z_y ? a.z_y + r0 * sm : nullptr;
    b.yf = a.yf ? a.yf + r0 * sm : nullptr;
    b.ws = a.ws + (h ? align_up(ws_layout(ctx->half[0]).total) : 0);
    return b;
}
}  // namespace

extern "C" size_t disco_workspace_bytes(const disco_ctx* ctx) { return ctx ? ws_layout(ctx).total : 0; }

namespace disco_host {
// caller's workspace if given (size-checked), else the context's own (grown on demand)
int acquire_ws(disco_ctx* ctx, void* workspace, size_t workspace_bytes, const WsLayout& l, char** ws_out, const char* who) {
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
    if (ctx->ref_ws == ws) ctx->ref_ws = nullptr;      // a steps = 1 state of disco_tango_reference in this workspace is overwritten
    *ws_out = ws;
    return 0;
}
// Both partial-sum blocks at the largest size any covariance call of this context can ask for with the present geometry:
// [R * Kl][chunks][F][P (P + 1) / 2] float4 with P = M + K - 1 and the largest of the three chunk counts.
int reserve_scratch(disco_ctx* ctx) {
    const disco_cfg& c = ctx->cfg;
    const size_t G = (size_t)c.rooms * ctx->Kl;
    const size_t P = (size_t)std::min(c.mics + c.nodes - 1, 16);
    const size_t NP = P * (P + 1) / 2;
    int chunks = std::max(cov_chunks(ctx), step2_chunks(ctx, (ctx->F - 1) / 64 + 1));
    if (c.mics >= 7) chunks = std::max(chunks, 2 * cov1_f64_chunks(ctx));                       // the (hi, lo) pairs of k_cov_loc_f64
    if (c.mics <= 8) chunks = std::max(chunks, stft_cov_chunks(ctx, nullptr));
    if (c.mics + c.nodes - 1 > 8) chunks = std::max(chunks, room_chunks(ctx));
    const size_t need = G * (size_t)chunks * ctx->F * NP * sizeof(float4);
    int rc = ensure_scratch(ctx, need);
    if (!rc) rc = ensure_scratch2(ctx, need);
    return rc;
}
}  // namespace disco_host

// offline_tango's y branch as a list of steps (tango.py:326-450 + 528); `a` names this context's slice of the batch
static void enhance_steps(disco_ctx* ctx, const PathArgs& a, Steps& st) {
    const disco_cfg& c = ctx->cfg;
    const WsLayout l = ws_layout(ctx);
    const float *y = a.y, *mask_z = a.mask_z, *mask_w = a.mask_w;
    float* out = a.out;
    disco_c32 *z_y = a.z_y, *yf = a.yf;
    disco_c32* X = (disco_c32*)(a.ws + l.X);
    disco_c32* z = z_y ? z_y : (disco_c32*)(a.ws + l.z);
    disco_c32* yo = yf ? yf : (disco_c32*)(a.ws + l.yf);
    disco_c32* w = (disco_c32*)(a.ws + l.w);
    disco_c32* w2 = (disco_c32*)(a.ws + l.w2);
    const int64_t G = (int64_t)c.rooms * c.nodes;
    const int M = c.mics, P2 = c.mics + c.nodes - 1;
    const bool same_mask = mask_w == mask_z;
    auto solve_pending = [ctx](disco_c32* w_out) { return [ctx, w_out](disco_stream s) { return disco_gevd_mwf_r1_pending(ctx, ctx->cfg.mu, w_out, nullptr, s); }; };

    if (c.nodes == 1 && same_mask && !z_y && !yf && c.n_fft == 512 && M <= 4) {
        // single node, enhanced output only (config C2): nothing is materialised -- one pass over the samples for the
        // statistics, one for filter + iSTFT with the spectra recomputed (get_z_signals.py:274-315 + tango.py:528)
        st.push_back({nullptr, false, [=](disco_stream s) { int ch = 1; return stft_cov_partials(ctx, y, mask_z, nullptr, &ch, s, false); }});
        st.push_back({"solve1", true, solve_pending(w)});
        st.push_back({"stft_apply_istft", false, [=](disco_stream s) { return stft_apply_istft(ctx, y, w, out, s); }});
        return;
    }
    // step 1 (tango.py:326-376): STFT + covariance in one pass, solve straight from the partial sums
    st.push_back({nullptr, false, [=](disco_stream s) { int ch = 1; return stft_cov_partials(ctx, y, mask_z, X, &ch, s); }});
    st.push_back({"solve1", true, solve_pending(w)});

    if (c.nodes > 1 && P2 <= 8 && !(c.flags & DISCO_FLAG_STAGED_STEP2)) {
        // step 2 on the on-chip z exchange (default whenever all nodes of a room share the GPU and P <= 8)
        // same mask array in both steps (oracle masks; a DNN mask re-used, tango.py:388-389): the leading M x M block of the
        // step-2 covariances IS the step-1 covariance still held as partial sums -> not recomputed
        st.push_back({"step2_cov", false, [=](disco_stream s) {
            int ch = 1;
            if (same_mask && ctx->loc_M == M) return disco_step2_cov_fused_reuse(ctx, X, mask_w, w, z_y, s);
            return step2_cov_partials(ctx, X, mask_w, w, z_y, &ch, s);
        }});
        st.push_back({"solve2", true, solve_pending(w2)});
        if (!yf && c.n_fft == 512) {   // yf not asked for: filter + iSTFT in one pass, yf stays on chip (shapes the kernel takes)
            if (step2_apply_istft_ok(ctx)) {
                st.push_back({"step2_apply_istft", false, [=](disco_stream s) { return disco_step2_apply_istft_fused(ctx, X, w, w2, out, s); }});
                return;
            }
        }
        st.push_back({"step2_apply", false, [=](disco_stream s) { return disco_step2_apply_fused(ctx, X, w, w2, nullptr, yo, s); }});
        st.push_back({"istft", false, [=](disco_stream s) { return disco_istft(ctx, yo, G, out, s); }});
        return;
    }
    if (c.nodes == 1 && same_mask) {
        if (!z_y && !yf && step2_apply_istft_ok(ctx)) {
            // single node, enhanced output only: filter + iSTFT in one pass over X, z never reaches HBM
            st.push_back({"step2_apply_istft", false, [=](disco_stream s) { return disco_step2_apply_istft_fused(ctx, X, w, w, out, s); }});
            return;
        }
        // single node, same mask: step 2 would rebuild the very same statistics from the very same inputs
        // (P = M, nothing to append), so w_glo == w_loc and yf == z_y bit for bit (config C2).
        st.push_back({"apply1", false, [=](disco_stream s) { return disco_apply(ctx, X, nullptr, w, M, 1, z, s); }});
        st.push_back({"istft", false, [=](disco_stream s) {
            if (yf) HIPCHK(ctx, hipMemcpyAsync(yf, z, (size_t)G * ctx->T * ctx->F * sizeof(c32), hipMemcpyDeviceToDevice, (hipStream_t)s));
            return disco_istft(ctx, z, G, out, s);
        }});
        return;
    }
    // exchange + step 2 (tango.py:378-450) with z materialised, mask_for_z = 'local': wide shapes take z + the step-2 statistics of a
    // whole room in one pass (k_room_cov) when the context's state allows it, else the filter pass + the covariance pass
    st.push_back({nullptr, false, [=](disco_stream s) {
        int ch = 1, rc;
        if (c.nodes > 1 && same_mask && room_cov_ok(ctx, X, mask_w))
            return STAGE(ctx, s, "room_cov2", room_cov_partials(ctx, X, mask_w, w, z, &ch, s));
        if ((rc = STAGE(ctx, s, "apply1", disco_apply(ctx, X, nullptr, w, M, 1, z, s)))) return rc;
        return STAGE(ctx, s, "cov2", cov_partials(ctx, X, mask_w, c.nodes > 1 ? z : nullptr, c.nodes > 1 ? z : nullptr, 1, P2, &ch, s, same_mask && c.nodes > 1));
    }});
    st.push_back({"solve2", true, solve_pending(w)});
    if (c.nodes > 1 && apply_istft_wide_ok(ctx)) {      // wide shapes: yf stays on chip (and goes out only when the caller asked for it)
        st.push_back({"apply2_istft", false, [=](disco_stream s) { return apply_istft_wide(ctx, X, z, w, yf, out, s); }});
        return;
    }
    st.push_back({"apply2", false, [=](disco_stream s) { return disco_apply(ctx, X, c.nodes > 1 ? z : nullptr, w, P2, 1, yo, s); }});
    st.push_back({"istft", false, [=](disco_stream s) { return disco_istft(ctx, yo, G, out, s); }});
}

extern "C" int disco_tango_enhance(disco_ctx* ctx, const float* y, const float* mask_z, const float* mask_w, float* out,
                                   disco_c32* z_y, disco_c32* yf, void* workspace, size_t workspace_bytes, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!y || !mask_z || !mask_w || !out) return fail(ctx, DISCO_E_ARG, "disco_tango_enhance: null argument");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_tango_enhance: node shard active, drive the staged calls around an all-gather of z");
    const WsLayout l = ws_layout(ctx);
    char* ws = nullptr;
    int rcw = acquire_ws(ctx, workspace, workspace_bytes, l, &ws, "disco_tango_enhance");
    if (rcw) return rcw;
    const PathArgs a{y, mask_z, mask_w, out, z_y, yf, ws};
    const disco_cfg& c0 = ctx->cfg;
    const bool fused_route = c0.nodes > 1 && c0.mics + c0.nodes - 1 <= 8 && !(c0.flags & DISCO_FLAG_STAGED_STEP2);
    // by default only where it was measured to pay: the fused route (C3: 19.67 -> 19.14 ms); forced (2 / 3) for every route
    if (overlap_applies(ctx) && ctx->half[0] && ctx->half[1] && (fused_route || ctx->opt[DISCO_OPT_OVERLAP_SOLVES] >= 2)) {
        Steps st[2];
        for (int h = 0; h < 2; ++h) enhance_steps(ctx->half[h], child_args(ctx, a, h), st[h]);
        return run_pipelined(ctx, st, s);
    }
    Steps st;
    enhance_steps(ctx, a, st);
    return run_steps(ctx, st, s);
}

// ---- reference outputs: all nine returns of offline_tango, device resident ------------------------------------------------
namespace disco_host {
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
}  // namespace disco_host

extern "C" size_t disco_reference_workspace_bytes(const disco_ctx* ctx) { return ctx ? ref_layout(ctx).total : 0; }

extern "C" size_t disco_owned_bytes(const disco_ctx* ctx) {
    if (!ctx) return 0;
    return ctx->scratch_bytes + ctx->scratch2_bytes + ctx->own_ws_bytes + ctx->conv_ws_bytes + disco_owned_bytes(ctx->half[0]) +
           disco_owned_bytes(ctx->half[1]);
}

extern "C" int disco_reserve(disco_ctx* ctx, int own_workspace) {
    DISCO_ENTER(ctx);
    if (own_workspace < 0 || own_workspace > 2) return fail(ctx, DISCO_E_ARG, "disco_reserve: own_workspace must be 0, 1 or 2");
    int rc = reserve_scratch(ctx);
    for (int h = 0; h < 2 && !rc; ++h)
        if (ctx->half[h] && (rc = reserve_scratch(ctx->half[h]))) return fail(ctx, rc, ctx->half[h]->err);
    if (rc || !own_workspace) return rc;
    size_t need = ws_layout(ctx).total;
    if (own_workspace == 2) need = std::max(need, ref_layout(ctx).total);
    if (ctx->own_ws_bytes < need) {
        if (ctx->own_ws) {
            if (ctx->ref_ws == ctx->own_ws) ctx->ref_ws = nullptr;
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
    if (steps == 2 && !(ctx->ref_ws == ws && ctx->ref_y == y && ctx->ref_s == s_img && ctx->ref_n == n_img))
        return fail(ctx, DISCO_E_ARG, "disco_tango_reference: steps = 2 needs the state a steps = 1 call with the same y, s, n left in the same "
                                      "workspace (no other whole-path call in between)");
    ctx->ref_ws = nullptr;
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
        if (steps == 1) {               // the state a later steps = 2 call continues from
            ctx->ref_ws = ws;
            ctx->ref_y = y;
            ctx->ref_s = s_img;
            ctx->ref_n = n_img;
        }
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
// ---- iterated (DANSE-style) continuation -------------------------------------------------------------------------------

static void iterated_steps(disco_ctx* ctx, const PathArgs& a, int iters, Steps& st) {
    const disco_cfg& c = ctx->cfg;
    const WsLayout l = ws_layout(ctx);
    const float *y = a.y, *mask_z = a.mask_z, *mask_w = a.mask_w;
    float* out = a.out;
    disco_c32* X = (disco_c32*)(a.ws + l.X);
    disco_c32* z = a.z_y ? a.z_y : (disco_c32*)(a.ws + l.z);
    disco_c32* yo = a.yf ? a.yf : (disco_c32*)(a.ws + l.yf);
    disco_c32* w_loc = (disco_c32*)(a.ws + l.w);
    disco_c32* w_glo = (disco_c32*)(a.ws + l.w2);
    const int64_t G = (int64_t)c.rooms * c.nodes;
    const int M = c.mics, P2 = c.mics + c.nodes - 1;
    const bool same_mask = mask_w == mask_z;
    st.push_back({nullptr, false, [=](disco_stream s) { int ch = 1; return stft_cov_partials(ctx, y, mask_z, X, &ch, s); }});
    st.push_back({"solve1", true, [=](disco_stream s) { return disco_gevd_mwf_r1_pending(ctx, c.mu, w_loc, nullptr, s); }});
    for (int it = 0; it < iters; ++it) {
        // compression with the current w_loc (step 1's, then the local part of the previous iteration's filter) and the step-2
        // statistics: ONE pass over X for every node of a room where the shape allows it (k_room_cov), else the filter pass
        // followed by the covariance pass that reads X again and the K - 1 remote z's
        const bool last_pass = it + 1 == iters;          // only the last pass's z is read again (by the filter pass; it is what z_y returns)
        st.push_back({nullptr, false, [=](disco_stream s) {
            int ch = 1, rc;
            if (same_mask && room_cov_ok(ctx, X, mask_w))
                return STAGE(ctx, s, "room_cov2", room_cov_partials(ctx, X, mask_w, w_loc, z, &ch, s, last_pass));
            if ((rc = STAGE(ctx, s, "apply1", disco_apply(ctx, X, nullptr, w_loc, M, 1, z, s)))) return rc;
            return STAGE(ctx, s, "cov2", cov_partials(ctx, X, mask_w, c.nodes > 1 ? z : nullptr, c.nodes > 1 ? z : nullptr, 1, P2, &ch, s,
                                                      same_mask && c.nodes > 1));
        }});
        const bool last = it + 1 == iters;
        st.push_back({"solve2", true, [=](disco_stream s) {
            int rc = disco_gevd_mwf_r1_pending(ctx, c.mu, w_glo, nullptr, s);
            if (rc || last) return rc;
            // yf of this iteration is not needed; the next one re-compresses with the local part of this iteration's filter
            const long long nb = (long long)G * ctx->F;
            hipLaunchKernelGGL(k_filter_head, dim3((unsigned)std::min<long long>((nb * M + 255) / 256, 65535)), dim3(256), 0, (hipStream_t)s,
                               (const c32*)w_glo, (c32*)w_loc, nb, M, P2);
            return check_launch(ctx, "k_filter_head");
        }});
    }
    if (c.nodes > 1 && apply_istft_wide_ok(ctx)) {
        st.push_back({"apply2_istft", false, [=](disco_stream s) { return apply_istft_wide(ctx, X, z, w_glo, a.yf, out, s); }});
        return;
    }
    st.push_back({"apply2", false, [=](disco_stream s) { return disco_apply(ctx, X, c.nodes > 1 ? z : nullptr, w_glo, P2, 1, yo, s); }});
    st.push_back({"istft", false, [=](disco_stream s) { return disco_istft(ctx, yo, G, out, s); }});
}

extern "C" int disco_tango_enhance_iterated(disco_ctx* ctx, const float* y, const float* mask_z, const float* mask_w, int iters,
                                            float* out, disco_c32* z_y, disco_c32* yf, void* workspace, size_t workspace_bytes,
                                            disco_stream s) {
    DISCO_ENTER(ctx);
    if (!y || !mask_z || !mask_w || !out || iters < 1) return fail(ctx, DISCO_E_ARG, "disco_tango_enhance_iterated: bad argument");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_tango_enhance_iterated: node shard active");
    const WsLayout l = ws_layout(ctx);
    char* ws = nullptr;
    int rc = acquire_ws(ctx, workspace, workspace_bytes, l, &ws, "disco_tango_enhance_iterated");
    if (rc) return rc;
    const PathArgs a{y, mask_z, mask_w, out, z_y, yf, ws};
    // The overlapped form on the wide shapes: with the LDS group solver it lost (C5: 46.4 ms plain, 48.0 / 48.4 ms overlapped -- the room
    // pass takes a whole CU per workgroup and the solver's LDS blocks compete with it); with the register / DPP solver (k_solve_dpp.h:
    // 15 KB of LDS per wave, float64 VALU only) it pays: 38.40 -> 37.88 ms (profiles/r03_o_C5_overlap*.json).  Default like the fused route.
    if (overlap_applies(ctx) && ctx->half[0] && ctx->half[1]) {
        Steps st[2];
        for (int h = 0; h < 2; ++h) iterated_steps(ctx->half[h], child_args(ctx, a, h), iters, st[h]);
        return run_pipelined(ctx, st, s);
    }
    Steps st;
    iterated_steps(ctx, a, iters, st);
    return run_steps(ctx, st, s);
}
