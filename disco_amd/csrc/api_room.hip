// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): one-pass room covariance for the wide shapes
#include "room_launch.h"

using namespace disco;
using namespace disco_host;

// Wide shapes (P = M + K - 1 > 8), all nodes of a room on this GPU, mask_for_z = 'local', step-1 partial sums of THIS X with
// THIS mask still in `scratch`: z of every node AND the step-2 partial sums of every node from ONE pass over X (k_room.h),
// instead of disco_apply + cov_partials; room_cov_ok says whether the shape and the context's state qualify.
namespace disco_host {
bool room_cov_ok(const disco_ctx* ctx, const disco_c32* X, const float* mask) {
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, K = c.nodes;
    bool shape = false;
#define X_(M_, K_) if (M == M_ && K == K_) shape = true;
    DISCO_FOR_ROOM(X_)
#undef X_
    const bool want = ctx->opt[DISCO_OPT_ROOM_COV] != 0;
    if (!want || !shape || M + K - 1 <= 8 || sharded(ctx) || !X || !mask) return false;
    if (!(ctx->loc_M == M && ctx->loc_X == X && ctx->loc_mask == mask)) return false;       // the leading M x M block must be step 1's
    return (long long)K * ctx->T * ctx->F * M <= 0x0fffffffLL;                               // 32-bit BYTE offsets inside a room (8 B per element)
}

// The persistent pass forms ONE total per node -- its workgroups walk items (room, tile of 4 bins, all frames), the time axis is split INSIDE
// the workgroup (8 sub-chunks across the lanes), the sub-chunks meet in float64 -- and hands it over as TWO float32 blocks (hi, lo).
int room_chunks(const disco_ctx*) { return 2; }

int room_cov_partials(disco_ctx* ctx, const disco_c32* X, const float* mask, const disco_c32* w_loc, disco_c32* z,
                             int* chunks_out, disco_stream s, bool store_z) {
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, K = c.nodes, P = M + K - 1;
    if (!w_loc || !z || !room_cov_ok(ctx, X, mask)) return fail(ctx, DISCO_E_ARG, "room covariance: shape / state does not qualify");
    const int chunks = room_chunks(ctx);
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
    a.R = c.rooms;
    a.store_z = store_z ? 1 : 0;
    {
        constexpr int nb = 32 / 8;             // bins per workgroup (k_room.h RoomGeomS<M, K, 8>)
        a.tiles = (ctx->F + nb - 1) / nb;
        const long long items = (long long)c.rooms * a.tiles;
        if (items > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "room covariance: batch too large for one launch");
        // one persistent workgroup per CU; a whole number of 64-workgroup blocks when the batch allows it (the kernel then keeps neighbouring
        // tiles on one XCD); a workgroup whose first item lies beyond the batch simply returns
        unsigned nwg = (unsigned)std::min<long long>(items, ctx->n_cu);
        if (items >= 64) nwg = std::max(64u, nwg / 64 * 64);
        if (!launch_room_s8(M, K, nwg, (hipStream_t)s, a)) return fail(ctx, DISCO_E_UNSUPPORTED, "room covariance: shape not instantiated");
    }
    *chunks_out = chunks;
    ctx->pending_chunks = chunks;
    ctx->pending_P = P;
    ctx->pending_skiploc = 1;
    return check_launch(ctx, "k_room_cov");
}
}  // namespace disco_host

// self-test of the asm primitives of the room pass (csrc/k_room.h: LDS-DMA loads, permlane swaps): see include/disco_hip.h
extern "C" int disco_selftest_room(disco_ctx* ctx, const float* src, int64_t n, float* out_hw, float* out_ref, disco_stream s) {
    static_assert(ROOM_SELFTEST_OPS == DISCO_ROOM_SELFTEST_OPS, "header and kernel disagree");
    if (!src || !out_hw || !out_ref || n < 256 || n % 256) return ctx ? fail(ctx, DISCO_E_ARG, "disco_selftest_room: bad argument (n: a multiple of 256)") : DISCO_E_ARG;
    DevGuard dev_guard_(ctx ? ctx->cfg.device : [] { int d = 0; (void)hipGetDevice(&d); return d; }());
    hipLaunchKernelGGL(k_room_selftest, dim3((unsigned)(n / 256)), dim3(64), 0, (hipStream_t)s, src, (long long)n, out_hw, out_ref);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : DISCO_E_HIP_BASE - (int)e;
}
