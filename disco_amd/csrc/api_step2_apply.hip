// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): step 2 with the z exchange on chip: filter
#include "host.h"
#include "k_fused.h"

using namespace disco;
using namespace disco_host;

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
