// Masked batch spatial covariance (tango.py:357-364 local, 433-440 global).
//
// One thread per frequency bin walks the frames of its (room, node, frame-chunk): consecutive lanes read
// consecutive bins of the frame-major STFT, i.e. one contiguous 8*M-byte vector per lane and a contiguous
// 512*M-byte span per wave-load.  The P(P+1)/2 upper-triangle entries of Rss and Rnn live in registers for
// the whole walk; the Nyquist bin (F = 64 n + 1) gets its own wave with lanes striding over frames and a
// shuffle reduction.  Chunk partials are summed, scaled by 1/T and mirrored by k_cov_finalize.
#pragma once
#include "common.h"
#include "pk.h"

namespace disco {

template <int P>
__device__ __forceinline__ constexpr int tri_index(int i, int j) { return i * P - (i * (i - 1)) / 2 + (j - i); }

// One frame's contribution when both statistics weigh the SAME vector u (step 1, and step 2 with mask_for_z = 'local'):
//   Rss += a u u^H,  Rnn += b u u^H  with a = m^2, b = (1-m)^2.  The product u_i conj(u_j) is formed once and added into
// both accumulators (8 instructions per off-diagonal pair, no per-row scaling).
// JMIN > 0 skips the pairs with j < JMIN, i.e. the leading JMIN x JMIN block (step 2 re-using the local block that
// step 1 already accumulated with the same mask).
template <int P, int JMIN = 0>
__device__ __forceinline__ void cov_accumulate_shared(const c32* u, float a, float b, c32* acc_s, c32* acc_n) {
    // off the diagonal: p = u_i conj(u_j) in two packed instructions (pk.h), then one packed fma per statistic with the weight
    // broadcast from the pair (a, b) by the operand selectors: 4 instructions per entry instead of 8
    const c32 ab = make_float2(a, b);
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
        for (int j = (i > JMIN ? i : JMIN); j < P; ++j) {
            const int q = tri_index<P>(i, j);
            if (j != i) {
                const c32 p = cmul_aconjb(u[i], u[j]);
                acc_s[q] = fma_by_half<0>(p, ab, acc_s[q]);
                acc_n[q] = fma_by_half<1>(p, ab, acc_n[q]);
            } else {
                const float pr = fmaf(u[i].x, u[i].x, u[i].y * u[i].y);
                acc_s[q].x = fmaf(a, pr, acc_s[q].x);
                acc_n[q].x = fmaf(b, pr, acc_n[q].x);
            }
        }
    }
}

struct CovArgs {
    const c32* X;        // [R][K][T][F][M]
    const float* mask;   // [R][K][T][F]
    const c32* Zs;       // [R][K][T][F] or null
    const c32* Zn;       // [R][K][T][F] or null
    float4* part;        // [R*Kl][chunks][F][NP] of (Rss.re, Rss.im, Rnn.re, Rnn.im) sums
    int K, T, F, chunks, mask_remote;
    // node shard (disco_set_node_shard): this context holds nodes [k0, k0 + Kl) of every room.  X, mask and part are
    // indexed by the LOCAL unit g = r*Kl + kl; only the remote rows Zs/Zn are indexed by global node (they come from the
    // all-gather of z).  Kl == K, k0 == 0 when all nodes of a room live here.
    int Kl, k0;
    // layout of Zs / Zn: planes [K / zblk][R][zblk] (zblk = K: the plain [R][K]; see z_plane in common.h)
    int zblk;
    long long R;
};

template <int M, int KR, bool SAMEZ>
__device__ __forceinline__ void cov_walk(const CovArgs& a, long long g, int f, int t_begin, int t_end, int t_step,
                                         c32* acc_s, c32* acc_n) {
    constexpr int P = M + KR;
    const int K = a.K, T = a.T, F = a.F;
    const long long r = g / a.Kl;
    const int k = a.k0 + (int)(g % a.Kl);
    const c32* Xg = a.X + (g * T * (long long)F) * M;
    const float* mg = a.mask + g * T * (long long)F;
    for (int t = t_begin; t < t_end; t += t_step) {
        const long long tf = (long long)t * F + f;
        const float m = mg[tf];
        const float mc = 1.f - m;
        c32 vs[P], vn[P];
        const c32* xp = Xg + tf * M;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const c32 x = xp[i];
            vs[i] = make_float2(m * x.x, m * x.y);
            vn[i] = make_float2(mc * x.x, mc * x.y);
        }
        if constexpr (KR > 0) {
            const float gs = a.mask_remote ? m : 1.f;
            const float gn = a.mask_remote ? mc : 1.f;
#pragma unroll
            for (int jj = 0; jj < KR; ++jj) {
                const int j = jj < k ? jj : jj + 1;               // concatenate_signals order: z_j (j<k), z_j (j>k)
                const long long zo = (z_plane(r, j, K, a.R, a.zblk) * T) * (long long)F + tf;
                const c32 zs = a.Zs[zo];
                const c32 zn = SAMEZ ? zs : a.Zn[zo];
                vs[M + jj] = make_float2(gs * zs.x, gs * zs.y);
                vn[M + jj] = make_float2(gn * zn.x, gn * zn.y);
            }
        }
#pragma unroll
        for (int i = 0; i < P; ++i) {
#pragma unroll
            for (int j = i; j < P; ++j) {
                const int q = tri_index<P>(i, j);
                // v_i conj(v_j)
                acc_s[q].x = fmaf(vs[i].x, vs[j].x, fmaf(vs[i].y, vs[j].y, acc_s[q].x));
                acc_n[q].x = fmaf(vn[i].x, vn[j].x, fmaf(vn[i].y, vn[j].y, acc_n[q].x));
                if (j != i) {
                    acc_s[q].y = fmaf(vs[i].y, vs[j].x, fmaf(-vs[i].x, vs[j].y, acc_s[q].y));
                    acc_n[q].y = fmaf(vn[i].y, vn[j].x, fmaf(-vn[i].x, vn[j].y, acc_n[q].y));
                }
            }
        }
    }
}

// grid = R*K*chunks blocks of NT = (F - 1) + 64 threads: threads [0, F-1) own bins, the last wave owns bin F-1.
// __launch_bounds__(NT) matters: without it hipcc budgets registers for 1024-thread blocks and spills the
// P(P+1) accumulators to scratch (measured: 64 VGPR + 280 B scratch, 32 ms instead of ~8 ms at C3).
template <int M, int KR, bool SAMEZ, int NT>
__global__ __launch_bounds__(NT) void k_cov(CovArgs a) {
    constexpr int P = M + KR, NP = P * (P + 1) / 2;
    const long long g = blockIdx.x / a.chunks;
    const int c = (int)(blockIdx.x % a.chunks);
    const int t0 = (int)(((long long)a.T * c) / a.chunks), t1 = (int)(((long long)a.T * (c + 1)) / a.chunks);
    const int nbin = a.F - 1;
    const bool nyq = (int)threadIdx.x >= nbin;
    const int lane = threadIdx.x & 63;
    c32 acc_s[NP], acc_n[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) acc_s[q] = acc_n[q] = make_float2(0.f, 0.f);
    if (!nyq) {
        cov_walk<M, KR, SAMEZ>(a, g, threadIdx.x, t0, t1, 1, acc_s, acc_n);
    } else {
        cov_walk<M, KR, SAMEZ>(a, g, nbin, t0 + lane, t1, 64, acc_s, acc_n);
    }
    if (nyq) {      // whole wave: reduce the 64 per-lane partial sums of the Nyquist bin
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                acc_s[q].x += __shfl_xor(acc_s[q].x, off);
                acc_s[q].y += __shfl_xor(acc_s[q].y, off);
                acc_n[q].x += __shfl_xor(acc_n[q].x, off);
                acc_n[q].y += __shfl_xor(acc_n[q].y, off);
            }
        }
    }
    if (!nyq || lane == 0) {
        const int f = nyq ? nbin : (int)threadIdx.x;
        float4* o = a.part + (((g * a.chunks + c) * a.F) + f) * (long long)NP;
#pragma unroll
        for (int q = 0; q < NP; ++q) o[q] = make_float4(acc_s[q].x, acc_s[q].y, acc_n[q].x, acc_n[q].y);
    }
}

// ---- 9 <= P <= 16 (e.g. 8 nodes x 8 mics: P = 15) ------------------------------------------------------------
// The 2 * P(P+1)/2 accumulators no longer fit one thread, so a workgroup of CB_S waves shares a 64-bin tile: every
// wave loads the same P-vector of its bin (the re-loads hit L1) and owns the pairs q with q % CB_S == wave.
// M, KR are run-time values; all register arrays are indexed statically (loops unrolled to the maximum, guarded).
constexpr int CB_S = 4;
constexpr int CB_PMAX = 16;

template <int SI, bool SAMEZ>
__device__ __forceinline__ void cov_big_walk(const CovArgs& a, int M, int KR, long long g, int f, bool live, int t0, int t1,
                                             int t_step, int t_off, c32* acc_s, c32* acc_n) {
    const int K = a.K, T = a.T, F = a.F, P = M + KR;
    const long long r = g / a.Kl;
    const int k = a.k0 + (int)(g % a.Kl);
    const c32* Xg = a.X + (g * T * (long long)F) * M;
    const float* mg = a.mask + g * T * (long long)F;
    // The kernel runs 2 waves per SIMD (2 x 34 complex accumulators per lane): the next frame's rows are requested one
    // iteration ahead, raw and unconditionally (clamped frame index), so that a wave always has a frame in flight.
    // (SAMEZ only: the second set of remote rows would not fit the register budget.)
    c32 xr[CB_PMAX], xq[CB_PMAX];
    float mr = 0.f;
    auto fetch = [&](int tu_, c32* xs_, c32* xn_, float& m_) {
        const int t_ = min(tu_ + t_off, t1 - 1);
        const long long tf_ = (long long)(t_ < t0 ? t0 : t_) * F + f;
        m_ = mg[tf_];
#pragma unroll
        for (int i = 0; i < CB_PMAX; ++i) {
            if (i < M) {
                xs_[i] = Xg[tf_ * M + i];
                if (!SAMEZ) xn_[i] = xs_[i];
            } else if (i < P) {
                const int jj = i - M;
                const int j = jj < k ? jj : jj + 1;
                const long long zo = (z_plane(r, j, K, a.R, a.zblk) * T) * (long long)F + tf_;
                xs_[i] = a.Zs[zo];
                if (!SAMEZ) xn_[i] = a.Zn[zo];
            } else {
                xs_[i] = make_float2(0.f, 0.f);
                if (!SAMEZ) xn_[i] = make_float2(0.f, 0.f);
            }
        }
    };
    if (SAMEZ) fetch(t0, xr, xq, mr);
    for (int tu = t0; tu < t1; tu += t_step) {
        const int t = tu + t_off;
        const bool ok = live && t < t1;
        c32 xc[CB_PMAX], xd[CB_PMAX];
        float mraw;
        if (SAMEZ) {
#pragma unroll
            for (int i = 0; i < CB_PMAX; ++i) xc[i] = xr[i];
            mraw = mr;
            fetch(tu + t_step, xr, xq, mr);                 // next frame (harmless clamp at the end of the chunk)
        } else {
            fetch(tu, xc, xd, mraw);
        }
        const float m = ok ? mraw : 0.f, mc = ok ? 1.f - mraw : 0.f;
        const float gs = a.mask_remote ? m : (ok ? 1.f : 0.f), gn = a.mask_remote ? mc : (ok ? 1.f : 0.f);
        c32 vs[CB_PMAX], vn[CB_PMAX];
#pragma unroll
        for (int i = 0; i < CB_PMAX; ++i) {
            const c32 xs = xc[i], xn = SAMEZ ? xc[i] : xd[i];
            const float ws = i < M ? m : (i < P ? gs : 0.f), wn = i < M ? mc : (i < P ? gn : 0.f);
            vs[i] = make_float2(ws * xs.x, ws * xs.y);
            vn[i] = make_float2(wn * xn.x, wn * xn.y);
        }
        int q = 0, slot = 0;
#pragma unroll
        for (int i = 0; i < CB_PMAX; ++i) {
#pragma unroll
            for (int j = i; j < CB_PMAX; ++j, ++q) {
                if (q % CB_S == SI) {                       // compile-time: this wave's share of the 136 (i, j) pairs
                    if (j < P) {                            // run-time, wave-uniform
                        acc_s[slot].x = fmaf(vs[i].x, vs[j].x, fmaf(vs[i].y, vs[j].y, acc_s[slot].x));
                        acc_n[slot].x = fmaf(vn[i].x, vn[j].x, fmaf(vn[i].y, vn[j].y, acc_n[slot].x));
                        if (j != i) {
                            acc_s[slot].y = fmaf(vs[i].y, vs[j].x, fmaf(-vs[i].x, vs[j].y, acc_s[slot].y));
                            acc_n[slot].y = fmaf(vn[i].y, vn[j].x, fmaf(-vn[i].x, vn[j].y, acc_n[slot].y));
                        }
                    }
                    ++slot;
                }
            }
        }
    }
}

template <int SI, bool SAMEZ>
__device__ __forceinline__ void cov_big_wave(const CovArgs& a, int M, int KR, long long g, int c, int tile, int lane) {
    constexpr int NSLOT = (CB_PMAX * (CB_PMAX + 1) / 2 + CB_S - 1) / CB_S;
    const int P = M + KR, NP = P * (P + 1) / 2;
    const int nbin = a.F - 1, tiles = (nbin + 63) / 64;
    const int t0 = (int)(((long long)a.T * c) / a.chunks), t1 = (int)(((long long)a.T * (c + 1)) / a.chunks);
    c32 acc_s[NSLOT], acc_n[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) acc_s[s] = acc_n[s] = make_float2(0.f, 0.f);
    const bool nyq = tile == tiles;
    int f = nyq ? nbin : tile * 64 + lane;
    const bool live = nyq || f < nbin;
    if (f > nbin) f = nbin;
    cov_big_walk<SI, SAMEZ>(a, M, KR, g, f, live, t0, t1, nyq ? 64 : 1, nyq ? lane : 0, acc_s, acc_n);
    if (nyq) {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                acc_s[s].x += __shfl_xor(acc_s[s].x, off);
                acc_s[s].y += __shfl_xor(acc_s[s].y, off);
                acc_n[s].x += __shfl_xor(acc_n[s].x, off);
                acc_n[s].y += __shfl_xor(acc_n[s].y, off);
            }
    }
    if (live && (!nyq || lane == 0)) {
        // scatter this wave's slots to their packed upper-triangle positions of the RUN-TIME P
        float4* o = a.part + (((g * a.chunks + c) * a.F) + f) * (long long)NP;
        int q = 0, slot = 0;
#pragma unroll
        for (int i = 0; i < CB_PMAX; ++i) {
#pragma unroll
            for (int j = i; j < CB_PMAX; ++j, ++q) {
                if (q % CB_S == SI) {
                    if (j < P) o[i * P - (i * (i - 1)) / 2 + (j - i)] = make_float4(acc_s[slot].x, acc_s[slot].y, acc_n[slot].x, acc_n[slot].y);
                    ++slot;
                }
            }
        }
    }
}

// grid = R*K * (tiles + 1) * chunks blocks of 64 * CB_S threads
template <bool SAMEZ>
__global__ __launch_bounds__(64 * CB_S) void k_cov_big(CovArgs a, int M, int KR) {
    const int nbin = a.F - 1, tiles = (nbin + 63) / 64;
    int bid = blockIdx.x;
    const int c = bid % a.chunks;
    bid /= a.chunks;
    const int tile = bid % (tiles + 1);
    const long long g = bid / (tiles + 1);
    const int lane = threadIdx.x & 63;
    switch (wave_id()) {
        case 0: cov_big_wave<0, SAMEZ>(a, M, KR, g, c, tile, lane); break;
        case 1: cov_big_wave<1, SAMEZ>(a, M, KR, g, c, tile, lane); break;
        case 2: cov_big_wave<2, SAMEZ>(a, M, KR, g, c, tile, lane); break;
        default: cov_big_wave<3, SAMEZ>(a, M, KR, g, c, tile, lane); break;
    }
}

// ---- 9 <= P <= 16, both statistics on the SAME vector (step 2 with mask_for_z = 'local', tango.py:416-418) -------------
// k_cov_big above gives every wave ALL P components of a bin and a quarter of the pairs: four times the loads, 136
// accumulators and 30 live inputs per lane, two waves per SIMD.  Here the P x P triangle is cut along the halves of the
// two component groups -- A = the node's own M channels, B = the K-1 remote z's -- and every wave of a workgroup owns one
// BLOCK of pairs, so it loads only the (at most two) half-groups its block touches:
//     wave 0..3 : A0 x B0, A0 x B1, A1 x B0, A1 x B1          wave 4 : tri(B0) + tri(B1)       wave 5 : B0 x B1
//     wave 6, 7 : tri(A0) + tri(A1), A0 x A1   -- only when the leading M x M block is wanted: with SKIPLOC it is the
//                 step-1 covariance (same mask), already in the context as partial sums (cf. k_step2_cov_fused)
// 8 x 7 (config C5): at most 16 pairs = 64 accumulators and 8 inputs per lane.  u_i conj(u_j) is formed once per pair and
// added into both statistics with the weights m^2 and (1-m)^2 (cov_accumulate_shared).  Lane = bin of a 64-bin tile; the
// Nyquist bin gets one more workgroup per (node, chunk) whose lanes stride over frames (as everywhere else).
template <int M, int KR, int X0, int X1, int Y0, int Y1, bool TRI>
struct CovSplitRole {
    // components [X0, X1) and [Y0, Y1) of v = [x_0..x_{M-1}, z_0..z_{KR-1}]; TRI: the two upper triangles, else the X x Y block
    static constexpr int NX = X1 - X0, NY = Y1 - Y0;
    static constexpr int NPAIR = TRI ? NX * (NX + 1) / 2 + NY * (NY + 1) / 2 : NX * NY;
    static constexpr int x0 = X0, x1 = X1, y0 = Y0, y1 = Y1;
    static constexpr bool tri = TRI;
};

template <int M, int KR, int C0, int C1>
__device__ __forceinline__ void cov_split_fetch(c32* u, const c32* __restrict__ xp, const c32* const* zp, int tf) {
    // components [C0, C1) of one (frame, bin); tf = t * F + f as a 32-bit offset into the node's [T][F] plane, so that the
    // wave-uniform plane pointers xp / zp stay in scalar registers (one VGPR of offset instead of a 64-bit pointer per row).
    // The mics among the components come as 16-byte pairs where the layout allows it (M even, even first index, even count).
    constexpr int XA = C0 < M ? C0 : M, XB = C1 < M ? C1 : M, ZA = C0 > M ? C0 : M;
    if constexpr (XB > XA) {
        if constexpr (M % 2 == 0 && XA % 2 == 0 && (XB - XA) % 2 == 0) {
            const float4* src = reinterpret_cast<const float4*>(xp + (long long)tf * M + XA);
#pragma unroll
            for (int p = 0; p < (XB - XA) / 2; ++p) {
                const float4 q = src[p];
                u[XA - C0 + 2 * p] = make_float2(q.x, q.y);
                u[XA - C0 + 2 * p + 1] = make_float2(q.z, q.w);
            }
        } else {
#pragma unroll
            for (int c = XA; c < XB; ++c) u[c - C0] = xp[(long long)tf * M + c];
        }
    }
#pragma unroll
    for (int c = ZA; c < C1; ++c) u[c - C0] = zp[c - M][tf];
}

// acc += (wa, wb) x { u_i conj(u_j) } over the role's pairs (TRI: the two upper triangles of X and Y, else the X x Y block)
#ifndef DISCO_COV_PK
#define DISCO_COV_PK 2
#endif
#if DISCO_COV_PK == 2
// on the instruction forms of pk.h: u_i conj(u_j) in two packed instructions, one packed fma per statistic (weights broadcast
// from the pair (wa, wb) by the operand selectors) -- as cov_accumulate_shared
__device__ __forceinline__ void cov_pair_acc(const c32 a, const c32 b, const float wa, const float wb, c32& as, c32& an) {
    const c32 p = cmul_aconjb(a, b), w2 = make_float2(wa, wb);
    as = fma_by_half<0>(p, w2, as);
    an = fma_by_half<1>(p, w2, an);
}
#elif DISCO_COV_PK && DISCO_PK && defined(__clang__)
// packed form: (pr, pi) = a conj(b) as v_pk_mul + v_pk_fma, then one v_pk_fma per part on the pair (Rss, Rnn) of sums -- 4
// instructions per pair of components instead of 8; the sums live as (acc_s.x, acc_n.x), (acc_s.y, acc_n.y)
__device__ __forceinline__ void cov_pair_acc(const c32 a, const c32 b, const float wa, const float wb, c32& as, c32& an) {
    const v2f w2 = {wa, wb};
    const v2f p = __builtin_elementwise_fma(v2f{a.x, a.x}, v2f{b.x, -b.y}, v2f{a.y, a.y} * v2f{b.y, b.x});
    const v2f re = __builtin_elementwise_fma(w2, v2f{p.x, p.x}, v2f{as.x, an.x});
    const v2f im = __builtin_elementwise_fma(w2, v2f{p.y, p.y}, v2f{as.y, an.y});
    as = make_float2(re.x, im.x);
    an = make_float2(re.y, im.y);
}
#else
__device__ __forceinline__ void cov_pair_acc(const c32 a, const c32 b, const float wa, const float wb, c32& as, c32& an) {
    const float pr = fmaf(a.x, b.x, a.y * b.y);          // a conj(b)
    const float pi = fmaf(a.x, -b.y, a.y * b.x);
    as.x = fmaf(wa, pr, as.x);
    an.x = fmaf(wb, pr, an.x);
    as.y = fmaf(wa, pi, as.y);
    an.y = fmaf(wb, pi, an.y);
}
#endif
__device__ __forceinline__ void cov_diag_acc(const c32 a, const float wa, const float wb, c32& as, c32& an) {
    const float pr = fmaf(a.x, a.x, a.y * a.y);
    as.x = fmaf(wa, pr, as.x);
    an.x = fmaf(wb, pr, an.x);
}

template <class Role, bool TRI, int NX, int NY>
__device__ __forceinline__ void cov_split_accumulate(const c32* ux, const c32* uy, const float wa, const float wb, c32* acc_s,
                                                     c32* acc_n) {
    int q = 0;
    if constexpr (TRI) {
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = i; j < NX; ++j, ++q) {
                if (j == i) cov_diag_acc(ux[i], wa, wb, acc_s[q], acc_n[q]);
                else cov_pair_acc(ux[i], ux[j], wa, wb, acc_s[q], acc_n[q]);
            }
#pragma unroll
        for (int i = 0; i < NY; ++i)
#pragma unroll
            for (int j = i; j < NY; ++j, ++q) {
                if (j == i) cov_diag_acc(uy[i], wa, wb, acc_s[q], acc_n[q]);
                else cov_pair_acc(uy[i], uy[j], wa, wb, acc_s[q], acc_n[q]);
            }
    } else {
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j < NY; ++j, ++q) cov_pair_acc(ux[i], uy[j], wa, wb, acc_s[q], acc_n[q]);      // i in X, j in Y
    }
}

// the role's sums -> their places in the upper triangle of the (chunk, bin) partial block o[NP]
template <int P, int X0, int Y0, bool TRI, int NX, int NY>
__device__ __forceinline__ void cov_split_store(float4* o, const c32* acc_s, const c32* acc_n) {
    int q = 0;
    if constexpr (TRI) {
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = i; j < NX; ++j, ++q)
                o[tri_index<P>(X0 + i, X0 + j)] = make_float4(acc_s[q].x, acc_s[q].y, acc_n[q].x, acc_n[q].y);
#pragma unroll
        for (int i = 0; i < NY; ++i)
#pragma unroll
            for (int j = i; j < NY; ++j, ++q)
                o[tri_index<P>(Y0 + i, Y0 + j)] = make_float4(acc_s[q].x, acc_s[q].y, acc_n[q].x, acc_n[q].y);
    } else {
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j < NY; ++j, ++q)                 // X precedes Y in v: (X0 + i, Y0 + j) is in the upper triangle
            o[tri_index<P>(X0 + i, Y0 + j)] = make_float4(acc_s[q].x, acc_s[q].y, acc_n[q].x, acc_n[q].y);
    }
}

// S > 1: the tile is 64 / S bins wide and a wave's lanes are (sub-chunk sc, bin): lane (sc, b) folds frames t0 + sc, t0 + sc + S, ... of
// the chunk -- S times shorter float32 sums per accumulator at the same register count (round 4: the step-1 statistics of the wide
// shapes summed 157 frames each, and the 8 x 8 block they produce is the leading block of every step-2 pencil; cf. k_room.h) -- and
// the S partial sums of an entry meet through the lane crossbar at the end, where the Nyquist tile's 64 always did.
template <int M, int KR, int X0, int X1, int Y0, int Y1, bool TRI, int S = 1>
__device__ __forceinline__ void cov_split_wave(const CovArgs& a, long long g, int c, int tile, int lane) {
    using Role = CovSplitRole<M, KR, X0, X1, Y0, Y1, TRI>;
    constexpr int P = M + KR, NP = P * (P + 1) / 2, NX = Role::NX, NY = Role::NY, NPAIR = Role::NPAIR;
    constexpr int NBT = 64 / S;                                                  // bins per tile
    static_assert(S == 1 || S == 2 || S == 4 || S == 8, "sub-chunks share a wave");
    if constexpr (NPAIR == 0) {
        return;
    } else {
        const int K = a.K, T = a.T, F = a.F;
        const int nbin = F - 1, tiles = (nbin + NBT - 1) / NBT;
        const int t0 = (int)(((long long)T * c) / a.chunks), t1 = (int)(((long long)T * (c + 1)) / a.chunks);
        const bool nyq = tile == tiles;
        int f = nyq ? nbin : tile * NBT + (lane & (NBT - 1));
        const bool live = nyq || f < nbin;
        if (f > nbin) f = nbin;
        const int t_step = nyq ? 64 : S, t_off = nyq ? lane : lane / NBT;
        const long long r = g / a.Kl;
        const int k = a.k0 + (int)(g % a.Kl);
        const c32* xp = a.X + (g * T * (long long)F) * M;                     // wave-uniform plane pointers (scalar registers)
        const float* mp = a.mask + g * T * (long long)F;
        const c32* zp[KR > 0 ? KR : 1];
#pragma unroll
        for (int jj = 0; jj < KR; ++jj) {
            const int j = jj < k ? jj : jj + 1;                                  // concatenate_signals order
            zp[jj] = a.Zs + (z_plane(r, j, K, a.R, a.zblk) * T) * (long long)F;
        }
        c32 acc_s[NPAIR], acc_n[NPAIR];
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) acc_s[q] = acc_n[q] = make_float2(0.f, 0.f);
        // one frame ahead: raw, unconditional loads (frame index clamped into the chunk), weighed 0 when out of range
        c32 ux[NX > 0 ? NX : 1], uy[NY > 0 ? NY : 1], nx[NX > 0 ? NX : 1], ny[NY > 0 ? NY : 1];
        float mcur, mnext;
        auto fetch = [&](int tu, c32* px, c32* py, float& m_) {
            int t_ = tu + t_off;
            t_ = t_ < t1 ? t_ : t1 - 1;
            const int tfm = t_ * F + f;                              // (frame, bin) offset inside the node's plane
            m_ = mp[tfm];
            cov_split_fetch<M, KR, X0, X1>(px, xp, zp, tfm);
            cov_split_fetch<M, KR, Y0, Y1>(py, xp, zp, tfm);
        };
        fetch(t0, ux, uy, mcur);
        for (int tu = t0; tu < t1; tu += t_step) {
            fetch(tu + t_step, nx, ny, mnext);                       // harmless clamp at the end of the chunk
            const bool ok = live && (tu + t_off) < t1;
            const float m = ok ? mcur : 0.f, mc = ok ? 1.f - mcur : 0.f;
            const float wa = m * m, wb = mc * mc;
            cov_split_accumulate<Role, TRI, NX, NY>(ux, uy, wa, wb, acc_s, acc_n);
#pragma unroll
            for (int i = 0; i < NX; ++i) ux[i] = nx[i];
#pragma unroll
            for (int i = 0; i < NY; ++i) uy[i] = ny[i];
            mcur = mnext;
        }
        if (nyq || S > 1) {          // lanes hold partial sums over disjoint frames of the same bin: all 64 (Nyquist tile) or the S sub-chunks
            const int stop = nyq ? 1 : NBT;
#pragma unroll
            for (int q = 0; q < NPAIR; ++q)
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    if (off < stop) break;
                    acc_s[q].x += __shfl_xor(acc_s[q].x, off);
                    acc_s[q].y += __shfl_xor(acc_s[q].y, off);
                    acc_n[q].x += __shfl_xor(acc_n[q].x, off);
                    acc_n[q].y += __shfl_xor(acc_n[q].y, off);
                }
        }
        if (live && (nyq ? lane == 0 : lane < NBT)) {
            float4* o = a.part + (((g * a.chunks + c) * F) + f) * (long long)NP;
            cov_split_store<P, X0, Y0, TRI, NX, NY>(o, acc_s, acc_n);
        }
    }
}

// ---- step-1 statistics of the wide shapes in FLOAT64 (round 4) -----------------------------------------------------------------------
// The M x M statistics of step 1 feed three solves of a two-iteration run -- the local filters, and the leading block of both step-2
// pencils -- and on C5 their float32 summation was what the output's distance from the float64 oracle followed: with the lanes-are-bins
// kernel above room 199 came out at 1.2e-4, with 4 / 8 time sub-chunks per wave at 9e-5 / 3-5e-5, whatever the step-2 pass did
// (profiles/r04_a_*, r04_b_*).  The pass is HBM-bound with arithmetic to spare (36 entries per bin and frame), so here the sums are simply
// formed in float64: the products u_i conj(u_j) stay float32 (their rounding errors are independent and average out over the frames), the
// weights m^2, (1 - m)^2 and the accumulation are float64 -- 2 conversions + 4 v_fma_f64 per entry where the float32 form has 2 packed
// fmas.  The M(M+1)/2 entries are dealt to FOUR waves (tri(A0), tri(A1), A0 x B0, A0 x B1 with A0 = [0, MA), A1 = [MA, M), B0 | B1 = A1:
// 32 float64 accumulators per lane at M = 8, four waves per SIMD stay resident), the total leaves as a (hi, lo) PAIR of float32 partial
// blocks (2 c, 2 c + 1 of 2 * chunks) which every solver adds in float64 -- so nothing is rounded to float32 on the way to the solve.
template <int M, int X0, int X1, int Y0, int Y1, bool TRI>
__device__ __forceinline__ void cov_loc_f64_wave(const CovArgs& a, long long g, int c, int tile, int lane) {
    using Role = CovSplitRole<M, 0, X0, X1, Y0, Y1, TRI>;
    constexpr int P = M, NP = P * (P + 1) / 2, NX = Role::NX, NY = Role::NY, NPAIR = Role::NPAIR;
    if constexpr (NPAIR == 0) {
        return;
    } else {
        const int T = a.T, F = a.F;
        const int nbin = F - 1, tiles = (nbin + 63) / 64;
        const int t0 = (int)(((long long)T * c) / a.chunks), t1 = (int)(((long long)T * (c + 1)) / a.chunks);
        const bool nyq = tile == tiles;                                      // the Nyquist bin: lanes are frames
        int f = nyq ? nbin : tile * 64 + lane;
        const bool live = nyq || f < nbin;
        if (f > nbin) f = nbin;
        const int t_step = nyq ? 64 : 1, t_off = nyq ? lane : 0;
        const c32* xp = a.X + (g * T * (long long)F) * M;                    // wave-uniform plane pointers (scalar registers)
        const float* mp = a.mask + g * T * (long long)F;
        const c32* const zp[1] = {nullptr};
        double sr[NPAIR], si[NPAIR], nr[NPAIR], ni[NPAIR];                   // Rss / Rnn entry q, real and imaginary part (imaginary parts of diagonal entries stay unused)
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) sr[q] = si[q] = nr[q] = ni[q] = 0.0;
        c32 ux[NX > 0 ? NX : 1], uy[NY > 0 ? NY : 1], nx[NX > 0 ? NX : 1], ny[NY > 0 ? NY : 1];
        float mcur, mnext;
        auto fetch = [&](int tu, c32* px, c32* py, float& m_) {
            int t_ = tu + t_off;
            t_ = t_ < t1 ? t_ : t1 - 1;
            const int tfm = t_ * F + f;                                      // (frame, bin) offset inside the node's plane
            m_ = mp[tfm];
            cov_split_fetch<M, 0, X0, X1>(px, xp, zp, tfm);
            cov_split_fetch<M, 0, Y0, Y1>(py, xp, zp, tfm);
        };
        auto pair = [&](const c32 u, const c32 v, const double wa, const double wb, int q) {
            const c32 p = cmul_aconjb(u, v);
            const double pr = (double)p.x, pi = (double)p.y;
            sr[q] = fma(wa, pr, sr[q]);
            nr[q] = fma(wb, pr, nr[q]);
            si[q] = fma(wa, pi, si[q]);
            ni[q] = fma(wb, pi, ni[q]);
        };
        auto diag = [&](const c32 u, const double wa, const double wb, int q) {
            const double pr = (double)fmaf(u.x, u.x, u.y * u.y);
            sr[q] = fma(wa, pr, sr[q]);
            nr[q] = fma(wb, pr, nr[q]);
        };
        fetch(t0, ux, uy, mcur);
        for (int tu = t0; tu < t1; tu += t_step) {
            fetch(tu + t_step, nx, ny, mnext);                               // harmless clamp at the end of the chunk
            const bool ok = live && (tu + t_off) < t1;
            const double m = ok ? (double)mcur : 0.0, mc = ok ? 1.0 - (double)mcur : 0.0;
            const double wa = m * m, wb = mc * mc;
            int q = 0;                                                       // (the order of cov_split_accumulate / cov_split_store)
            if constexpr (TRI) {
#pragma unroll
                for (int i = 0; i < NX; ++i)
#pragma unroll
                    for (int j = i; j < NX; ++j, ++q) {
                        if (j == i) diag(ux[i], wa, wb, q);
                        else pair(ux[i], ux[j], wa, wb, q);
                    }
#pragma unroll
                for (int i = 0; i < NY; ++i)
#pragma unroll
                    for (int j = i; j < NY; ++j, ++q) {
                        if (j == i) diag(uy[i], wa, wb, q);
                        else pair(uy[i], uy[j], wa, wb, q);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < NX; ++i)
#pragma unroll
                    for (int j = 0; j < NY; ++j, ++q) pair(ux[i], uy[j], wa, wb, q);
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) ux[i] = nx[i];
#pragma unroll
            for (int i = 0; i < NY; ++i) uy[i] = ny[i];
            mcur = mnext;
        }
        if (nyq) {          // lanes hold partial sums over disjoint frames of the same bin
#pragma unroll
            for (int q = 0; q < NPAIR; ++q)
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    sr[q] += __shfl_xor(sr[q], off);
                    si[q] += __shfl_xor(si[q], off);
                    nr[q] += __shfl_xor(nr[q], off);
                    ni[q] += __shfl_xor(ni[q], off);
                }
        }
        if (live && (!nyq || lane == 0)) {
            c32 hs[NPAIR], hn[NPAIR], ls[NPAIR], ln[NPAIR];
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) {
                hs[q] = make_float2((float)sr[q], (float)si[q]);
                hn[q] = make_float2((float)nr[q], (float)ni[q]);
                ls[q] = make_float2((float)(sr[q] - (double)hs[q].x), (float)(si[q] - (double)hs[q].y));
                ln[q] = make_float2((float)(nr[q] - (double)hn[q].x), (float)(ni[q] - (double)hn[q].y));
            }
            float4* o = a.part + (((g * (2 * a.chunks) + 2 * c) * F) + f) * (long long)NP;
            cov_split_store<P, X0, Y0, TRI, NX, NY>(o, hs, hn);
            cov_split_store<P, X0, Y0, TRI, NX, NY>(o + (long long)F * NP, ls, ln);
        }
    }
}

// The same wave roles with the tile's spectra FETCHED ONCE PER WORKGROUP (M = 8; round 4, late): every lane loading its own bin's mics is a
// 16-byte access every 64 bytes, 32 lines per instruction, and the four roles fetch 2 + 2 + 3 + 3 granules of the same 4 -- 10 KB of L1
// requests per 4 KB frame-tile (k_apply_mq's finding: 5.5 -> 4.6 ms for contiguous loads).  Here thread i of the workgroup loads granule
// i of the frame-tile (256 granules = 64 bins x 4: 1 KiB contiguous per wave), three frames ahead in registers (4 registers per frame),
// parks it in one of two LDS tiles -- written in load order, read bin-major, XOR-swizzled like k_apply_mq's so that both directions are
// conflict-free -- and after ONE barrier per frame every role reads its mics from there.  The arithmetic is cov_loc_f64_wave's.
#ifndef DISCO_COV64_AHEAD
#define DISCO_COV64_AHEAD 3
#endif
template <int M>
struct alignas(16) CovLocTile {
    float4 x[2][64 * (M / 2)];
    float m[2][64];
};
template <int M, int X0, int X1, int Y0, int Y1, bool TRI>
__device__ __forceinline__ void cov_loc_f64_wave_tile(const CovArgs& a, long long g, int c, int tile, int lane, CovLocTile<M>& sh) {
    using Role = CovSplitRole<M, 0, X0, X1, Y0, Y1, TRI>;
    constexpr int P = M, NP = P * (P + 1) / 2, NX = Role::NX, NY = Role::NY, NPAIR = Role::NPAIR;
    constexpr int MH = M / 2, BPR = 16 / MH, D = DISCO_COV64_AHEAD;
    static_assert(M == 8 && NPAIR > 0 && X0 % 2 == 0 && X1 % 2 == 0 && Y0 % 2 == 0 && Y1 % 2 == 0, "whole granules per role; 256 threads = 256 granules");
    const int T = a.T, F = a.F, nbin = F - 1;
    const int t0 = (int)(((long long)T * c) / a.chunks), t1 = (int)(((long long)T * (c + 1)) / a.chunks);
    const int f = tile * 64 + lane;
    const bool live = f < nbin;
    // loader: granule gi = threadIdx.x of the frame-tile = (bin lb, granule lp); bins past the last one re-read it (their weights are zero)
    const int gi = threadIdx.x, lb = gi / MH, lp = gi % MH;
    const int lgo = min(tile * 64 + lb, nbin) * MH + lp;
    const int lpos = lb * MH + (lp ^ ((lb / BPR) % MH));
    const int swz = (lane / BPR) % MH;
    const float4* xq = reinterpret_cast<const float4*>(a.X + (g * T * (long long)F) * M);
    const float* mp = a.mask + g * T * (long long)F + min(f, nbin);
    const bool mload = threadIdx.x < 64;
    double sr[NPAIR], si[NPAIR], nr[NPAIR], ni[NPAIR];
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) sr[q] = si[q] = nr[q] = ni[q] = 0.0;
    float pq[D][4], pm[D];                                                   // (scalars on purpose, cf. k_apply_mq)
    auto fetch = [&](int t, int slot) {
        const int t_ = t < t1 ? t : t1 - 1;
        const float4 v = xq[(long long)t_ * F * MH + lgo];
        pq[slot][0] = v.x;
        pq[slot][1] = v.y;
        pq[slot][2] = v.z;
        pq[slot][3] = v.w;
        pm[slot] = mload ? mp[(long long)t_ * F] : 0.f;
    };
    auto pair = [&](const c32 u, const c32 v, const double wa, const double wb, int q) {
        const c32 p = cmul_aconjb(u, v);
        const double pr = (double)p.x, pi = (double)p.y;
        sr[q] = fma(wa, pr, sr[q]);
        nr[q] = fma(wb, pr, nr[q]);
        si[q] = fma(wa, pi, si[q]);
        ni[q] = fma(wb, pi, ni[q]);
    };
    auto diag = [&](const c32 u, const double wa, const double wb, int q) {
        const double pr = (double)fmaf(u.x, u.x, u.y * u.y);
        sr[q] = fma(wa, pr, sr[q]);
        nr[q] = fma(wb, pr, nr[q]);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(t0 + d, d);
    for (int t = t0; t < t1; ++t) {
        const int s = (t - t0) & 1;
        sh.x[s][lpos] = make_float4(pq[0][0], pq[0][1], pq[0][2], pq[0][3]);
        if (mload) sh.m[s][lane] = pm[0];
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) pq[d][e] = pq[d + 1][e];
            pm[d] = pm[d + 1];
        }
        fetch(t + D, D - 1);
        __syncthreads();                                 // (the tile this frame's successor's successor overwrites was read before the next barrier)
        c32 ux[NX > 0 ? NX : 1], uy[NY > 0 ? NY : 1];
#pragma unroll
        for (int p_ = 0; p_ < NX / 2; ++p_) {
            const float4 v = sh.x[s][lane * MH + ((X0 / 2 + p_) ^ swz)];
            ux[2 * p_] = make_float2(v.x, v.y);
            ux[2 * p_ + 1] = make_float2(v.z, v.w);
        }
#pragma unroll
        for (int p_ = 0; p_ < NY / 2; ++p_) {
            const float4 v = sh.x[s][lane * MH + ((Y0 / 2 + p_) ^ swz)];
            uy[2 * p_] = make_float2(v.x, v.y);
            uy[2 * p_ + 1] = make_float2(v.z, v.w);
        }
        const float mcur = sh.m[s][lane];
        const double m = live ? (double)mcur : 0.0, mc = live ? 1.0 - (double)mcur : 0.0;
        const double wa = m * m, wb = mc * mc;
        int q = 0;                                       // (the order of cov_split_accumulate / cov_split_store)
        if constexpr (TRI) {
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int j = i; j < NX; ++j, ++q) {
                    if (j == i) diag(ux[i], wa, wb, q);
                    else pair(ux[i], ux[j], wa, wb, q);
                }
#pragma unroll
            for (int i = 0; i < NY; ++i)
#pragma unroll
                for (int j = i; j < NY; ++j, ++q) {
                    if (j == i) diag(uy[i], wa, wb, q);
                    else pair(uy[i], uy[j], wa, wb, q);
                }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int j = 0; j < NY; ++j, ++q) pair(ux[i], uy[j], wa, wb, q);
        }
    }
    if (live) {
        c32 hs[NPAIR], hn[NPAIR], ls[NPAIR], ln[NPAIR];
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) {
            hs[q] = make_float2((float)sr[q], (float)si[q]);
            hn[q] = make_float2((float)nr[q], (float)ni[q]);
            ls[q] = make_float2((float)(sr[q] - (double)hs[q].x), (float)(si[q] - (double)hs[q].y));
            ln[q] = make_float2((float)(nr[q] - (double)hn[q].x), (float)(ni[q] - (double)hn[q].y));
        }
        float4* o = a.part + (((g * (2 * a.chunks) + 2 * c) * F) + f) * (long long)NP;
        cov_split_store<P, X0, Y0, TRI, NX, NY>(o, hs, hn);
        cov_split_store<P, X0, Y0, TRI, NX, NY>(o + (long long)F * NP, ls, ln);
    }
}

// grid = R*Kl * (tiles + 1) * chunks blocks of 4 waves, tiles = ceil((F - 1) / 64); partial blocks: 2 * chunks
template <int M>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(256) void k_cov_loc_f64(CovArgs a) {
    constexpr int MA = (M + 1) / 2, MB = MA + (M - MA + 1) / 2;
    const int nbin = a.F - 1, tiles = (nbin + 63) / 64;
    int bid = blockIdx.x;
    const int c = bid % a.chunks;
    bid /= a.chunks;
    const int tile = bid % (tiles + 1);
    const long long g = bid / (tiles + 1);
    const int lane = threadIdx.x & 63;
    if constexpr (M == 8) {
        if (tile < tiles) {                              // (uniform over the workgroup) the 64-bin tiles: one fetch per workgroup
            __shared__ CovLocTile<M> sh;
            switch (wave_id()) {
                case 0: cov_loc_f64_wave_tile<M, 0, MA, MA, MA, true>(a, g, c, tile, lane, sh); break;
                case 1: cov_loc_f64_wave_tile<M, MA, M, M, M, true>(a, g, c, tile, lane, sh); break;
                case 2: cov_loc_f64_wave_tile<M, 0, MA, MA, MB, false>(a, g, c, tile, lane, sh); break;
                default: cov_loc_f64_wave_tile<M, 0, MA, MB, M, false>(a, g, c, tile, lane, sh); break;
            }
            return;
        }
    }
    switch (wave_id()) {
        case 0: cov_loc_f64_wave<M, 0, MA, MA, MA, true>(a, g, c, tile, lane); break;          // tri(A0)
        case 1: cov_loc_f64_wave<M, MA, M, M, M, true>(a, g, c, tile, lane); break;            // tri(A1)
        case 2: cov_loc_f64_wave<M, 0, MA, MA, MB, false>(a, g, c, tile, lane); break;         // A0 x B0
        default: cov_loc_f64_wave<M, 0, MA, MB, M, false>(a, g, c, tile, lane); break;         // A0 x B1
    }
}

// waves per workgroup: six blocks of pairs that involve the remote rows (none when KR = 0: the step-1 shape), two for the
// local M x M block (none with SKIPLOC)
template <int KR, bool SKIPLOC>
constexpr int cov_split_waves() { return (KR > 0 ? 6 : 0) + (SKIPLOC ? 0 : 2); }

// role of wave `role` of a workgroup: fn(CovSplitRole<...>{}) with the block of pairs it owns
template <int M, int KR, bool SKIPLOC, class Fn>
__device__ __forceinline__ void cov_split_roles(const int role, Fn&& fn) {
    constexpr int MA = (M + 1) / 2, KB = (KR + 1) / 2, P = M + KR;
    switch (role) {
        case 0: fn(CovSplitRole<M, KR, 0, MA, M, M + KB, false>{}); break;
        case 1: fn(CovSplitRole<M, KR, 0, MA, M + KB, P, false>{}); break;
        case 2: fn(CovSplitRole<M, KR, MA, M, M, M + KB, false>{}); break;
        case 3: fn(CovSplitRole<M, KR, MA, M, M + KB, P, false>{}); break;
        case 4: fn(CovSplitRole<M, KR, M, M + KB, M + KB, P, true>{}); break;
        case 5: fn(CovSplitRole<M, KR, M, M + KB, M + KB, P, false>{}); break;
        case 6:
            if constexpr (!SKIPLOC) fn(CovSplitRole<M, KR, 0, MA, MA, M, true>{});
            break;
        default:
            if constexpr (!SKIPLOC) fn(CovSplitRole<M, KR, 0, MA, MA, M, false>{});
            break;
    }
}

// ---- the same partition with the frames staged through LDS ---------------------------------------------------------------------
// In k_cov_split every wave fetches the (up to two) half-groups of v its block of pairs touches: 2.9 x the tile's bytes leave
// L1/L2, and because the waves of a workgroup drift apart, 1.36 x reach the fabric (PMC, C5).  Here the workgroup's waves share
// the fetch: DISCO_COV_STAGE_FRAMES frames of the tile (mics as they lie in X, the remote rows plane by plane, the mask) are
// loaded ONCE, 64 lanes x 16 / 8 / 4 bytes per wave-slot dealt round-robin to the waves, parked in registers during the
// arithmetic on the previous stage, written to the other of two LDS buffers and published by one barrier per stage.  Needs
// M even (16-byte granules of X) and F - 1 a multiple of 64 (every n_fft this library supports); the Nyquist tile, one bin
// over many frames, keeps the direct path of cov_split_wave.
#ifndef DISCO_COV_STAGE_FRAMES
#define DISCO_COV_STAGE_FRAMES 2
#endif
#ifndef DISCO_COV_XCD
#define DISCO_COV_XCD 8               // XCDs the workgroup ids are dealt over (0: plain tile-fastest ids)
#endif
#ifndef DISCO_COV_LDS_WPE
#define DISCO_COV_LDS_WPE 3           // waves per SIMD the register allocation leaves room for
#endif

template <int M, int KR, int S>
struct alignas(16) CovStage {
    static constexpr int XP = M + 2;          // pitch of a bin's mic row: (M + 2) * 8 B keeps the lanes' ds_read_b128 off each other's banks
    c32 xs[S][64][XP];
    c32 zs[S][KR > 0 ? KR : 1][64];
    float ms[S][64];
};

template <int M, int KR, int S, int NW>
struct CovStageLoader {
    static constexpr int XH = M / 2, WSX = S * XH, WSZ = S * KR, WS = WSX + WSZ + S, NS = (WS + NW - 1) / NW;
    float4 r[NS];
    // wave-slot `slot` (wave-uniform): X granules first, then the remote rows, then the mask
    __device__ __forceinline__ void load(const CovArgs& a, const long long g, const int f0, const int ts, const int t1, const int wid,
                                         const int lane) {
        const int T = a.T, F = a.F;
        const long long r_ = g / a.Kl;
        const int k = a.k0 + (int)(g % a.Kl);
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            const int slot = wid + n * NW;
            if (slot < WSX) {
                int t = ts + slot / XH;
                t = t < t1 ? t : t1 - 1;
                const float4* src = reinterpret_cast<const float4*>(a.X + ((g * T + t) * (long long)F + f0) * M);
                r[n] = src[(slot % XH) * 64 + lane];
            } else if (slot < WSX + WSZ) {
                if constexpr (KR > 0) {
                    const int zz = slot - WSX, jj = zz % KR;
                    int t = ts + zz / KR;
                    t = t < t1 ? t : t1 - 1;
                    const int j = jj < k ? jj : jj + 1;                          // concatenate_signals order
                    const c32 v = a.Zs[(z_plane(r_, j, a.K, a.R, a.zblk) * T + t) * (long long)F + f0 + lane];
                    r[n] = make_float4(v.x, v.y, 0.f, 0.f);
                }
            } else if (slot < WS) {
                int t = ts + (slot - WSX - WSZ);
                t = t < t1 ? t : t1 - 1;
                r[n] = make_float4(a.mask[(g * T + t) * (long long)F + f0 + lane], 0.f, 0.f, 0.f);
            }
        }
    }
    __device__ __forceinline__ void store(CovStage<M, KR, S>& st, const int wid, const int lane) const {
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            const int slot = wid + n * NW;
            if (slot < WSX) {
                const int i = (slot % XH) * 64 + lane;                           // float4 granule i of the frame's [64][M] block
                *reinterpret_cast<float4*>(&st.xs[slot / XH][i / XH][2 * (i % XH)]) = r[n];
            } else if (slot < WSX + WSZ) {
                if constexpr (KR > 0) {
                    const int zz = slot - WSX;
                    st.zs[zz / KR][zz % KR][lane] = make_float2(r[n].x, r[n].y);
                }
            } else if (slot < WS) {
                st.ms[slot - WSX - WSZ][lane] = r[n].x;
            }
        }
    }
};

// components [C0, C1) of the lane's bin in frame s_ of a stage
template <int M, int KR, int S, int C0, int C1>
__device__ __forceinline__ void cov_stage_read(c32* u, const CovStage<M, KR, S>& st, const int s_, const int lane) {
    constexpr int XA = C0 < M ? C0 : M, XB = C1 < M ? C1 : M, ZA = C0 > M ? C0 : M;
    if constexpr (XB > XA) {
        if constexpr (XA % 2 == 0 && (XB - XA) % 2 == 0) {
#pragma unroll
            for (int p = 0; p < (XB - XA) / 2; ++p) {
                const float4 q = *reinterpret_cast<const float4*>(&st.xs[s_][lane][XA + 2 * p]);
                u[XA - C0 + 2 * p] = make_float2(q.x, q.y);
                u[XA - C0 + 2 * p + 1] = make_float2(q.z, q.w);
            }
        } else {
#pragma unroll
            for (int c = XA; c < XB; ++c) u[c - C0] = st.xs[s_][lane][c];
        }
    }
#pragma unroll
    for (int c = ZA; c < C1; ++c) u[c - C0] = st.zs[s_][c - M][lane];
}

template <int M, int KR, int NW, class Role>
__device__ __forceinline__ void cov_split_wave_lds(const CovArgs& a, const long long g, const int c, const int tile, const int lane,
                                                   const int wid, CovStage<M, KR, DISCO_COV_STAGE_FRAMES>* sh) {
    constexpr int S = DISCO_COV_STAGE_FRAMES, P = M + KR, NP = P * (P + 1) / 2;
    constexpr int NX = Role::NX, NY = Role::NY, NPAIR = Role::NPAIR, NPA = NPAIR > 0 ? NPAIR : 1;
    const int T = a.T, F = a.F;
    const int t0 = (int)(((long long)T * c) / a.chunks), t1 = (int)(((long long)T * (c + 1)) / a.chunks);
    const int f0 = tile * 64;
    c32 acc_s[NPA], acc_n[NPA];
#pragma unroll
    for (int q = 0; q < NPA; ++q) acc_s[q] = acc_n[q] = make_float2(0.f, 0.f);
    CovStageLoader<M, KR, S, NW> ld;
    ld.load(a, g, f0, t0, t1, wid, lane);
    ld.store(sh[0], wid, lane);
    __syncthreads();
    int b = 0;
    for (int ts = t0; ts < t1; ts += S, b ^= 1) {          // every wave of the workgroup walks the same stages: the barriers match
        const bool more = ts + S < t1;
        if (more) ld.load(a, g, f0, ts + S, t1, wid, lane);          // in flight during the arithmetic below
        if constexpr (NPAIR > 0) {
#pragma unroll
            for (int s_ = 0; s_ < S; ++s_) {
                c32 ux[NX > 0 ? NX : 1], uy[NY > 0 ? NY : 1];
                cov_stage_read<M, KR, S, Role::x0, Role::x1>(ux, sh[b], s_, lane);
                cov_stage_read<M, KR, S, Role::y0, Role::y1>(uy, sh[b], s_, lane);
                const float mk = sh[b].ms[s_][lane];
                const bool ok = ts + s_ < t1;
                const float m = ok ? mk : 0.f, mc = ok ? 1.f - mk : 0.f;
                cov_split_accumulate<Role, Role::tri, NX, NY>(ux, uy, m * m, mc * mc, acc_s, acc_n);
            }
        }
        if (more) ld.store(sh[b ^ 1], wid, lane);
        __syncthreads();
    }
    if constexpr (NPAIR > 0) {
        float4* o = a.part + (((g * a.chunks + c) * F) + f0 + lane) * (long long)NP;
        cov_split_store<P, Role::x0, Role::y0, Role::tri, NX, NY>(o, acc_s, acc_n);
    }
}

template <int M, int KR, bool SKIPLOC>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__((64 * cov_split_waves<KR, SKIPLOC>()), DISCO_COV_LDS_WPE) void k_cov_split_lds(CovArgs a) {
    static_assert(M % 2 == 0, "16-byte granules of X");     // KR = 0: the step-1 statistics of a wide node (M = 8), same staging
    constexpr int NW = cov_split_waves<KR, SKIPLOC>();
    __shared__ CovStage<M, KR, DISCO_COV_STAGE_FRAMES> sh[2];
    const int nbin = a.F - 1, tiles = nbin / 64;          // the launcher checks nbin % 64 == 0
    // Which workgroup does what: the Kl nodes of a room read the same K - 1 remote rows of a (tile, chunk), so they are made
    // NEIGHBOURS ON ONE XCD (one L2): the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, hence id b is
    // logical item (b % 8) * (grid / 8) + b / 8, and logical items run node-fastest.  (Tile-fastest ids, as k_cov_split's,
    // spread the 8 nodes of a room over 8 L2s and every remote row crosses the fabric up to K - 1 times.)
#if DISCO_COV_XCD
    const long long n_items = (long long)a.R * a.Kl * (tiles + 1) * a.chunks;
    long long item = (long long)(blockIdx.x % DISCO_COV_XCD) * (gridDim.x / DISCO_COV_XCD) + blockIdx.x / DISCO_COV_XCD;
    if (item >= n_items) return;                           // the grid is padded to a multiple of 8
    const int kl = (int)(item % a.Kl);
    item /= a.Kl;
    const int c = (int)(item % a.chunks);
    item /= a.chunks;
    const int tile = (int)(item % (tiles + 1));
    const long long g = (item / (tiles + 1)) * a.Kl + kl;
#else
    int bid = blockIdx.x;
    const int c = bid % a.chunks;
    bid /= a.chunks;
    const int tile = bid % (tiles + 1);
    const long long g = bid / (tiles + 1);
#endif
    const int lane = threadIdx.x & 63;
    const int wid = wave_id();
    const int role = wid + (KR > 0 ? 0 : 6);               // (as in k_cov_split: without remote rows only the local block's two roles exist)
    if (tile == tiles) {                                   // the Nyquist bin: lanes are frames there, no barrier on that path
        cov_split_roles<M, KR, SKIPLOC>(role, [&](auto tag) {
            using R_ = decltype(tag);
            cov_split_wave<M, KR, R_::x0, R_::x1, R_::y0, R_::y1, R_::tri>(a, g, c, tile, lane);
        });
        return;
    }
    cov_split_roles<M, KR, SKIPLOC>(role, [&](auto tag) {
        cov_split_wave_lds<M, KR, NW, decltype(tag)>(a, g, c, tile, lane, wid, sh);
    });
}

// part [n_gf/F][chunks][F][NP] -> Rss, Rnn [n_gf][P][P], mean over T, Hermitian mirror.
static __global__ void k_cov_finalize(const float4* __restrict__ part, c32* __restrict__ Rss, c32* __restrict__ Rnn,
                               long long n_gf, int F, int chunks, int P, float inv_T) {
    const int NP = P * (P + 1) / 2;
    for (long long gf = (long long)blockIdx.x * blockDim.x + threadIdx.x; gf < n_gf; gf += (long long)gridDim.x * blockDim.x) {
        const long long g = gf / F;
        const int f = (int)(gf % F);
        int q = 0;
        for (int i = 0; i < P; ++i)
            for (int j = i; j < P; ++j, ++q) {
                // the chunk sums are combined in float64 and rounded once, exactly as the solvers' own loaders do (k_solve.h,
                // k_solve_small.h): the matrices handed out are bit-identical to the pencils the pending solve works on
                double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
                for (int c = 0; c < chunks; ++c) {
                    const float4 p = part[(((g * chunks + c) * F) + f) * (long long)NP + q];
                    sx += (double)p.x;
                    sy += (double)p.y;
                    sz += (double)p.z;
                    sw += (double)p.w;
                }
                const double it = (double)inv_T;
                float4 s = make_float4((float)(sx * it), (float)(sy * it), (float)(sz * it), (float)(sw * it));
                if (i == j) s.y = s.w = 0.f;
                c32* rs = Rss + gf * P * P;
                c32* rn = Rnn + gf * P * P;
                rs[i * P + j] = make_float2(s.x, s.y);
                rn[i * P + j] = make_float2(s.z, s.w);
                if (i != j) {
                    rs[j * P + i] = make_float2(s.x, -s.y);
                    rn[j * P + i] = make_float2(s.z, -s.w);
                }
            }
    }
}

}  // namespace disco
