// Step-2 statistics of ALL nodes of a room in one pass over X (tango.py:369-374 + 382-386 + 433-440, P = M + K - 1 > 8).
//
// The staged route for wide shapes is two passes per step-2 iteration: k_apply (X -> z_k = w_k^H x_k, every node) and
// k_cov_split_lds (X and the K - 1 remote z's -> node k's covariances): X crosses the memory system twice and every z row is
// written once and read K - 1 times.  Here ONE workgroup owns (room, 32-bin tile, frame chunk) for all K nodes:
//   * the loader lanes fetch the room's frame of the tile -- K contiguous runs of 32 * M spectra -- 16 bytes per lane,
//     park them in registers during the arithmetic on the previous stage and write them to the other of two LDS buffers;
//   * on the way each loader lane multiplies its two mics with its two (register-resident) filter taps; the M/2 lanes of a
//     (node, bin) add up through cross-lane moves: that IS z_k(t, f), written to LDS for the covariances and once to HBM
//     (the filter pass of the last iteration reads it);
//   * after one barrier per stage a lane is (bin, slot): slot A(k, h) folds the 4 mics [4h, 4h + 4) of node k against the
//     K - 1 remote z's (<= 28 pairs), slot B(k) folds the upper triangle of the remote z's of node k (<= 28 pairs) -- both
//     statistics at once with the node's own mask (mask_for_z = 'local'): 112 accumulator registers per lane.
// The leading M x M block of every node is the step-1 covariance (same mask, same X): it is NOT formed here -- the solver
// takes it from the step-1 partial sums (SolveSrc::part_loc), exactly as after k_cov_split_lds<.., SKIPLOC = true>.
// Output: partial sums in the layout of every other covariance kernel, part[(g * chunks + c) * F + f][tri(i, j)] for the
// entries with j >= M, so k_gevd_mwf_r1 reads them unchanged.
#pragma once
#include "common.h"
#include "k_cov.h"
#include "pk.h"

namespace disco {

#ifndef DISCO_ROOM_STAGE_FRAMES
#define DISCO_ROOM_STAGE_FRAMES 1
#endif
#ifndef DISCO_ROOM_WPE
#define DISCO_ROOM_WPE 3              // waves per SIMD the register allocation must leave room for (a 12-wave workgroup needs 3)
#endif

template <int M, int K>
struct RoomGeom {
    static_assert(M % 4 == 0 && K % 2 == 0 && K >= 2 && K <= 8, "4-mic slots, two slots per wave, <= 28 pairs per slot");
    static constexpr int KR = K - 1, P = M + KR, NP = P * (P + 1) / 2;
    static constexpr int NB = 32;                       // bins per workgroup (half a wave)
    static constexpr int NA = M / 4;                    // A slots per node
    static constexpr int WA = K * NA / 2, WB = K / 2;   // waves of A slots, waves of B slots
    static constexpr int NT = 64 * (WA + WB);
    static constexpr int MH = M / 2;                    // 16-byte granules (two mics) per bin
    static constexpr int NITEMS = K * NB * MH;          // granules of one frame of the tile
    static constexpr int NL = (NITEMS + NT - 1) / NT;   // loader rounds
    static_assert(NITEMS % 64 == 0 && (NITEMS - (NL - 1) * NT) % 64 == 0, "whole waves in every loader round");
    static_assert(K * NB <= NT, "one mask value per thread");
    static constexpr int XP = M + 2;                    // pitch of a bin's mic row in LDS (keeps ds_read_b128 of neighbouring lanes off each other's banks)
    static constexpr int S = DISCO_ROOM_STAGE_FRAMES;
};

template <int M, int K>
struct alignas(16) RoomStage {
    using Gm = RoomGeom<M, K>;
    c32 xs[Gm::S][K][Gm::NB][Gm::XP];
    c32 zs[Gm::S][K][Gm::NB];
    float ms[Gm::S][K][Gm::NB];
};

struct RoomArgs {
    const c32* X;        // [R][K][T][F][M]
    const float* mask;   // [R][K][T][F]
    const c32* w;        // [R][K][F][M]   step-1 filters (or the local part of the previous iteration's)
    c32* z;              // [R][K][T][F]   out
    float4* part;        // [R*K][chunks][F][NP]
    int T, F, chunks, tiles;
    long long R;
};

// IS_A: the role of the calling WAVE (slots A or B).  The whole stage loop is instantiated per role -- a role branch inside one
// loop makes hipcc carry the 112 accumulators through both arms with copies and spill the prefetch registers (584 bytes of
// scratch per lane); every wave executes the same number of barriers in either instantiation.
template <int M, int K, bool IS_A>
__device__ __forceinline__ void room_cov_run(const RoomArgs& a, RoomStage<M, K>* sh) {
    using Gm = RoomGeom<M, K>;
    constexpr int KR = Gm::KR, P = Gm::P, NP = Gm::NP, NB = Gm::NB, NA = Gm::NA, WA = Gm::WA, NT = Gm::NT, MH = Gm::MH;
    constexpr int NITEMS = Gm::NITEMS, NL = Gm::NL, S = Gm::S;
    const int T = a.T, F = a.F;
    long long item = blockIdx.x;
    const int c = (int)(item % a.chunks);
    item /= a.chunks;
    const int tile = (int)(item % a.tiles);
    const long long room = item / a.tiles;
    const int t0 = (int)(((long long)T * c) / a.chunks), t1 = (int)(((long long)T * (c + 1)) / a.chunks);
    const int f0 = tile * NB;
    const int tid = threadIdx.x, wid = wave_id(), lane = tid & 63;
    const int bin = lane & (NB - 1), half = lane >> 5;
    const bool live = f0 + bin < F;

    // ---- loader state: granule `it = tid + r * NT` of the tile's frame (32-bit offsets from the room's wave-uniform bases),
    // its two filter taps, the thread's mask element
    const c32* Xr = a.X + (room * K * T) * (long long)F * M;
    c32* Zr = a.z + (room * K * T) * (long long)F;
    const float* Mr = a.mask + (room * K * T) * (long long)F;
    int lxo[NL], lzo[NL], lxs[NL], lzs[NL];
    c32 lw0[NL], lw1[NL];
    bool lact[NL];
#pragma unroll
    for (int r = 0; r < NL; ++r) {
        const int it = tid + r * NT;
        lact[r] = wid * 64 + r * NT < NITEMS;            // whole waves (static_assert above): a scalar condition
        const int it_ = lact[r] ? it : 0;
        const int lk = it_ / (NB * MH), rem = it_ % (NB * MH), lbin = rem / MH, lp = rem % MH;
        const int lf = min(f0 + lbin, F - 1);            // clamped: unconditional loads, weighed 0 / not written when out of range
        lxo[r] = ((lk * T) * F + lf) * M + 2 * lp;       // + t * F * M
        lzo[r] = (lp == 0 && f0 + lbin < F) ? (lk * T) * F + f0 + lbin : -1;      // + t * F; -1: this lane does not write z
        lxs[r] = (lk * NB + lbin) * Gm::XP + 2 * lp;     // into xs[s_] (c32 units)
        lzs[r] = lp == 0 ? lk * NB + lbin : -1;          // into zs[s_]
        const c32* wp = a.w + ((room * K + lk) * F + lf) * (long long)M + 2 * lp;
        lw0[r] = wp[0];
        lw1[r] = wp[1];
    }
    const bool mact = wid * 64 < K * NB;                 // whole waves: K * NB is a multiple of 64
    const int mk = mact ? tid / NB : 0, mb = tid % NB;
    const int mo = (mk * T) * F + min(f0 + mb, F - 1);

    float4 lr[S][NL];
    float lm[S];
    auto load = [&](int ts) {
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            int t = ts + s_;
            t = t < t1 ? t : t1 - 1;
#pragma unroll
            for (int r = 0; r < NL; ++r)
                if (lact[r]) lr[s_][r] = *reinterpret_cast<const float4*>(Xr + lxo[r] + t * F * M);
            if (mact) lm[s_] = Mr[mo + t * F];
        }
    };
    auto store = [&](RoomStage<M, K>& st, int ts) {
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            const int t = ts + s_;
#pragma unroll
            for (int r = 0; r < NL; ++r) {
                if (lact[r]) {                          // whole waves
                    const float4 q = lr[s_][r];
                    *reinterpret_cast<float4*>(&st.xs[s_][0][0][0] + lxs[r]) = q;
                    // z = w^H x: this lane's two taps, then the MH lanes of the (node, bin)
                    c32 p = cfma_conj(lw0[r], make_float2(q.x, q.y), make_float2(0.f, 0.f));
                    p = cfma_conj(lw1[r], make_float2(q.z, q.w), p);
#pragma unroll
                    for (int off = 1; off < MH; off <<= 1) {
                        p.x += __shfl_xor(p.x, off);
                        p.y += __shfl_xor(p.y, off);
                    }
                    if (lzs[r] >= 0) (&st.zs[s_][0][0])[lzs[r]] = p;
                    if (lzo[r] >= 0 && t < t1) Zr[lzo[r] + t * F] = p;
                }
            }
            if (mact) st.ms[s_][mk][mb] = lm[s_];
        }
    };

    // ---- the lane's slot
    constexpr bool is_a = IS_A;
    const int slot = (is_a ? wid : wid - WA) * 2 + half;
    const int k = is_a ? slot / NA : slot;
    const int h = is_a ? slot % NA : 0;
    constexpr int NACC = IS_A ? 4 * KR : KR * (KR + 1) / 2;
    c32 acc_s[NACC], acc_n[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc_s[q] = acc_n[q] = make_float2(0.f, 0.f);

    load(t0);
    store(sh[0], t0);
    __syncthreads();
    int b = 0;
    for (int ts = t0; ts < t1; ts += S, b ^= 1) {
        const bool more = ts + S < t1;
        if (more) load(ts + S);                         // in flight during the arithmetic below
        const RoomStage<M, K>& st = sh[b];
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            const float mkv = st.ms[s_][k][bin];
            const bool ok = live && ts + s_ < t1;
            const float m = ok ? mkv : 0.f, mc = ok ? 1.f - mkv : 0.f;
            const float wa = m * m, wb = mc * mc;
            if constexpr (is_a) {
                c32 x[4];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const float4 q = *reinterpret_cast<const float4*>(&st.xs[s_][k][bin][4 * h + 2 * p]);
                    x[2 * p] = make_float2(q.x, q.y);
                    x[2 * p + 1] = make_float2(q.z, q.w);
                }
#pragma unroll
                for (int jj = 0; jj < KR; ++jj) {
                    const int j = jj < k ? jj : jj + 1;                   // concatenate_signals order
                    const c32 z = st.zs[s_][j][bin];
#pragma unroll
                    for (int i = 0; i < 4; ++i) cov_pair_acc(x[i], z, wa, wb, acc_s[i * KR + jj], acc_n[i * KR + jj]);
                }
            } else {
                c32 z[KR];
#pragma unroll
                for (int jj = 0; jj < KR; ++jj) z[jj] = st.zs[s_][jj < k ? jj : jj + 1][bin];
                int q = 0;
#pragma unroll
                for (int i = 0; i < KR; ++i)
#pragma unroll
                    for (int j = i; j < KR; ++j, ++q) {
                        if (j == i) cov_diag_acc(z[i], wa, wb, acc_s[q], acc_n[q]);
                        else cov_pair_acc(z[i], z[j], wa, wb, acc_s[q], acc_n[q]);
                    }
            }
        }
        if (more) store(sh[b ^ 1], ts + S);
        __syncthreads();
    }
    if (live) {
        float4* o = a.part + ((((room * K + k) * a.chunks + c) * F) + f0 + bin) * (long long)NP;
        if constexpr (is_a) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < KR; ++jj) {
                    const int q = i * KR + jj;
                    o[tri_index<P>(4 * h + i, M + jj)] = make_float4(acc_s[q].x, acc_s[q].y, acc_n[q].x, acc_n[q].y);
                }
        } else {
            int q = 0;
#pragma unroll
            for (int i = 0; i < KR; ++i)
#pragma unroll
                for (int j = i; j < KR; ++j, ++q)
                    o[tri_index<P>(M + i, M + j)] = make_float4(acc_s[q].x, acc_s[q].y, acc_n[q].x, acc_n[q].y);
        }
    }
}

template <int M, int K>
__global__ __launch_bounds__((RoomGeom<M, K>::NT), DISCO_ROOM_WPE) void k_room_cov(RoomArgs a) {
    __shared__ RoomStage<M, K> sh[2];
    if (wave_id() < RoomGeom<M, K>::WA) room_cov_run<M, K, true>(a, sh);
    else room_cov_run<M, K, false>(a, sh);
}

}  // namespace disco
