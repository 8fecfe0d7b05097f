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

// NB_: bins per workgroup.  32 = two slots per wave, one 12-wave workgroup per CU (C5 shape).  16 = four slots per wave: half the
// lanes per workgroup, so TWO workgroups share a CU and run out of phase -- each still meets its own barrier once per frame, but no
// longer the whole CU at once (the DMA variant only; needs whole waves of A slots and of B slots: K a multiple of 4).  Built, parity-green
// and slower (see room_tile16_shape): option "room_tile16", default 0.
template <int M, int K, int NB_ = 32>
struct RoomGeom {
    static_assert(NB_ == 32 || NB_ == 16, "half or quarter of a wave");
    static constexpr int NB = NB_;                      // bins per workgroup
    static constexpr int SPW = 64 / NB;                 // slots per wave
    static_assert(M % 4 == 0 && K % 2 == 0 && K >= 2 && K <= 8, "4-mic slots, <= 28 pairs per slot");
    static constexpr int KR = K - 1, P = M + KR, NP = P * (P + 1) / 2;
    static constexpr int NA = M / 4;                    // A slots per node
    static_assert((K * NA) % SPW == 0 && K % SPW == 0, "whole waves of A slots and of B slots");
    static constexpr int WA = K * NA / SPW, WB = K / SPW;   // waves of A slots, waves of B slots
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
__global__ DISCO_KERNEL_ALIGN __launch_bounds__((RoomGeom<M, K>::NT), DISCO_ROOM_WPE) void k_room_cov(RoomArgs a) {
    __shared__ RoomStage<M, K> sh[2];
    if (wave_id() < RoomGeom<M, K>::WA) room_cov_run<M, K, true>(a, sh);
    else room_cov_run<M, K, false>(a, sh);
}

// ---- the same pass with the frames fetched by LDS-DMA ---------------------------------------------------------------------------
// k_room_cov keeps ONE frame of the tile in flight per workgroup (registers: 112 of a lane's 168 are accumulators, a second
// frame does not fit) and there is one workgroup per CU: 4.4 MB in flight over the chip (8.6 ms per C5 launch; this variant: 7.1 ms,
// the two passes it replaces: 11.0 ms).  Here the spectra and masks go from HBM straight into an LDS ring of
// DISCO_ROOM_DEPTH frames (global_load_lds_dwordx4 / _dword: no registers), issued DISCO_ROOM_AHEAD = 3 frames ahead:
//   iteration t:  issue frame t + 3  ->  fold frame t  ->  wait until only that issue is outstanding (frame t + 2 has landed)
//                 ->  form z(t + 1) from the ring (own granule + taps, cross-lane sum), publish it  ->  barrier.
// An LDS-DMA wave-load writes 64 lanes x 16 B to consecutive LDS bytes, so a lane's LDS position is fixed and the granule it
// FETCHES is chosen instead: position (bin, lp') holds granule lp = lp' ^ swz(bin) -- an XOR swizzle that keeps the 16-byte
// reads of the covariance lanes (same granule, 16 consecutive bins) on distinct banks without padding.
// The waits are counted by hand (the instructions are inline asm: hipcc's own LDS-DMA tracking would wait for vmcnt(0) before every
// LDS read, the ring index being a run-time value): after the wait of iteration t at most the loads of frame t + 3 are
// outstanding; VMEM operations return in order, the z stores of the previous iteration are older than that issue.  The kernel
// must not spill (a scratch access in the loop would shift the count): build.py checks the resource usage.
#ifndef DISCO_ROOM_AHEAD
#define DISCO_ROOM_AHEAD 3              // frames between a frame's LDS-DMA issue and its fold.  3 = one iteration between an issue and the wait
                                        // for it; 4 / 5 (two / three iterations, 16 KB of LDS each) measured 14.12 / 14.19 ms against 14.08 ms per
                                        // C5 step (profiles/r03_n_*): the pass is not waiting for its loads
#endif
#ifndef DISCO_ROOM_FPB
#define DISCO_ROOM_FPB 2                // frames per barrier.  2: six ring slots, two frames issued / formed / folded per iteration, one wait for
                                        // both -- 13.4 against 14.0 ms per C5 step for 1 (profiles/r03_t_*), bit-identical sums
#endif
#ifndef DISCO_ROOM_DEPTH
#define DISCO_ROOM_DEPTH (DISCO_ROOM_FPB == 2 ? 6 : DISCO_ROOM_AHEAD + 1)
#endif

// Shapes for which the DMA variant also exists on 16-bin tiles (option "room_tile16").  Measured SLOWER on the MI355X -- 16.8 against
// 14.05 ms per C5 step (profiles/r03_s_*): the pass is bound by what a frame costs a workgroup whatever its width (barrier, LDS-DMA
// issue, z formation), not by phase-locked waves.  Kept selectable, and tested, as the record of it.
template <int M, int K>
constexpr bool room_tile16_shape() { return K % 4 == 0; }

template <int M, int K, int NB_>
struct alignas(16) RoomRing {
    using Gm = RoomGeom<M, K, NB_>;
    float4 xs[DISCO_ROOM_DEPTH][Gm::NITEMS];           // granules, linear in the loader's item index
    float ms[DISCO_ROOM_DEPTH][K * Gm::NB];
    c32 zs[2 * DISCO_ROOM_FPB][K][Gm::NB];
    c32 wt[K][Gm::NB][M];
};

// 64 lanes x 16 (4) bytes, lane i from gbase + off[i] (gbase wave-uniform: an SGPR pair, off a 32-bit byte offset: no per-lane
// 64-bit address arithmetic), to the LDS bytes [lds_wave, lds_wave + 1024 (256)) in lane order; lds_wave is wave-uniform.
// M0 (the LDS-DMA destination base) is compiler-reserved: saved, written and restored inside the one statement that reads it;
// the nops cover "SALU writes an SGPR -> VMEM reads it as base" (5 wait states) for a base the compiler has just formed.
// hipcc neither counts these loads nor waits for them (vm_wait below does).
__device__ __forceinline__ void lds_dma16(const void* gbase, unsigned off, void* lds_wave, int lane) {
#if defined(__clang__)
    (void)lane;
    const unsigned l = (unsigned)(unsigned long long)lds_wave;            // low half of a flat LDS address = the LDS byte address
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(off), "s"(gbase), "s"(l)
                 : "memory");
#else
    reinterpret_cast<float4*>(lds_wave)[lane] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(gbase) + off);
#endif
}
__device__ __forceinline__ void lds_dma4(const void* gbase, unsigned off, void* lds_wave, int lane) {
#if defined(__clang__)
    (void)lane;
    const unsigned l = (unsigned)(unsigned long long)lds_wave;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(off), "s"(gbase), "s"(l)
                 : "memory");
#else
    reinterpret_cast<float*>(lds_wave)[lane] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(gbase) + off);
#endif
}
// at most N vector-memory operations of this wave still outstanding (N <= 9 here)
__device__ __forceinline__ void vm_wait(int n) {
#if defined(__clang__)
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    }
#else
    (void)n;
#endif
}

template <int M, int K, int NB_, bool IS_A>
__device__ __forceinline__ void room_cov_dma_run(const RoomArgs& a, RoomRing<M, K, NB_>& sh) {
    using Gm = RoomGeom<M, K, NB_>;
    constexpr int KR = Gm::KR, P = Gm::P, NP = Gm::NP, NB = Gm::NB, NA = Gm::NA, WA = Gm::WA, NT = Gm::NT, MH = Gm::MH;
    constexpr int NITEMS = Gm::NITEMS, NL = Gm::NL, D = DISCO_ROOM_DEPTH;
    constexpr int BPR = 16 / MH;                       // bins per 256-byte bank row of granules
    static_assert(DISCO_ROOM_FPB == 2 ? D == 6 : (D == DISCO_ROOM_AHEAD + 1 && (DISCO_ROOM_AHEAD - 2) * (NL + 1) <= 9), "AHEAD frames ahead; vm_wait knows 0..9");
    const int T = a.T, F = a.F;
    long long item = blockIdx.x;
    const int c = (int)(item % a.chunks);
    item /= a.chunks;
    const int tile = (int)(item % a.tiles);
    const long long room = item / a.tiles;
    const int t0 = (int)(((long long)T * c) / a.chunks), t1 = (int)(((long long)T * (c + 1)) / a.chunks);
    const int f0 = tile * NB;
    const int tid = threadIdx.x, wid = wave_id(), lane = tid & 63;
    const int bin = lane & (NB - 1), sub = lane / NB;
    const bool live = f0 + bin < F;

    const c32* Xr = a.X + (room * K * T) * (long long)F * M;
    c32* Zr = a.z + (room * K * T) * (long long)F;
    const float* Mr = a.mask + (room * K * T) * (long long)F;
    // loader lane: LDS position it = tid + r * NT  <->  node lk, bin lbin, granule lp = lp' ^ swz(lbin); bins beyond F - 1 fetch (and
    // later re-write) bin F - 1: every load and store is issued by every lane, the counts the waits rely on are exact
    unsigned lxo[NL];
    int lwt[NL], lzs[NL], lzo[NL];                      // taps in wt (c32 units), z slot in zs (-1: not this lane's), z offset in the room's z block
    bool lact[NL];
    int nload = 0;
#pragma unroll
    for (int r = 0; r < NL; ++r) {
        const int it = tid + r * NT;
        lact[r] = wid * 64 + r * NT < NITEMS;          // whole waves: a scalar condition
        nload += lact[r] ? 1 : 0;
        const int it_ = lact[r] ? it : 0;
        const int lk = it_ / (NB * MH), rem = it_ % (NB * MH), lbin = rem / MH, lp = (rem % MH) ^ ((lbin / BPR) % MH);
        const int lf = min(f0 + lbin, F - 1);
        lxo[r] = (unsigned)((((lk * T) * F + lf) * M + 2 * lp) * 8);           // bytes; + t * F * M * 8
        lwt[r] = (lk * NB + lbin) * M + 2 * lp;
        lzs[r] = (rem % MH == 0) ? lk * NB + lbin : -1;
        lzo[r] = (lk * T) * F + lf;                                              // + t * F; bins beyond F - 1 repeat bin F - 1's value
        if (lact[r]) {
            const float4 wq = *reinterpret_cast<const float4*>(a.w + ((room * K + lk) * F + lf) * (long long)M + 2 * lp);
            *reinterpret_cast<float4*>(&sh.wt[0][0][0] + (lk * NB + lbin) * M + 2 * lp) = wq;
        }
    }
    const bool mact = wid * 64 < K * NB;               // whole waves
    nload += mact ? 1 : 0;
    const unsigned mo = (unsigned)((((mact ? tid / NB : 0) * T) * F + min(f0 + tid % NB, F - 1)) * 4);

    auto issue = [&](int t, int slot_) {
        t = t < t1 ? t : t1 - 1;
#pragma unroll
        for (int r = 0; r < NL; ++r)
            if (lact[r]) lds_dma16(Xr + (long long)t * F * M, lxo[r], &sh.xs[slot_][wid * 64 + r * NT], lane);       // the frame's base is scalar
        if (mact) lds_dma4(Mr + (long long)t * F, mo, &sh.ms[slot_][wid * 64], lane);
    };
    // z(t) = w^H x of every (node, bin) of the tile from ring slot `slot_`: a lane's own granule and taps, then the MH lanes of the bin
    c32 zreg[NL];                                       // z of the frame just formed, on its way to HBM (stored after the counted wait)
    auto form_z = [&](int t, int slot_) {
#pragma unroll
        for (int r = 0; r < NL; ++r) {
            if (lact[r]) {
                const float4 q = sh.xs[slot_][tid + r * NT];
                const float4 wq = *reinterpret_cast<const float4*>(&sh.wt[0][0][0] + lwt[r]);
                c32 p = cfma_conj(make_float2(wq.x, wq.y), make_float2(q.x, q.y), make_float2(0.f, 0.f));
                p = cfma_conj(make_float2(wq.z, wq.w), make_float2(q.z, q.w), p);
                static_assert(MH == 2 || MH == 4, "the lanes of a bin are (part of) a quad");
                p.x = quad_xor_add<1>(p.x);
                p.y = quad_xor_add<1>(p.y);
                if constexpr (MH == 4) {
                    p.x = quad_xor_add<2>(p.x);
                    p.y = quad_xor_add<2>(p.y);
                }
                if (lzs[r] >= 0) (&sh.zs[t & (2 * DISCO_ROOM_FPB - 1)][0][0])[lzs[r]] = p;
                zreg[r] = p;
            }
        }
    };
    auto store_z = [&](int t) {
#pragma unroll
        for (int r = 0; r < NL; ++r)
            if (lact[r] && lzs[r] >= 0) Zr[lzo[r] + t * F] = zreg[r];
    };

    // ---- the lane's slot
    // slot = SPW * (wave within its role) + sub.  A: k = slot / NA, h = slot % NA; B: k = slot.  kw = the wave's first node (a
    // SCALAR), dk = k - kw in {0, 1} (always 0 when the two halves of a wave share a node, NA even): the remote row jj of a lane
    // is node jj + (jj >= kw + dk), which differs between the halves only for jj == kw -- every other LDS offset of a remote
    // row is a scalar instead of a per-lane select.
    // With SPW slots per wave: slot = SPW * (wave within its role) + sub; the wave's nodes are kw ... kw + DKMAX and only the remote
    // rows jj in [kw, kw + DKMAX) take a per-lane select.
    constexpr bool is_a = IS_A;
    constexpr int SPW = Gm::SPW;
    constexpr int DKMAX = IS_A ? (SPW - 1) / NA : SPW - 1;
    const int wr = is_a ? wid : wid - WA;
    const int kw = is_a ? (SPW * wr) / NA : SPW * wr;
    const int dk = is_a ? sub / NA : sub;
    const int k = kw + dk;
    const int h = is_a ? sub % NA : 0;
    auto remote = [&](int jj) {                          // concatenate_signals order: jj -> node jj (jj < k), jj + 1 (jj >= k)
        int j = jj + (jj >= kw + DKMAX ? 1 : 0);
        if (DKMAX > 0 && jj >= kw && jj < kw + DKMAX) j = jj + (jj >= k ? 1 : 0);
        return j;
    };
    const int swz = (bin / BPR) % MH;
    constexpr int NACC = IS_A ? 4 * KR : KR * (KR + 1) / 2;
    c32 acc_s[NACC], acc_n[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc_s[q] = acc_n[q] = make_float2(0.f, 0.f);

    auto fold = [&](int t, int slot_) {
        const float mkv = sh.ms[slot_][k * NB + bin];
        const float m = live ? mkv : 0.f, mc = live ? 1.f - mkv : 0.f;
        const float wa = m * m, wb = mc * mc;
        const c32(*zs)[NB] = sh.zs[t & (2 * DISCO_ROOM_FPB - 1)];
        if constexpr (is_a) {
            c32 x[4];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float4 q = sh.xs[slot_][(k * NB + bin) * MH + ((2 * h + p) ^ swz)];
                x[2 * p] = make_float2(q.x, q.y);
                x[2 * p + 1] = make_float2(q.z, q.w);
            }
            c32 z[KR];                                                    // all remote rows requested up front: one LDS latency per frame, not K - 1
#pragma unroll
            for (int jj = 0; jj < KR; ++jj) z[jj] = zs[remote(jj)][bin];
#pragma unroll
            for (int jj = 0; jj < KR; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) cov_pair_acc(x[i], z[jj], wa, wb, acc_s[i * KR + jj], acc_n[i * KR + jj]);
        } else {
            c32 z[KR];
#pragma unroll
            for (int jj = 0; jj < KR; ++jj) z[jj] = zs[remote(jj)][bin];
            int q = 0;
#pragma unroll
            for (int i = 0; i < KR; ++i)
#pragma unroll
                for (int j = i; j < KR; ++j, ++q) {
                    if (j == i) cov_diag_acc(z[i], wa, wb, acc_s[q], acc_n[q]);
                    else cov_pair_acc(z[i], z[j], wa, wb, acc_s[q], acc_n[q]);
                }
        }
    };

    if constexpr (DISCO_ROOM_FPB == 2) {
        // Two frames per barrier: at the top of an iteration z(t), z(t + 1) are published and frames t + 2, t + 3 are in LDS; the
        // iteration issues t + 4, t + 5 (into the slots of t - 2, t - 1), forms and stores z(t + 2), z(t + 3), folds t and t + 1 and
        // waits for everything it issued.  Half the barriers, loop overhead and waits per frame; a load has two form_z + two folds to land.
#pragma unroll
        for (int i = 0; i < 4; ++i) issue(t0 + i, i);
        vm_wait(0);
        __syncthreads();                                // frames t0 ... t0 + 3 and the taps are in place
        form_z(t0, 0);
        store_z(t0);
        if (t0 + 1 < t1) {
            form_z(t0 + 1, 1);
            store_z(t0 + 1);
        }
        __syncthreads();
        int s0 = 0;
        for (int t = t0; t < t1; t += 2) {
            issue(t + 4, (s0 + 4) % D);
            issue(t + 5, (s0 + 5) % D);
            if (t + 2 < t1) {
                form_z(t + 2, (s0 + 2) % D);
                store_z(t + 2);
            }
            if (t + 3 < t1) {
                form_z(t + 3, (s0 + 3) % D);
                store_z(t + 3);
            }
            fold(t, s0);
            if (t + 1 < t1) fold(t + 1, (s0 + 1) % D);
            vm_wait(0);
            __syncthreads();
            s0 = (s0 + 2) % D;
        }
    } else {
    constexpr int AH = DISCO_ROOM_AHEAD;                // frames a load is issued ahead of its fold
#pragma unroll
    for (int i = 0; i < AH; ++i) issue(t0 + i, i);
    vm_wait((AH - 2) * nload);                          // frames t0 and t0 + 1 have landed (this wave's part)
    __syncthreads();                                    // ... and everybody else's; the taps are in place
    form_z(t0, 0);
    store_z(t0);
    __syncthreads();
    int s0 = 0;                                         // ring slot of frame t
    for (int t = t0; t < t1; ++t) {
        issue(t + AH, (s0 + AH) % D);                   // the slot frame t - 1 was folded from (D = AH + 1)
        if (t + 1 < t1) form_z(t + 1, (s0 + 1) % D);    // frame t + 1 landed (and was published) an iteration ago; into the OTHER z buffer
        fold(t, s0);
        vm_wait((AH - 2) * nload);                      // only the last AH - 2 issues may still be in flight: frame t + 2 is in LDS (the z stores
                                                        // between them are younger than frame t + 2 either way: whether or not stores retire in
                                                        // order with loads, AH - 2 issues' worth of outstanding operations cannot include it)
        if (t + 1 < t1) store_z(t + 1);                 // after the counted wait: the stores never stand between an issue and its wait
        __syncthreads();
        s0 = (s0 + 1) % D;
    }
    }
    vm_wait(0);                                         // no LDS-DMA may outlive the workgroup's LDS allocation
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

template <int M, int K, int NB_>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__((RoomGeom<M, K, NB_>::NT), DISCO_ROOM_WPE) void k_room_cov_dma(RoomArgs a) {
    __shared__ RoomRing<M, K, NB_> sh;
    if (wave_id() < RoomGeom<M, K, NB_>::WA) room_cov_dma_run<M, K, NB_, true>(a, sh);
    else room_cov_dma_run<M, K, NB_, false>(a, sh);
}

}  // namespace disco
