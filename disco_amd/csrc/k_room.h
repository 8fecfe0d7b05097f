// Step-2 statistics of ALL nodes of a room in one pass over X (tango.py:369-374 + 382-386 + 433-440, P = M + K - 1 > 8).
//
// The staged route for wide shapes is two passes per step-2 iteration: k_apply (X -> z_k = w_k^H x_k, every node) and
// k_cov_split_lds (X and the K - 1 remote z's -> node k's covariances): X crosses the memory system twice and every z row is
// written once and read K - 1 times.  Here ONE workgroup owns a room's tile of bins for all K nodes:
//   * the frames of the tile -- K contiguous runs of spectra -- go from HBM into an LDS ring by LDS-DMA;
//   * every loader lane multiplies its two mics with its two filter taps; the M/2 lanes of a (node, bin) add up through cross-lane
//     moves: that IS z_k(t, f), written to LDS for the covariances and (last pass only) to HBM;
//   * a lane is (bin, slot, time sub-chunk): slot A(k, h) folds the 4 mics [4h, 4h + 4) of node k against the K - 1 remote z's
//     (<= 28 pairs), slot B(k) folds the upper triangle of the remote z's of node k (<= 28 pairs) -- both statistics at once with
//     the node's own mask (mask_for_z = 'local'): 112 accumulator registers per lane.
// The leading M x M block of every node is the step-1 covariance (same mask, same X): it is NOT formed here -- the solver
// takes it from the step-1 partial sums (SolveSrc::part_loc), exactly as after k_cov_split_lds<.., SKIPLOC = true>.
// Output: partial sums in the layout of every other covariance kernel, TWO blocks per node -- the float32 head and the remainder of totals
// formed in float64 --, part[(g * 2 + {0, 1}) * F + f][tri(i, j)] for the entries with j >= M.
// (Round 3's register-staged form of this pass -- one frame in flight per workgroup, one workgroup per (room, tile, chunk): 7.6 against
// 6.4 ms per C5 launch -- was removed in round 5; git history and profiles/r03_design_and_experiment_log.md have it.)
#pragma once
#include "common.h"
#include "k_cov.h"
#include "pk.h"

namespace disco {

#ifndef DISCO_ROOM_WPE
#define DISCO_ROOM_WPE 3              // waves per SIMD the register allocation must leave room for (a 12-wave workgroup needs 3)
#endif

struct RoomArgs {
    const c32* X;        // [R][K][T][F][M]
    const float* mask;   // [R][K][T][F]
    const c32* w;        // [R][K][F][M]   step-1 filters (or the local part of the previous iteration's)
    c32* z;              // [R][K][T][F]   out
    float4* part;        // [R*K][2][F][NP]   (hi, lo)
    int T, F, chunks, tiles;
    long long R;
    int store_z;         // 0: z is only formed on chip (an iteration whose z nobody reads: the next pass re-compresses with new filters)
};

// ---- ONE PERSISTENT workgroup per CU, frames fetched by LDS-DMA, time sub-chunks across its lanes ---------------------------------
// 112 of a lane's 168 registers are accumulators: a second frame of look-ahead does not fit them.  The spectra and masks therefore go
// from HBM straight into an LDS ring (global_load_lds_dwordx4 / _dword: no registers) two iterations ahead of their use.
//
// Round 4 -- what changed and why.  C5's two-iteration output was 2.3e-4 from the float64 oracle because every accumulator summed
// 157 frames sequentially in float32 (profiles/r03_c5_accumulation.txt); more frame chunks per node cured it at 1.2 GB of partial
// sums written and read back per extra chunk and launch.  A second accumulation level does not fit the registers (112 + 112).
// What does fit is MORE, SHORTER sums at the same register count: a workgroup now owns 32 / SUB bins and SUB CONSECUTIVE FRAMES at a
// time -- lane = (bin, slot, sub-chunk sc), the lane folds frames t0 + SUB u + sc, u = 0, 1, ... -- so a lane's sums are SUB times
// shorter, the bytes per barrier, the ring, the loader rounds and the arithmetic per lane are exactly what they were, and the SUB
// partial sums of an entry meet INSIDE the wave at the end (v_permlane32_swap / v_permlane16_swap: the two halves swap one register
// each and add, so every level also halves the entries a lane is left to store).  SUB = 8 sums 20 frames per accumulator where
// round 3 summed 157, and one TOTAL per node reaches HBM (as a (hi, lo) pair of blocks since the end of round 5: `finish` below).
// Shorter items would pay the per-workgroup prologue / epilogue (about 20 us of a 250 us item in round 3: dispatch, taps, ring fill,
// first z, 28 uncoalesced 16-byte stores per lane) four to eight times as often, so the workgroup is PERSISTENT: it walks items
// blockIdx.x, blockIdx.x + gridDim.x, ... (an item: a room's tile, all its frames) and the ring never drains -- the loads of the next item's first frames (and its taps, into
// the other tap buffer) are issued while the current item's last frames are folded, the sums of the finished item are reduced and
// stored under the first iteration of the next.
//
// Per iteration (two groups of SUB frames):
//   issue   groups of the iteration after next  ->  ring slots (s0 + 4, s0 + 5) % 6       [+ the next item's taps when it begins]
//   [reduce + store + clear the sums of the item whose last groups were folded in the previous iteration]
//   form    z of the next iteration's groups from the ring (own granule + taps, cross-lane sum), publish in LDS, store to HBM
//   fold    this iteration's two groups
//   s_waitcnt vmcnt(0); barrier
// An LDS-DMA wave-load writes 64 lanes x 16 B to consecutive LDS bytes, so a lane's LDS position is fixed and the granule it
// FETCHES is chosen instead: position (sc, node, bin, lp') holds granule lp = lp' ^ swz(bin, sc) -- an XOR swizzle that keeps the
// 16-byte reads of the covariance lanes on distinct banks without padding.  The waits are written by hand (the loads are inline
// asm: hipcc's own LDS-DMA tracking would wait for vmcnt(0) before every LDS read, the ring index being a run-time value); every
// iteration waits for everything it issued, so nothing depends on the order in which loads and stores retire.  build.py refuses the
// kernel beyond 64 bytes of scratch per lane (a few per-lane loader constants reloaded per iteration; the accumulators must stay in registers).
#ifndef DISCO_ROOM_EXP
#define DISCO_ROOM_EXP 0
#endif
#ifndef DISCO_ROOM_DEPTH
#define DISCO_ROOM_DEPTH 6              // ring slots: two groups per iteration, issued two iterations ahead
#endif

template <int M, int K, int SUB_>
struct RoomGeomS {
    static_assert(SUB_ == 2 || SUB_ == 4 || SUB_ == 8, "sub-chunks share a wave");
    static_assert(M % 4 == 0 && K % 2 == 0 && K >= 2 && K <= 8, "4-mic slots, two slots per wave, <= 28 pairs per slot");
    static constexpr int SUB = SUB_, NB = 32 / SUB_;    // frames per group, bins per workgroup
    static constexpr int KR = K - 1, P = M + KR, NP = P * (P + 1) / 2;
    static constexpr int NA = M / 4;                    // A slots per node
    static constexpr int WA = K * NA / 2, WB = K / 2;   // waves of A slots, waves of B slots (two slots per wave)
    static constexpr int NT = 64 * (WA + WB);
    static constexpr int MH = M / 2;                    // 16-byte granules (two mics) per bin
    static constexpr int NROW = K * NB * MH;            // granules of one frame of the tile
    static constexpr int NITEMS = SUB * NROW;           // ... of a group of SUB frames: 32 K MH whatever SUB is
    static constexpr int NL = (NITEMS + NT - 1) / NT;   // loader rounds
    static_assert(NITEMS % 64 == 0 && (NITEMS - (NL - 1) * NT) % 64 == 0, "whole waves in every loader round");
    static constexpr int NMASK = SUB * K * NB;          // mask values of a group: 32 K
    static constexpr int NMW = NMASK / 64;              // ... in wave-loads
    static_assert(NMASK % 64 == 0 && NMW <= WA + WB, "one mask wave-load per wave");
    static constexpr int NTAPP = (NROW + 63) / 64 * 64; // tap granules of an item, padded to whole wave-loads
    static constexpr int BPR = 16 / MH;                 // bins per 256-byte bank row of granules
    static constexpr int LH = SUB >= 4 ? 2 : (SUB == 2 ? 1 : 0);      // halving levels of the final reduction (lane bits 5, 4)
};

template <int M, int K, int SUB>
struct alignas(16) RoomRingS {
    using Gm = RoomGeomS<M, K, SUB>;
    float4 xs[DISCO_ROOM_DEPTH][Gm::NITEMS];           // granules, linear in the loader's item index
    float4 wt[2][Gm::NTAPP];                           // the step-1 filters of the current and of the next item, as granules
    float ms[DISCO_ROOM_DEPTH][Gm::NMW][64 + 8];       // one padded row per wave-load (the SUB sub-chunks of a wave read different rows)
    c32 zs[4][SUB][K + 1][Gm::NB];                     // z of four groups in flight; the + 1 row keeps the sub-chunks of a wave off each other's banks
};

// 64 lanes x 16 (4) bytes, lane i from gbase + off[i] (gbase wave-uniform: an SGPR pair, off a 32-bit byte offset: no per-lane
// 64-bit address arithmetic), to the LDS bytes [lds_wave, lds_wave + 1024 (256)) in lane order; lds_wave is wave-uniform.
// M0 (the LDS-DMA destination base) is compiler-reserved: saved, written and restored inside the one statement that reads it;
// the nops cover "SALU writes an SGPR -> VMEM reads it as base" (5 wait states) for a base the compiler has just formed.
// hipcc neither counts these loads nor waits for them (vm_wait_all below does).
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
// every vector-memory operation of this wave has completed (LDS-DMA loads have landed, stores are out)
__device__ __forceinline__ void vm_wait_all() {
#if defined(__clang__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// 2 x 2 transpose between two registers and lane bit BIT, then the sum: lanes with the bit clear are left with a + a' (a' = the partner
// lane's a), lanes with it set with b + b'.  One swap + one add per PAIR of registers, and the survivors are split between the halves.
template <int BIT>
__device__ __forceinline__ float lane_swap_add(float a, float b, int lane) {
#if defined(__clang__)
    (void)lane;
    static_assert(BIT == 32 || BIT == 16, "permlane swaps");
    if constexpr (BIT == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#else
    const float pa = __shfl_xor(a, BIT), pb = __shfl_xor(b, BIT);
    return (lane & BIT) ? pb + b : a + pa;
#endif
}
// the same for a float64 value: its two words are swapped separately
template <int BIT>
__device__ __forceinline__ double lane_swap_add64(double a, double b, int lane) {
#if defined(__clang__)
    (void)lane;
    static_assert(BIT == 32 || BIT == 16, "permlane swaps");
    const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
    unsigned l0, l1, h0, h1;
    if constexpr (BIT == 32) {
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
        l0 = lo[0], l1 = lo[1], h0 = hi[0], h1 = hi[1];
    } else {
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
        l0 = lo[0], l1 = lo[1], h0 = hi[0], h1 = hi[1];
    }
    return __longlong_as_double((long long)((unsigned long long)h0 << 32 | l0)) + __longlong_as_double((long long)((unsigned long long)h1 << 32 | l1));
#else
    const double pa = __shfl_xor(a, BIT), pb = __shfl_xor(b, BIT);
    return (lane & BIT) ? pb + b : a + pa;
#endif
}
template <int M, int K, int SUB, bool IS_A>
__device__ __forceinline__ void room_cov_dma_run(const RoomArgs& a, RoomRingS<M, K, SUB>& sh) {
    using Gm = RoomGeomS<M, K, SUB>;
    constexpr int KR = Gm::KR, P = Gm::P, NP = Gm::NP, NB = Gm::NB, NA = Gm::NA, WA = Gm::WA, NT = Gm::NT, MH = Gm::MH;
    constexpr int NROW = Gm::NROW, NITEMS = Gm::NITEMS, NL = Gm::NL, D = DISCO_ROOM_DEPTH, BPR = Gm::BPR, LH = Gm::LH;
    constexpr int LOGM = M == 8 ? 3 : 2;
    static_assert(D == 6, "two groups per iteration, two iterations ahead");
    static_assert(M == 8 || M == 4, "byte offsets of z are derived from those of X by a shift");
    const int T = a.T, F = a.F;
    const unsigned FM8 = (unsigned)(F * M * 8);
    const int tid = threadIdx.x, wid = wave_id(), lane = tid & 63;
    const int bin = lane & (NB - 1), sub = (lane / NB) & 1, sc = lane / (2 * NB);
    // An item is (room, tile of NB bins), all T frames: item I = room * tiles + tile.  (Frame chunks across workgroups are not needed
    // here: SUB sub-chunks live inside the workgroup, and rooms x tiles fills the chip for every batch but toy ones.)
    const int n_items = (int)(a.R * a.tiles);
    const int J = (T + 2 * SUB - 1) / (2 * SUB);         // iterations per item: two groups of SUB frames each

    // ---- loader lane: LDS position it = tid + r * NT  <->  sub-chunk lsc, node lk, bin lbin, granule lp = lp' ^ swz(lbin, lsc).
    // Bins beyond F - 1 fetch (and later re-write) bin F - 1, frames beyond T - 1 frame T - 1: every load and store is issued by every
    // lane of a whole wave, and what the surplus lanes write is the value that is there already.
    // Per round three registers: lxa = byte offset of the granule inside its room for bin 0 of the tile and frame 0 of the group
    // ((node, granule) part), lpk = lbin | lsc << 8 | (lp' == 0) << 16, lwt = the lane's tap granule in wt.
    unsigned lxa[NL];
    int lpk[NL], lwt[NL];
    bool lact[NL];
#pragma unroll
    for (int r = 0; r < NL; ++r) {
        const int it = tid + r * NT;
        lact[r] = wid * 64 + r * NT < NITEMS;          // whole waves: a scalar condition
        const int it_ = lact[r] ? it : 0;
        const int lsc = it_ / NROW, rem = it_ % NROW, lk = rem / (NB * MH), lbin = (rem / MH) % NB, lpp = rem % MH;
        const int lp = lpp ^ ((lbin / BPR + lsc) % MH);
        lxa[r] = (unsigned)(((lk * T) * F * M + 2 * lp) * 8);
        lpk[r] = lbin | lsc << 8 | (lpp == 0 ? 1 << 16 : 0);
        lwt[r] = (lk * NB + lbin) * MH + (lp ^ ((lbin / BPR) % MH));
    }
    const bool mact = wid < Gm::NMW;                   // wave w fetches mask row w of a group: value (msc, mk, mb) = tid
    const unsigned mxa = (unsigned)((((tid / NB) % K) * T) * F * 4);

    // group u of the item (room, f0) into ring slot `slot_`
    auto issue = [&](int room, int f0, int u, int slot_) {
        const int tb = u * SUB, tbc = tb < T ? tb : T - 1, nv1 = T - 1 - tbc;         // sub-chunks beyond nv1 repeat frame T - 1
        const int bmax = F - 1 - f0;                                                  // bins beyond bmax repeat bin F - 1
        const c32* gx = a.X + (((long long)room * K * T + tbc) * F + f0) * M;
#pragma unroll
        for (int r = 0; r < NL; ++r)
            if (lact[r]) {
                const unsigned off = lxa[r] + (unsigned)min(lpk[r] & 255, bmax) * (unsigned)(M * 8) + (unsigned)min((lpk[r] >> 8) & 255, nv1) * FM8;
                lds_dma16(gx, off, &sh.xs[slot_][wid * 64 + r * NT], lane);
            }
        if (mact) {
            const float* gm = a.mask + ((long long)room * K * T + tbc) * F + f0;
            const unsigned off = mxa + (unsigned)min(tid % NB, bmax) * 4u + (unsigned)min(tid / (K * NB), nv1) * (unsigned)(F * 4);
            lds_dma4(gm, off, &sh.ms[slot_][wid][0], lane);
        }
    };
    // the item's step-1 filters: tap granules at positions [0, NROW) of the item order (sub-chunk 0's swizzle), whole waves
    auto issue_taps = [&](int room, int f0, int buf) {
        const c32* gw = a.w + ((long long)room * K * F + f0) * M;
        const int bmax = F - 1 - f0;
        int tid_ = tid;                                  // opaque: the offsets are formed here, once per item, not carried through the loop
#if defined(__clang__)
        asm volatile("" : "+v"(tid_));
#endif
#pragma unroll
        for (int r = 0; r < (Gm::NTAPP + NT - 1) / NT; ++r)
            if (wid * 64 + r * NT < Gm::NTAPP) {
                const int tit = min(tid_ + r * NT, NROW - 1), tk = tit / (NB * MH), tbin = (tit / MH) % NB, tp = (tit % MH) ^ ((tbin / BPR) % MH);
                lds_dma16(gw, (unsigned)(((tk * F + min(tbin, bmax)) * M + 2 * tp) * 8), &sh.wt[buf][wid * 64 + r * NT], lane);
            }
    };
    // z of group u = w^H x of every (frame, node, bin) of the group from ring slot `slot_`: a lane's own granule and taps, then the
    // MH lanes of the bin; published in zs[zb] and stored
    auto form_z = [&](int room, int f0, int u, int slot_, int zb, int buf) {
        const int tb = u * SUB, tbc = tb < T ? tb : T - 1, nv1 = T - 1 - tbc;
        const int bmax = F - 1 - f0;
        char* gz = reinterpret_cast<char*>(a.z + ((long long)room * K * T + tbc) * F + f0);
#pragma unroll
        for (int r = 0; r < NL; ++r) {
            if (lact[r]) {
                const float4 q = sh.xs[slot_][tid + r * NT];
                const float4 wq = sh.wt[buf][lwt[r]];
                c32 p = cfma_conj(make_float2(wq.x, wq.y), make_float2(q.x, q.y), make_float2(0.f, 0.f));
                p = cfma_conj(make_float2(wq.z, wq.w), make_float2(q.z, q.w), p);
                static_assert(MH == 2 || MH == 4, "the lanes of a bin are (part of) a quad");
                p.x = quad_xor_add<1>(p.x);
                p.y = quad_xor_add<1>(p.y);
                if constexpr (MH == 4) {
                    p.x = quad_xor_add<2>(p.x);
                    p.y = quad_xor_add<2>(p.y);
                }
                if (lpk[r] >> 16) {
                    const int lbin = lpk[r] & 255, lsc = (lpk[r] >> 8) & 255;
                    // the node's place: lxa = (lk T F M + 2 lp) 8 with 2 lp < M  =>  (lxa >> log2 M) & ~7 = lk T F 8 (bytes of z), and
                    // lwt / (NB MH) = lk (the z rows of a sub-chunk are padded to K + 1)
                    (&sh.zs[zb][0][0][0])[(lsc * (K + 1) + lwt[r] / (NB * MH)) * NB + lbin] = p;
                    if (a.store_z) *reinterpret_cast<c32*>(gz + ((lxa[r] >> LOGM) & ~7u) + (unsigned)((min(lbin, bmax) + min(lsc, nv1) * F) * 8)) = p;
                }
            }
        }
    };

    // ---- the lane's slot
    // slot = 2 * (wave within its role) + sub.  A: k = slot / NA, h = slot % NA; B: k = slot.  kw = the wave's first node (a
    // SCALAR), dk = k - kw in {0, 1} (always 0 when the two slots of a wave share a node, NA even): the remote row jj of a lane
    // is node jj + (jj >= kw + dk), which differs between the slots only for jj == kw -- every other LDS offset of a remote
    // row is a scalar instead of a per-lane select.
    constexpr bool is_a = IS_A;
    constexpr int DKMAX = IS_A ? 1 / NA : 1;
    const int wr = is_a ? wid : wid - WA;
    const int kw = is_a ? (2 * wr) / NA : 2 * wr;
    const int dk = is_a ? sub / NA : sub;
    const int k = kw + dk;
    const int h = is_a ? sub % NA : 0;
    auto remote = [&](int jj) {                          // concatenate_signals order: jj -> node jj (jj < k), jj + 1 (jj >= k)
        int j = jj + (jj >= kw + DKMAX ? 1 : 0);
        if (DKMAX > 0 && jj >= kw && jj < kw + DKMAX) j = jj + (jj >= k ? 1 : 0);
        return j;
    };
    const int swz = (bin / BPR + sc) % MH;
    const int fim = (sc * K + k) * NB + bin;             // the lane's (sub-chunk, node, bin) in the item order of a group
    constexpr int NACC = IS_A ? 4 * KR : KR * (KR + 1) / 2;
    c32 acc_s[NACC], acc_n[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc_s[q] = acc_n[q] = make_float2(0.f, 0.f);

    auto fold = [&](int f0, int u, int slot_, int zb) {
        const float mkv = sh.ms[slot_][fim / 64][fim % 64];
        const bool ok = f0 + bin < F && u * SUB + sc < T;
        const float m = ok ? mkv : 0.f, mc = ok ? 1.f - mkv : 0.f;
        const float wa = m * m, wb = mc * mc;
        const c32(*zs)[NB] = sh.zs[zb][sc];
        if constexpr (is_a) {
            c32 x[4];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float4 q = sh.xs[slot_][fim * MH + ((2 * h + p) ^ swz)];
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

    // The SUB partial sums of every entry meet inside the wave IN FLOAT64 and leave as a (hi, lo) pair of float32 blocks (round 5, late).
    // A lane's float32 sums run over T / SUB frames; what the solve cannot take is their TOTAL rounded to float32 -- an entry of a pencil
    // with cond(Rnn) ~ 1e5 rounded at 6e-8 (profiles/r05_c5_accumulation.txt: the same sums combined in float64 and handed over unrounded
    // put C5's worst room at 1.3 - 3.2e-5 over the summation orders where the float32 tree put it at 2.3 - 8.4e-5).  So: lane bit 5, then
    // bit 4, by swap-and-add on float64 values (every level leaves a lane half of its entries: the upper half's totals go to the lanes with
    // the bit set), the third sub-chunk bit (lane bit 3) by a plain add; four entries at a time (32 registers; the float32 sums they came
    // from are dead by then).  A lane then stores whole 16-byte entries -- for A slots one row of KR contiguous ones -- twice: the float32
    // head of the total into block 0, the remainder into block 1 (the solvers add the blocks of an entry in float64), and clears its sums.
    // Entries are numbered q = i KR + jj (A: rows 4 h + i against the remote columns) / the upper triangle of the remote block
    // row by row (B): in both cases consecutive q of a row are consecutive in the packed triangle, and for B so is the whole run.
    auto finish = [&](int room, int f0) {
        static_assert(SUB == 8 && LH == 2, "three sub-chunk bits: lane bits 5, 4 (swaps) and 3 (add)");
        constexpr int NACCP = (NACC + 3) / 4 * 4, EPL = NACCP / 4;
        // Where the lane's entries go is recomputed here from an OPAQUE copy of the thread index: derived once before the loop, these
        // values would be carried through it in registers the fold needs (hipcc hoists them out and then spills them).
        int tid_ = tid;
#if defined(__clang__)
        asm volatile("" : "+v"(tid_));
#endif
        const int lane_ = tid_ & 63, bin_ = lane_ & (NB - 1), sub_ = (lane_ / NB) & 1;
        const int k_ = is_a ? (2 * wr + sub_) / NA : 2 * wr + sub_, h_ = is_a ? sub_ % NA : 0;
        const int g = ((lane_ >> 5) & 1) * 2 + ((lane_ >> 4) & 1);
        const bool writer = f0 + bin_ < F && (lane_ & 8) == 0;
        float4* o = a.part + ((((long long)room * K + k_) * 2) * F + min(f0 + bin_, F - 1)) * (long long)NP;
        const long long lo_block = (long long)F * NP;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            double tot[4];                              // Rss.re, Rss.im, Rnn.re, Rnn.im of the entry this lane is left with
#pragma unroll
            for (int comp = 0; comp < 4; ++comp) {
                double v[4];                            // entries e + {0, 1, 2, 3} EPL: level 1 pairs (0, 2) and (1, 3), level 2 their results
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int q = e + x * EPL;
                    const c32 src_ = q < NACC ? (comp < 2 ? acc_s[q < NACC ? q : 0] : acc_n[q < NACC ? q : 0]) : make_float2(0.f, 0.f);
                    v[x] = (double)((comp & 1) ? src_.y : src_.x);
                }
                const double l0 = lane_swap_add64<32>(v[0], v[2], lane), l1 = lane_swap_add64<32>(v[1], v[3], lane);
                double t = lane_swap_add64<16>(l0, l1, lane);
                t += __shfl_xor(t, 8);
                tot[comp] = t;
            }
            const float4 hi = make_float4((float)tot[0], (float)tot[1], (float)tot[2], (float)tot[3]);
            const float4 lo = make_float4((float)(tot[0] - (double)hi.x), (float)(tot[1] - (double)hi.y), (float)(tot[2] - (double)hi.z),
                                          (float)(tot[3] - (double)hi.w));
            const int q = g * EPL + e;
            long long at = -1;
            if constexpr (is_a) {
                const int r = 4 * h_ + q / KR, jj = q % KR;               // row of the node's own mic, remote column
                at = r * P - (r * (r - 1)) / 2 + (M - r) + jj;
            } else {
                if (q < NACC) at = tri_index<P>(M, M) + q;
            }
            if (writer && at >= 0) {
                o[at] = hi;
                o[lo_block + at] = lo;
            }
        }
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc_s[q] = acc_n[q] = make_float2(0.f, 0.f);
    };

    // ---- the pipeline over this workgroup's items blockIdx.x, + gridDim.x, ...  Three positions, one iteration apart: (Ii, ji) is
    // issued, (If, jf) formed, (Id, jd) folded; room / first bin of each are cached.  All of it is wave-uniform.
    // Which items: round n of the grid covers items [n G, (n + 1) G) (G workgroups), dealt so that the 8 NEIGHBOURING TILES of a room run
    // at the same time ON ONE XCD (one L2): a tile is 32 / SUB bins wide, so 8 / SUB ... 8 tiles share every 128-byte line of the masks and
    // of z.  The hardware deals consecutive workgroup ids round-robin to the 8 XCDs; with the plain map (item = id + n G) neighbouring tiles
    // sat on 8 different L2s, each of which fetched the shared mask lines for itself and wrote its 32-byte piece of a z line separately:
    // 29.1 GB read per C5 launch for 18.5 GB of inputs (profiles/r04_m_C5_pmc_traffic.json).  Needs G a multiple of 64; else the plain map.
    const int G = gridDim.x;
    auto item_of = [&](int n) {
        const int w = blockIdx.x;
        if (G % 64 != 0) return n * G + w;
        const int xcd = w % 8, s_ = w / 8, per = G / 64;          // s_: the workgroup's place on its XCD, per: blocks of 8 items per XCD and round
        return ((n * per + s_ / 8) * 8 + xcd) * 8 + s_ % 8;
    };
    int nr = 0;                                         // round of the issue position
    int Ii = item_of(0), ji = 0, ni = 0;                // item, iteration inside it, ordinal of the item (its tap buffer is n & 1)
    if (Ii >= n_items) return;
    int ri = Ii / a.tiles, fi = (Ii % a.tiles) * NB;
    int rf = ri, ff = fi, jf = 0, nf = 0;
    int rd = ri, fd = fi, jd = 0;
    bool vi = true, vf = true;                          // position still inside the workgroup's items
    auto advance_issue = [&]() {
        if (++ji == J) {
            Ii = item_of(++nr);
            vi = Ii < n_items;
            if (vi) {
                ri = Ii / a.tiles;
                fi = (Ii % a.tiles) * NB;
                ji = 0;
                ++ni;
            }
        }
    };
    // prologue: the first item's taps, the first two iterations' groups
    issue_taps(ri, fi, 0);
    issue(ri, fi, 0, 0);
    issue(ri, fi, 1, 1);
    advance_issue();
    rf = ri, ff = fi, jf = ji, nf = ni, vf = vi;        // the form position of the first loop iteration is what is issued second
    if (vi) {
        if (ji == 0) issue_taps(ri, fi, ni & 1);
        issue(ri, fi, 2 * ji, 2);
        issue(ri, fi, 2 * ji + 1, 3);
        advance_issue();
    }
    vm_wait_all();
    __syncthreads();                                    // groups 0 ... 3 and the taps are in place
    form_z(rd, fd, 0, 0, 0, 0);
    form_z(rd, fd, 1, 1, 1, 0);
    __syncthreads();
    int s0 = 0, zb0 = 0;                                // ring slot / z buffer of the fold position's first group
    int rp = rd, fp = fd;                               // the item whose last groups were folded in the previous iteration (pending)
    bool pending = false;
    // DISCO_ROOM_EXP (default 0): TIMING-ONLY builds with parts of the loop body removed -- bit 0 the folds, 1 the z formation, 2 the loads,
    // 3 the barrier, 4 the wait for the loads (raw s_barrier); the results are garbage (tools/gpu/mk_variant.sh roomexp1 "-DDISCO_ROOM_EXP=1" api_room_s8; profiles/r04_u_room_parts.txt)
    while (true) {
        if (vi && !(DISCO_ROOM_EXP & 4)) {
            if (ji == 0) issue_taps(ri, fi, ni & 1);    // (the form position left that buffer's item an iteration ago)
            issue(ri, fi, 2 * ji, (s0 + 4) % D);
            issue(ri, fi, 2 * ji + 1, (s0 + 5) % D);
        }
        if (pending) finish(rp, fp);
        if (vf && !(DISCO_ROOM_EXP & 2)) {
            form_z(rf, ff, 2 * jf, (s0 + 2) % D, zb0 ^ 2, nf & 1);
            form_z(rf, ff, 2 * jf + 1, (s0 + 3) % D, (zb0 ^ 2) + 1, nf & 1);
        }
        if (!(DISCO_ROOM_EXP & 1)) {
            fold(fd, 2 * jd, s0, zb0);
            fold(fd, 2 * jd + 1, (s0 + 1) % D, zb0 + 1);
        }
        pending = jd == J - 1;
        rp = rd, fp = fd;
#if DISCO_ROOM_EXP & 16
        __builtin_amdgcn_s_barrier();                  // (bit 4: the loads are never waited for, raw barrier: is the loop waiting for them?)
#else
        vm_wait_all();
        if (!(DISCO_ROOM_EXP & 8)) __syncthreads();
#endif
        if (!vf) break;                                 // the fold position was this workgroup's last iteration
        s0 = (s0 + 2) % D;
        zb0 ^= 2;
        rd = rf, fd = ff, jd = jf;
        rf = ri, ff = fi, jf = ji, nf = ni, vf = vi;
        if (vi) advance_issue();
    }
    finish(rp, fp);
}

template <int M, int K, int SUB>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__((RoomGeomS<M, K, SUB>::NT), DISCO_ROOM_WPE) void k_room_cov_dma(RoomArgs a) {
    __shared__ RoomRingS<M, K, SUB> sh;
    if (wave_id() < RoomGeomS<M, K, SUB>::WA) room_cov_dma_run<M, K, SUB, true>(a, sh);
    else room_cov_dma_run<M, K, SUB, false>(a, sh);
}

// ---- self-test of the room pass's asm primitives (disco_selftest_room; no reference counterpart) ------------------------------------
// lds_dma16 / lds_dma4 (global_load_lds with the M0 dance) + vm_wait_all against plain loads, lane_swap_add<32 / 16> and lane_swap_add64 (v_permlane32_swap /
// v_permlane16_swap) against their __shfl_xor statements: row q of out_hw through the instruction form, of out_ref through plain C++.
// Lane l of wave v fetches a PERMUTED granule (the kernel's own use: a lane's LDS place is fixed, the granule it fetches is chosen).
constexpr int ROOM_SELFTEST_OPS = 6;
static __global__ __launch_bounds__(64) void k_room_selftest(const float* __restrict__ src, long long n, float* __restrict__ out_hw, float* __restrict__ out_ref) {
    __shared__ float4 s16[64];
    __shared__ float s4[64];
    const int lane = threadIdx.x;
    const long long base = (long long)blockIdx.x * 256;          // 64 lanes x 4 floats per wave
    if (base + 256 > n) return;
    const int perm = (lane * 5 + 3) & 63;                        // a permutation of the lanes (5 is odd)
    lds_dma16(src + base, (unsigned)(perm * 16), &s16[0], lane);
    lds_dma4(src + base, (unsigned)(perm * 4), &s4[0], lane);
    vm_wait_all();
    __syncthreads();
    const float4 g16 = s16[lane];
    const float g4 = s4[lane];
    const float4 r16 = *reinterpret_cast<const float4*>(src + base + perm * 4);
    const float r4 = src[base + perm];
    const float a = src[base + lane], b = src[base + 64 + lane];
    const float sw32 = lane_swap_add<32>(a, b, lane), sw16 = lane_swap_add<16>(a, b, lane);
    const float pa32 = __shfl_xor(a, 32), pb32 = __shfl_xor(b, 32), pa16 = __shfl_xor(a, 16), pb16 = __shfl_xor(b, 16);
    const long long o = ((long long)blockIdx.x * 64 + lane) * ROOM_SELFTEST_OPS;
    out_hw[o + 0] = g16.x + 2.f * g16.y + 3.f * g16.z + 5.f * g16.w;
    out_ref[o + 0] = r16.x + 2.f * r16.y + 3.f * r16.z + 5.f * r16.w;
    out_hw[o + 1] = g4;
    out_ref[o + 1] = r4;
    out_hw[o + 2] = sw32;
    out_ref[o + 2] = (lane & 32) ? pb32 + b : a + pa32;
    out_hw[o + 3] = sw16;
    out_ref[o + 3] = (lane & 16) ? pb16 + b : a + pa16;
    // the float64 form (`finish`): operands whose low words matter, the result's two float32 halves' sum is compared (both words took part)
    const double da = (double)a * 1.0000001192092896 + (double)b * 1e-9, db = (double)b * 0.9999998807907104 - (double)a * 1e-9;
    const double dw32 = lane_swap_add64<32>(da, db, lane), dw16 = lane_swap_add64<16>(da, db, lane);
    const double qa32 = __shfl_xor(da, 32), qb32 = __shfl_xor(db, 32), qa16 = __shfl_xor(da, 16), qb16 = __shfl_xor(db, 16);
    const double dr32 = (lane & 32) ? qb32 + db : da + qa32, dr16 = (lane & 16) ? qb16 + db : da + qa16;
    out_hw[o + 4] = (float)(dw32 - (double)(float)dw32) * 1e6f + (float)dw32;
    out_ref[o + 4] = (float)(dr32 - (double)(float)dr32) * 1e6f + (float)dr32;
    out_hw[o + 5] = (float)(dw16 - (double)(float)dw16) * 1e6f + (float)dw16;
    out_ref[o + 5] = (float)(dr16 - (double)(float)dr16) * 1e6f + (float)dr16;
}

}  // namespace disco
