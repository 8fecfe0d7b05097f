// Rank-1 GEVD-MWF solve for 9 <= P <= 16 with the matrices in REGISTERS and every "entry of another lane's column" read through
// DPP row_newbcast (dpp64.h) instead of LDS -- same algorithm, same formulas and the same breakdown rules as gevd_solve_group
// (k_solve.h; intern_filter(..., type='gevd', rank=1), disco_theque/se_utils/internal_formulas.py:56-73).
//
// Why: the LDS form of the P = 15 solve (C5: 820 800 pencils per launch) spent 7.2 ms per launch with 2 039 M VALU and 392 M LDS
// instructions -- one broadcast ds_read_b128 per complex multiply-add, i.e. ~5 ms of LDS time at 128 B/clk/CU against ~1.9 ms of
// float64 multiply-adds (profiles/r03_f_C5_pmc_alu.json).  A row of 16 lanes is exactly one pencil here (lane j = row / column j),
// and everything the solve reads from other lanes is "entry i of lane k's array" with i and k known at compile time: that is
// `v_fmac_f64_dpp ... row_newbcast:k` on the register that holds entry i, at the issue rate of a plain v_fma_f64.
//
//   Cholesky of Rnn      lane j holds row j of the lower triangle; column c: s_j = A[j][c] - sum_{k<c} L[j][k] conj(L[c][k]),
//                        L[c][k] = lane c's entry k; the pivot is lane c's s, broadcast
//   Y = L^-1 Rxx         lane j holds column j of Y; row i: y[i] = (Rxx[i][j] - sum_{k<i} L[i][k] y[k]) / L[i][i], L[i][k] = lane i's entry k
//   C = L^-1 Y^H         needs ROW j of Y in lane j: the one transposition, through LDS (P writes + P reads per lane); then as above
//   squarings            (B^2)[i][j] = sum_k B[i][k] B[k][j]: B[i][k] = lane k's entry i, B[k][j] = own entry k; 4 P^2 v_fmac_f64_dpp each
//   power steps, back substitution, Rayleigh quotient: the VECTORS stay distributed (lane j holds component j): one
//                        complex multiply-add per lane and step instead of P (the LDS form computed all of q in every lane).
//                        The back substitution needs column j of L in lane j -- L is parked in LDS by rows before the
//                        squarings (its registers are needed there) and read back by columns after them: the second transposition.
// Per pencil and lane: ~3 P^2/2 + S (P^2 + ~2 P) + ~8 P complex multiply-adds for S squarings; no LDS inside any O(P^2) loop.
#pragma once
#include "dpp64.h"
#include "k_solve.h"

namespace disco {

template <int P>
struct DppSolveGeom {
    static_assert(P >= 9 && P <= 16, "one pencil per 16-lane row");
    static constexpr int THREADS = 64, PROBS = 4;
    static constexpr int PW = P | 1;                    // odd row pitch (16-byte words): transposed reads spread over the banks
    static constexpr int WORDS = P * PW + PW;           // + one row of zeros (lanes >= P; the "not part of column j" entries of the parked L)
    // the same block first holds the staged pencil (group_stage_part): (hi, lo) of P (P + 1) / 2 entries + 1 float4, 16 bytes like a c64
    static constexpr int BLOCK = WORDS > P * (P + 1) + 1 ? WORDS : P * (P + 1) + 1;
};

// The group's pencil from the covariance kernels' chunk partials (SolveSrc, k_solve.h), fetched by the GROUP into its LDS block:
// solve_load_row has every lane walk its own row of the packed triangle -- P entries x chunks dependent 16-byte loads per lane, 1 500 of
// the 8 000 instructions of a P = 15 solve and its longest stall.  Here the 16 lanes read the pencil's P (P + 1) / 2 entries linearly
// (lane, lane + 16, ...: 256 contiguous bytes per instruction and group, all chunks of an entry in flight together) and combine the
// blocks of an entry in float64.
//
// Round 5: nothing is rounded at the solver's door any more.  (Round 4 measured what that rounding costs: the float64 oracle itself moves
// by 3.7e-5 on C5's worst room when its exact covariances are merely rounded to complex64, profiles/r04_c5_accumulation.txt.)
//   * The sums are handed over UNSCALED -- w and t1 do not change when Rxx and Rnn are scaled together (d0 is their ratio, q scales with
//     1 / L, t1 = q L00 conj(v0[0])) -- the multiplication by 1 / T cost every entry a rounding.
//   * EVERY entry is staged as a (hi, lo) pair of float4: the float64 total of its blocks, split.  The leading M_loc x M_loc block is the
//     step-1 statistics, which the wide shapes accumulate in float64 and store as (hi, lo) pairs of blocks (k_cov_loc_f64); the other
//     entries are the room pass's totals, formed in float64 from its float32 sub-chunk sums and stored the same way (k_room.h, `finish`).
// `stage`: hi[NP], lo[NP], one zero word = 2 NP + 1 float4.
template <int P>
__device__ __forceinline__ void group_stage_part(const SolveSrc& src, long long pid, int j, bool live, float4* stage) {
    constexpr int NP = P * (P + 1) / 2, SLOTS = (NP + 15) / 16;
    const long long pidc = live ? pid : 0;              // a pencil that does not exist reads pencil 0 (and stages zeros)
    const long long g = pidc / src.F;
    const int f = (int)(pidc % src.F);
    const int ML = src.M_loc, NPL = ML * (ML + 1) / 2;
    const double it = live ? 1.0 : 0.0;
    auto put = [&](int at, double a0, double a1, double a2, double a3) {
        a0 *= it, a1 *= it, a2 *= it, a3 *= it;
        const float4 hi = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
        stage[at] = hi;
        stage[NP + at] = make_float4((float)(a0 - (double)hi.x), (float)(a1 - (double)hi.y), (float)(a2 - (double)hi.z), (float)(a3 - (double)hi.w));
    };
    {
        const float4* pb = src.part + ((g * src.chunks) * src.F + f) * (long long)NP;
        const long long cs = (long long)src.F * NP;
        int qs[SLOTS];                                  // the last slot is ragged: its spare lanes re-read the last entry (and store nothing)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) qs[s] = 16 * s + j < NP ? 16 * s + j : NP - 1;
        double acc[SLOTS][4];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) acc[s][0] = acc[s][1] = acc[s][2] = acc[s][3] = 0.0;
        for (int ch = 0; ch < src.chunks; ++ch) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const float4 v = pb[ch * cs + qs[s]];
                acc[s][0] += (double)v.x;
                acc[s][1] += (double)v.y;
                acc[s][2] += (double)v.z;
                acc[s][3] += (double)v.w;
            }
        }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (16 * s + j < NP) put(16 * s + j, acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
    }
    if (ML > 0) {                                       // leading M_loc x M_loc block: the step-1 partial sums OVER what the main blocks held there
        DISCO_GROUP_SYNC();                             // (program order between the lanes' stores to the same words)
        const float4* pl = src.part_loc + ((g * src.chunks_loc) * src.F + f) * (long long)NPL;
        const long long csl = (long long)src.F * NPL;
        for (int q = j; q < NPL; q += 16) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            for (int ch = 0; ch < src.chunks_loc; ++ch) {
                const float4 v = pl[ch * csl + q];
                a0 += (double)v.x;
                a1 += (double)v.y;
                a2 += (double)v.z;
                a3 += (double)v.w;
            }
            int r = 0, q0 = 0;                          // q = tri_ML(r, c): the row whose run [q0, q0 + ML - r) holds q
            while (q0 + ML - r <= q) q0 += ML - r, ++r;
            const int c = r + (q - q0);
            put(r * P - (r * (r - 1)) / 2 + (c - r), a0, a1, a2, a3);
        }
    }
    if (j == 0) stage[2 * NP] = make_float4(0.f, 0.f, 0.f, 0.f);           // what the lanes >= P read
    DISCO_GROUP_SYNC();
}
// Row j of the staged pencil in ONE pass over its entries: Rnn's row in float64 (`b`: the Cholesky takes it at once), Rxx's row as its
// float32 head `a_hi` (what crosses the squarings for the Rayleigh quotient: d0 only enters through d0 / (d0 + mu)) plus the `lo` halves
// `a_lo`, from which the caller forms the float64 row when the whitening begins.  Entry (j, c) sits at upper-triangle coordinates
// (lo, hi) = (min, max); lower triangle = conj(upper); the diagonal is real.  Branch-free on purpose (integer selects as masks, signs as
// factors): hipcc otherwise wraps every entry in its own exec-masked block with a wait of its own.
// (P = 16 has no register left for 16 `lo` halves of Rxx -- the kernel may not spill, build.py -- and no room-pass shape: it keeps those of
// the leading 8 columns, the widest step-1 block; Rnn's row takes every `lo` either way.)
template <int P>
constexpr int dpp_lo_cols() { return P <= 15 ? P : 8; }
template <int P>
__device__ __forceinline__ void group_stage_rows(const float4* stage, int j, c64* b, c32* a_hi, c32* a_lo) {
    constexpr int NP = P * (P + 1) / 2, ZERO = 2 * NP;
    const auto isel = [](bool c, int x, int y) { return y + ((x - y) & -(int)c); };
    const int base_j = j * P - (j * (j - 1)) / 2 - j;                 // + c for c >= j
#pragma unroll
    for (int c = 0; c < P; ++c) {
        const bool up = c >= j;
        const int idx = isel(up, base_j + c, (c * P - (c * (c - 1)) / 2 - c) + j);
        float4 v = stage[isel(j < P, idx, ZERO)];
        float4 l = stage[isel(j < P, idx + NP, ZERO)];
        DISCO_CONSUME(v.y);                             // all four words are read, whatever the selects below keep
        DISCO_CONSUME(v.w);
        DISCO_CONSUME(l.y);
        DISCO_CONSUME(l.w);
        const float sg = c == j ? 0.f : (up ? 1.f : -1.f);
        a_hi[c] = make_float2(v.x, v.y * sg);
        if (c < dpp_lo_cols<P>()) a_lo[c] = make_float2(l.x, l.y * sg);
        b[c] = make_double2((double)v.z + (double)l.z, (double)(v.w * sg) + (double)(l.w * sg));
    }
    DISCO_GROUP_SYNC();                                 // the block is reused (transposition of Y) once every lane has its rows
}

template <int P, bool FROM_PART>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(DppSolveGeom<P>::THREADS, 2) void k_gevd_mwf_r1_dpp(SolveSrc src, long long n_prob, double mu,
                                                                                              c32* __restrict__ w_out, c32* __restrict__ t1_out) {
    using DG = DppSolveGeom<P>;
    constexpr int PW = DG::PW;
    __shared__ c64 s_M[DG::PROBS][DG::BLOCK];
    const int j = threadIdx.x & 15;                     // row / column owned by this lane
    const int slot = threadIdx.x >> 4;
    const long long pid = (long long)blockIdx.x * DG::PROBS + slot;
    const bool live = pid < n_prob;
    const bool col = live && j < P;
    c64* Mm = s_M[slot];

    c32 rowA[P];                                        // row j of Rxx in float32: all that crosses the squarings (the Rayleigh quotient's)
    c64 a[P];                                           // row j of Rnn, then of its Cholesky factor
    c32 rowA_lo[dpp_lo_cols<P>()];                      // FROM_PART: the `lo` halves of its entries
    if constexpr (FROM_PART) {
        group_stage_part<P>(src, pid, j, live, reinterpret_cast<float4*>(Mm));
        group_stage_rows<P>(reinterpret_cast<const float4*>(Mm), j, a, rowA, rowA_lo);
#pragma unroll
        for (int c = 0; c < P; ++c) a[c].x = (c == j && !live) ? 1.0 : a[c].x;           // a pencil that does not exist: Rxx = 0, Rnn = I
    } else {
        c32 rowB[P];
        if (col) {
            solve_load_row<P, false>(src, pid, j, rowA, rowB);
        } else {
#pragma unroll
            for (int c = 0; c < P; ++c) {
                rowA[c] = make_float2(0.f, 0.f);
                rowB[c] = make_float2(c == j ? 1.f : 0.f, 0.f);
            }
        }
#pragma unroll
        for (int c = 0; c < P; ++c) a[c] = make_double2((double)rowB[c].x, (double)rowB[c].y);
    }
    DISCO_DPP_SETTLE();

    // ---- Cholesky of Rnn, right-looking, lane j on row j: a[c] is A[j][c] until column c is due, then L[j][c] (meaningful for
    // c < j; lane c's own a[c] ends as L[c][c] or 0, lanes above the diagonal carry values nobody reads).  Column c: the pivot is lane
    // c's a[c]; every lane scales its entry; the trailing update A[j][r] -= L[j][c] conj(L[r][c]) reads L[r][c] as lane r's a[c] and
    // touches a different accumulator with every multiply-add.  Pivot floor / zeroed column on breakdown: as group_cholesky_factor.
    double rdj = 1.0, dgj = 1.0;                        // 1 / L[j][j] and L[j][j] of the lane's own row
    {
        double dorig = 0.0;                             // Rnn[j][j]
#pragma unroll
        for (int c = 0; c < P; ++c)
            if (c == j) dorig = a[c].x;
        const BcReal dv(dorig);
        static_for<0, P>([&](auto C) {
            constexpr int c = decltype(C)::value;
            const double d2 = BcReal(a[c].x).template get<c>();
            const double fl = fmax(1e-7 * dv.template get<c>(), 1e-30);
            const bool brk = !(d2 >= fl);               // also true for NaN
            const double d2c = brk ? fl : d2;
            const double rd = rsqrt64(d2c);
            rdj = j == c ? rd : rdj;
            dgj = j == c ? d2c * rd : dgj;
            a[c] = zscale(a[c], brk ? 0.0 : rd);
            if constexpr (c < P - 1) {
                const BcVec lc(a[c]);                   // column c of L, one entry per lane
                static_for<c + 1, P>([&](auto R) { lc.template fma<decltype(R)::value, Z_SUB_OCS>(a[decltype(R)::value], a[c]); });
            }
        });
    }
    const BcReal rdv(rdj);

    // ---- column j of Y = L^-1 Rxx (Rxx[i][j] = conj(Rxx[j][i]): row j of Rxx), right-looking: row i is finished by its own
    // diagonal, then every later row r subtracts L[r][i] y[i] -- L[r][i] is lane r's a[i]
    c64 g[P];
    if constexpr (FROM_PART) {
#pragma unroll
        for (int i = 0; i < P; ++i) {                   // row j of Rxx in float64 from (hi, lo)
            constexpr int LC = dpp_lo_cols<P>();
            const double lx = i < LC ? (double)rowA_lo[i < LC ? i : 0].x : 0.0, ly = i < LC ? (double)rowA_lo[i < LC ? i : 0].y : 0.0;
            g[i] = make_double2((double)rowA[i].x + lx, -((double)rowA[i].y + ly));
        }
    } else {
#pragma unroll
        for (int i = 0; i < P; ++i) g[i] = make_double2((double)rowA[i].x, -(double)rowA[i].y);
    }
    auto forward = [&]() {
        static_for<0, P>([&](auto I) {
            constexpr int i = decltype(I)::value;
            g[i] = zscale(g[i], rdv.template get<i>());
            if constexpr (i < P - 1) {
                const BcVec li(a[i]);
                static_for<i + 1, P>([&](auto R) { li.template fma<decltype(R)::value, Z_SUB_SO>(g[decltype(R)::value], g[i]); });
            }
        });
    };
    forward();
    // ---- the transposition: lane j needs conj(row j of Y) as the right-hand side of column j of C = L^-1 Y^H (lanes >= P: the zero row)
    if (j < P) {
#pragma unroll
        for (int i = 0; i < P; ++i) Mm[i * PW + j] = g[i];
    }
    if (j < PW) Mm[P * PW + j] = make_double2(0.0, 0.0);
    DISCO_GROUP_SYNC();
    {
        const c64* rowp = Mm + (j < P ? j : P) * PW;
#pragma unroll
        for (int i = 0; i < P; ++i) g[i] = make_double2(rowp[i].x, -rowp[i].y);
    }
    DISCO_GROUP_SYNC();
    DISCO_DPP_SETTLE();
    forward();
    // ---- L leaves the registers: parked by rows, read back by columns for the back substitution
    if (j < P) {
#pragma unroll
        for (int k = 0; k < P; ++k) Mm[j * PW + k] = a[k];
    }
    DISCO_GROUP_SYNC();
    DISCO_DPP_SETTLE();

    // ---- dominant eigenpair of C by repeated squaring of B = C / tr C (see gevd_solve_group / group_dominant for the reasoning:
    // complex trace as the normaliser, DISCO_SQUARING_DONE, the power-step finish, the wave-uniform exit)
    bool done;
    {
        double dj = 0.0;
#pragma unroll
        for (int i = 0; i < P; ++i)
            if (i == j) dj = g[i].x;
        const double trl = BcReal(dj).sum();
        const bool ok = trl > 0.0 && trl < 1.7e308;            // false for NaN / inf / the zero matrix (Rxx = 0)
        const double rt = ok ? rcp64(trl) : 0.0;
#pragma unroll
        for (int i = 0; i < P; ++i) g[i] = zsel(ok, zscale(g[i], rt), make_double2(0.0, 0.0));
        done = !ok;
    }
    for (int it = 0; it < DISCO_SQUARINGS_MAX; ++it) {
        DISCO_DPP_SETTLE();
        c64 nn[P];
#pragma unroll
        for (int i = 0; i < P; ++i) nn[i] = make_double2(0.0, 0.0);
        static_for<0, P>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const BcRow<k, P, P, (k > 0)> bk(g);                // column k of B (pinned and settled once, by the view of column 0)
#pragma unroll
            for (int i = 0; i < P; ++i) bk.template fma<Z_ADD_SO>(nn[i], i, g[k]);
        });
        c64 tc = make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < P; ++i)
            if (i == j) tc = nn[i];
        tc.x = BcReal(tc.x).sum();
        tc.y = BcReal(tc.y).sum();
        const double den = tc.x * tc.x + tc.y * tc.y;
        const double rden = den > 0.0 ? rcp64(den) : 0.0;
        const c64 itau = make_double2(tc.x * rden, -tc.y * rden);
        if (!done) {                                           // a finished pencil idles until the slowest of its wave is through
#pragma unroll
            for (int i = 0; i < P; ++i) g[i] = zmul(nn[i], itau);
        }
        done = done || (1.0 - tc.x < DISCO_SQUARING_DONE) || !(den > 0.0);
        if (!__any(!done)) break;
    }
    DISCO_DPP_SETTLE();

    // ---- B ~ v0 v0^H: the longest column (lowest index among equals); lane j needs ITS component of it, B[j][bj] = conj(B[bj][j])
    c64 v;
    bool have;
    {
        double nrm = 0.0;
#pragma unroll
        for (int i = 0; i < P; ++i) nrm += g[i].x * g[i].x + g[i].y * g[i].y;
        const BcReal nv(nrm);
        double best = nv.template get<0>();
        int bj = 0;
        static_for<1, P>([&](auto K) {
            const double nk = nv.template get<decltype(K)::value>();
            const bool up = nk > best;
            best = up ? nk : best;
            bj = up ? decltype(K)::value : bj;
        });
        have = best > 0.0;
        const double rb = have ? rsqrt64(best) : 0.0;
        c64 vj = make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < P; ++i)
            if (i == bj) vj = make_double2(g[i].x, -g[i].y);
        v = zsel(have, zscale(vj, rb), make_double2(j == 0 ? 1.0 : 0.0, 0.0));
    }
    // ---- power steps on the kept square: (B v)_j = sum_i conj(B[i][j]) v_i
#pragma unroll 1
    for (int s = 0; s < DISCO_POWER_STEPS; ++s) {
        const BcVec bv(v);
        c64 u[2] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0)};      // two chains: consecutive statements touch different registers
        static_for<0, P>([&](auto I) { bv.template fma<decltype(I)::value, Z_ADD_COS>(u[decltype(I)::value & 1], g[decltype(I)::value]); });
        v = zsel(have, zadd(u[0], u[1]), v);
    }
    if (DISCO_POWER_STEPS > 0) {
        const double n2 = BcReal(v.x * v.x + v.y * v.y).sum();
        const double rn = rsqrt64(n2);
        v = zsel(have, zscale(v, rn), v);
    }

    // ---- q = L^-H v0, distributed: step k finishes q_k in lane k, the lanes above it (i < k) subtract conj(L[k][i]) q_k
    c64 q;
    {
        c64 lcol[P];                                           // column j of L below the diagonal; zero elsewhere
#pragma unroll
        for (int k = 0; k < P; ++k) lcol[k] = Mm[k > j ? k * PW + j : P * PW];
        c64 acc = v;
        static_for<0, P - 1>([&](auto KK) {
            constexpr int k = P - 1 - decltype(KK)::value;     // P-1 ... 1
            const BcVec qk(zscale(acc, rdj));
            qk.template fma<k, Z_SUB_COS>(acc, lcol[k]);
        });
        q = zscale(acc, rdj);
    }
    const double l00 = BcReal(dgj).template get<0>();
    // ---- d0 = q^H Rxx q: lane j forms (Rxx q)_j from its row of Rxx
    const BcVec bq(q);
    double d0;
    {
        // the row of Rxx crosses the squarings as float32: without this hipcc keeps the float64 conversions made for the first
        // substitution alive instead (4 P registers instead of 2 P; spills at P = 16)
#pragma unroll
        for (int c = 0; c < P; ++c) {
            DISCO_CONSUME(rowA[c].x);
            DISCO_CONSUME(rowA[c].y);
        }
        c64 sj[2] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
        static_for<0, P>([&](auto Cc) {
            constexpr int c = decltype(Cc)::value;
            bq.template fma<c, Z_ADD_SO>(sj[c & 1], make_double2((double)rowA[c].x, (double)rowA[c].y));
        });
        const c64 rq = zadd(sj[0], sj[1]);
        const double e = j < P ? q.x * rq.x + q.y * rq.y : 0.0;   // Re(conj(q_j) (Rxx q)_j)
        const double es = BcReal(e).sum();
        d0 = have ? es : 0.0;
    }
    const c64 v00 = BcVec(v).template get<0>();
    const double dcl = fmin(fmax(d0, SOLVE_EPS), SOLVE_ETA);
    const c64 gsc = make_double2(l00 * v00.x, -l00 * v00.y);   // L[0,0] conj(v0[0]) = (Q^-1)[0,0]
    const double gain = dcl / (dcl + mu);
    const c64 t1 = zmul(q, gsc);
    if (col) {
        if (t1_out) t1_out[pid * P + j] = make_float2((float)t1.x, (float)t1.y);
        w_out[pid * P + j] = make_float2((float)(t1.x * gain), (float)(t1.y * gain));
    }
}

}  // namespace disco
