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
};

// Row j of both Hermitian matrices of the group's pencil from the covariance kernels' chunk partials (SolveSrc, k_solve.h), loaded
// by the GROUP: solve_load_row has every lane walk its own row of the packed triangle -- P entries x chunks dependent 16-byte loads
// per lane, 1 500 of the 8 000 instructions of a P = 15 solve and its longest stall.  Here the 16 lanes read the pencil's
// P (P + 1) / 2 entries linearly (lane, lane + 16, ...: 256 contiguous bytes per instruction and group, all chunks of an entry in flight
// together), combine the chunks in float64, round ONCE to float32 -- the same arithmetic, bit for bit, as solve_load_row -- and
// leave the triangle in the group's LDS block, from which lane j picks row j.  `stage`: P (P + 1) / 2 + M_loc (M_loc + 1) / 2 + 1 float4
// (SolveSrc's second source: the leading M_loc x M_loc block from the step-1 partial sums, as the room pass leaves them).
template <int P>
__device__ __forceinline__ void group_load_rows_part(const SolveSrc& src, long long pid, int j, bool live, float4* stage, c32* rs, c32* rn) {
    constexpr int NP = P * (P + 1) / 2, SLOTS = (NP + 15) / 16;
    const long long pidc = live ? pid : 0;              // a pencil that does not exist reads pencil 0 (its rows are replaced by the caller)
    const long long g = pidc / src.F;
    const int f = (int)(pidc % src.F);
    const int ML = src.M_loc, NPL = ML * (ML + 1) / 2;
    const double it = live ? (double)src.inv_T : 0.0;   // a pencil that does not exist: all zeros
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
            if (16 * s + j < NP) stage[16 * s + j] = make_float4((float)(acc[s][0] * it), (float)(acc[s][1] * it), (float)(acc[s][2] * it), (float)(acc[s][3] * it));
    }
    if (ML > 0) {                                       // leading M_loc x M_loc block: the step-1 partial sums (room pass, k_room.h)
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
            stage[NP + q] = make_float4((float)(a0 * it), (float)(a1 * it), (float)(a2 * it), (float)(a3 * it));
        }
    }
    if (j == 0) stage[NP + NPL] = make_float4(0.f, 0.f, 0.f, 0.f);         // what the lanes >= P read
    DISCO_GROUP_SYNC();
    // row j: entry (j, c) sits at upper-triangle coordinates (lo, hi) = (min, max); lower triangle = conj(upper); the diagonal is
    // real.  Branch-free on purpose (integer selects as masks, signs as factors): hipcc otherwise wraps every entry in its own
    // exec-masked block with a wait of its own.
    const auto isel = [](bool c, int x, int y) { return y + ((x - y) & -(int)c); };
    const int base_j = j * P - (j * (j - 1)) / 2 - j;                 // + c for c >= j
    const int base_l = NP + j * ML - (j * (j - 1)) / 2 - j;
#pragma unroll
    for (int c = 0; c < P; ++c) {
        const bool up = c >= j;
        int idx = isel(up, base_j + c, (c * P - (c * (c - 1)) / 2 - c) + j);
        if (ML > 0) {                                   // (uniform)
            const int il = isel(up, base_l + c, NP + (c * ML - (c * (c - 1)) / 2 - c) + j);
            idx = isel((up ? c : j) < ML, il, idx);
        }
        idx = isel(j < P, idx, NP + NPL);
        float4 v = stage[idx];
        DISCO_CONSUME(v.y);                             // all four words are read, whatever the selects below keep
        DISCO_CONSUME(v.w);
        const float sg = c == j ? 0.f : (up ? 1.f : -1.f);
        rs[c] = make_float2(v.x, v.y * sg);
        rn[c] = make_float2(v.z, v.w * sg);
    }
    DISCO_GROUP_SYNC();                                 // the block is reused (transposition of Y) once every lane has its row
}

template <int P, bool FROM_PART>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(DppSolveGeom<P>::THREADS, 2) void k_gevd_mwf_r1_dpp(SolveSrc src, long long n_prob, double mu,
                                                                                              c32* __restrict__ w_out, c32* __restrict__ t1_out) {
    using DG = DppSolveGeom<P>;
    constexpr int PW = DG::PW;
    __shared__ c64 s_M[DG::PROBS][DG::WORDS];
    const int j = threadIdx.x & 15;                     // row / column owned by this lane
    const int slot = threadIdx.x >> 4;
    const long long pid = (long long)blockIdx.x * DG::PROBS + slot;
    const bool live = pid < n_prob;
    const bool col = live && j < P;
    c64* Mm = s_M[slot];

    c32 rowA[P], rowB[P];                               // row j of Rxx / Rnn
    if constexpr (FROM_PART) {
        group_load_rows_part<P>(src, pid, j, live, reinterpret_cast<float4*>(Mm), rowA, rowB);
#pragma unroll
        for (int c = 0; c < P; ++c) rowB[c].x = (c == j && !live) ? 1.f : rowB[c].x;      // a pencil that does not exist: Rxx = 0, Rnn = I
    } else if (col) {
        solve_load_row<P, false>(src, pid, j, rowA, rowB);
    } else {
#pragma unroll
        for (int c = 0; c < P; ++c) {
            rowA[c] = make_float2(0.f, 0.f);
            rowB[c] = make_float2(c == j ? 1.f : 0.f, 0.f);
        }
    }
    DISCO_DPP_SETTLE();

    // ---- Cholesky of Rnn, right-looking, lane j on row j: a[c] is A[j][c] until column c is due, then L[j][c] (meaningful for
    // c < j; lane c's own a[c] ends as L[c][c] or 0, lanes above the diagonal carry values nobody reads).  Column c: the pivot is lane
    // c's a[c]; every lane scales its entry; the trailing update A[j][r] -= L[j][c] conj(L[r][c]) reads L[r][c] as lane r's a[c] and
    // touches a different accumulator with every multiply-add.  Pivot floor / zeroed column on breakdown: as group_cholesky_factor.
    c64 a[P];
#pragma unroll
    for (int c = 0; c < P; ++c) a[c] = make_double2((double)rowB[c].x, (double)rowB[c].y);
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
#pragma unroll
    for (int i = 0; i < P; ++i) g[i] = make_double2((double)rowA[i].x, -(double)rowA[i].y);
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
