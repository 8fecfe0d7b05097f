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
    static constexpr int WORDS = P * PW + 1;            // + one zero word (the "not part of column j" entries of the parked L)
};

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
    const int jr = j < P ? j : 0;
    c64* Mm = s_M[slot];

    c32 rowA[P], rowB[P];                               // row j of Rxx / Rnn
    if (col) {
        solve_load_row<P, FROM_PART>(src, pid, j, rowA, rowB);
    } else {
#pragma unroll
        for (int c = 0; c < P; ++c) {
            rowA[c] = make_float2(0.f, 0.f);
            rowB[c] = make_float2(c == j ? 1.f : 0.f, 0.f);
        }
    }
    DISCO_DPP_SETTLE();

    // ---- Cholesky of Rnn by rows.  l[c] of lane j: L[j][c] for c < j, (1 / L[j][j], L[j][j]) for c == j, never read for c > j.
    // Pivot floor and zeroed column on breakdown: as group_cholesky_factor.
    c64 l[P];
#pragma unroll
    for (int c = 0; c < P; ++c) l[c] = make_double2((double)rowB[c].x, (double)rowB[c].y);
    static_for<0, P>([&](auto C) {
        constexpr int c = decltype(C)::value;
        const BcRow<c, c + 1, P> lc(l);                 // row c of L so far (entries < c), and A[c][c]
        const double a_cc = lc.re(c);
        c64 s = l[c];
#pragma unroll
        for (int k = 0; k < c; ++k) lc.template fma<Z_SUB_OCS>(s, k, l[k]);
        const double d2 = BcReal(s.x).template get<c>();
        const double fl = fmax(1e-7 * a_cc, 1e-30);
        const bool brk = !(d2 >= fl);                   // also true for NaN
        const double d2c = brk ? fl : d2;
        const double rd = rsqrt64(d2c);
        const c64 below = zsel(brk, make_double2(0.0, 0.0), zscale(s, rd));
        l[c] = zsel(j == c, make_double2(rd, d2c * rd), below);
    });

    // ---- column j of Y = L^-1 Rxx   (Rxx[i][j] = conj(Rxx[j][i]): row j of Rxx)
    c64 g[P];
    static_for<0, P>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const BcRow<i, i + 1, P> li(l);
        c64 a = make_double2((double)rowA[i].x, -(double)rowA[i].y);
#pragma unroll
        for (int k = 0; k < i; ++k) li.template fma<Z_SUB_SO>(a, k, g[k]);
        g[i] = zscale(a, li.re(i));
    });
    // ---- the transposition: lane j needs conj(row j of Y) as the right-hand side of column j of C = L^-1 Y^H
    if (j < P) {
#pragma unroll
        for (int i = 0; i < P; ++i) Mm[i * PW + j] = g[i];
    }
    DISCO_GROUP_SYNC();
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const c64 t = Mm[jr * PW + i];
        g[i] = zsel(j < P, make_double2(t.x, -t.y), make_double2(0.0, 0.0));
    }
    DISCO_GROUP_SYNC();
    DISCO_DPP_SETTLE();
    static_for<0, P>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const BcRow<i, i + 1, P> li(l);
        c64 a = g[i];
#pragma unroll
        for (int k = 0; k < i; ++k) li.template fma<Z_SUB_SO>(a, k, g[k]);
        g[i] = zscale(a, li.re(i));
    });
    // ---- L leaves the registers: parked by rows, read back by columns for the back substitution
    if (j < P) {
#pragma unroll
        for (int k = 0; k < P; ++k) Mm[j * PW + k] = l[k];
    }
    if (j == 0) Mm[P * PW] = make_double2(0.0, 0.0);
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
        const double trl = BcReal(dj).template sum<P>();
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
            const BcRow<k, P, P> bk(g);                         // column k of B
#pragma unroll
            for (int i = 0; i < P; ++i) bk.template fma<Z_ADD_SO>(nn[i], i, g[k]);
        });
        c64 tc = make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < P; ++i)
            if (i == j) tc = nn[i];
        tc.x = BcReal(tc.x).template sum<P>();
        tc.y = BcReal(tc.y).template sum<P>();
        const double den = tc.x * tc.x + tc.y * tc.y;
        const double rden = den > 0.0 ? rcp64(den) : 0.0;
        const c64 itau = make_double2(tc.x * rden, -tc.y * rden);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            g[i] = zsel(done, g[i], zmul(nn[i], itau));                           // a finished pencil idles until the slowest of its wave is through
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
        c64 u = make_double2(0.0, 0.0);
        static_for<0, P>([&](auto I) { bv.template fma<decltype(I)::value, Z_ADD_COS>(u, g[decltype(I)::value]); });
        v = zsel(have, u, v);
    }
    if (DISCO_POWER_STEPS > 0) {
        const double n2 = BcReal(v.x * v.x + v.y * v.y).template sum<P>();
        const double rn = rsqrt64(n2);
        v = zsel(have, zscale(v, rn), v);
    }

    // ---- q = L^-H v0, distributed: step k finishes q_k in lane k, the lanes above it (i < k) subtract conj(L[k][i]) q_k
    c64 q;
    {
        c64 lcol[P];                                           // column j of L below the diagonal; zero elsewhere
#pragma unroll
        for (int k = 0; k < P; ++k) lcol[k] = Mm[k > j ? k * PW + j : P * PW];
        const double rdj = Mm[jr * PW + jr].x;                // 1 / L[j][j]
        c64 acc = v;
        static_for<0, P - 1>([&](auto KK) {
            constexpr int k = P - 1 - decltype(KK)::value;     // P-1 ... 1
            const BcVec qk(zscale(acc, rdj));
            qk.template fma<k, Z_SUB_COS>(acc, lcol[k]);
        });
        q = zscale(acc, rdj);
    }
    const double l00 = Mm[0].y;
    // ---- d0 = q^H Rxx q: lane j forms (Rxx q)_j from its row of Rxx
    const BcVec bq(q);
    double d0;
    {
        c64 sj = make_double2(0.0, 0.0);
        static_for<0, P>([&](auto Cc) {
            constexpr int c = decltype(Cc)::value;
            bq.template fma<c, Z_ADD_SO>(sj, make_double2((double)rowA[c].x, (double)rowA[c].y));
        });
        const double e = j < P ? q.x * sj.x + q.y * sj.y : 0.0;   // Re(conj(q_j) (Rxx q)_j)
        const double es = BcReal(e).template sum<P>();
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
