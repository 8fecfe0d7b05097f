// Rank-1 GEVD-MWF for SMALL pencils (P <= 4: every step-1 solve of a node with up to 4 microphones), one THREAD per
// problem.  Same algorithm and the same float64 arithmetic as k_solve.h (Cholesky whitening, dominant eigenpair by
// repeated squaring, back substitution; internal_formulas.py:56-73) -- but at this size a group of 4 lanes spends more
// instructions on hand-offs (LDS round trips, fences, shuffle reductions) than on arithmetic, so everything lives in the
// registers of one thread: no LDS or shuffles in the arithmetic (only the wave-cooperative fetch of the partial sums goes
// through LDS), 64 problems per wave instead of 16.  The Hermitian matrices are held as (real diagonal, strict lower
// triangle); B^2 is formed on that half only, which also keeps B exactly Hermitian.
#pragma once
#include "k_solve.h"

namespace disco {

template <int P>
struct HermReg {                       // Hermitian P x P: d[i] = A[i][i] (real), o[i(i-1)/2 + k] = A[i][k], i > k
    double d[P];
    c64 o[P > 1 ? P * (P - 1) / 2 : 1];
    __device__ __forceinline__ c64 at(int i, int k) const {       // i, k compile-time after unrolling
        if (i == k) return make_double2(d[i], 0.0);
        if (i > k) return o[i * (i - 1) / 2 + k];
        const c64 v = o[k * (k - 1) / 2 + i];
        return make_double2(v.x, -v.y);
    }
};

// Rxx, Rnn given as (diag, strict lower triangle) in float32.  w, t1: this problem's P filter entries.
// Contains a wave-wide vote: every lane of the wave must call it (dead lanes pass Rxx = 0, Rnn = I).
template <int P>
__device__ __forceinline__ void gevd_solve_thread(const float* a_d, const c32* a_o, const float* b_d, const c32* b_o, const double mu,
                                                  c64* w, c64* t1) {
    constexpr int NO = P > 1 ? P * (P - 1) / 2 : 1;
    auto lo = [](int i, int k) { return i * (i - 1) / 2 + k; };
    auto A = [&](int i, int k) -> c64 {        // Rxx[i][k]
        if (i == k) return make_double2((double)a_d[i], 0.0);
        if (i > k) return make_double2((double)a_o[i * (i - 1) / 2 + k].x, (double)a_o[i * (i - 1) / 2 + k].y);
        return make_double2((double)a_o[k * (k - 1) / 2 + i].x, -(double)a_o[k * (k - 1) / 2 + i].y);
    };
    // ---- Cholesky Rnn = L L^H with the pivot floor of k_solve.h (numerically singular noise statistics)
    double Ld[P], rL[P];
    c64 Lo[NO];
#pragma unroll
    for (int c = 0; c < P; ++c) {
        const double a_cc = (double)b_d[c];
        double d2 = a_cc;
#pragma unroll
        for (int k = 0; k < c; ++k) d2 -= Lo[lo(c, k)].x * Lo[lo(c, k)].x + Lo[lo(c, k)].y * Lo[lo(c, k)].y;
        const double fl = fmax(1e-7 * a_cc, 1e-30);
        const bool brk = !(d2 >= fl);
        const double d2c = brk ? fl : d2;
        const double rd = rsqrt64(d2c);
        rL[c] = rd;
        Ld[c] = d2c * rd;
#pragma unroll
        for (int i = c + 1; i < P; ++i) {
            c64 s = make_double2((double)b_o[lo(i, c)].x, (double)b_o[lo(i, c)].y);
#pragma unroll
            for (int k = 0; k < c; ++k) s = zsub(s, zmulc(Lo[lo(i, k)], Lo[lo(c, k)]));
            Lo[lo(i, c)] = brk ? make_double2(0.0, 0.0) : zscale(s, rd);
        }
    }
    // ---- C = L^-1 Rxx L^-H, column by column; only the lower triangle is kept
    HermReg<P> B;
    double tr = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        // column j of C directly: C[:, j] = L^-1 (Rxx (L^-H e_j)); u = L^-H e_j has entries 0..j only
        c64 yj[P], u[P];
#pragma unroll
        for (int i = P - 1; i >= 0; --i) {
            c64 a = make_double2(i == j ? 1.0 : 0.0, 0.0);
#pragma unroll
            for (int k = i + 1; k < P; ++k)
                if (k <= j) a = zsub(a, zmul(make_double2(Lo[lo(k, i)].x, -Lo[lo(k, i)].y), u[k]));
            u[i] = i <= j ? zscale(a, rL[i]) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int i = 0; i < P; ++i) {            // yj = Rxx u
            c64 a = make_double2(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < P; ++k)
                if (k <= j) a = zadd(a, zmul(A(i, k), u[k]));
            yj[i] = a;
        }
#pragma unroll
        for (int i = 0; i < P; ++i) {            // c = L^-1 yj (in place)
            c64 a = yj[i];
#pragma unroll
            for (int k = 0; k < i; ++k) a = zsub(a, zmul(Lo[lo(i, k)], yj[k]));
            yj[i] = zscale(a, rL[i]);
        }
        B.d[j] = yj[j].x;
        tr += yj[j].x;
#pragma unroll
        for (int i = j + 1; i < P; ++i) B.o[lo(i, j)] = yj[i];
    }
    // ---- dominant eigenpair by repeated squaring of B = C / tr C (see k_solve.h); tau = tr(B^2) = ||B||_F^2 is real here
    const bool ok = tr > 0.0 && tr < 1.7e308;
    {
        const double rt = ok ? rcp64(tr) : 0.0;
#pragma unroll
        for (int i = 0; i < P; ++i) B.d[i] = ok ? B.d[i] * rt : 0.0;
#pragma unroll
        for (int q = 0; q < NO; ++q) B.o[q] = ok ? zscale(B.o[q], rt) : make_double2(0.0, 0.0);
    }
    bool done = !ok || P == 1;
    for (int it = 0; it < DISCO_SQUARINGS_MAX; ++it) {
        if (!__any(!done)) break;
        HermReg<P> S;
        double tau = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            double dj = 0.0;
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const c64 b = B.at(j, k);
                dj = fma(b.x, b.x, fma(b.y, b.y, dj));
            }
            S.d[j] = dj;
            tau += dj;
#pragma unroll
            for (int i = j + 1; i < P; ++i) {
                c64 a = make_double2(0.0, 0.0);
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    const c64 x = B.at(i, k), y = B.at(k, j);
                    a.x = fma(x.x, y.x, fma(-x.y, y.y, a.x));
                    a.y = fma(x.x, y.y, fma(x.y, y.x, a.y));
                }
                S.o[lo(i, j)] = a;
            }
        }
        const double rtau = tau > 0.0 ? rcp64(tau) : 0.0;
        if (!done) {
#pragma unroll
            for (int i = 0; i < P; ++i) B.d[i] = S.d[i] * rtau;
#pragma unroll
            for (int q = 0; q < NO; ++q) B.o[q] = zscale(S.o[q], rtau);
        }
        done = done || (1.0 - tau < DISCO_SQUARING_DONE) || !(tau > 0.0);
    }
    // ---- B = v0 v0^H: the longest column (ties: lowest index), normalised
    c64 v0[P];
    double best = -1.0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        double nj = 0.0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const c64 b = B.at(i, j);
            nj += b.x * b.x + b.y * b.y;
        }
        if (nj > best) {
            best = nj;
#pragma unroll
            for (int i = 0; i < P; ++i) v0[i] = B.at(i, j);
        }
    }
    const bool have = best > 0.0;
    const double rb = have ? rsqrt64(best) : 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) v0[i] = have ? zscale(v0[i], rb) : make_double2(i == 0 ? 1.0 : 0.0, 0.0);
    // ---- power steps v <- B v on the kept square (k_solve.h: DISCO_POWER_STEPS), then unit length
    if (P > 1) {
#pragma unroll 1
        for (int s = 0; s < DISCO_POWER_STEPS; ++s) {
            c64 u[P];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                c64 a = make_double2(0.0, 0.0);
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    const c64 b = B.at(i, k);
                    a.x = fma(b.x, v0[k].x, fma(-b.y, v0[k].y, a.x));
                    a.y = fma(b.x, v0[k].y, fma(b.y, v0[k].x, a.y));
                }
                u[i] = a;
            }
            if (have) {
#pragma unroll
                for (int i = 0; i < P; ++i) v0[i] = u[i];
            }
        }
        if (DISCO_POWER_STEPS > 0 && have) {
            double n2 = 0.0;
#pragma unroll
            for (int i = 0; i < P; ++i) n2 = fma(v0[i].x, v0[i].x, fma(v0[i].y, v0[i].y, n2));
            const double rn = rsqrt64(n2);
#pragma unroll
            for (int i = 0; i < P; ++i) v0[i] = zscale(v0[i], rn);
        }
    }
    // ---- q = L^-H v0, d0 = q^H Rxx q (q^H Rnn q = 1), t1 = q L[0][0] conj(v0[0]), w = t1 d0 / (d0 + mu)
    c64 q[P];
#pragma unroll
    for (int i = P - 1; i >= 0; --i) {
        c64 a = v0[i];
#pragma unroll
        for (int k = i + 1; k < P; ++k) a = zsub(a, zmul(make_double2(Lo[lo(k, i)].x, -Lo[lo(k, i)].y), q[k]));
        q[i] = zscale(a, rL[i]);
    }
    double d0 = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        c64 sj = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < P; ++k) sj = zadd(sj, zmul(A(i, k), q[k]));
        d0 += q[i].x * sj.x + q[i].y * sj.y;
    }
    d0 = have ? d0 : 0.0;
    const double dcl = fmin(fmax(d0, SOLVE_EPS), SOLVE_ETA);
    const double gain = dcl / (dcl + mu);
    const c64 gsc = make_double2(Ld[0] * v0[0].x, -Ld[0] * v0[0].y);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        t1[i] = zmul(q[i], gsc);
        w[i] = zscale(t1[i], gain);
    }
}

// (diag, strict lower triangle) of both matrices of problem pid, from full row-major matrices or from chunk partials
template <int P, bool FROM_PART>
__device__ __forceinline__ void solve_load_tri(const SolveSrc& src, long long pid, float* a_d, c32* a_o, float* b_d, c32* b_o) {
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int k = 0; k <= i; ++k) {
            c32 rs, rn;                          // R[i][k], i >= k
            if constexpr (!FROM_PART) {
                rs = src.Rss[pid * P * P + i * P + k];
                rn = src.Rnn[pid * P * P + i * P + k];
            } else {
                constexpr int NP = P * (P + 1) / 2;
                const long long g = pid / src.F;
                const int f = (int)(pid % src.F);
                const bool loc = i < src.M_loc;                                   // upper-triangle entry (k, i): k <= i < M_loc
                const int Pq = loc ? src.M_loc : P;
                const int q = k * Pq - (k * (k - 1)) / 2 + (i - k);
                const float4* base = loc ? src.part_loc : src.part;
                const int nch = loc ? src.chunks_loc : src.chunks;
                const long long npq = loc ? (long long)(src.M_loc * (src.M_loc + 1) / 2) : (long long)NP;
                double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;                      // float64 combination, one rounding (as k_solve.h)
                for (int ch = 0; ch < nch; ++ch) {
                    const float4 v = base[(((g * nch + ch) * src.F) + f) * npq + q];
                    sx += (double)v.x;
                    sy += (double)v.y;
                    sz += (double)v.z;
                    sw += (double)v.w;
                }
                const double it = (double)src.inv_T;
                rs = make_float2((float)(sx * it), -(float)(sy * it));             // stored (k, i) -> R[i][k] = conj
                rn = make_float2((float)(sz * it), -(float)(sw * it));
            }
            if (i == k) {
                a_d[i] = rs.x;
                b_d[i] = rn.x;
            } else {
                a_o[i * (i - 1) / 2 + k] = rs;
                b_o[i * (i - 1) / 2 + k] = rn;
            }
        }
}

// threads per workgroup: two waves, one for the larger pencils (the wave-cooperative fetch stages 64 * NP float4 per wave in LDS)
template <int P>
constexpr int solve_small_threads() { return P <= 6 ? 128 : 64; }

template <int P, bool FROM_PART>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(solve_small_threads<P>()) void k_gevd_mwf_r1_thread(SolveSrc src, long long n_prob, double mu,
                                                                              c32* __restrict__ w_out, c32* __restrict__ t1_out) {
    constexpr int NO = P > 1 ? P * (P - 1) / 2 : 1;
    constexpr int SOLVE_SMALL_THREADS = solve_small_threads<P>();
    const long long pid = (long long)blockIdx.x * SOLVE_SMALL_THREADS + threadIdx.x;
    const bool live = pid < n_prob;
    float a_d[P], b_d[P];
    c32 a_o[NO], b_o[NO];
    bool loaded = false;
    if constexpr (FROM_PART) {
        // The partial sums of a wave's 64 pencils are 64 * NP consecutive float4 per chunk (pencils are (g, f)-major like the
        // partial array): fetched lane-linearly, summed over the chunks, and handed over through LDS -- a thread loading its own
        // NP entries reads 16 bytes of every 160, ten times over, and the lines do not survive in L1/L2 between the ten
        // (PMC, C3: 2.07 GB fetched for 0.33 GB of partial sums).  Blocks that carry a step-1 block (M_loc) keep the direct path.
        constexpr int NP = P * (P + 1) / 2;
        __shared__ float4 s_tile[SOLVE_SMALL_THREADS / 64][64 * NP];
        if (src.M_loc == 0) {                                             // block-uniform
            const int lane = threadIdx.x & 63, wv = wave_id();
            const long long pc = live ? pid : n_prob - 1;                 // dead lanes stand in for the last pencil
            const long long off = ((pc / src.F) * src.chunks * src.F + pc % src.F) * (long long)NP;      // chunk 0 of this lane's pencil
            const int off_lo = (int)(unsigned)(off & 0xffffffffLL), off_hi = (int)(off >> 32);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int e = lane + 64 * k, pe = e / NP, q = e % NP;     // element e of the wave's block belongs to lane pe's pencil
                const long long o = ((long long)__shfl(off_hi, pe) << 32) | (unsigned)__shfl(off_lo, pe);
                const float4* ptr = src.part + o + q;
                // the chunk sums are combined in float64 and rounded ONCE, as in every other loader (k_solve.h, k_solve_dpp.h, k_cov_finalize)
                double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
                for (int ch = 0; ch < src.chunks; ++ch) {
                    const float4 v = ptr[(long long)ch * src.F * NP];
                    sx += (double)v.x;
                    sy += (double)v.y;
                    sz += (double)v.z;
                    sw += (double)v.w;
                }
                const double it = (double)src.inv_T;
                s_tile[wv][e] = make_float4((float)(sx * it), (float)(sy * it), (float)(sz * it), (float)(sw * it));
            }
            DISCO_GROUP_SYNC();                                           // wave-local hand-over
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int k = 0; k <= i; ++k) {
                    const float4 v = s_tile[wv][lane * NP + (k * P - (k * (k - 1)) / 2 + (i - k))];      // stored (k, i): R[i][k] = conj
                    if (i == k) {
                        a_d[i] = v.x;
                        b_d[i] = v.z;
                    } else {
                        a_o[i * (i - 1) / 2 + k] = make_float2(v.x, -v.y);
                        b_o[i * (i - 1) / 2 + k] = make_float2(v.z, -v.w);
                    }
                }
            loaded = true;
        }
    }
    if (loaded) {
        if (!live) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                a_d[i] = 0.f;
                b_d[i] = 1.f;
            }
#pragma unroll
            for (int q = 0; q < NO; ++q) a_o[q] = b_o[q] = make_float2(0.f, 0.f);
        }
    } else if (live) {
        solve_load_tri<P, FROM_PART>(src, pid, a_d, a_o, b_d, b_o);
    } else {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            a_d[i] = 0.f;
            b_d[i] = 1.f;
        }
#pragma unroll
        for (int q = 0; q < NO; ++q) a_o[q] = b_o[q] = make_float2(0.f, 0.f);
    }
    c64 w[P], t1[P];
    gevd_solve_thread<P>(a_d, a_o, b_d, b_o, mu, w, t1);
    if (live) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            w_out[pid * P + i] = make_float2((float)w[i].x, (float)w[i].y);
            if (t1_out) t1_out[pid * P + i] = make_float2((float)t1[i].x, (float)t1[i].y);
        }
    }
}

}  // namespace disco
