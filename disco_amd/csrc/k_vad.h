// 'ivad' mask: get_mask's branch tango.py:217-221 = vad_oracle_batch (sigproc_utils.py:12-55) on the
// target's time signal, sampled every hop and tiled over frequency.
//   x = ts - mean(ts);  x2 = |x^2|;  thr = 0.001 * quantile(x2, 0.99)          (numpy 'linear' interpolation)
//   window n = samples [n hop, n hop + win) (the last one shorter): active iff #(x2 > thr) >= int(len / 2)
//   vad[sample] = 1 if any active window covers it;  mask[f, t] = vad[t hop] for t < ceil(L / hop), else 0
// One workgroup per signal.  The 0.99-quantile needs two neighbouring order statistics of ~L values: a 4-pass radix
// select on the float bit patterns (x2 >= 0, so unsigned order = numeric order) with a 256-bin LDS histogram per pass,
// then one pass for the next larger value.  The signal (<= 640 kB) stays in L2 across the passes.
#pragma once
#include "common.h"

namespace disco {

constexpr int VAD_THREADS = 256;
constexpr int VAD_MAX_SEG = 4096;          // hop segments per signal (LDS counters): L <= 4096 * hop

struct VadShared {
    int hist[256];
    int cnt[VAD_MAX_SEG];
    double red[VAD_THREADS / 64];
    unsigned redu[VAD_THREADS / 64];
    int redi[VAD_THREADS / 64];
    unsigned prefix;
    int k_rem;
    float mean, thr;
};

__device__ __forceinline__ float vad_x2(const float* __restrict__ x, int i, float mean) {
    const float d = x[i] - mean;
    return fabsf(d * d);
}

// k-th smallest (0-based) of x2[0..n): radix select, most significant byte first
__device__ __forceinline__ unsigned vad_select(VadShared& sh, const float* __restrict__ x, int n, float mean, int k) {
    const int tid = threadIdx.x;
    if (tid == 0) {
        sh.prefix = 0u;
        sh.k_rem = k;
    }
    unsigned maskbits = 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
        sh.hist[tid] = 0;                               // VAD_THREADS == 256 bins
        __syncthreads();
        const unsigned prefix = sh.prefix;
        for (int i = tid; i < n; i += VAD_THREADS) {
            const unsigned b = __float_as_uint(vad_x2(x, i, mean));
            if ((b & maskbits) == prefix) atomicAdd(&sh.hist[(b >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int kr = sh.k_rem, d = 0;
            while (d < 255 && kr >= sh.hist[d]) {
                kr -= sh.hist[d];
                ++d;
            }
            sh.k_rem = kr;
            sh.prefix = prefix | ((unsigned)d << shift);
        }
        maskbits |= 255u << shift;
        __syncthreads();
    }
    return sh.prefix;
}

static __global__ __launch_bounds__(VAD_THREADS) void k_vad_mask(const float* __restrict__ ts, float* __restrict__ mask, int L, int T,
                                                          int F, int win, int hop, float thr_rel, float quant, int rat) {
    __shared__ VadShared sh;
    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const long long sig = blockIdx.x;
    const float* x = ts + sig * L;
    // ---- mean (float64 accumulation, rounded to the float32 np.mean returns for a float32 signal)
    double s = 0.0;
    for (int i = tid; i < L; i += VAD_THREADS) s += (double)x[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) sh.red[w] = s;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < VAD_THREADS / 64; ++i) t += sh.red[i];
        sh.mean = (float)(t / (double)L);
    }
    __syncthreads();
    const float mean = sh.mean;
    // ---- quantile: virtual index (L-1) q, linear interpolation between the two neighbouring order statistics
    const double vidx = (double)(L - 1) * (double)quant;
    const int k_lo = (int)vidx;
    const double frac = vidx - (double)k_lo;
    const unsigned b_lo = vad_select(sh, x, L, mean, k_lo);
    // the next order statistic: v_lo again if more than k_lo + 1 values are <= v_lo, else the smallest value above it
    int c_le = 0;
    unsigned m_gt = 0xffffffffu;
    for (int i = tid; i < L; i += VAD_THREADS) {
        const unsigned b = __float_as_uint(vad_x2(x, i, mean));
        c_le += b <= b_lo ? 1 : 0;
        if (b > b_lo && b < m_gt) m_gt = b;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        c_le += __shfl_xor(c_le, off, 64);
        const unsigned o = __shfl_xor(m_gt, off, 64);
        m_gt = o < m_gt ? o : m_gt;
    }
    if (lane == 0) {
        sh.redi[w] = c_le;
        sh.redu[w] = m_gt;
    }
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        unsigned m = 0xffffffffu;
        for (int i = 0; i < VAD_THREADS / 64; ++i) {
            c += sh.redi[i];
            m = sh.redu[i] < m ? sh.redu[i] : m;
        }
        const float v_lo = __uint_as_float(b_lo);
        const float v_hi = (c > k_lo + 1 || k_lo + 1 >= L) ? v_lo : __uint_as_float(m);
        const float q = (float)((double)v_lo + ((double)v_hi - (double)v_lo) * frac);
        sh.thr = thr_rel * q;
    }
    __syncthreads();
    const float thr = sh.thr;
    // ---- #(x2 > thr) per hop segment
    const int n_seg = (L + hop - 1) / hop;
    for (int sg = w; sg < n_seg; sg += VAD_THREADS / 64) {
        int c = 0;
        const int lo = sg * hop, hi = min(lo + hop, L);
        for (int i = lo + lane; i < hi; i += 64) c += vad_x2(x, i, mean) > thr ? 1 : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
        if (lane == 0) sh.cnt[sg] = c;
    }
    __syncthreads();
    // ---- frame decisions + tiling over frequency.  Window n covers segments n .. n + win/hop - 1 (clipped at L).
    const int spw = win / hop;                                               // segments per window (2)
    const int n_win = (int)ceil(((double)L - (double)win) / (double)hop + 1.0);
    float* mg = mask + sig * (long long)T * F;
    for (int t = w; t < T; t += VAD_THREADS / 64) {
        float v = 0.f;
        if (t < n_seg) {                                                     // vad[::hop] has ceil(L / hop) entries
            // sample t*hop lies in windows n = t - spw + 1 .. t
            for (int nwin = max(0, t - spw + 1); nwin <= t && nwin < n_win; ++nwin) {
                int c = 0;
                for (int q2 = 0; q2 < spw; ++q2) c += (nwin + q2 < n_seg) ? sh.cnt[nwin + q2] : 0;
                const int len = min(nwin * hop + win, L) - nwin * hop;
                if (c >= len / rat) v = 1.f;
            }
        }
        for (int f = lane; f < F; f += 64) mg[(long long)t * F + f] = v;
    }
}

}  // namespace disco
