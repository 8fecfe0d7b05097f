// STFT / oracle-mask / iSTFT kernels (librosa semantics, see oracle/stft_oracle.py for the restated
// algorithm and the reference call sites: tango.py:335-342, 528-539; math_utils.py:134-152).
#pragma once
#include "fft.h"

namespace disco {

constexpr int STFT_WAVES = 4;     // waves (= frame/channel-pair items in flight) per block
constexpr int STFT_ITERS = 4;     // items per wave per block (amortises the table loads)

template <int N>
struct StftShared {
    c32 tw[N];
    float win[N];
    c32 buf[STFT_WAVES][fft_buf_len<N>()];
};

template <int N>
__device__ __forceinline__ void load_tables(StftShared<N>& sh, const float* __restrict__ win, const c32* __restrict__ tw) {
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        sh.tw[i] = tw[i];
        sh.win[i] = win[i];
    }
    __syncthreads();
}

// windowed, centre-padded frame t of the channel pair (xa, xb) into FFT slots: v[e] = (a, b)[lane + 64 e]
template <int N>
__device__ __forceinline__ void load_frame_pair(c32* v, const float* __restrict__ xa, const float* __restrict__ xb,
                                                int t, int L, int pad_mode, const float* win, int lane) {
    constexpr int E = FftPlan<N>::E, H = N / 2;
    const int p0 = t * H - N / 2;
    const bool interior = (p0 >= 0) && (p0 + N <= L);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int n = lane + 64 * e;
        const float w = win[n];
        float a, b = 0.f;
        if (interior) {
            a = xa[p0 + n];
            if (xb) b = xb[p0 + n];
        } else {
            a = load_padded(xa, p0 + n, L, pad_mode);
            if (xb) b = load_padded(xb, p0 + n, L, pad_mode);
        }
        v[e] = make_float2(a * w, b * w);
    }
}

// x: [n_sig][chans][L] -> X: [n_sig][T][F][chans].  One wave per (signal group, frame, channel pair).
template <int N>
__global__ __launch_bounds__(256) void k_stft(const float* __restrict__ x, c32* __restrict__ X,
                                               const float* __restrict__ win, const c32* __restrict__ tw,
                                               int chans, int L, int T, int pad_mode, long long n_items) {
    constexpr int E = FftPlan<N>::E, F = N / 2 + 1;
    __shared__ StftShared<N> sh;
    load_tables<N>(sh, win, tw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int MP = (chans + 1) >> 1;
    for (int it = 0; it < STFT_ITERS; ++it) {
        const long long item = ((long long)blockIdx.x * STFT_ITERS + it) * STFT_WAVES + wave;
        const bool active = item < n_items;
        const int pair = (int)(item % MP);
        const long long gt = item / MP;
        const int t = (int)(gt % T);
        const long long g = gt / T;
        const int ca = 2 * pair, cb = 2 * pair + 1;
        c32 v[E];
        if (active) {
            const float* xa = x + (g * chans + ca) * (long long)L;
            const float* xb = (cb < chans) ? x + (g * chans + cb) * (long long)L : nullptr;
            load_frame_pair<N>(v, xa, xb, t, L, pad_mode, sh.win, lane);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = make_float2(0.f, 0.f);
        }
        fft_wave<N>(v, sh.tw, sh.buf[wave], lane);
        c32* Xo = X + ((g * T + t) * (long long)F) * chans;
        rfft_pair_untangle<N>(v, sh.buf[wave], lane, [&](int f, c32 A, c32 B) {
            if (active) {
                Xo[(long long)f * chans + ca] = A;
                if (cb < chans) Xo[(long long)f * chans + cb] = B;
            }
        });
    }
}

__device__ __forceinline__ float tf_mask_value(c32 S, c32 Nn, int mask_type, int mask_pow, float thr_lin) {
    const float as = sqrtf(S.x * S.x + S.y * S.y);
    if (mask_type == DISCO_MASK_IAM) {
        const c32 y = cadd(S, Nn);
        float r = as / sqrtf(y.x * y.x + y.y * y.y);
        float m = r;
        for (int i = 1; i < mask_pow; ++i) m *= r;
        return mask_pow == 0 ? 1.f : m;
    }
    const float an = fmaxf(sqrtf(Nn.x * Nn.x + Nn.y * Nn.y), 2.220446049250313e-16f);
    const float r = as / an;
    float xi = r;
    for (int i = 1; i < mask_pow; ++i) xi *= r;
    if (mask_pow == 0) xi = 1.f;
    if (mask_type == DISCO_MASK_IBM) return xi >= thr_lin ? 1.f : 0.f;
    return xi / (1.f + xi);
}

// s_ref, n_ref: [n_sig][L] -> mask [n_sig][T][F]; the pair (s, n) shares one complex FFT.
template <int N>
__global__ __launch_bounds__(256) void k_mask_oracle(const float* __restrict__ s_ref, const float* __restrict__ n_ref,
                                                      float* __restrict__ mask, const float* __restrict__ win,
                                                      const c32* __restrict__ tw, int L, int T, int pad_mode,
                                                      int mask_type, int mask_pow, float thr_lin, long long n_items) {
    constexpr int E = FftPlan<N>::E, F = N / 2 + 1;
    __shared__ StftShared<N> sh;
    load_tables<N>(sh, win, tw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int it = 0; it < STFT_ITERS; ++it) {
        const long long item = ((long long)blockIdx.x * STFT_ITERS + it) * STFT_WAVES + wave;
        const bool active = item < n_items;
        const int t = (int)(item % T);
        const long long g = item / T;
        c32 v[E];
        if (active) {
            load_frame_pair<N>(v, s_ref + g * (long long)L, n_ref + g * (long long)L, t, L, pad_mode, sh.win, lane);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = make_float2(0.f, 0.f);
        }
        fft_wave<N>(v, sh.tw, sh.buf[wave], lane);
        float* mo = mask + (g * T + t) * (long long)F;
        rfft_pair_untangle<N>(v, sh.buf[wave], lane, [&](int f, c32 A, c32 B) {
            if (active) mo[f] = tf_mask_value(A, B, mask_type, mask_pow, thr_lin);
        });
    }
}

__global__ void k_tf_mask(const c32* __restrict__ S, const c32* __restrict__ Nn, float* __restrict__ mask,
                          long long n, int mask_type, int mask_pow, float thr_lin) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        mask[i] = tf_mask_value(S[i], Nn[i], mask_type, mask_pow, thr_lin);
}

// ---- iSTFT --------------------------------------------------------------------------------------------
// Z: [n_sig][T][F] -> out: [n_sig][L].  A block owns ISTFT_FRAMES consecutive frames of one signal (each wave
// inverts two frames with one complex FFT), overlap-adds them in LDS and emits the ISTFT_FRAMES - 1 hop
// segments that are complete inside the block; consecutive blocks overlap by one frame.
constexpr int ISTFT_FRAMES = 2 * STFT_WAVES;
constexpr int ISTFT_SEGS = ISTFT_FRAMES - 1;

template <int N>
struct IstftShared {
    c32 tw[N];
    float win[N];
    c32 buf[STFT_WAVES][fft_buf_len<N>()];     // after the transform each wave parks its two time frames here
};
template <int N>
__device__ __forceinline__ float* istft_frame(IstftShared<N>& sh, int j) {
    return reinterpret_cast<float*>(sh.buf[j >> 1]) + (j & 1) * N;
}

template <int N>
__global__ __launch_bounds__(256) void k_istft(const c32* __restrict__ Z, float* __restrict__ out,
                                                const float* __restrict__ win, const c32* __restrict__ tw,
                                                int L, int T, int blocks_per_sig) {
    constexpr int E = FftPlan<N>::E, F = N / 2 + 1, H = N / 2;
    __shared__ IstftShared<N> sh;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        sh.tw[i] = tw[i];
        sh.win[i] = win[i];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long g = blockIdx.x / blocks_per_sig;
    const int seg0 = (int)(blockIdx.x % blocks_per_sig) * ISTFT_SEGS;       // first output segment == first frame
    const int ta = seg0 + 2 * wave, tb = ta + 1;
    const c32* Za = Z + (g * T + ta) * (long long)F;
    const c32* Zb = Z + (g * T + tb) * (long long)F;
    const bool has_a = ta < T, has_b = tb < T;
    // V[n] = A~[n] + i B~[n] with A~, B~ the Hermitian extensions; the inverse transform is conj(FFT(conj V)) / N
    c32 v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int n = lane + 64 * e;
        const bool upper = n > N / 2;
        const int f = upper ? N - n : n;
        c32 a = has_a ? Za[f] : make_float2(0.f, 0.f);
        c32 b = has_b ? Zb[f] : make_float2(0.f, 0.f);
        if (f == 0 || f == N / 2) {      // irfft ignores the imaginary part of DC and Nyquist
            a.y = 0.f;
            b.y = 0.f;
        }
        if (upper) {
            a.y = -a.y;
            b.y = -b.y;
        }
        const c32 V = make_float2(a.x - b.y, a.y + b.x);
        v[e] = cconj(V);
    }
    fft_wave<N>(v, sh.tw, sh.buf[wave], lane);
    const float inv = 1.0f / N;
    DISCO_LDS_WAR();                                   // every lane is past its last read of buf[wave]
    float* fa = istft_frame<N>(sh, 2 * wave);
    float* fb = istft_frame<N>(sh, 2 * wave + 1);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int n = lane + 64 * e;
        const float w = sh.win[n] * inv;
        fa[n] = v[e].x * w;                            // Re(conj(out)) =  out.x
        fb[n] = -v[e].y * w;                           // Im(conj(out)) = -out.y
    }
    __syncthreads();
    float* o = out + g * (long long)L;
    for (int i = threadIdx.x; i < ISTFT_SEGS * H; i += blockDim.x) {
        const int j = i / H, n = i - j * H;
        const int seg = seg0 + j;
        const long long pos = (long long)seg * H + n;
        if (seg < T && pos < L) {
            const float w0 = sh.win[H + n], w1 = sh.win[n];
            float wss = w0 * w0;
            if (seg + 1 < T) wss += w1 * w1;
            float val = istft_frame<N>(sh, j)[H + n] + istft_frame<N>(sh, j + 1)[n];
            if (wss > 1.17549435e-38f) val /= wss;
            o[pos] = val;
        }
    }
}

}  // namespace disco
