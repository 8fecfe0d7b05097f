// Where a wave FFT's time goes on gfx950: the transform of fft.h in a loop (registers -> registers, wave-private LDS exchange),
// (a) complete, (b) butterflies only (exchanges removed: wrong numbers, VALU time), (c) exchanges only (LDS time),
// (d) complete + the two-real-channels untangle, at 1..4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize
#include "fft_plans.h"
#include <cstdio>
#include <cstdlib>
using namespace disco;

template <int MODE>
__global__ __launch_bounds__(256) void k_fft_rate(const c32* __restrict__ tw, c32* __restrict__ out, int iters) {
    constexpr int N = 512, E = 8;
    __shared__ c32 bufs[4][fft_buf_len<N>()];
    const int wave = wave_id(), lane = threadIdx.x & 63;
    c32* buf = bufs[wave];
    WaveTw<N> wtw;
    wtw.init(tw, lane);
    c32 v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = make_float2(1.f + lane + e, 0.5f * e);
    c32 acc = make_float2(0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 3) {
            fft_wave<N>(v, wtw, buf, lane);
        } else if constexpr (MODE == 1) {          // butterflies and twiddles only
#pragma unroll
            for (int pss = 0; pss < 3; ++pss) {
                if (pss > 0) {
#pragma unroll
                    for (int r = 1; r < 8; ++r) v[r] = cmul_pk(v[r], pss == 1 ? wtw.t1[r - 1] : wtw.t2[r - 1]);
                }
                dft8(v);
#pragma unroll
                for (int e = 0; e < E; ++e) DISCO_CONSUME(v[e].x);
            }
        } else if constexpr (MODE == 2) {          // the two exchanges only
#pragma unroll
            for (int pss = 0; pss < 2; ++pss) {
                DISCO_LDS_WAR();
#pragma unroll
                for (int r = 0; r < 8; ++r) buf[fft_pad<N>(lane * 8 + r)] = v[r];
                DISCO_LDS_RAW();
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = buf[fft_pad<N>(lane + 64 * r)];
            }
        }
        if constexpr (MODE == 3) {
            rfft_pair_untangle<N>(v, buf, lane, [&](int j, int, c32 a, c32 b) { acc = cadd(acc, cadd(a, b)); });
        }
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = make_float2(v[e].x * 0.04f, v[e].y * 0.04f);      // keeps the values bounded; 16 extra multiplies
    }
#pragma unroll
    for (int e = 0; e < E; ++e) acc = cadd(acc, v[e]);
    out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// The round-4 plan: two transforms per wave, 16 points per lane, ONE exchange (fft_wave_2x512).  MODE 0: complete, 1: butterflies,
// twiddles and the lane transposes only, 2: the exchange only.  Reported per TRANSFORM (a wave iteration is two).
template <int MODE>
__global__ __launch_bounds__(256) void k_fft2x_rate(const c32* __restrict__ tw, c32* __restrict__ out, int iters) {
    __shared__ c32 bufs[4][FFT2X_BUF];
    const int wave = wave_id(), lane = threadIdx.x & 63;
    c32* buf = bufs[wave];
    WaveTw2x wtw;
    wtw.init(tw, lane);
    c32 v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = make_float2(1.f + lane + e, 0.5f * e);
    c32 acc = make_float2(0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            fft_wave_2x512(v, wtw, buf, lane);
        } else if constexpr (MODE == 1) {
            dft16(v);
#pragma unroll
            for (int k = 1; k < 16; ++k) v[k] = cmul_pk(v[k], (k & 1) ? wtw.w1 : ((k & 2) ? wtw.w2 : ((k & 4) ? wtw.w4 : wtw.w8)));
#pragma unroll
            for (int k = 0; k < 11; ++k) acc = cmul_pk(acc, wtw.w1);                      // the products that form the 11 composite twiddles
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                c32 a = v[s], b = v[s + 8];
                xlane_swap<16>(a, b, lane);
                v[s] = cadd(a, b);
                v[s + 8] = cmul_pk(csub(a, b), wtw.r);
            }
            dft16(v);
#pragma unroll
            for (int e = 0; e < 16; ++e) DISCO_CONSUME(v[e].x);
        } else {
            const int hh = (lane >> 4) & 1, j = lane & 15;
            c32* rows = buf + (lane >> 5) * (32 * FFT2X_PITCH);
            DISCO_LDS_WAR();
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                rows[(s + 8 * hh) * FFT2X_PITCH + j] = v[s];
                rows[(s + 8 * hh + 16) * FFT2X_PITCH + j] = v[s + 8];
            }
            DISCO_LDS_RAW();
            const float4* rd = reinterpret_cast<const float4*>(rows + (lane & 31) * FFT2X_PITCH);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 t = rd[q];
                v[2 * q] = make_float2(t.x, t.y);
                v[2 * q + 1] = make_float2(t.z, t.w);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = make_float2(v[e].x * 0.04f, v[e].y * 0.04f);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc = cadd(acc, v[e]);
    out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
static void run2x(const char* name, const c32* tw) {
    const int iters = 1000;
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = 256 * wps;
        c32* out;
        (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_fft2x_rate<MODE>, dim3(blocks), dim3(256), 0, 0, tw, out, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_fft2x_rate<MODE>, dim3(blocks), dim3(256), 0, 0, tw, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s waves/SIMD %d : %7.3f ms  -> %7.1f ns per transform per SIMD\n", name, wps, ms, ms * 1e6 / (2.0 * iters * wps));
        (void)hipFree(out);
    }
}

__global__ __launch_bounds__(64) void k_fft2x_check(const c32* __restrict__ tw, const c32* __restrict__ x, c32* __restrict__ X) {
    __shared__ c32 buf[FFT2X_BUF];
    const int lane = threadIdx.x, half = lane >> 5, lp = lane & 31;
    WaveTw2x t;
    t.init(tw, lane);
    c32 v[16];
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) v[s_] = x[half * 512 + lp + 32 * s_];
    fft_wave_2x512(v, t, buf, lane);
#pragma unroll
    for (int e = 0; e < 16; ++e) X[half * 512 + lp + 32 * e] = v[e];
}

static void check2x(const c32* tw) {
    const int N = 512;
    c32* hx = (c32*)malloc(2 * N * sizeof(c32));
    srand(2);
    for (int j = 0; j < 2 * N; ++j) hx[j] = make_float2(rand() / (float)RAND_MAX - 0.5f, rand() / (float)RAND_MAX - 0.5f);
    c32 *dx, *dX;
    (void)hipMalloc(&dx, 2 * N * sizeof(c32));
    (void)hipMalloc(&dX, 2 * N * sizeof(c32));
    (void)hipMemcpy(dx, hx, 2 * N * sizeof(c32), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_fft2x_check, dim3(1), dim3(64), 0, 0, tw, dx, dX);
    c32* hX = (c32*)malloc(2 * N * sizeof(c32));
    (void)hipMemcpy(hX, dX, 2 * N * sizeof(c32), hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int h = 0; h < 2; ++h)
        for (int k = 0; k < N; ++k) {
            double re = 0, im = 0;
            for (int n = 0; n < N; ++n) {
                const double c = cos(-2.0 * M_PI * n * k / N), s_ = sin(-2.0 * M_PI * n * k / N);
                re += hx[h * N + n].x * c - hx[h * N + n].y * s_;
                im += hx[h * N + n].x * s_ + hx[h * N + n].y * c;
            }
            worst = fmax(worst, hypot(hX[h * N + k].x - re, hX[h * N + k].y - im));
            scale = fmax(scale, hypot(re, im));
        }
    printf("check: fft_wave_2x512 max abs err %.3e (spectrum scale %.2f) -> %s\n", worst, scale, worst < 1e-5 * scale ? "OK" : "WRONG");
}

template <int MODE>
static void run(const char* name, const c32* tw) {
    const int iters = 2000;
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = 256 * wps;
        c32* out;
        (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_fft_rate<MODE>, dim3(blocks), dim3(256), 0, 0, tw, out, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_fft_rate<MODE>, dim3(blocks), dim3(256), 0, 0, tw, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: wps waves x iters transforms
        printf("%-44s waves/SIMD %d : %7.3f ms  -> %7.1f ns per transform per SIMD\n", name, wps, ms, ms * 1e6 / ((double)iters * wps));
        (void)hipFree(out);
    }
}

// one transform of a known input against a float64 DFT on the host (pins the cross-lane instruction semantics)
__global__ __launch_bounds__(64) void k_fft_check(const c32* __restrict__ tw, const c32* __restrict__ x, c32* __restrict__ X, c32* __restrict__ AB) {
    constexpr int N = 512, E = 8;
    __shared__ c32 buf[fft_buf_len<N>()];
    const int lane = threadIdx.x;
    WaveTw<N> wtw;
    wtw.init(tw, lane);
    c32 v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = x[lane + 64 * e];
    fft_wave<N>(v, wtw, buf, lane);
#pragma unroll
    for (int e = 0; e < E; ++e) X[lane + 64 * e] = v[e];
    rfft_pair_untangle<N>(v, buf, lane, [&](int, int f, c32 a, c32 b) {
        AB[f] = a;
        AB[N + f] = b;
    });
}

static void check(const c32* tw) {
    const int N = 512;
    c32* hx = (c32*)malloc(N * sizeof(c32));
    srand(1);
    for (int j = 0; j < N; ++j) hx[j] = make_float2(rand() / (float)RAND_MAX - 0.5f, rand() / (float)RAND_MAX - 0.5f);
    c32 *dx, *dX, *dAB;
    (void)hipMalloc(&dx, N * sizeof(c32));
    (void)hipMalloc(&dX, N * sizeof(c32));
    (void)hipMalloc(&dAB, 2 * N * sizeof(c32));
    (void)hipMemset(dAB, 0, 2 * N * sizeof(c32));
    (void)hipMemcpy(dx, hx, N * sizeof(c32), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_fft_check, dim3(1), dim3(64), 0, 0, tw, dx, dX, dAB);
    c32* hX = (c32*)malloc(N * sizeof(c32));
    c32* hAB = (c32*)malloc(2 * N * sizeof(c32));
    (void)hipMemcpy(hX, dX, N * sizeof(c32), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hAB, dAB, 2 * N * sizeof(c32), hipMemcpyDeviceToHost);
    double worst = 0, worst_ab = 0, scale = 0;
    for (int k = 0; k < N; ++k) {
        double re = 0, im = 0, are = 0, aim = 0, bre = 0, bim = 0;
        for (int n = 0; n < N; ++n) {
            const double c = cos(-2.0 * M_PI * n * k / N), s_ = sin(-2.0 * M_PI * n * k / N);
            re += hx[n].x * c - hx[n].y * s_;
            im += hx[n].x * s_ + hx[n].y * c;
            are += hx[n].x * c;  aim += hx[n].x * s_;      // spectrum of the real part
            bre += hx[n].y * c;  bim += hx[n].y * s_;      // spectrum of the imaginary part
        }
        worst = fmax(worst, hypot(hX[k].x - re, hX[k].y - im));
        scale = fmax(scale, hypot(re, im));
        if (k <= N / 2) {      // untangle returns 2 A, 2 B (the halving is the caller's)
            worst_ab = fmax(worst_ab, hypot(0.5 * hAB[k].x - are, 0.5 * hAB[k].y - aim));
            worst_ab = fmax(worst_ab, hypot(0.5 * hAB[N + k].x - bre, 0.5 * hAB[N + k].y - bim));
        }
    }
    printf("check: fft_wave<512> max abs err %.3e, untangled pair max abs err %.3e (spectrum scale %.2f) -> %s\n", worst, worst_ab, scale,
           (worst < 1e-4 * scale && worst_ab < 1e-4 * scale) ? "OK" : "WRONG");
}

int main() {
    const int N = 512;
    c32* h = (c32*)malloc(N * sizeof(c32));
    for (int j = 0; j < N; ++j) h[j] = make_float2((float)cos(-2.0 * M_PI * j / N), (float)sin(-2.0 * M_PI * j / N));
    c32* tw;
    (void)hipMalloc(&tw, N * sizeof(c32));
    (void)hipMemcpy(tw, h, N * sizeof(c32), hipMemcpyHostToDevice);
    check(tw);
    run<0>("fft_wave<512> complete", tw);
    run<1>("butterflies + twiddles only (VALU)", tw);
    run<2>("two LDS exchanges only", tw);
    run<3>("fft_wave<512> + rfft_pair_untangle", tw);
    check2x(tw);
    run2x<0>("fft_wave_2x512 complete (one exchange)", tw);
    run2x<1>("  its butterflies + twiddles + transposes", tw);
    run2x<2>("  its one LDS exchange", tw);
    return 0;
}
