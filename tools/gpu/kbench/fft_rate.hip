// Where a wave FFT's time goes on gfx950: the transform of fft.h in a loop (registers -> registers, wave-private LDS exchange),
// (a) complete, (b) butterflies only (exchanges removed: wrong numbers, VALU time), (c) exchanges only (LDS time),
// (d) complete + the two-real-channels untangle, at 1..4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize
#include "../../../disco_amd/csrc/fft.h"
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
    return 0;
}
