// v_fmac_f64_dpp row_newbcast probe for gfx950: (1) what the instruction computes (lane n of each row of 16 as src0, with and
// without the neg modifier), (2) wave64 issue rate against plain v_fma_f64 / v_fmac_f64 and against the LDS broadcast read it is
// to replace (ds_read_b128 + 4 v_fma_f64 per complex multiply-add).
// Standalone: hipcc --offload-arch=gfx950 -O3 dpp64_rate.hip -o dpp64_rate
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

__global__ void k_sem(const double* a, const double* b, double* out) {
    const int t = threadIdx.x;
    double acc0 = 100.0 + t, acc1 = 100.0 + t, acc2 = 100.0 + t;
    const double av = a[t], bv = b[t];
    asm volatile("s_nop 4\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc0) : "v"(av), "v"(bv));
    asm volatile("s_nop 4\n v_fmac_f64_dpp %0, -%1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "+v"(acc1) : "v"(av), "v"(bv));
    asm volatile("s_nop 4\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(acc2) : "v"(av), "v"(bv));
    out[t] = acc0;
    out[64 + t] = acc1;
    out[128 + t] = acc2;
}

template <int V>
__global__ __launch_bounds__(256) void k_rate(double* out, long long* cyc, int iters) {
    __shared__ double2 lds[256];
    double a[8], b = 1.0000001, c = 1e-9;
    lds[threadIdx.x] = make_double2(1.0 + 1e-9 * threadIdx.x, 1e-9);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (double)threadIdx.x + i;
    const double2* lp = &lds[(threadIdx.x & ~15)];
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (V == 0) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if constexpr (V == 1) {
#define X(i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if constexpr (V == 2) {
#define X(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if constexpr (V == 3) {      // with the neg modifier and varying lanes
#define X(i) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c), "n"(i + 4));
                REP8(X)
#undef X
            } else if constexpr (V == 4) {      // what it replaces: one broadcast ds_read_b128 per 4 v_fma_f64
                double2 v0, v1;
                asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)" : "=v"(v0), "=v"(v1) : "v"((unsigned)(size_t)(lp + 2 * u)));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[0]) : "v"(v0.x), "v"(c));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[1]) : "v"(v0.y), "v"(c));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[2]) : "v"(v0.x), "v"(b));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[3]) : "v"(v0.y), "v"(b));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[4]) : "v"(v1.x), "v"(c));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[5]) : "v"(v1.y), "v"(c));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[6]) : "v"(v1.x), "v"(b));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[7]) : "v"(v1.y), "v"(b));
            } else if constexpr (V == 5) {      // dependent chain (latency)
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                             : "+v"(a[0]) : "v"(b), "v"(c));
            } else if constexpr (V == 6) {      // v_mov_b64_dpp (the all-gather form)
#define X(i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            }
        }
    }
    const long long t1 = clock64();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char* name, int instr_per_u) {
    const int iters = 2000;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        double* out;
        long long* cyc;
        hipMalloc(&out, (size_t)blocks * 256 * 8);
        hipMalloc(&cyc, (size_t)blocks * 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k_rate<V>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rate<V>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        long long* h = (long long*)malloc((size_t)blocks * 8);
        hipMemcpy(h, cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (int i = 0; i < blocks; ++i) avg += (double)h[i];
        avg /= blocks;
        const double n_instr = (double)iters * 4 * instr_per_u;
        printf("%-40s waves/SIMD %d : %8.3f ms  wall ns per VALU instr per SIMD %6.3f\n", name, wps, ms, ms * 1e6 / (n_instr * wps));
        free(h);
        hipFree(out);
        hipFree(cyc);
    }
}

int main() {
    double ha[64], hb[64], ho[192];
    for (int i = 0; i < 64; ++i) {
        ha[i] = 1.0 + i;
        hb[i] = 0.5 + 0.25 * i;
    }
    double *da, *db, *dout;
    hipMalloc(&da, 512);
    hipMalloc(&db, 512);
    hipMalloc(&dout, 192 * 8);
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, da, db, dout);
    hipMemcpy(ho, dout, 192 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        const int row = t & ~15;
        const double e0 = std::fma(ha[row + 5], hb[t], 100.0 + t), e1 = std::fma(-ha[row + 15], hb[t], 100.0 + t), e2 = std::fma(ha[row], hb[t], 100.0 + t);
        if (ho[t] != e0 || ho[64 + t] != e1 || ho[128 + t] != e2) {
            if (bad < 6) printf("lane %d: got %g %g %g  expected %g %g %g\n", t, ho[t], ho[64 + t], ho[128 + t], e0, e1, e2);
            ++bad;
        }
    }
    printf("semantics: acc += src0[lane n of the row] * src1[own lane], neg on src0: %s (%d lanes differ)\n", bad ? "NO" : "yes", bad);
    run<0>("v_fma_f64", 8);
    run<1>("v_fmac_f64", 8);
    run<2>("v_fmac_f64_dpp row_newbcast", 8);
    run<3>("v_fmac_f64_dpp neg, lanes 4..11", 8);
    run<4>("2 ds_read_b128 + 8 v_fma_f64", 8);
    run<5>("v_fmac_f64_dpp dependent chain", 8);
    run<6>("v_mov_b64_dpp row_newbcast", 8);
    return 0;
}
