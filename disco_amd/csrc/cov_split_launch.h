// Launches of the block-partitioned covariance kernels k_cov_split / k_cov_split_lds (k_cov.h), shared by api_cov_split_m*.hip:
// each of those units instantiates the shapes of ONE mic count (the kernels are large; three units build in parallel).
#pragma once
#include "host.h"
#include "k_cov.h"

namespace disco_host {
using namespace disco;


// Step-1 shapes (KR = 0, M >= 7) run k_cov_loc_f64: float64 accumulators, 2 * chunks partial blocks.  (Their float32 forms -- lanes = bins,
// or 4 / 8 time sub-chunks across the lanes, option "cov1_mode" of round 4 -- were what C5's distance from the float64 oracle followed and
// were removed in round 5: profiles/r04_c_c5_variants_cov1_f64.json.)
template <int M, int KR>
static void launch_cov_split(bool skiploc, int sub, unsigned nblk, hipStream_t st, const CovArgs& a) {
    // even M (every shape with remote rows, and the step-1 shape M = 8) with F - 1 a multiple of 64 (both FFT sizes of this library): frames
    // staged through LDS once per workgroup (k_cov.h; with remote rows 7.3 ms per C5 launch, the per-wave fetches of k_cov_split: 9.6 ms)
    if constexpr (M % 2 == 0) {
        if (KR > 0) {
            const unsigned nb = DISCO_COV_XCD ? (nblk + DISCO_COV_XCD - 1) / DISCO_COV_XCD * DISCO_COV_XCD : nblk;      // see the kernel's id -> item map
            if constexpr (KR > 0) {
                if (skiploc) {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov_split_lds<M, KR, true>), dim3(nb), dim3(64 * cov_split_waves<KR, true>()), 0, st, a);
                    return;
                }
            }
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov_split_lds<M, KR, false>), dim3(nb), dim3(64 * cov_split_waves<KR, false>()), 0, st, a);
            return;
        }
    }
    if constexpr (KR == 0) {        // step-1 statistics of the wide shapes: float64 accumulators, (hi, lo) pairs of partial blocks
        (void)sub;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov_loc_f64<M>), dim3(nblk), dim3(256), 0, st, a);
    }
}


// (M, KR) shapes for which the block-partitioned kernels are instantiated: 9 <= M + KR <= 16, and the step-1 shapes (KR = 0)
// whose 2 * M(M+1)/2 complex accumulators no longer fit one thread without spilling (M >= 7)
#define DISCO_FOR_SPLIT_M8(X_) X_(7, 0) X_(8, 0) X_(8, 1) X_(8, 2) X_(8, 3) X_(8, 4) X_(8, 5) X_(8, 6) X_(8, 7) X_(8, 8)
#define DISCO_FOR_SPLIT_M4(X_) X_(4, 5) X_(4, 6) X_(4, 7) X_(4, 8) X_(4, 9) X_(4, 10) X_(4, 11) X_(4, 12)
#define DISCO_FOR_SPLIT_M2(X_) X_(2, 7) X_(2, 8) X_(2, 9) X_(2, 10) X_(2, 11) X_(2, 12) X_(2, 13) X_(2, 14)

bool launch_cov_split_m8(int M, int KR, bool skiploc, int sub, unsigned nblk, hipStream_t st, const CovArgs& a);
bool launch_cov_split_m4(int M, int KR, bool skiploc, int sub, unsigned nblk, hipStream_t st, const CovArgs& a);
bool launch_cov_split_m2(int M, int KR, bool skiploc, int sub, unsigned nblk, hipStream_t st, const CovArgs& a);

#define DISCO_DEFINE_SPLIT_LAUNCHER(NAME_, TABLE_)                                                          \
    bool NAME_(int M, int KR, bool skiploc, int sub, unsigned nblk, hipStream_t st, const CovArgs& a) {     \
        TABLE_(DISCO_SPLIT_CASE_)                                                                           \
        return false;                                                                                       \
    }
#define DISCO_SPLIT_CASE_(M_, KR_)                        \
    if (M == M_ && KR == KR_) {                           \
        launch_cov_split<M_, KR_>(skiploc, sub, nblk, st, a);  \
        return true;                                      \
    }
}  // namespace disco_host
