// libdisco_hip.so -- host side of the C ABI (gfx950 only): dispatch of the block-partitioned covariance kernels to the unit
// that instantiates the shape (api_cov_split_m8 / _m4 / _m2.hip)
#include "cov_split_launch.h"

namespace disco_host {
bool cov_split_shape(int M, int KR) {
#define X_(M_, KR_) if (M == M_ && KR == KR_) return true;
    DISCO_FOR_SPLIT_M8(X_) DISCO_FOR_SPLIT_M4(X_) DISCO_FOR_SPLIT_M2(X_)
#undef X_
    return false;
}

bool launch_cov_split_shape(int M, int KR, bool skiploc, int sub, unsigned nblk, hipStream_t st, const CovArgs& a) {
    return launch_cov_split_m8(M, KR, skiploc, sub, nblk, st, a) || launch_cov_split_m4(M, KR, skiploc, sub, nblk, st, a) ||
           launch_cov_split_m2(M, KR, skiploc, sub, nblk, st, a);
}
}  // namespace disco_host
