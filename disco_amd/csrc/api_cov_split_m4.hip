// libdisco_hip.so -- host side of the C ABI (gfx950 only): block-partitioned covariance kernels, shapes of table M4
#include "cov_split_launch.h"

namespace disco_host {
DISCO_DEFINE_SPLIT_LAUNCHER(launch_cov_split_m4, DISCO_FOR_SPLIT_M4)
}  // namespace disco_host
