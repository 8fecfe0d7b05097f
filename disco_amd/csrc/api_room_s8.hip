// libdisco_hip.so -- host side of the C ABI (gfx950 only): the persistent room pass with 8 time sub-chunk(s) per workgroup
#include "room_launch.h"

namespace disco_host {
DISCO_DEFINE_ROOM_LAUNCHER(launch_room_s8, 8)
}  // namespace disco_host
