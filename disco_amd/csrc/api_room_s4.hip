// libdisco_hip.so -- host side of the C ABI (gfx950 only): the persistent room pass with 4 time sub-chunk(s) per workgroup
#include "room_launch.h"

namespace disco_host {
DISCO_DEFINE_ROOM_LAUNCHER(launch_room_s4, 4)
}  // namespace disco_host
