// rt_internal.h -- shared by the host-side translation units of librt_hip.so (not part of the ABI).
#pragma once
#include <string>

namespace rt {
// Records the calling thread's error text (what rt_last_error() returns) and hands back `code`.
int set_error(const std::string &message, int code);
}

#include "../../include/rt_hip.h"

namespace rt {
// rt_scene_create for several devices at once: the description is assembled (octrees built, boxes
// formed) once, then every device gets its copy, the copies made side by side by one host thread
// per device.  out[i] = the handle on devices[i]; on failure no handle is left behind.
int create_scenes(const rt_scene_desc *desc, const int *devices, int n, rt_scene **out);
}
