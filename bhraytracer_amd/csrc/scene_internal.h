// scene_internal.h — the opaque bhrt_scene handle behind include/bhrt.h
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "bhrt.h"
#include "scene_host.h"

namespace bhrt {
struct DeviceState; // device_state.h (HIP side)
void SetError(const std::string &msg);
void DestroyDeviceState(DeviceState *d); // defined in the HIP TU
} // namespace bhrt

struct bhrt_scene {
    bhrt::FlatScene flat;
    uint32_t n_triangles = 0, n_bvh_nodes = 0, max_bvh_depth = 0;
    bhrt::DeviceState *dev = nullptr;
};
