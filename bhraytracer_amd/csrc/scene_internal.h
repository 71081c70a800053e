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
// No C++ exception crosses the C ABI: every `int bhrt_*` entry point is a function-try-block that ends in this handler
// (std::bad_alloc from a file-sized allocation, std::length_error, ...): sets bhrt_last_error() and returns an error code.
int AbiException();
void DestroyDeviceState(DeviceState *d); // defined in the HIP TU
} // namespace bhrt

struct bhrt_scene {
    bhrt::FlatScene flat;
    uint32_t n_triangles = 0, n_bvh_nodes = 0, max_bvh_depth = 0;
    bhrt::DeviceState *dev = nullptr;
};
