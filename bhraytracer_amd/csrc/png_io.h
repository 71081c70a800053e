// png_io.h — see png_io.cpp
#pragma once
#include <stdint.h>
#include <vector>

namespace bhrt {
// 8-bit greyscale (comps=1) or RGB (comps=3), row-major, like RenderImage::SavePNG (scene.h:634-644)
bool SavePng(const char *path, const uint8_t *pixels, int w, int h, int comps);
// decode to 8-bit RGB like lodepng::decode(..., LCT_RGB) at Texture.cpp:76
bool LoadPngRgb(const char *path, std::vector<uint8_t> &rgb, int &w, int &h);
} // namespace bhrt
