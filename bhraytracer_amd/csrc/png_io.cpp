// png_io.cpp — 8-bit PNG writer and a PNG reader (every colour type and bit depth, not interlaced) on top of zlib.
// Replaces the vendored lodepng the reference calls at Scenes/scene.h:634-644
// (lodepng::encode, LCT_RGB / LCT_GREY, 8 bit) and Textures/Texture.cpp:76
// (lodepng::decode to LCT_RGB).  Pixel bytes are what matters for parity, not the
// compressed stream: any conforming decoder returns the same bytes.
#include "png_io.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

namespace bhrt {

static void put32(std::vector<uint8_t> &o, uint32_t v)
{
    o.push_back(v >> 24); o.push_back(v >> 16); o.push_back(v >> 8); o.push_back(v);
}
static void chunk(std::vector<uint8_t> &o, const char *type, const uint8_t *data, size_t n)
{
    put32(o, (uint32_t)n);
    size_t b = o.size();
    o.insert(o.end(), type, type + 4);
    if (n) o.insert(o.end(), data, data + n);
    put32(o, (uint32_t)crc32(0, o.data() + b, (uInt)(n + 4)));
}

bool SavePng(const char *path, const uint8_t *pixels, int w, int h, int comps)
{
    if (!(comps == 1 || comps == 3) || w <= 0 || h <= 0) return false;
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * ((size_t)w * comps + 1));
    for (int y = 0; y < h; y++) {
        raw.push_back(0); // filter: none
        raw.insert(raw.end(), pixels + (size_t)y * w * comps, pixels + (size_t)(y + 1) * w * comps);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    std::vector<uint8_t> o = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    uint8_t ihdr[13];
    ihdr[0] = w >> 24; ihdr[1] = w >> 16; ihdr[2] = w >> 8; ihdr[3] = w;
    ihdr[4] = h >> 24; ihdr[5] = h >> 16; ihdr[6] = h >> 8; ihdr[7] = h;
    ihdr[8] = 8; ihdr[9] = comps == 3 ? 2 : 0; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk(o, "IHDR", ihdr, 13);
    chunk(o, "IDAT", comp.data(), clen);
    chunk(o, "IEND", nullptr, 0);
    FILE *fp = fopen(path, "wb");
    if (!fp) return false;
    bool ok = fwrite(o.data(), 1, o.size(), fp) == o.size();
    fclose(fp);
    return ok;
}

static int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool LoadPngRgb(const char *path, std::vector<uint8_t> &rgb, int &w, int &h)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return false;
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(fp);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) return false;
    size_t p = 8;
    int bitdepth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    w = h = 0;
    while (p + 12 <= file.size()) {
        uint32_t len = (file[p] << 24) | (file[p + 1] << 16) | (file[p + 2] << 8) | file[p + 3];
        const char *type = (const char *)&file[p + 4];
        const uint8_t *d = &file[p + 8];
        if (p + 12 + len > file.size()) return false;
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = (d[0] << 24) | (d[1] << 16) | (d[2] << 8) | d[3];
            h = (d[4] << 24) | (d[5] << 16) | (d[6] << 8) | d[7];
            bitdepth = d[8]; ctype = d[9]; interlace = d[12];
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(type, "IEND", 4)) break;
        p += 12 + len;
    }
    // every colour type and bit depth of the PNG specification, as lodepng::decode(..., LCT_RGB, 8) (Texture.cpp:76) accepts them;
    // Adam7-interlaced files are not read (the caller warns like for any texture that fails to load)
    if (w <= 0 || h <= 0 || w > 32768 || h > 32768 || (size_t)w * h > ((size_t)1 << 28) || interlace != 0) return false;
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) return false;
    const bool depth_ok = ctype == 0 ? (bitdepth == 1 || bitdepth == 2 || bitdepth == 4 || bitdepth == 8 || bitdepth == 16)
                        : ctype == 3 ? (bitdepth == 1 || bitdepth == 2 || bitdepth == 4 || bitdepth == 8) : (bitdepth == 8 || bitdepth == 16);
    if (!depth_ok) return false;
    const size_t bits_px = (size_t)ch * bitdepth, bpp = bits_px >= 8 ? bits_px / 8 : 1; // filter distance in bytes
    const size_t stride = ((size_t)w * bits_px + 7) / 8;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rl = (uLongf)raw.size();
    if (uncompress(raw.data(), &rl, idat.data(), (uLong)idat.size()) != Z_OK || rl != raw.size()) return false;
    std::vector<uint8_t> img(stride * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *in = &raw[(stride + 1) * y];
        int f = in[0];
        in++;
        uint8_t *out = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; x++) {
            int a = x >= bpp ? out[x - bpp] : 0;
            int b = up ? up[x] : 0;
            int c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = in[x];
            switch (f) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return false;
            }
            out[x] = (uint8_t)v;
        }
    }
    rgb.resize((size_t)w * h * 3);
    // sample k of row y as an 8-bit value: 16-bit samples keep their high byte, samples below 8 bits are scaled to 0..255
    // (grey) or kept as palette indices — the conversions lodepng applies
    auto sample = [&](const uint8_t *row, size_t k, bool scale) -> unsigned {
        if (bitdepth == 8) return row[k];
        if (bitdepth == 16) return row[2 * k];
        const size_t bit = k * bitdepth;
        const unsigned v = (row[bit >> 3] >> (8 - bitdepth - (bit & 7))) & ((1u << bitdepth) - 1u);
        return scale ? v * 255u / ((1u << bitdepth) - 1u) : v;
    };
    for (int y = 0; y < h; y++) {
        const uint8_t *row = &img[stride * y];
        for (int x = 0; x < w; x++) {
            uint8_t *o = &rgb[((size_t)y * w + x) * 3];
            switch (ctype) {
            case 0: case 4: o[0] = o[1] = o[2] = (uint8_t)sample(row, (size_t)x * ch, true); break;
            case 2: case 6: o[0] = (uint8_t)sample(row, (size_t)x * ch, true); o[1] = (uint8_t)sample(row, (size_t)x * ch + 1, true); o[2] = (uint8_t)sample(row, (size_t)x * ch + 2, true); break;
            case 3: {
                const size_t k = (size_t)sample(row, (size_t)x, false) * 3;
                if (k + 2 < plte.size()) { o[0] = plte[k]; o[1] = plte[k + 1]; o[2] = plte[k + 2]; }
                else o[0] = o[1] = o[2] = 0;
            } break;
            }
        }
    }
    return true;
}

} // namespace bhrt
