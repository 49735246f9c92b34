// Minimal PNG reader/writer (8-bit RGB / RGBA, non-interlaced) over zlib: enough for the block textures the registry
// loads (the reference uses the `image` crate, src/graphics/texture_array.rs:88-93) and for dumping rendered frames.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace vx {

struct Image8 {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;  // row 0 = top, as stored in the file
};

inline uint32_t png_be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

inline bool png_read(const std::string& path, Image8& out, std::string& err) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<uint8_t> d;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    std::fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 33 || std::memcmp(d.data(), sig, 8) != 0) { err = path + ": not a PNG"; return false; }
    uint32_t w = 0, h = 0, channels = 0;
    std::vector<uint8_t> idat;
    for (size_t pos = 8; pos + 12 <= d.size();) {
        const uint32_t len = png_be32(&d[pos]);
        const char* type = reinterpret_cast<const char*>(&d[pos + 4]);
        if (pos + 12 + len > d.size()) break;
        const uint8_t* body = &d[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            w = png_be32(body); h = png_be32(body + 4);
            const uint8_t depth = body[8], ctype = body[9], interlace = body[12];
            if (depth != 8 || interlace != 0 || (ctype != 2 && ctype != 6)) { err = path + ": only 8-bit RGB/RGBA non-interlaced PNGs are supported"; return false; }
            channels = ctype == 6 ? 4 : 3;
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (!w || !h || !channels) { err = path + ": missing IHDR"; return false; }
    const size_t stride = size_t(w) * channels;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = uLongf(raw.size());
    if (uncompress(raw.data(), &raw_len, idat.data(), uLong(idat.size())) != Z_OK || raw_len != raw.size()) { err = path + ": inflate failed"; return false; }
    std::vector<uint8_t> pix(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t filter = raw[(stride + 1) * y];
        const uint8_t* src = &raw[(stride + 1) * y + 1];
        uint8_t* cur = &pix[stride * y];
        const uint8_t* up = y ? &pix[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= channels ? cur[i - channels] : 0, b = up ? up[i] : 0, c = (up && i >= channels) ? up[i - channels] : 0;
            int pred = 0;
            switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) / 2; break;
                case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: err = path + ": bad filter"; return false;
            }
            cur[i] = uint8_t(src[i] + pred);
        }
    }
    out.width = w; out.height = h;
    out.rgba.resize(size_t(w) * h * 4);
    for (size_t i = 0; i < size_t(w) * h; ++i) {
        out.rgba[i * 4 + 0] = pix[i * channels + 0];
        out.rgba[i * 4 + 1] = pix[i * channels + 1];
        out.rgba[i * 4 + 2] = pix[i * channels + 2];
        out.rgba[i * 4 + 3] = channels == 4 ? pix[i * channels + 3] : 255;
    }
    return true;
}

inline bool png_write(const std::string& path, const Image8& img) {
    std::vector<uint8_t> raw((size_t(img.width) * 4 + 1) * img.height);
    for (uint32_t y = 0; y < img.height; ++y) {
        raw[(size_t(img.width) * 4 + 1) * y] = 0;
        std::memcpy(&raw[(size_t(img.width) * 4 + 1) * y + 1], &img.rgba[size_t(img.width) * 4 * y], size_t(img.width) * 4);
    }
    uLongf clen = compressBound(uLong(raw.size()));
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), uLong(raw.size()), 6) != Z_OK) return false;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
        uint8_t hdr[8] = {uint8_t(len >> 24), uint8_t(len >> 16), uint8_t(len >> 8), uint8_t(len), uint8_t(type[0]), uint8_t(type[1]), uint8_t(type[2]), uint8_t(type[3])};
        std::fwrite(hdr, 1, 8, f);
        if (len) std::fwrite(data, 1, len, f);
        uLong crc = crc32(0, hdr + 4, 4);
        if (len) crc = crc32(crc, data, len);
        const uint8_t c[4] = {uint8_t(crc >> 24), uint8_t(crc >> 16), uint8_t(crc >> 8), uint8_t(crc)};
        std::fwrite(c, 1, 4, f);
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::fwrite(sig, 1, 8, f);
    uint8_t ihdr[13] = {uint8_t(img.width >> 24), uint8_t(img.width >> 16), uint8_t(img.width >> 8), uint8_t(img.width),
                        uint8_t(img.height >> 24), uint8_t(img.height >> 16), uint8_t(img.height >> 8), uint8_t(img.height), 8, 6, 0, 0, 0};
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), uint32_t(clen));
    chunk("IEND", nullptr, 0);
    std::fclose(f);
    return true;
}

}  // namespace vx
