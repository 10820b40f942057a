// png.cpp -- dependency-free RGBA8 PNG writer.
//
// The reference saves its images through stb_image_write (an absent submodule):
// SaveImage("out_cpu.png", ...) and stbi_write_jpg("out_gpu.png", ...),
// /root/reference/source/main.cpp:53, 80-81.  This writer produces a valid PNG
// (8-bit RGBA, filter 0 on every row) whose zlib stream uses stored blocks, so
// it needs no compression library; pixels are written exactly as given, row 0
// at the top.
#include "../../include/raymarch_hip.h"

#include <cstdio>
#include <cstring>
#include <vector>

namespace {

uint32_t crc_table[256];
bool crc_ready = false;

void crc_init() {
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_ready = true;
}

uint32_t crc_update(uint32_t crc, const uint8_t *data, size_t n) {
    for (size_t i = 0; i < n; i++) crc = crc_table[(crc ^ data[i]) & 0xffu] ^ (crc >> 8);
    return crc;
}

void put_be32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}

bool write_chunk(FILE *f, const char tag[4], const std::vector<uint8_t> &payload) {
    std::vector<uint8_t> head;
    put_be32(head, static_cast<uint32_t>(payload.size()));
    head.insert(head.end(), tag, tag + 4);
    uint32_t crc = crc_update(0xffffffffu, reinterpret_cast<const uint8_t *>(tag), 4);
    if (!payload.empty()) crc = crc_update(crc, payload.data(), payload.size());
    crc ^= 0xffffffffu;
    std::vector<uint8_t> tail;
    put_be32(tail, crc);
    if (fwrite(head.data(), 1, head.size(), f) != head.size()) return false;
    if (!payload.empty() && fwrite(payload.data(), 1, payload.size(), f) != payload.size()) return false;
    return fwrite(tail.data(), 1, tail.size(), f) == tail.size();
}

}  // namespace

// the C entry point (api.cpp: rm_write_png) adds the error text
int rm_write_png_file(const char *path, const uint8_t *rgba, uint32_t width, uint32_t height) {
    if (!path || !rgba || width == 0 || height == 0) return RM_ERR_INVALID;
    if (!crc_ready) crc_init();

    // raw scanlines: filter byte 0 + width*4 bytes
    const size_t stride = (size_t)width * 4 + 1;
    std::vector<uint8_t> raw(stride * height);
    for (uint32_t y = 0; y < height; y++) {
        raw[y * stride] = 0;
        std::memcpy(&raw[y * stride + 1], rgba + (size_t)y * width * 4, (size_t)width * 4);
    }
    // zlib container with stored (uncompressed) deflate blocks
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;  // adler32
    size_t pos = 0;
    while (pos < raw.size()) {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        const bool last = pos + n == raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(uint8_t(n & 0xff)); z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n & 0xff)); z.push_back(uint8_t((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; i++) {
            a = (a + raw[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        pos += n;
    }
    put_be32(z, (b << 16) | a);

    FILE *f = fopen(path, "wb");
    if (!f) return RM_ERR_IO;
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    bool ok = fwrite(sig, 1, 8, f) == 8;
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    ihdr.push_back(8);  // bit depth
    ihdr.push_back(6);  // colour type RGBA
    ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    ok = ok && write_chunk(f, "IHDR", ihdr);
    ok = ok && write_chunk(f, "IDAT", z);
    ok = ok && write_chunk(f, "IEND", std::vector<uint8_t>());
    ok = (fclose(f) == 0) && ok;
    return ok ? RM_OK : RM_ERR_IO;
}
