/* HDR image readers for the `envmap` emitter of the scene loader — the C++ twin of ppg_host/imageio.py: OpenEXR scanline files
 * (compression NONE / ZIPS / ZIP, half or float channels; zlib), PFM and Radiance RGBE (.hdr), each to linear RGB float, top row
 * first.  Mitsuba reads these through its Bitmap class (bitmap.cpp); only the pixel values matter here. */
#pragma once
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace ppg {

struct HdrImage {
    int width = 0, height = 0;
    std::vector<float> rgb;  // [height * width * 3]
};

inline std::vector<unsigned char> slurp(const std::string &path, const char *what) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(std::string(what) + ": file '" + path + "' not found");
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

inline float halfToFloat(uint16_t h) {
    const uint32_t sign = (uint32_t)(h >> 15) << 31, exp = (h >> 10) & 31, man = h & 1023;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {  // subnormal
            int e = -1;
            uint32_t m = man;
            do { ++e; m <<= 1; } while (!(m & 1024));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (m & 1023) << 13;
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
    else bits = sign | (exp + 127 - 15) << 23 | man << 13;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

inline HdrImage readPFM(const std::string &path) {
    const std::vector<unsigned char> buf = slurp(path, "envmap");
    size_t off = 0;
    std::string parts[4];
    for (int k = 0; k < 4; ++k) {
        while (off < buf.size() && isspace(buf[off])) ++off;
        while (off < buf.size() && !isspace(buf[off])) parts[k] += (char)buf[off++];
    }
    ++off;
    if (parts[0] != "PF" && parts[0] != "Pf") throw std::runtime_error(path + ": not a PFM file");
    HdrImage img;
    img.width = std::stoi(parts[1]); img.height = std::stoi(parts[2]);
    const int c = parts[0] == "PF" ? 3 : 1;
    const bool little = std::stof(parts[3]) < 0;
    if (buf.size() < off + (size_t)img.width * img.height * c * 4) throw std::runtime_error(path + ": truncated PFM file");
    img.rgb.resize((size_t)img.width * img.height * 3);
    for (int y = 0; y < img.height; ++y)
        for (int x = 0; x < img.width; ++x)
            for (int k = 0; k < 3; ++k) {
                const unsigned char *p = &buf[off + (((size_t)(img.height - 1 - y) * img.width + x) * c + (c == 3 ? k : 0)) * 4];  // bottom row first
                unsigned char b[4] = {p[0], p[1], p[2], p[3]};
                if (!little) { b[0] = p[3]; b[1] = p[2]; b[2] = p[1]; b[3] = p[0]; }
                memcpy(&img.rgb[((size_t)y * img.width + x) * 3 + k], b, 4);
            }
    return img;
}

inline HdrImage readHDR(const std::string &path) {
    const std::vector<unsigned char> buf = slurp(path, "envmap");
    const std::string head(buf.begin(), buf.begin() + std::min<size_t>(buf.size(), 10));
    if (head.compare(0, 10, "#?RADIANCE") != 0 && head.compare(0, 6, "#?RGBE") != 0) throw std::runtime_error(path + ": not a Radiance HDR file");
    size_t e = 0;
    while (e + 1 < buf.size() && !(buf[e] == '\n' && buf[e + 1] == '\n')) ++e;
    e += 2;
    size_t nl = e;
    while (nl < buf.size() && buf[nl] != '\n') ++nl;
    char sy[8], sx[8];
    int H, W;
    if (sscanf(std::string(buf.begin() + e, buf.begin() + nl).c_str(), "%7s %d %7s %d", sy, &H, sx, &W) != 4 || strcmp(sy, "-Y") || strcmp(sx, "+X"))
        throw std::runtime_error(path + ": only the standard -Y H +X W orientation is supported");
    size_t off = nl + 1;
    std::vector<unsigned char> row((size_t)W * 4);
    HdrImage img;
    img.width = W; img.height = H;
    img.rgb.resize((size_t)W * H * 3);
    for (int y = 0; y < H; ++y) {
        if (W >= 8 && W < 32768 && off + 4 <= buf.size() && buf[off] == 2 && buf[off + 1] == 2 && ((buf[off + 2] << 8) | buf[off + 3]) == W) {
            off += 4;
            for (int ch = 0; ch < 4; ++ch)
                for (int x = 0; x < W;) {
                    if (off >= buf.size()) throw std::runtime_error(path + ": truncated HDR file");
                    int n = buf[off++];
                    if (n > 128) { n -= 128; for (int k = 0; k < n && x < W; ++k) row[(size_t)(x++) * 4 + ch] = buf[off]; ++off; }
                    else for (int k = 0; k < n && x < W; ++k) row[(size_t)(x++) * 4 + ch] = buf[off++];
                }
        } else {
            if (off + (size_t)W * 4 > buf.size()) throw std::runtime_error(path + ": truncated HDR file");
            memcpy(row.data(), &buf[off], (size_t)W * 4);
            off += (size_t)W * 4;
        }
        for (int x = 0; x < W; ++x) {
            const int ex = row[(size_t)x * 4 + 3];
            const float f = ex > 0 ? std::ldexp(1.0f, ex - (128 + 8)) : 0.0f;
            for (int k = 0; k < 3; ++k) img.rgb[((size_t)y * W + x) * 3 + k] = (float)row[(size_t)x * 4 + k] * f;
        }
    }
    return img;
}

inline HdrImage readEXR(const std::string &path) {
    const std::vector<unsigned char> buf = slurp(path, "envmap");
    uint32_t magic = 0;
    if (buf.size() < 8) throw std::runtime_error(path + ": not an EXR file");
    memcpy(&magic, buf.data(), 4);
    if (magic != 20000630u) throw std::runtime_error(path + ": not an EXR file");
    size_t off = 8;
    std::map<std::string, std::vector<unsigned char>> attrs;
    while (buf[off] != 0) {
        std::string name, type;
        while (buf[off]) name += (char)buf[off++];
        ++off;
        while (buf[off]) type += (char)buf[off++];
        ++off;
        int32_t size;
        memcpy(&size, &buf[off], 4);
        off += 4;
        attrs[name].assign(buf.begin() + off, buf.begin() + off + size);
        off += size;
    }
    ++off;
    const int comp = attrs["compression"][0];
    if (comp != 0 && comp != 2 && comp != 3) throw std::runtime_error(path + ": EXR compression " + std::to_string(comp) + " is not supported (NONE, ZIPS, ZIP)");
    int32_t box[4];
    memcpy(box, attrs["dataWindow"].data(), 16);
    const int W = box[2] - box[0] + 1, H = box[3] - box[1] + 1;
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans;
    {
        const std::vector<unsigned char> &d = attrs["channels"];
        size_t o = 0;
        while (d[o] != 0) {
            Chan c;
            while (d[o]) c.name += (char)d[o++];
            ++o;
            int32_t t;
            memcpy(&t, &d[o], 4);
            if (t != 1 && t != 2) throw std::runtime_error(path + ": channel " + c.name + " is neither half nor float");
            c.type = t;
            o += 16;
            chans.push_back(c);
        }
    }
    const int lines = comp == 3 ? 16 : 1, nblocks = (H + lines - 1) / lines;
    size_t lineBytes = 0;
    for (auto &c : chans) lineBytes += (size_t)(c.type == 1 ? 2 : 4) * W;
    HdrImage img;
    img.width = W; img.height = H;
    img.rgb.assign((size_t)W * H * 3, 0.0f);
    std::vector<unsigned char> raw, tmp;
    for (int b = 0; b < nblocks; ++b) {
        uint64_t bo;
        memcpy(&bo, &buf[off + 8 * (size_t)b], 8);
        int32_t y, size;
        memcpy(&y, &buf[bo], 4);
        memcpy(&size, &buf[bo + 4], 4);
        const int nl = std::min(lines, box[3] - y + 1);
        const size_t expect = (size_t)nl * lineBytes;
        const unsigned char *src = &buf[bo + 8];
        if (comp != 0 && (size_t)size < expect) {
            tmp.resize(expect);
            uLongf got = (uLongf)expect;
            if (uncompress(tmp.data(), &got, src, (uLong)size) != Z_OK || got != expect) throw std::runtime_error(path + ": corrupt EXR block");
            for (size_t k = 1; k < expect; ++k) tmp[k] = (unsigned char)(tmp[k - 1] + tmp[k] - 128);  // undo the predictor ...
            raw.resize(expect);
            const size_t half = (expect + 1) / 2;
            for (size_t k = 0; k < expect; ++k) raw[k] = (k & 1) ? tmp[half + k / 2] : tmp[k / 2];  // ... and the byte interleave
            src = raw.data();
        }
        size_t p = 0;
        for (int ly = 0; ly < nl; ++ly)
            for (auto &c : chans) {
                const int k = c.name == "R" ? 0 : c.name == "G" ? 1 : c.name == "B" ? 2 : c.name == "Y" ? 3 : -1;
                for (int x = 0; x < W; ++x) {
                    float v;
                    if (c.type == 1) { uint16_t h; memcpy(&h, src + p, 2); v = halfToFloat(h); p += 2; }
                    else { memcpy(&v, src + p, 4); p += 4; }
                    float *px = &img.rgb[((size_t)(y - box[1] + ly) * W + x) * 3];
                    if (k >= 0 && k < 3) px[k] = v;
                    else if (k == 3) px[0] = px[1] = px[2] = v;
                }
            }
    }
    return img;
}

inline HdrImage readHdrImage(const std::string &path) {
    std::string ext = path.substr(path.find_last_of('.') == std::string::npos ? path.size() : path.find_last_of('.') + 1);
    for (auto &ch : ext) ch = (char)tolower(ch);
    if (ext == "exr") return readEXR(path);
    if (ext == "pfm") return readPFM(path);
    if (ext == "hdr" || ext == "pic" || ext == "rgbe") return readHDR(path);
    slurp(path, "envmap");
    throw std::runtime_error(path + ": unsupported image format (exr, pfm, hdr)");
}

}  // namespace ppg
