// PNG (8-bit, non-interlaced) read/write on zlib, PFM write, and the sRGB output stage.
// Stands in for stb_image / sutil::saveImage, which the reference uses but does not vendor
// (loader/texture_load.h:7-20, renderer/renderer.h:1291-1302).
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace hjr {

static uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static void put_be32(std::vector<unsigned char>& v, uint32_t x)
{
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16)); v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
static void chunk(std::vector<unsigned char>& out, const char* type, const unsigned char* data, size_t len)
{
    put_be32(out, (uint32_t)len);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), data, data + len);
    uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(len + 4));
    put_be32(out, crc);
}

bool write_png(const std::string& path, const uint8_t* rgba, uint32_t w, uint32_t h, bool flip_y, std::string& err)
{
    if (!rgba || w == 0 || h == 0) { err = "empty image"; return false; }
    std::vector<unsigned char> raw((size_t)h * (1 + (size_t)w * 4));
    for (uint32_t y = 0; y < h; y++) {
        uint32_t sy = flip_y ? (h - 1 - y) : y;
        unsigned char* row = &raw[(size_t)y * (1 + (size_t)w * 4)];
        row[0] = 0;
        memcpy(row + 1, rgba + (size_t)sy * w * 4, (size_t)w * 4);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = "zlib compress failed"; return false; }
    std::vector<unsigned char> out;
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    out.insert(out.end(), sig, sig + 8);
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, w); put_be32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    chunk(out, "IDAT", comp.data(), clen);
    chunk(out, "IEND", nullptr, 0);
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot open " + path + " for writing"; return false; }
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    if (!ok) err = "short write to " + path;
    return ok;
}

static int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Decodes to 8-bit RGBA (stbi_load(..., 4) semantics: grey replicated, missing alpha = 255).
bool read_png_rgba8(const std::string& path, std::vector<uint8_t>& rgba, int& w, int& h, std::string& err)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (buf.size() < 8 || memcmp(buf.data(), sig, 8) != 0) { err = path + ": not a PNG"; return false; }
    size_t pos = 8;
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte, trns;
    bool have_ihdr = false;
    while (pos + 12 <= buf.size()) {
        uint32_t len = be32(&buf[pos]);
        if (pos + 12 + (size_t)len > buf.size()) { err = path + ": truncated chunk"; return false; }
        const unsigned char* type = &buf[pos + 4];
        const unsigned char* data = &buf[pos + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) { W = be32(data); H = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; have_ihdr = true; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || W == 0 || H == 0 || W > 65535 || H > 65535) { err = path + ": bad IHDR"; return false; }
    if (interlace) { err = path + ": interlaced PNG not supported"; return false; }
    if (depth != 8 && !(depth == 16)) { err = path + ": only 8/16-bit PNG supported"; return false; }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || (ctype == 3 && depth != 8)) { err = path + ": unsupported colour type"; return false; }
    size_t bpp = (size_t)ch * (size_t)(depth / 8), stride = bpp * W;
    std::vector<unsigned char> raw((stride + 1) * H);
    uLongf rlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rlen, idat.data(), (uLong)idat.size()) != Z_OK || rlen != raw.size()) { err = path + ": zlib inflate failed"; return false; }
    std::vector<unsigned char> img(stride * H);
    for (uint32_t y = 0; y < H; y++) {
        const unsigned char* in = &raw[(stride + 1) * y];
        unsigned char* out = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        int ft = in[0];
        for (size_t x = 0; x < stride; x++) {
            int a = x >= bpp ? out[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0, v = in[1 + x];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: err = path + ": bad filter type"; return false;
            }
            out[x] = (unsigned char)v;
        }
    }
    w = (int)W; h = (int)H;
    rgba.assign((size_t)W * H * 4, 255);
    size_t step = (size_t)(depth / 8);
    for (size_t i = 0; i < (size_t)W * H; i++) {
        const unsigned char* p = &img[i * bpp];
        uint8_t* o = &rgba[i * 4];
        switch (ctype) {
        case 0: o[0] = o[1] = o[2] = p[0]; break;
        case 2: o[0] = p[0]; o[1] = p[step]; o[2] = p[2 * step]; break;
        case 3: {
            size_t k = p[0];
            if (3 * k + 2 < plte.size()) { o[0] = plte[3 * k]; o[1] = plte[3 * k + 1]; o[2] = plte[3 * k + 2]; }
            if (k < trns.size()) o[3] = trns[k];
            break;
        }
        case 4: o[0] = o[1] = o[2] = p[0]; o[3] = p[step]; break;
        case 6: o[0] = p[0]; o[1] = p[step]; o[2] = p[2 * step]; o[3] = p[3 * step]; break;
        }
    }
    return true;
}

// Radiance .hdr (RGBE, flat or new-style RLE scanlines, -Y H +X W) -> float RGBA with a = 0, as HDRTexture builds it from
// stbi_loadf (renderer/texture.h:67-88).  RGBE -> float: m * 2^(e - 136), e == 0 -> 0 (stb_image's conversion).
bool read_hdr_rgba32f(const std::string& path, std::vector<float>& rgba, int& w, int& h, std::string& err)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    size_t pos = 0;
    auto getline = [&](std::string& line) {
        line.clear();
        while (pos < buf.size() && buf[pos] != '\n') line += (char)buf[pos++];
        if (pos < buf.size()) pos++;
    };
    std::string line;
    getline(line);
    if (line.rfind("#?RADIANCE", 0) != 0 && line.rfind("#?RGBE", 0) != 0) { err = path + ": not a Radiance HDR file"; return false; }
    bool fmt = false;
    for (;;) {
        if (pos >= buf.size()) { err = path + ": truncated header"; return false; }
        getline(line);
        if (line.empty()) break;
        if (line == "FORMAT=32-bit_rle_rgbe") fmt = true;
    }
    if (!fmt) { err = path + ": unsupported HDR format"; return false; }
    getline(line);
    int W = 0, H = 0;
    if (sscanf(line.c_str(), "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0 || W > 65535 || H > 65535) { err = path + ": unsupported HDR orientation"; return false; }
    w = W; h = H;
    rgba.assign((size_t)W * H * 4, 0.0f);
    std::vector<unsigned char> scan((size_t)W * 4);
    auto put = [&](int y) {
        for (int x = 0; x < W; x++) {
            const unsigned char* p = &scan[(size_t)x * 4];
            float* o = &rgba[((size_t)y * W + x) * 4];
            if (p[3] != 0) {
                float sc = ldexpf(1.0f, (int)p[3] - 136);
                o[0] = p[0] * sc; o[1] = p[1] * sc; o[2] = p[2] * sc;
            }
        }
    };
    for (int y = 0; y < H; y++) {
        if (pos + 4 > buf.size()) { err = path + ": truncated pixel data"; return false; }
        if (W < 8 || W >= 32768 || buf[pos] != 2 || buf[pos + 1] != 2 || (buf[pos + 2] & 0x80)) { // flat scanline
            if (pos + (size_t)W * 4 > buf.size()) { err = path + ": truncated pixel data"; return false; }
            memcpy(scan.data(), &buf[pos], (size_t)W * 4);
            pos += (size_t)W * 4;
        } else {
            if ((((int)buf[pos + 2]) << 8 | buf[pos + 3]) != W) { err = path + ": bad RLE scanline width"; return false; }
            pos += 4;
            for (int c = 0; c < 4; c++) {
                int x = 0;
                while (x < W) {
                    if (pos >= buf.size()) { err = path + ": truncated RLE data"; return false; }
                    int count = buf[pos++];
                    if (count > 128) {
                        count -= 128;
                        if (pos >= buf.size() || x + count > W) { err = path + ": corrupt RLE run"; return false; }
                        unsigned char v = buf[pos++];
                        for (int k = 0; k < count; k++) scan[(size_t)(x++) * 4 + c] = v;
                    } else {
                        if (count == 0 || pos + (size_t)count > buf.size() || x + count > W) { err = path + ": corrupt RLE literal"; return false; }
                        for (int k = 0; k < count; k++) scan[(size_t)(x++) * 4 + c] = buf[pos++];
                    }
                }
            }
        }
        put(y);
    }
    return true;
}

bool write_pfm(const std::string& path, const float* rgba, uint32_t w, uint32_t h, std::string& err)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot open " + path + " for writing"; return false; }
    fprintf(f, "PF\n%u %u\n-1.0\n", w, h); // little endian; PFM rows run bottom-to-top == our row 0 at the bottom
    std::vector<float> row((size_t)w * 3);
    for (uint32_t y = 0; y < h; y++) {
        for (uint32_t x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) row[(size_t)x * 3 + c] = rgba[((size_t)y * w + x) * 4 + c];
        fwrite(row.data(), sizeof(float), row.size(), f);
    }
    fclose(f);
    return true;
}

// float4ConvertColor = toSRGB + quantizeUnsignedChar (renderer/renderer.h:73-101).
// A negative or NaN channel is undefined behaviour in the reference ((unsigned)(negative float)); defined as 0 here.
void float4_to_srgb8(const float* rgba, uint8_t* out, uint32_t n)
{
    const float invGamma = 1.0f / 2.4f;
    for (uint32_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            float col = rgba[4 * (size_t)i + c];
            float powed = std::pow(col, invGamma);
            float sr = col < 0.0031308f ? 12.92f * col : 1.055f * powed - 0.055f;
            float q = sr * 256.0f;
            uint32_t u = (q > 0.0f) ? ((q >= 4294967040.0f) ? 4294967040u : (uint32_t)q) : 0u;
            out[4 * (size_t)i + c] = (uint8_t)(u < 255u ? u : 255u);
        }
        out[4 * (size_t)i + 3] = 255;
    }
}


/* kernel/color.h:10-29.  The double literals (1.0, 0.0) promote the surrounding sub-expressions exactly as written there. */
static float tonemap_uchimura1(float x, float P, float a, float m, float l, float c, float b)
{
    float l0 = ((P - m) * l) / a;
    float S0 = m + l0;
    float S1 = m + a * l0;
    float C2 = (a * P) / (P - S1);
    float CP = -C2 / P;
    /* smoothstep(0.0, m, x) (math.h:113-116), step(m + l0, x) (math.h:118-120) */
    float sx = fmaxf(0.0f, fminf((x - 0.0f) / (m - 0.0f), 1.0f));
    float w0 = (float)(1.0 - (double)(sx * sx * (3.0f - 2.0f * sx)));
    float w2 = (float)((m + l0) < x);
    float w1 = (float)(1.0 - (double)w0 - (double)w2);
    float T = (float)((double)m * pow((double)(x / m), (double)c) + (double)b);
    float S = (float)((double)P - (double)(P - S1) * exp((double)(CP * (x - S0))));
    float L = m + a * (x - m);
    return T * w0 + L * w1 + S * w2;
}
static float tonemap_uchimura(float x) { return tonemap_uchimura1(x, 1.0f, 1.0f, 0.22f, 0.4f, 1.33f, 0.0f); } /* color.h:31-39 */
static float tonemap_aces(float x) /* color.h:55-63 */
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return fmaxf(0.0f, fminf((x * (a * x + b)) / (x * (c * x + d) + e), 1.0f));
}

void tonemap_to_srgb8(const float* rgba, uint8_t* out, uint32_t n, int mode)
{
    float px[4];
    for (uint32_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            float v = rgba[4 * (size_t)i + c];
            px[c] = mode == 1 ? tonemap_uchimura(v) : (mode == 2 ? tonemap_aces(v) : v);
        }
        px[3] = 1.0f;
        float4_to_srgb8(px, out + 4 * (size_t)i, 1);
    }
}

} // namespace hjr
