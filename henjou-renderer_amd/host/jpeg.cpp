// Baseline JPEG reader for material textures.  The reference decodes textures with stb_image's stbi_load(..., STBI_rgb_alpha)
// (renderer/texture.h:22-38, via loader/texture_load.h:7-20); stb_image is an un-vendored dependency (it ships inside the OptiX
// SDK's support/tinygltf), so this restates ITU-T T.81 sequential Huffman decoding together with stb_image's published
// arithmetic choices, which decide the texel values: 12-bit fixed-point "islow" IDCT with a DC-only column shortcut,
// triangle-filter ("fancy") chroma upsampling with round-half-up, and the 20-bit fixed-point YCbCr -> RGB step.  No stb_image
// binary exists here to compare with: texel parity with the reference is unpinned; tests compare against an
// independent decoder within 3 LSB.  Supported: SOF0 / SOF1 (8-bit, Huffman), 1 or 3 components, sampling factors 1..4,
// restart intervals, Adobe APP14 transform 0 (RGB).  Progressive / arithmetic / lossless / 12-bit streams are rejected.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace hjr {
namespace {

const uint8_t kZigzag[64 + 15] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63 // a corrupt run cannot index past the block
};

struct Huff {
    bool present = false;
    uint8_t fast[512];    // code of <= 9 bits -> symbol index, 255 = longer
    uint16_t code[256];
    uint8_t values[256];
    uint8_t size[257];
    uint32_t maxcode[18]; // left-justified to 16 bits
    int delta[17];
    bool build(const uint8_t* count) // T.81 Annex C
    {
        int k = 0;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < count[i]; j++) { if (k >= 256) return false; size[k++] = (uint8_t)(i + 1); }
        size[k] = 0;
        int code_ = 0;
        k = 0;
        for (int j = 1; j <= 16; j++) {
            delta[j] = k - code_;
            if (size[k] == j) {
                while (size[k] == j) code[k++] = (uint16_t)(code_++);
                if (code_ - 1 >= (1 << j)) return false;
            }
            maxcode[j] = (uint32_t)code_ << (16 - j);
            code_ <<= 1;
        }
        maxcode[17] = 0xffffffffu;
        memset(fast, 255, sizeof(fast));
        for (int i = 0; i < k; i++) {
            int s = size[i];
            if (s <= 9) {
                int c = code[i] << (9 - s), m = 1 << (9 - s);
                for (int j = 0; j < m; j++) fast[c + j] = (uint8_t)i;
            }
        }
        present = true;
        return true;
    }
};

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;
    std::vector<uint8_t> data;
};

struct Jpeg {
    const uint8_t* p = nullptr;
    const uint8_t* end = nullptr;
    Huff hdc[4], hac[4];
    uint16_t dequant[4][64];
    bool have_q[4] = { false, false, false, false };
    int img_x = 0, img_y = 0, ncomp = 0, hmax = 1, vmax = 1, mcu_x = 0, mcu_y = 0, mcu_w = 0, mcu_h = 0;
    Comp comp[3];
    int restart_interval = 0, todo = 0;
    int app14_transform = -1;
    // entropy-coded segment reader
    uint32_t code_buffer = 0;
    int code_bits = 0;
    uint8_t marker = 0xff; // 0xff = none pending
    bool nomore = false;
    std::string err;

    int get8() { return p < end ? *p++ : 0; }
    int get16() { int a = get8(); return (a << 8) | get8(); }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }

    void grow()
    {
        do {
            unsigned b = nomore ? 0u : (unsigned)get8();
            if (b == 0xff) {
                int c = get8();
                while (c == 0xff) c = get8();
                if (c != 0) { marker = (uint8_t)c; nomore = true; return; }
            }
            code_buffer |= b << (24 - code_bits);
            code_bits += 8;
        } while (code_bits <= 24);
    }
    int decode(const Huff& h)
    {
        if (code_bits < 16) grow();
        int c = (int)((code_buffer >> 23) & 511);
        int k = h.fast[c];
        if (k < 255) {
            int s = h.size[k];
            if (s > code_bits) return -1;
            code_buffer <<= s;
            code_bits -= s;
            return h.values[k];
        }
        uint32_t temp = code_buffer >> 16;
        int s;
        for (s = 10;; s++)
            if (temp < h.maxcode[s]) break;
        if (s == 17) { code_bits -= 16; return -1; }
        if (s > code_bits) return -1;
        c = (int)((code_buffer >> (32 - s)) & ((1u << s) - 1u)) + h.delta[s];
        if (c < 0 || c > 255) return -1;
        code_buffer <<= s;
        code_bits -= s;
        return h.values[c];
    }
    int extend_receive(int n) // T.81 F.2.2.1 EXTEND(RECEIVE(n), n)
    {
        if (n == 0) return 0;
        if (code_bits < n) grow();
        if (code_bits < n) return 0;
        const uint32_t v = code_buffer >> (32 - n);
        code_buffer <<= n;
        code_bits -= n;
        return (v >> (n - 1)) ? (int)v : (int)v - (int)((1u << n) - 1u);
    }
    bool decode_block(short data[64], Comp& c)
    {
        memset(data, 0, 64 * sizeof(short));
        const Huff& hd = hdc[c.hd];
        const Huff& ha = hac[c.ha];
        const uint16_t* dq = dequant[c.tq];
        int t = decode(hd);
        if (t < 0 || t > 15) return fail("bad Huffman code");
        int diff = t ? extend_receive(t) : 0;
        int dc = c.dc_pred + diff;
        if (dc < -65536 || dc > 65535) return fail("bad DC difference");
        c.dc_pred = dc;
        data[0] = (short)((long long)dc * dq[0]);
        int k = 1;
        do {
            int rs = decode(ha);
            if (rs < 0) return fail("bad Huffman code");
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break;
                k += 16;
            } else {
                k += r;
                int zig = kZigzag[k++];
                data[zig] = (short)((long long)extend_receive(s) * dq[zig]);
            }
        } while (k < 64);
        return true;
    }
    void reset_entropy()
    {
        code_bits = 0; code_buffer = 0; nomore = false; marker = 0xff;
        for (int i = 0; i < 3; i++) comp[i].dc_pred = 0;
        todo = restart_interval ? restart_interval : 0x7fffffff;
    }
};

inline int f2f(float x) { return (int)(x * 4096 + 0.5); }
inline uint8_t clamp8(long long x) { return x < 0 ? 0 : (x > 255 ? 255 : (uint8_t)x); }

#define HJR_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                   \
    long long t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3; /* 64-bit: corrupt coefficients must not overflow */ \
    p2 = s2; p3 = s6;                                                                  \
    p1 = (p2 + p3) * f2f(0.5411961f);                                                  \
    t2 = p1 + p3 * f2f(-1.847759065f);                                                 \
    t3 = p1 + p2 * f2f(0.765366865f);                                                  \
    p2 = s0; p3 = s4;                                                                  \
    t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                                      \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                            \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                            \
    p5 = (p3 + p4) * f2f(1.175875602f);                                                \
    t0 = t0 * f2f(0.298631336f); t1 = t1 * f2f(2.053119869f);                          \
    t2 = t2 * f2f(3.072711026f); t3 = t3 * f2f(1.501321110f);                          \
    p1 = p5 + p1 * f2f(-0.899976223f); p2 = p5 + p2 * f2f(-2.562915447f);              \
    p3 = p3 * f2f(-1.961570560f); p4 = p4 * f2f(-0.390180644f);                        \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

void idct_block(uint8_t* out, int stride, const short d[64])
{
    long long val[64];
    for (int i = 0; i < 8; i++) { // columns
        const short* s = d + i;
        long long* v = val + i;
        if (s[8] == 0 && s[16] == 0 && s[24] == 0 && s[32] == 0 && s[40] == 0 && s[48] == 0 && s[56] == 0) {
            long long dcterm = s[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
        } else {
            HJR_IDCT_1D(s[0], s[8], s[16], s[24], s[32], s[40], s[48], s[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    for (int i = 0; i < 8; i++) { // rows, + 128 level shift, to 8 bits
        const long long* v = val + 8 * i;
        uint8_t* o = out + (size_t)i * stride;
        HJR_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
        o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
        o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
        o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
    }
}

// chroma upsampling of one output row
inline uint8_t div4(int x) { return (uint8_t)(x >> 2); }
inline uint8_t div16(int x) { return (uint8_t)(x >> 4); }
void resample_row(uint8_t* out, const uint8_t* near_, const uint8_t* far_, int w, int hs, int vs)
{
    if (hs == 1 && vs == 1) { memcpy(out, near_, (size_t)w); return; }
    if (hs == 1 && vs == 2) { for (int i = 0; i < w; i++) out[i] = div4(3 * near_[i] + far_[i] + 2); return; }
    if (hs == 2 && vs == 1) {
        const uint8_t* in = near_;
        if (w == 1) { out[0] = out[1] = in[0]; return; }
        out[0] = in[0];
        out[1] = div4(in[0] * 3 + in[1] + 2);
        int i;
        for (i = 1; i < w - 1; i++) {
            int n = 3 * in[i] + 2;
            out[i * 2 + 0] = div4(n + in[i - 1]);
            out[i * 2 + 1] = div4(n + in[i + 1]);
        }
        out[i * 2 + 0] = div4(in[w - 2] * 3 + in[w - 1] + 2); // stb_image's weighting of the last pair, kept as published
        out[i * 2 + 1] = in[w - 1];
        return;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = div4(3 * near_[0] + far_[0] + 2); return; }
        int t1 = 3 * near_[0] + far_[0];
        out[0] = div4(t1 + 2);
        for (int i = 1; i < w; i++) {
            int t0 = t1;
            t1 = 3 * near_[i] + far_[i];
            out[i * 2 - 1] = div16(3 * t0 + t1 + 8);
            out[i * 2] = div16(3 * t1 + t0 + 8);
        }
        out[w * 2 - 1] = div4(t1 + 2);
        return;
    }
    for (int i = 0; i < w; i++) // other ratios: nearest
        for (int j = 0; j < hs; j++) out[i * hs + j] = near_[i];
}

inline int float2fixed(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }

} // namespace

bool read_jpeg_rgba8(const std::vector<uint8_t>& file, std::vector<uint8_t>& rgba, int& w, int& h, std::string& err)
{
    Jpeg z;
    z.p = file.data();
    z.end = file.data() + file.size();
    if (file.size() < 4 || z.get8() != 0xff || z.get8() != 0xd8) { err = "not a JPEG file"; return false; }
    bool have_frame = false, done = false;
    auto next_marker = [&]() -> int {
        if (z.marker != 0xff) { int m = z.marker; z.marker = 0xff; return m; }
        int x = z.get8();
        if (x != 0xff) return -1;
        while (x == 0xff) x = z.get8();
        return x;
    };
    while (!done) {
        int m = next_marker();
        while (m == -1 && z.p < z.end) m = next_marker(); // garbage between segments
        if (m < 0) { err = "JPEG: no end-of-image marker"; return false; }
        switch (m) {
        case 0xd9: done = true; break;
        case 0xdb: { // DQT
            int L = z.get16() - 2;
            while (L > 0) {
                int q = z.get8(), p16 = q >> 4, t = q & 15;
                if ((p16 != 0 && p16 != 1) || t > 3) { err = "JPEG: bad DQT"; return false; }
                for (int i = 0; i < 64; i++) z.dequant[t][kZigzag[i]] = (uint16_t)(p16 ? z.get16() : z.get8());
                z.have_q[t] = true;
                L -= p16 ? 129 : 65;
            }
            if (L != 0) { err = "JPEG: bad DQT length"; return false; }
            break;
        }
        case 0xc4: { // DHT
            int L = z.get16() - 2;
            while (L > 0) {
                int q = z.get8(), tc = q >> 4, th = q & 15;
                uint8_t count[16];
                int n = 0;
                if (tc > 1 || th > 3) { err = "JPEG: bad DHT"; return false; }
                for (int i = 0; i < 16; i++) { count[i] = (uint8_t)z.get8(); n += count[i]; }
                if (n > 256) { err = "JPEG: bad DHT"; return false; }
                Huff& H = tc == 0 ? z.hdc[th] : z.hac[th];
                if (!H.build(count)) { err = "JPEG: bad code lengths"; return false; }
                for (int i = 0; i < n; i++) H.values[i] = (uint8_t)z.get8();
                L -= 17 + n;
            }
            if (L != 0) { err = "JPEG: bad DHT length"; return false; }
            break;
        }
        case 0xdd: if (z.get16() != 4) { err = "JPEG: bad DRI"; return false; } z.restart_interval = z.get16(); break;
        case 0xc0: case 0xc1: { // SOF0 / SOF1
            if (have_frame) { err = "JPEG: two frames"; return false; }
            int L = z.get16();
            if (z.get8() != 8) { err = "JPEG: only 8-bit samples are supported"; return false; }
            z.img_y = z.get16(); z.img_x = z.get16();
            z.ncomp = z.get8();
            if (z.img_x <= 0 || z.img_y <= 0) { err = "JPEG: empty image"; return false; }
            if (z.img_x > 16384 || z.img_y > 16384) { err = "JPEG: image larger than 16384 x 16384"; return false; }
            if (z.ncomp != 1 && z.ncomp != 3) { err = "JPEG: 1 or 3 components expected"; return false; }
            if (L != 8 + 3 * z.ncomp) { err = "JPEG: bad SOF length"; return false; }
            for (int i = 0; i < z.ncomp; i++) {
                Comp& c = z.comp[i];
                c.id = z.get8();
                int q = z.get8();
                c.h = q >> 4; c.v = q & 15;
                c.tq = z.get8();
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) { err = "JPEG: bad component"; return false; }
                z.hmax = c.h > z.hmax ? c.h : z.hmax;
                z.vmax = c.v > z.vmax ? c.v : z.vmax;
            }
            for (int i = 0; i < z.ncomp; i++)
                if (z.hmax % z.comp[i].h || z.vmax % z.comp[i].v) { err = "JPEG: fractional sampling ratio"; return false; }
            z.mcu_w = z.hmax * 8; z.mcu_h = z.vmax * 8;
            z.mcu_x = (z.img_x + z.mcu_w - 1) / z.mcu_w;
            z.mcu_y = (z.img_y + z.mcu_h - 1) / z.mcu_h;
            for (int i = 0; i < z.ncomp; i++) {
                Comp& c = z.comp[i];
                c.x = (z.img_x * c.h + z.hmax - 1) / z.hmax;
                c.y = (z.img_y * c.v + z.vmax - 1) / z.vmax;
                c.w2 = z.mcu_x * c.h * 8;
                c.h2 = z.mcu_y * c.v * 8;
                c.data.assign((size_t)c.w2 * c.h2, 0);
            }
            have_frame = true;
            break;
        }
        case 0xc2: err = "progressive JPEG is not supported (re-encode as baseline)"; return false;
        case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
            err = "JPEG: unsupported coding process (only baseline / extended sequential Huffman)"; return false;
        case 0xda: { // SOS + entropy-coded data
            if (!have_frame) { err = "JPEG: scan before frame header"; return false; }
            int L = z.get16();
            int ns = z.get8();
            if (ns != z.ncomp || L != 6 + 2 * ns) { err = "JPEG: only one interleaved scan over all components is supported"; return false; }
            int order[3] = { 0, 0, 0 };
            for (int i = 0; i < ns; i++) {
                int id = z.get8(), q = z.get8(), which = -1;
                for (int k = 0; k < z.ncomp; k++) if (z.comp[k].id == id) which = k;
                if (which < 0) { err = "JPEG: bad scan component"; return false; }
                z.comp[which].hd = q >> 4; z.comp[which].ha = q & 15;
                if (z.comp[which].hd > 3 || z.comp[which].ha > 3) { err = "JPEG: bad table index"; return false; }
                if (!z.hdc[z.comp[which].hd].present || !z.hac[z.comp[which].ha].present || !z.have_q[z.comp[which].tq]) { err = "JPEG: missing table"; return false; }
                order[i] = which;
            }
            if (z.get8() != 0) { err = "JPEG: bad spectral selection"; return false; }
            z.get8();
            if (z.get8() != 0) { err = "JPEG: bad successive approximation"; return false; }
            z.reset_entropy();
            short data[64];
            if (ns == 1) { // non-interleaved: blocks of the component's own pixel grid
                Comp& c = z.comp[order[0]];
                const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
                for (int j = 0; j < bh; j++)
                    for (int i = 0; i < bw; i++) {
                        if (!z.decode_block(data, c)) { err = "JPEG: " + z.err; return false; }
                        idct_block(&c.data[(size_t)c.w2 * j * 8 + (size_t)i * 8], c.w2, data);
                        if (--z.todo <= 0) {
                            if (z.code_bits < 24) z.grow();
                            if (!(z.marker >= 0xd0 && z.marker <= 0xd7)) { j = bh; break; }
                            z.reset_entropy();
                        }
                    }
            } else {
                bool stop = false;
                for (int j = 0; j < z.mcu_y && !stop; j++)
                    for (int i = 0; i < z.mcu_x && !stop; i++) {
                        for (int k = 0; k < ns; k++) {
                            Comp& c = z.comp[order[k]];
                            for (int y = 0; y < c.v; y++)
                                for (int x = 0; x < c.h; x++) {
                                    const int x2 = (i * c.h + x) * 8, y2 = (j * c.v + y) * 8;
                                    if (!z.decode_block(data, c)) { err = "JPEG: " + z.err; return false; }
                                    idct_block(&c.data[(size_t)c.w2 * y2 + x2], c.w2, data);
                                }
                        }
                        if (--z.todo <= 0) {
                            if (z.code_bits < 24) z.grow();
                            if (!(z.marker >= 0xd0 && z.marker <= 0xd7)) { stop = true; break; }
                            z.reset_entropy();
                        }
                    }
            }
            if (z.marker == 0xff) { // skip to the next marker
                while (z.p < z.end) {
                    int x = z.get8();
                    if (x == 0xff) {
                        int y = z.get8();
                        while (y == 0xff) y = z.get8();
                        if (y != 0 && !(y >= 0xd0 && y <= 0xd7)) { z.marker = (uint8_t)y; break; }
                    }
                }
                if (z.marker == 0xff) done = true; // truncated file: keep what was decoded, like stb_image
            }
            break;
        }
        case 0xee: { // APP14 "Adobe"
            int L = z.get16() - 2;
            if (L >= 12) {
                static const char tag[6] = { 'A', 'd', 'o', 'b', 'e', 0 };
                bool ok = true;
                for (int i = 0; i < 6; i++) if (z.get8() != (uint8_t)tag[i]) ok = false;
                L -= 6;
                if (ok) { z.get8(); z.get16(); z.get16(); z.app14_transform = z.get8(); L -= 6; }
            }
            if (z.p + L > z.end) { err = "JPEG: truncated segment"; return false; }
            z.p += L;
            break;
        }
        default:
            if ((m >= 0xe0 && m <= 0xef) || m == 0xfe || (m >= 0xd0 && m <= 0xd7) || m == 0x01) {
                if (m >= 0xd0 && m <= 0xd7) break;
                if (m == 0x01) break;
                int L = z.get16();
                if (L < 2 || z.p + (L - 2) > z.end) { err = "JPEG: truncated segment"; return false; }
                z.p += L - 2;
            } else { err = "JPEG: unknown marker"; return false; }
        }
    }
    if (!have_frame) { err = "JPEG: no frame"; return false; }

    // output stage: upsample each component to full resolution row by row, then to RGBA
    w = z.img_x; h = z.img_y;
    rgba.assign((size_t)w * h * 4, 255);
    struct Res { const uint8_t* line0; const uint8_t* line1; int hs, vs, w_lores, ystep, ypos; std::vector<uint8_t> buf; } rs[3];
    for (int k = 0; k < z.ncomp; k++) {
        Res& r = rs[k];
        r.hs = z.hmax / z.comp[k].h; r.vs = z.vmax / z.comp[k].v;
        r.ystep = r.vs >> 1;
        r.w_lores = (w + r.hs - 1) / r.hs;
        r.ypos = 0;
        r.line0 = r.line1 = z.comp[k].data.data();
        r.buf.assign((size_t)w + 8 * 4 + 3, 0);
    }
    const bool rgb_direct = z.ncomp == 3 && z.app14_transform == 0;
    for (int j = 0; j < h; j++) {
        const uint8_t* row[3] = { nullptr, nullptr, nullptr };
        for (int k = 0; k < z.ncomp; k++) {
            Res& r = rs[k];
            const bool y_bot = r.ystep >= (r.vs >> 1);
            resample_row(r.buf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
            row[k] = r.buf.data();
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < z.comp[k].y) r.line1 += z.comp[k].w2;
            }
        }
        uint8_t* o = &rgba[(size_t)j * w * 4];
        if (z.ncomp == 1) for (int i = 0; i < w; i++) { o[4 * i] = o[4 * i + 1] = o[4 * i + 2] = row[0][i]; }
        else if (rgb_direct) for (int i = 0; i < w; i++) { o[4 * i] = row[0][i]; o[4 * i + 1] = row[1][i]; o[4 * i + 2] = row[2][i]; }
        else
            for (int i = 0; i < w; i++) {
                const int y_fixed = (row[0][i] << 20) + (1 << 19);
                const int cb = row[1][i] - 128, cr = row[2][i] - 128;
                int r = y_fixed + cr * float2fixed(1.40200f);
                int g = y_fixed + (cr * -float2fixed(0.71414f)) + ((cb * -float2fixed(0.34414f)) & 0xffff0000);
                int b = y_fixed + cb * float2fixed(1.77200f);
                r >>= 20; g >>= 20; b >>= 20;
                o[4 * i] = clamp8(r); o[4 * i + 1] = clamp8(g); o[4 * i + 2] = clamp8(b);
            }
    }
    return true;
}

// Texture(filename, type) front end: PNG or baseline JPEG by signature (stbi_load accepts both, renderer/texture.h:22-38)
bool read_png_rgba8(const std::string& path, std::vector<uint8_t>& rgba, int& w, int& h, std::string& err);
bool read_image_rgba8(const std::string& path, std::vector<uint8_t>& rgba, int& w, int& h, std::string& err)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    unsigned char sig[2] = { 0, 0 };
    const size_t got = fread(sig, 1, 2, f);
    if (got == 2 && sig[0] == 0xff && sig[1] == 0xd8) {
        std::vector<uint8_t> file;
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (n <= 0) { fclose(f); err = "cannot read " + path; return false; }
        file.resize((size_t)n);
        const size_t rd = fread(file.data(), 1, (size_t)n, f);
        fclose(f);
        if (rd != (size_t)n) { err = "cannot read " + path; return false; }
        return read_jpeg_rgba8(file, rgba, w, h, err);
    }
    fclose(f);
    return read_png_rgba8(path, rgba, w, h, err);
}

} // namespace hjr
