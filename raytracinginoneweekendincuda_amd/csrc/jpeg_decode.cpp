// jpeg_decode.cpp -- the host-side image ingest of the reference: RtwImage::Load (R/RtwImage.h:51-87) reads earthmap.jpg with
// stb_image's stbi_loadf (R/StbImageImpl.cpp, R/external/stb_image.h v2.30) and hands ImageTexture bytes.  This file
// restates the part of that pipeline a sequential Huffman JPEG goes through, so that the library (and rtow) can read the
// reference's texture file itself and arrive at the SAME bytes:
//
//   * the entropy-coded data are decoded as ITU T.81 prescribes (Annex F: Huffman, DC prediction, zig-zag, restart
//     intervals) -- any conforming decoder yields the same coefficients;
//   * what is NOT prescribed by the standard is restated from stb_image, the reference's decoder: dequantisation into
//     16-bit coefficients, its fixed-point inverse DCT (the jidctint "islow" factorisation with 12-bit constants, two
//     extra bits kept between the passes, R/external/stb_image.h:2426-2524), its chroma upsampling filters (:3465-3528)
//     and its reduced-precision YCbCr -> RGB (:3659-3687);
//   * then stbi_loadf's LDR -> HDR step and RtwImage's FloatToByte: rt_rtwimage_bytes (scene_builder.cpp).
//
// tests/test_jpeg.py compares the result byte for byte with the reference's own stb build (oracle/_ref) on the reference's
// earthmap.jpg and on JPEGs of every sampling layout, with and without restart intervals, and with committed fixtures
// where the reference is not available.  Progressive and arithmetic-coded files, 12-bit samples and CMYK are not decoded
// (the caller gets an error and ImageTexture its cyan fallback, R/Texture.h:113-114).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtow.h"
#include "scene_host.h"

namespace rtow {
namespace {

// position of the k-th coefficient of the zig-zag sequence in the 8x8 block (T.81 figure A.6)
const uint8_t kNatural[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                   6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                   39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                   // a corrupt run may step past 63: land inside the block
                                   63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct HuffTable {  // T.81 Annex C / F.2.2.3: codes of each length are consecutive
    bool present = false;
    int mincode[17], maxcode[18], valptr[17];
    uint8_t values[256];
    bool build(const int count[16])
    {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k;
            mincode[len] = code;
            code += count[len - 1];
            k += count[len - 1];
            maxcode[len] = count[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) return false;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        present = true;
        return k <= 256;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dc_pred = 0;
    int blocks_w = 0, blocks_h = 0;  // plane size in blocks (whole MCUs)
    int rows = 0;                    // rows of samples that belong to the image: ceil(height * v / vmax)
    std::vector<uint8_t> plane;      // blocks_w * 8 wide
};

struct Decoder {
    const uint8_t *p, *end;
    std::string error;
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1;
    Component comp[4];
    uint16_t quant[4][64];  // natural order
    HuffTable dc[4], ac[4];
    int restart_interval = 0;
    bool jfif = false, frame_seen = false;
    int adobe_transform = -1;
    // bit reader over the entropy-coded segment
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    bool hit_marker = false;
    int marker = -1;

    bool fail(const char *what)
    {
        if (error.empty()) error = what;
        return false;
    }
    int get8() { return p < end ? *p++ : 0; }
    int get16()
    {
        int a = get8();
        return (a << 8) | get8();
    }

    void fill()
    {
        while (bitcnt <= 24) {
            int b = 0;
            if (!hit_marker) {
                b = get8();
                if (b == 0xFF) {
                    int c = get8();
                    while (c == 0xFF) c = get8();  // fill bytes
                    if (c != 0) {                   // a marker ends the segment: zeros from here on
                        marker = c;
                        hit_marker = true;
                        b = 0;
                    }
                }
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int bit()
    {
        if (bitcnt < 1) fill();
        int b = (int)(bitbuf >> 31);
        bitbuf <<= 1;
        bitcnt--;
        return b;
    }
    int receive(int n)  // n bits, most significant first
    {
        if (n == 0) return 0;
        if (bitcnt < n) fill();
        int v = (int)(bitbuf >> (32 - n));
        bitbuf <<= n;
        bitcnt -= n;
        return v;
    }
    static int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }  // T.81 F.2.2.1
    int decode(const HuffTable &t)
    {
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (code << 1) | bit();
            if (t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len]) return t.values[t.valptr[len] + code - t.mincode[len]];
        }
        return -1;
    }

    bool decode_block(Component &c, short out[64])
    {
        std::memset(out, 0, 64 * sizeof(short));
        const HuffTable &hd = dc[c.td], &ha = ac[c.ta];
        const uint16_t *q = quant[c.tq];
        int t = decode(hd);
        if (t < 0 || t > 15) return fail("bad huffman code");
        int diff = t ? extend(receive(t), t) : 0;
        c.dc_pred += diff;
        out[0] = (short)(c.dc_pred * q[0]);
        for (int k = 1; k < 64;) {
            int rs = decode(ha);
            if (rs < 0) return fail("bad huffman code");
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xF0) break;  // end of block
                k += 16;
            } else {
                k += r;
                int pos = kNatural[k++];
                out[pos] = (short)(extend(receive(s), s) * q[pos]);
            }
        }
        return true;
    }

    // ---- what stb_image computes with the coefficients (R/external/stb_image.h:2426-2524) ----
    static uint8_t clamp8(int x) { return (unsigned)x > 255u ? (x < 0 ? 0 : 255) : (uint8_t)x; }
    // one 8-point pass of the "islow" inverse DCT with constants scaled by 4096: x0..x3 even part, t0..t3 odd part
    struct Pass {
        int x0, x1, x2, x3, t0, t1, t2, t3;
    };
    static Pass idct8(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7)
    {
        // round(c * 4096) of the factorisation's constants, as (int)(c * 4096 + 0.5) gives them
        constexpr int C0_541 = 2217, Cm1_848 = -7567, C0_765 = 3135, C1_176 = 4816, C0_299 = 1223, C2_053 = 8410, C3_073 = 12586,
                      C1_501 = 6149, Cm0_900 = -3685, Cm2_563 = -10497, Cm1_962 = -8034, Cm0_390 = -1597;
        Pass r;
        int p2 = s2, p3 = s6;
        int p1 = (p2 + p3) * C0_541;
        int t2 = p1 + p3 * Cm1_848, t3 = p1 + p2 * C0_765;
        p2 = s0;
        p3 = s4;
        int t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
        r.x0 = t0 + t3; r.x3 = t0 - t3; r.x1 = t1 + t2; r.x2 = t1 - t2;
        t0 = s7; t1 = s5; t2 = s3; t3 = s1;
        p3 = t0 + t2;
        int p4 = t1 + t3;
        p1 = t0 + t3;
        p2 = t1 + t2;
        int p5 = (p3 + p4) * C1_176;
        t0 *= C0_299; t1 *= C2_053; t2 *= C3_073; t3 *= C1_501;
        p1 = p5 + p1 * Cm0_900;
        p2 = p5 + p2 * Cm2_563;
        p3 *= Cm1_962;
        p4 *= Cm0_390;
        r.t3 = t3 + p1 + p4; r.t2 = t2 + p2 + p3; r.t1 = t1 + p2 + p4; r.t0 = t0 + p1 + p3;
        return r;
    }
    static void idct_block(uint8_t *out, int stride, const short d[64])
    {
        int v[64];
        for (int i = 0; i < 8; i++) {  // columns; a column of zeros below its first entry is flat
            if (!d[i + 8] && !d[i + 16] && !d[i + 24] && !d[i + 32] && !d[i + 40] && !d[i + 48] && !d[i + 56]) {
                int flat = d[i] * 4;
                for (int k = 0; k < 8; k++) v[i + 8 * k] = flat;
                continue;
            }
            Pass a = idct8(d[i], d[i + 8], d[i + 16], d[i + 24], d[i + 32], d[i + 40], d[i + 48], d[i + 56]);
            a.x0 += 512; a.x1 += 512; a.x2 += 512; a.x3 += 512;  // down by 10 of the 12 bits: two are kept for the second pass
            v[i] = (a.x0 + a.t3) >> 10; v[i + 56] = (a.x0 - a.t3) >> 10;
            v[i + 8] = (a.x1 + a.t2) >> 10; v[i + 48] = (a.x1 - a.t2) >> 10;
            v[i + 16] = (a.x2 + a.t1) >> 10; v[i + 40] = (a.x2 - a.t1) >> 10;
            v[i + 24] = (a.x3 + a.t0) >> 10; v[i + 32] = (a.x3 - a.t0) >> 10;
        }
        for (int i = 0; i < 8; i++) {  // rows: 17 bits to drop (12 + 2 + 3), rounded, level shift of 128 added before the shift
            const int *w = v + 8 * i;
            uint8_t *o = out + (size_t)stride * i;
            Pass a = idct8(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
            const int bias = 65536 + (128 << 17);
            a.x0 += bias; a.x1 += bias; a.x2 += bias; a.x3 += bias;
            o[0] = clamp8((a.x0 + a.t3) >> 17); o[7] = clamp8((a.x0 - a.t3) >> 17);
            o[1] = clamp8((a.x1 + a.t2) >> 17); o[6] = clamp8((a.x1 - a.t2) >> 17);
            o[2] = clamp8((a.x2 + a.t1) >> 17); o[5] = clamp8((a.x2 - a.t1) >> 17);
            o[3] = clamp8((a.x3 + a.t0) >> 17); o[4] = clamp8((a.x3 - a.t0) >> 17);
        }
    }

    // ---- markers ----
    bool read_tables_and_frame(int m)
    {
        int len = get16() - 2;
        if (len < 0 || p + len > end) return fail("bad segment length");
        const uint8_t *seg_end = p + len;
        switch (m) {
        case 0xDB:  // DQT
            while (p < seg_end) {
                int pq = get8(), t = pq & 15, wide = pq >> 4;
                if (t > 3 || wide > 1) return fail("bad DQT");
                for (int i = 0; i < 64; i++) quant[t][kNatural[i]] = (uint16_t)(wide ? get16() : get8());
            }
            break;
        case 0xC4:  // DHT
            while (p < seg_end) {
                int tc = get8(), th = tc & 15, count[16], total = 0;
                tc >>= 4;
                if (tc > 1 || th > 3) return fail("bad DHT");
                for (int i = 0; i < 16; i++) total += count[i] = get8();
                if (total > 256) return fail("bad DHT");
                HuffTable &t = tc ? ac[th] : dc[th];
                if (!t.build(count)) return fail("bad code lengths");
                for (int i = 0; i < total; i++) t.values[i] = (uint8_t)get8();
            }
            break;
        case 0xDD: restart_interval = get16(); break;
        case 0xE0:
            if (len >= 5 && !std::memcmp(p, "JFIF\0", 5)) jfif = true;
            break;
        case 0xEE:
            if (len >= 12 && !std::memcmp(p, "Adobe\0", 6)) adobe_transform = p[11];
            break;
        case 0xC0:
        case 0xC1: {  // baseline / extended sequential, Huffman
            if (frame_seen) return fail("second frame header");
            if (get8() != 8) return fail("only 8-bit samples");
            height = get16();
            width = get16();
            ncomp = get8();
            if (width <= 0 || height <= 0) return fail("empty image");
            if (ncomp != 1 && ncomp != 3) return fail("1 or 3 components only (no CMYK)");
            for (int k = 0; k < ncomp; k++) {
                Component &c = comp[k];
                c.id = get8();
                int hv = get8();
                c.h = hv >> 4;
                c.v = hv & 15;
                c.tq = get8();
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad component");
                hmax = c.h > hmax ? c.h : hmax;
                vmax = c.v > vmax ? c.v : vmax;
            }
            for (int k = 0; k < ncomp; k++)
                if (hmax % comp[k].h || vmax % comp[k].v) return fail("fractional sampling ratio");
            if ((int64_t)width * height > ((int64_t)1 << 28)) return fail("image too large");
            const int mcus_x = (width + 8 * hmax - 1) / (8 * hmax), mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
            for (int k = 0; k < ncomp; k++) {
                Component &c = comp[k];
                c.blocks_w = mcus_x * c.h;
                c.blocks_h = mcus_y * c.v;
                c.rows = (height * c.v + vmax - 1) / vmax;
                c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
            }
            frame_seen = true;
            break;
        }
        default: break;  // APPn, COM, ...: skipped
        }
        p = seg_end;
        return error.empty();
    }

    bool read_scan()
    {
        int len = get16();
        int ns = get8();
        if (!frame_seen || ns < 1 || ns > ncomp || len != 6 + 2 * ns) return fail("bad SOS");
        int order[4];
        for (int i = 0; i < ns; i++) {
            int id = get8(), tt = get8(), which = -1;
            for (int k = 0; k < ncomp; k++)
                if (comp[k].id == id) which = k;
            if (which < 0) return fail("bad SOS component");
            comp[which].td = tt >> 4;
            comp[which].ta = tt & 15;
            if (comp[which].td > 3 || comp[which].ta > 3 || !dc[comp[which].td].present || !ac[comp[which].ta].present)
                return fail("missing huffman table");
            order[i] = which;
        }
        int ss = get8(), se = get8(), ahal = get8();
        if (ss != 0 || se != 63 || ahal != 0) return fail("not a sequential scan");
        bitbuf = 0;
        bitcnt = 0;
        hit_marker = false;
        marker = -1;
        for (int k = 0; k < ncomp; k++) comp[k].dc_pred = 0;
        int todo = restart_interval ? restart_interval : 0x7FFFFFFF;
        short block[64];
        auto restart_due = [&]() -> bool {  // true: go on decoding
            if (--todo > 0) return true;
            // the interval is over: the next thing in the stream must be RSTn (T.81 E.1.4)
            bitbuf = 0;
            bitcnt = 0;
            if (!hit_marker) {
                while (p < end && *p != 0xFF) p++;  // (conforming streams are already there)
                while (p + 1 < end && p[1] == 0xFF) p++;
                if (p + 1 < end) {
                    marker = p[1];
                    p += 2;
                    hit_marker = true;
                }
            }
            if (marker < 0xD0 || marker > 0xD7) return false;  // no restart marker: the scan is over
            hit_marker = false;
            marker = -1;
            for (int k = 0; k < ncomp; k++) comp[k].dc_pred = 0;
            todo = restart_interval;
            return true;
        };
        if (ns == 1) {  // one component: its blocks in raster order, only those that hold image samples (T.81 A.2.2)
            Component &c = comp[order[0]];
            const int bw = ((width * c.h + hmax - 1) / hmax + 7) >> 3, bh = ((height * c.v + vmax - 1) / vmax + 7) >> 3;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    if (!decode_block(c, block)) return false;
                    idct_block(c.plane.data() + ((size_t)by * 8 * c.blocks_w + bx) * 8, c.blocks_w * 8, block);
                    if (!restart_due()) return true;
                }
        } else {  // interleaved: MCU by MCU, in each the components' blocks (T.81 A.2.3)
            const int mcus_x = comp[0].blocks_w / comp[0].h, mcus_y = comp[0].blocks_h / comp[0].v;
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    for (int i = 0; i < ns; i++) {
                        Component &c = comp[order[i]];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++) {
                                if (!decode_block(c, block)) return false;
                                idct_block(c.plane.data() + ((size_t)(my * c.v + y) * 8 * c.blocks_w + (mx * c.h + x)) * 8, c.blocks_w * 8, block);
                            }
                    }
                    if (!restart_due()) return true;
                }
        }
        return true;
    }

    bool parse()
    {
        if (get8() != 0xFF || get8() != 0xD8) return fail("not a JPEG file");
        for (;;) {
            int m;
            if (hit_marker && marker >= 0) {  // the scan ended at a marker
                m = marker;
                hit_marker = false;
                marker = -1;
            } else {
                int b = get8();
                while (b != 0xFF && p < end) b = get8();  // (bytes between segments: skipped)
                m = get8();
                while (m == 0xFF && p < end) m = get8();
                if (p >= end && m != 0xD9) return frame_seen ? true : fail("no image data");
            }
            if (m == 0xD9) return frame_seen ? true : fail("no image data");
            if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC))
                return fail("progressive, lossless and arithmetic-coded JPEGs are not decoded");
            if (m == 0xDA) {
                if (!read_scan()) return false;
                continue;
            }
            if (m >= 0xD0 && m <= 0xD7) continue;  // a stray restart marker
            if (!read_tables_and_frame(m)) return false;
        }
    }

    // ---- output: chroma upsampling as stb_image does it (R/external/stb_image.h:3465-3528,3646-3656), then colour ----
    static const uint8_t *resample(uint8_t *out, const uint8_t *near_row, const uint8_t *far_row, int w, int hs, int vs)
    {
        if (hs == 1 && vs == 1) return near_row;
        if (hs == 1 && vs == 2) {  // vertical: 3/4 of the nearer row, 1/4 of the farther
            for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near_row[i] + far_row[i] + 2) >> 2);
            return out;
        }
        if (hs == 2 && vs == 1) {  // horizontal: 3/4 nearer sample, 1/4 neighbour; the ends copied
            const uint8_t *in = near_row;
            if (w == 1) {
                out[0] = out[1] = in[0];
                return out;
            }
            out[0] = in[0];
            out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
            int i;
            for (i = 1; i < w - 1; i++) {
                int n = 3 * in[i] + 2;
                out[i * 2] = (uint8_t)((n + in[i - 1]) >> 2);
                out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
            }
            out[i * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
            out[i * 2 + 1] = in[w - 1];
            return out;
        }
        if (hs == 2 && vs == 2) {  // both: vertical first (t = 3 near + far, kept at 4x), then horizontal on those sums
            if (w == 1) {
                out[0] = out[1] = (uint8_t)((3 * near_row[0] + far_row[0] + 2) >> 2);
                return out;
            }
            int t1 = 3 * near_row[0] + far_row[0];
            out[0] = (uint8_t)((t1 + 2) >> 2);
            for (int i = 1; i < w; i++) {
                int t0 = t1;
                t1 = 3 * near_row[i] + far_row[i];
                out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
                out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
            }
            out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
            return out;
        }
        for (int i = 0; i < w; i++)  // any other ratio: nearest neighbour along the row
            for (int j = 0; j < hs; j++) out[i * hs + j] = near_row[i];
        return out;
    }

    bool to_rgb(std::vector<uint8_t> &rgb)
    {
        rgb.assign((size_t)width * height * 3, 0);
        struct Row {
            int hs, vs, ystep, ypos, w_lores;
            const uint8_t *line0, *line1;
            std::vector<uint8_t> buf;
        } rows[4];
        for (int k = 0; k < ncomp; k++) {
            Row &r = rows[k];
            r.hs = hmax / comp[k].h;
            r.vs = vmax / comp[k].v;
            r.ystep = r.vs >> 1;
            r.ypos = 0;
            r.w_lores = (width + r.hs - 1) / r.hs;
            r.line0 = r.line1 = comp[k].plane.data();
            r.buf.assign((size_t)width + 3 + r.hs, 0);
        }
        bool is_rgb = false;
        if (ncomp == 3) {
            const bool named_rgb = comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B';
            is_rgb = named_rgb || (adobe_transform == 0 && !jfif);
        }
        // (int)(c * 4096.0f + 0.5f) << 8 for 1.402, 0.71414, 0.34414, 1.772
        constexpr int kCrR = 5743 << 8, kCrG = 2925 << 8, kCbG = 1410 << 8, kCbB = 7258 << 8;
        for (int j = 0; j < height; j++) {
            const uint8_t *line[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int k = 0; k < ncomp; k++) {
                Row &r = rows[k];
                const bool bottom = r.ystep >= (r.vs >> 1);
                line[k] = resample(r.buf.data(), bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < comp[k].rows) r.line1 += (size_t)comp[k].blocks_w * 8;
                }
            }
            uint8_t *out = rgb.data() + (size_t)j * width * 3;
            if (ncomp == 1) {
                for (int i = 0; i < width; i++) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = line[0][i];
            } else if (is_rgb) {
                for (int i = 0; i < width; i++) {
                    out[3 * i] = line[0][i];
                    out[3 * i + 1] = line[1][i];
                    out[3 * i + 2] = line[2][i];
                }
            } else {
                for (int i = 0; i < width; i++) {
                    const int y_fixed = (line[0][i] << 20) + (1 << 19);
                    const int cb = line[1][i] - 128, cr = line[2][i] - 128;
                    int r = y_fixed + cr * kCrR;
                    int g = y_fixed + cr * -kCrG + (int)((unsigned)(cb * -kCbG) & 0xffff0000u);  // the low half dropped, as the SIMD form of the reference's decoder does
                    int b = y_fixed + cb * kCbB;
                    out[3 * i] = clamp8(r >> 20);
                    out[3 * i + 1] = clamp8(g >> 20);
                    out[3 * i + 2] = clamp8(b >> 20);
                }
            }
        }
        return true;
    }
};

}  // namespace
}  // namespace rtow

using namespace rtow;

extern "C" {

int rt_jpeg_decode(const unsigned char *data, size_t size, unsigned char **rgb_out, int *width, int *height)
{
    if (!data || !rgb_out || !width || !height) return fail(RT_ERR_INVALID, "rt_jpeg_decode: null argument");
    *rgb_out = nullptr;
    Decoder d;
    d.p = data;
    d.end = data + size;
    std::memset(d.quant, 0, sizeof d.quant);
    std::vector<uint8_t> rgb;
    if (!d.parse() || !d.to_rgb(rgb)) return fail(RT_ERR_UNSUPPORTED, "rt_jpeg_decode: " + (d.error.empty() ? std::string("corrupt file") : d.error));
    unsigned char *out = static_cast<unsigned char *>(std::malloc(rgb.size() ? rgb.size() : 1));
    if (!out) return fail(RT_ERR_INVALID, "rt_jpeg_decode: out of memory");
    std::memcpy(out, rgb.data(), rgb.size());
    *rgb_out = out;
    *width = d.width;
    *height = d.height;
    return RT_OK;
}

int rt_rtwimage_load(const char *path, unsigned char **rgb_out, int *width, int *height)
{
    if (!path || !rgb_out || !width || !height) return fail(RT_ERR_INVALID, "rt_rtwimage_load: null argument");
    *rgb_out = nullptr;
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return fail(RT_ERR_INVALID, std::string("rt_rtwimage_load: could not open '") + path + "'");
    std::vector<unsigned char> bytes;
    unsigned char chunk[65536];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof chunk, fp)) > 0) bytes.insert(bytes.end(), chunk, chunk + got);
    std::fclose(fp);
    if (int rc = rt_jpeg_decode(bytes.data(), bytes.size(), rgb_out, width, height)) return rc;
    rt_rtwimage_bytes(*rgb_out, (size_t)*width * (size_t)*height * 3, *rgb_out);  // stbi_loadf's linearisation, then FloatToByte
    return RT_OK;
}

void rt_image_free(unsigned char *rgb) { std::free(rgb); }

}  // extern "C"
