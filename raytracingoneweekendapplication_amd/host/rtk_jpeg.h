// rtk_jpeg.h -- JPEG decoding (sequential and progressive) for image_texture.
//
// The reference loads textures through rtw_image (rtw_stb_image.h:23-66), i.e.
// stbi_loadf of the vendored stb_image.h: earthmap.jpg (main.cpp:152,311;
// 1024x512, 4:4:4) and male_texture.jpg (main.cpp:383; 2048x2048, 4:2:0) are
// baseline, 8-bit, three-component JFIF files.  The texels are INPUTS of the
// sample loop, so for the drop-in to render what the reference renders the decoded
// bytes must be the bytes stb_image produces, not merely a faithful JPEG decode:
// JPEG leaves the inverse DCT, the chroma interpolation and the colour conversion
// to the implementation, and they differ between libraries in the last bit.
// This file therefore fixes those three choices to the ones stb_image makes
// (written from the JPEG standard, ITU T.81, and the published algorithms named
// below -- no code is taken from stb_image):
//
//   inverse DCT     Loeffler-Ligtenberg-Moschytz 8-point factorisation in 32-bit
//                   integers with 12-bit constants; column pass keeps 2 extra bits
//                   (+512 >> 10), row pass rounds at 17 bits with the +128 level
//                   shift folded in, then clamps to 0..255.
//   chroma upsample 4:2:0 / 4:2:2: "triangle" filter, 3/4 near + 1/4 far in each
//                   direction, 16ths with +8 rounding (edges: (3 near + far + 2) >> 2);
//                   4:4:4: none.  Other sampling ratios: pixel replication.
//   YCbCr -> RGB    BT.601 full range in 20-bit fixed point: R = Y + 1.402 Cr,
//                   G = Y - 0.71414 Cr - 0.34414 Cb (the Cb term truncated to its
//                   upper 16 bits), B = Y + 1.772 Cb, with +0.5 before the shift.
//
// tests/test_image_decode.py pins the result byte for byte against the
// reference's own loader (oracle/_ref) on generated fixtures and, where
// /root/reference exists, on earthmap.jpg and male_texture.jpg themselves.
//
// Supported: SOF0 / SOF1 (sequential) and SOF2 (progressive: spectral selection
// and successive approximation, T.81 G.1.2) with 8-bit precision, 1 or 3
// components, any sampling factors up to 4, restart intervals, 16-bit DQT
// entries.  Not supported (load fails, width() == 0 as rtw_stb_image.h:62 has it on
// a failed load): arithmetic coding, lossless / hierarchical, 12-bit samples, CMYK.
#ifndef RTK_JPEG_H
#define RTK_JPEG_H

#include <cstdint>
#include <cstring>
#include <vector>

namespace rtk {

class jpeg_decoder {
public:
    // Decodes `data` to interleaved RGB8 (grey images are replicated to three channels, as stbi_load with
    // req_comp = 3 does).  Returns false on malformed or unsupported input.
    bool decode(const uint8_t* data, size_t size, int& width, int& height, std::vector<uint8_t>& rgb) {
        src = data;
        end = data + size;
        if (size < 4 || src[0] != 0xFF || src[1] != 0xD8) return false;
        src += 2;
        restart_interval = 0;
        n_comp = 0;
        bool have_frame = false;
        for (;;) {
            int m = next_marker();
            if (m < 0) return false;
            if (m == 0xD9) return false;  // EOI before any scan
            if (m == 0xDA) {
                if (!have_frame || !read_scan_header()) return false;
                if (!decode_scan()) return false;
                break;  // baseline, interleaved: one scan carries the image (non-interleaved files loop below)
            }
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (have_frame || !read_frame_header()) return false;
                have_frame = true;
                progressive = m == 0xC2;
                if (progressive)
                    for (int c = 0; c < n_comp; c++) comp[c].coef.assign(size_t(comp[c].blocks_w) * comp[c].blocks_h * 64, 0);
            } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                return false;  // lossless / hierarchical / arithmetic coding
            } else if (m == 0xC4) {
                if (!read_huffman_tables()) return false;
            } else if (m == 0xDB) {
                if (!read_quant_tables()) return false;
            } else if (m == 0xDD) {
                if (segment_length() != 4) return false;
                restart_interval = (src[2] << 8) | src[3];
                src += 4;
            } else {
                if (!skip_segment()) return false;
            }
        }
        // Non-interleaved baseline files carry one scan per component: keep reading scans until EOI.
        for (;;) {
            int m = next_marker();
            if (m < 0 || m == 0xD9) break;
            if (m == 0xDA) {
                if (!read_scan_header() || !decode_scan()) return false;
            } else if (m == 0xC4) {
                if (!read_huffman_tables()) return false;
            } else if (m == 0xDB) {
                if (!read_quant_tables()) return false;
            } else if (m == 0xDD) {
                if (segment_length() != 4) return false;
                restart_interval = (src[2] << 8) | src[3];
                src += 4;
            } else if (!skip_segment()) {
                break;
            }
        }
        if (progressive) finish_progressive();
        width = img_w;
        height = img_h;
        to_rgb(rgb);
        return true;
    }

private:
    struct huffman {
        // canonical code: for each length L (1..16) the first code value, the index of its first symbol and the count
        int32_t first_code[17], first_index[17], count[17];
        uint8_t symbols[256];
        bool present = false;
    };
    struct component {
        int id, h, v, tq, td, ta;
        int blocks_w, blocks_h;  // in 8x8 blocks, padded to whole MCUs
        int plane_w, plane_h;
        int dc_pred;
        std::vector<uint8_t> plane;
        std::vector<int16_t> coef;  // progressive only: the coefficients of every block, accumulated over the scans
    };

    const uint8_t* src = nullptr;
    const uint8_t* end = nullptr;
    int img_w = 0, img_h = 0, n_comp = 0, h_max = 1, v_max = 1, restart_interval = 0;
    component comp[4];
    uint16_t quant[4][64];
    huffman dc_tab[4], ac_tab[4];
    int scan_n = 0, scan_comp[4];
    bool progressive = false;
    int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;
    // bit reader
    uint32_t bit_buf = 0;
    int bit_cnt = 0;
    bool hit_marker = false;

    static const uint8_t* zigzag() {
        static const uint8_t z[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                      41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                      30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        return z;
    }

    int segment_length() const { return (end - src >= 2) ? ((src[0] << 8) | src[1]) : -1; }
    bool skip_segment() {
        int len = segment_length();
        if (len < 2 || end - src < len) return false;
        src += len;
        return true;
    }
    int next_marker() {
        while (src < end && *src != 0xFF) src++;  // tolerate stray bytes between segments
        while (src < end && *src == 0xFF) src++;  // fill bytes
        if (src >= end) return -1;
        return *src++;
    }

    bool read_quant_tables() {
        int len = segment_length();
        if (len < 2 || end - src < len) return false;
        const uint8_t* p = src + 2;
        const uint8_t* stop = src + len;
        while (p < stop) {
            int precision = *p >> 4, id = *p & 15;
            p++;
            if (id > 3 || precision > 1) return false;
            if (stop - p < (precision ? 128 : 64)) return false;
            for (int k = 0; k < 64; k++) {
                int v = precision ? ((p[0] << 8) | p[1]) : p[0];
                p += precision ? 2 : 1;
                quant[id][zigzag()[k]] = uint16_t(v);
            }
        }
        src = stop;
        return true;
    }

    bool read_huffman_tables() {
        int len = segment_length();
        if (len < 2 || end - src < len) return false;
        const uint8_t* p = src + 2;
        const uint8_t* stop = src + len;
        while (p < stop) {
            if (stop - p < 17) return false;
            int cls = *p >> 4, id = *p & 15;
            p++;
            if (cls > 1 || id > 3) return false;
            huffman& t = cls ? ac_tab[id] : dc_tab[id];
            int total = 0, code = 0;
            for (int L = 1; L <= 16; L++) {
                t.count[L] = p[L - 1];
                t.first_index[L] = total;
                t.first_code[L] = code;
                total += t.count[L];
                code = (code + t.count[L]) << 1;
            }
            p += 16;
            if (total > 256 || stop - p < total) return false;
            std::memcpy(t.symbols, p, size_t(total));
            p += total;
            t.present = true;
        }
        src = stop;
        return true;
    }

    bool read_frame_header() {
        int len = segment_length();
        if (len < 8 || end - src < len) return false;
        const uint8_t* p = src + 2;
        if (p[0] != 8) return false;  // sample precision
        img_h = (p[1] << 8) | p[2];
        img_w = (p[3] << 8) | p[4];
        n_comp = p[5];
        if (img_w <= 0 || img_h <= 0 || (n_comp != 1 && n_comp != 3) || len < 8 + 3 * n_comp) return false;
        // The planes below are sized from these 16-bit fields: believe them only if the file could hold such an image.
        // A baseline scan costs every 8x8 block at least two bits of entropy-coded data (a DC code and an end-of-block
        // code): at most 256 pixels per byte of file.  A PROGRESSIVE file gets by on one bit per block (its DC-first scan;
        // the AC scans of a flat image collapse into end-of-band runs), i.e. up to 512 pixels per byte -- a uniform
        // 1024x1024 progressive image is ~3 KB and the reference's stb_image loads it -- so the bound is 1024 pixels per byte
        // whatever the frame type (slack of two over the progressive minimum).  Nothing above 2^27 pixels is decoded at all.
        if (uint64_t(img_w) * uint64_t(img_h) > (uint64_t(1) << 27) || uint64_t(img_w) * uint64_t(img_h) > uint64_t(end - src) * 1024 + 4096) return false;
        h_max = v_max = 1;
        for (int c = 0; c < n_comp; c++) {
            comp[c].id = p[6 + 3 * c];
            comp[c].h = p[7 + 3 * c] >> 4;
            comp[c].v = p[7 + 3 * c] & 15;
            comp[c].tq = p[8 + 3 * c];
            if (comp[c].h < 1 || comp[c].h > 4 || comp[c].v < 1 || comp[c].v > 4 || comp[c].tq > 3) return false;
            if (comp[c].h > h_max) h_max = comp[c].h;
            if (comp[c].v > v_max) v_max = comp[c].v;
        }
        const int mcu_w = 8 * h_max, mcu_h = 8 * v_max;
        const int mcus_x = (img_w + mcu_w - 1) / mcu_w, mcus_y = (img_h + mcu_h - 1) / mcu_h;
        for (int c = 0; c < n_comp; c++) {
            comp[c].blocks_w = mcus_x * comp[c].h;
            comp[c].blocks_h = mcus_y * comp[c].v;
            comp[c].plane_w = comp[c].blocks_w * 8;
            comp[c].plane_h = comp[c].blocks_h * 8;
            comp[c].plane.assign(size_t(comp[c].plane_w) * comp[c].plane_h, 0);
        }
        src += len;
        return true;
    }

    bool read_scan_header() {
        int len = segment_length();
        if (len < 6 || end - src < len) return false;
        const uint8_t* p = src + 2;
        scan_n = p[0];
        if (scan_n < 1 || scan_n > n_comp || len != 6 + 2 * scan_n) return false;
        for (int k = 0; k < scan_n; k++) {
            int id = p[1 + 2 * k], which = -1;
            for (int c = 0; c < n_comp; c++)
                if (comp[c].id == id) which = c;
            if (which < 0) return false;
            comp[which].td = p[2 + 2 * k] >> 4;
            comp[which].ta = p[2 + 2 * k] & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) return false;
            scan_comp[k] = which;
        }
        const uint8_t* q = p + 1 + 2 * scan_n;
        spec_start = q[0];
        spec_end = q[1];
        succ_high = q[2] >> 4;
        succ_low = q[2] & 15;
        if (progressive) {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) return false;
            if (spec_start == 0 && spec_end != 0) return false;   // a DC scan carries the DC coefficient only
            if (spec_start != 0 && scan_n != 1) return false;     // AC scans are not interleaved
        } else if (spec_start != 0 || spec_end != 63 || q[2] != 0) {
            return false;  // spectral selection / approximation in a sequential scan
        }
        src += len;
        return true;
    }

    // ---- entropy-coded segment
    void reset_bits() {
        bit_buf = 0;
        bit_cnt = 0;
        hit_marker = false;
    }
    void fill_bits() {
        while (bit_cnt <= 24) {
            uint32_t byte = 0;
            if (!hit_marker && src < end) {
                byte = *src;
                if (byte == 0xFF) {
                    uint32_t nxt = (src + 1 < end) ? src[1] : 0xD9;
                    if (nxt == 0) {
                        src += 2;  // stuffed zero
                    } else {
                        hit_marker = true;  // a marker: feed zeros from here on
                        byte = 0;
                    }
                } else {
                    src++;
                }
            }
            bit_buf |= byte << (24 - bit_cnt);
            bit_cnt += 8;
        }
    }
    int get_bits(int n) {
        if (n == 0) return 0;
        if (bit_cnt < n) fill_bits();
        int v = int(bit_buf >> (32 - n));
        bit_buf <<= n;
        bit_cnt -= n;
        return v;
    }
    int decode_symbol(const huffman& t) {
        if (bit_cnt < 16) fill_bits();
        int code = 0;
        for (int L = 1; L <= 16; L++) {
            code = (code << 1) | int((bit_buf >> (32 - L)) & 1u);
            if (t.count[L] && code - t.first_code[L] < t.count[L] && code >= t.first_code[L]) {
                bit_buf <<= L;
                bit_cnt -= L;
                return t.symbols[t.first_index[L] + code - t.first_code[L]];
            }
        }
        return -1;
    }
    static int extend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }  // T.81 F.2.2.1

    bool decode_block(component& c, int16_t coef[64]) {
        const huffman& dc = dc_tab[c.td];
        const huffman& ac = ac_tab[c.ta];
        if (!dc.present || !ac.present) return false;
        std::memset(coef, 0, 64 * sizeof(int16_t));
        int s = decode_symbol(dc);
        if (s < 0 || s > 15) return false;
        int diff = s ? extend(get_bits(s), s) : 0;
        c.dc_pred = int(uint32_t(c.dc_pred) + uint32_t(diff));  // wrapping: a damaged stream must not overflow a signed int
        const uint16_t* q = quant[c.tq];
        coef[0] = int16_t(uint32_t(c.dc_pred) * uint32_t(q[0]));
        for (int k = 1; k < 64;) {
            int rs = decode_symbol(ac);
            if (rs < 0) return false;
            int run = rs >> 4, size = rs & 15;
            if (size == 0) {
                if (run != 15) break;  // end of block
                k += 16;
                continue;
            }
            k += run;
            if (k > 63) return false;
            int pos = zigzag()[k];
            coef[pos] = int16_t(extend(get_bits(size), size) * q[pos]);
            k++;
        }
        return true;
    }

    // 8-point inverse DCT, Loeffler-Ligtenberg-Moschytz factorisation, 12-bit fixed-point constants.  `shifted`
    // inputs: even part scaled by 4096; returns the eight butterfly outputs (before the final pairing) in e[] / o[].
    // All IDCT arithmetic is done in uint32_t: the same bits as stb_image's int arithmetic on every decodable file, and
    // defined (wrapping) on a damaged one whose coefficients would overflow a signed int.  sar() is the arithmetic shift.
    using u32 = uint32_t;
    static int32_t sar(u32 v, int n) { return int32_t(v) >> n; }
    static void idct_1d(const u32 s[8], u32 e[4], u32 o[4]) {
        constexpr u32 c0_541 = 2217, c1_847 = u32(-7567), c0_765 = 3135, c1_175 = 4816, c0_298 = 1223, c2_053 = 8410, c3_072 = 12586,
                      c1_501 = 6149, c0_899 = u32(-3685), c2_562 = u32(-10497), c1_961 = u32(-8034), c0_390 = u32(-1597);
        // even part
        u32 z = (s[2] + s[6]) * c0_541;
        u32 t2 = z + s[6] * c1_847, t3 = z + s[2] * c0_765;
        u32 t0 = (s[0] + s[4]) * 4096u, t1 = (s[0] - s[4]) * 4096u;
        e[0] = t0 + t3;
        e[3] = t0 - t3;
        e[1] = t1 + t2;
        e[2] = t1 - t2;
        // odd part
        u32 a0 = s[7], a1 = s[5], a2 = s[3], a3 = s[1];
        u32 p3 = a0 + a2, p4 = a1 + a3, p1 = a0 + a3, p2 = a1 + a2;
        u32 p5 = (p3 + p4) * c1_175;
        a0 *= c0_298;
        a1 *= c2_053;
        a2 *= c3_072;
        a3 *= c1_501;
        p1 = p5 + p1 * c0_899;
        p2 = p5 + p2 * c2_562;
        p3 *= c1_961;
        p4 *= c0_390;
        o[3] = a3 + p1 + p4;
        o[2] = a2 + p2 + p3;
        o[1] = a1 + p2 + p4;
        o[0] = a0 + p1 + p3;
    }
    static uint8_t clamp8(int32_t v) { return uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v)); }

    static void idct_block(const int16_t coef[64], uint8_t* out, int stride) {
        u32 tmp[64];
        for (int x = 0; x < 8; x++) {  // columns: keep two fractional bits
            u32 s[8], e[4], o[4];
            for (int k = 0; k < 8; k++) s[k] = u32(int32_t(coef[x + 8 * k]));
            idct_1d(s, e, o);
            for (int k = 0; k < 4; k++) e[k] += 512u;
            tmp[x + 0] = u32(sar(e[0] + o[3], 10));
            tmp[x + 56] = u32(sar(e[0] - o[3], 10));
            tmp[x + 8] = u32(sar(e[1] + o[2], 10));
            tmp[x + 48] = u32(sar(e[1] - o[2], 10));
            tmp[x + 16] = u32(sar(e[2] + o[1], 10));
            tmp[x + 40] = u32(sar(e[2] - o[1], 10));
            tmp[x + 24] = u32(sar(e[3] + o[0], 10));
            tmp[x + 32] = u32(sar(e[3] - o[0], 10));
        }
        for (int y = 0; y < 8; y++) {  // rows: round at 17 bits, +128 level shift folded in
            u32 e[4], o[4];
            idct_1d(tmp + 8 * y, e, o);
            for (int k = 0; k < 4; k++) e[k] += 65536u + (128u << 17);
            uint8_t* row = out + y * stride;
            row[0] = clamp8(sar(e[0] + o[3], 17));
            row[7] = clamp8(sar(e[0] - o[3], 17));
            row[1] = clamp8(sar(e[1] + o[2], 17));
            row[6] = clamp8(sar(e[1] - o[2], 17));
            row[2] = clamp8(sar(e[2] + o[1], 17));
            row[5] = clamp8(sar(e[2] - o[1], 17));
            row[3] = clamp8(sar(e[3] + o[0], 17));
            row[4] = clamp8(sar(e[3] - o[0], 17));
        }
    }

    // ---- progressive (SOF2) scans, T.81 G.1.2: coefficients accumulate in comp.coef and are transformed at the end
    bool decode_block_prog_dc(component& c, int16_t* data) {
        if (succ_high == 0) {
            const huffman& dc = dc_tab[c.td];
            if (!dc.present) return false;
            int s = decode_symbol(dc);
            if (s < 0 || s > 15) return false;
            int diff = s ? extend(get_bits(s), s) : 0;
            c.dc_pred = int(uint32_t(c.dc_pred) + uint32_t(diff));  // wrapping: a damaged stream must not overflow a signed int
            data[0] = int16_t(uint32_t(c.dc_pred) << succ_low);
        } else if (get_bits(1)) {
            data[0] = int16_t(data[0] + (1 << succ_low));
        }
        return true;
    }
    bool decode_block_prog_ac(component& c, int16_t* data) {
        const huffman& ac = ac_tab[c.ta];
        if (!ac.present) return false;
        if (succ_high == 0) {  // first pass of this band
            if (eob_run) {
                --eob_run;
                return true;
            }
            int k = spec_start;
            do {
                int rs = decode_symbol(ac);
                if (rs < 0) return false;
                int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = (1 << r);
                        if (r) eob_run += get_bits(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) return false;
                    data[zigzag()[k++]] = int16_t(extend(get_bits(s), s) * (1 << succ_low));
                }
            } while (k <= spec_end);
            return true;
        }
        // refinement pass: one more bit for the coefficients already non-zero, new +-1 coefficients in between
        const int16_t bit = int16_t(1 << succ_low);
        auto refine = [&](int16_t* p) {
            if (get_bits(1) && (*p & bit) == 0) *p = int16_t(*p > 0 ? *p + bit : *p - bit);
        };
        if (eob_run) {
            --eob_run;
            for (int k = spec_start; k <= spec_end; k++) {
                int16_t* p = &data[zigzag()[k]];
                if (*p != 0) refine(p);
            }
            return true;
        }
        int k = spec_start;
        do {
            int rs = decode_symbol(ac);
            if (rs < 0) return false;
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    eob_run = (1 << r) - 1;
                    if (r) eob_run += get_bits(r);
                    r = 64;  // run to the end of the band
                }           // else: sixteen zero coefficients to skip
            } else {
                if (s != 1) return false;
                s = get_bits(1) ? bit : -bit;
            }
            while (k <= spec_end) {
                int16_t* p = &data[zigzag()[k++]];
                if (*p != 0) {
                    refine(p);
                } else {
                    if (r == 0) {
                        *p = int16_t(s);
                        break;
                    }
                    --r;
                }
            }
        } while (k <= spec_end);
        return true;
    }
    bool decode_scan_progressive() {
        reset_bits();
        eob_run = 0;
        for (int c = 0; c < n_comp; c++) comp[c].dc_pred = 0;
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        auto after_unit = [&]() {
            if (--todo > 0) return;
            reset_bits_to_marker();
            if (src + 1 < end && src[0] == 0xFF && src[1] >= 0xD0 && src[1] <= 0xD7) {
                src += 2;
                reset_bits();
                eob_run = 0;
                for (int c = 0; c < n_comp; c++) comp[c].dc_pred = 0;
                todo = restart_interval ? restart_interval : 0x7fffffff;
            }
        };
        if (scan_n == 1) {
            component& c = comp[scan_comp[0]];
            const int w = (((img_w * c.h + h_max - 1) / h_max) + 7) >> 3, h = (((img_h * c.v + v_max - 1) / v_max) + 7) >> 3;
            for (int by = 0; by < h; by++)
                for (int bx = 0; bx < w; bx++) {
                    int16_t* data = c.coef.data() + (size_t(by) * c.blocks_w + bx) * 64;
                    if (!(spec_start == 0 ? decode_block_prog_dc(c, data) : decode_block_prog_ac(c, data))) return false;
                    after_unit();
                }
        } else {  // interleaved: DC only
            const int mcus_x = comp[0].blocks_w / comp[0].h, mcus_y = comp[0].blocks_h / comp[0].v;
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    for (int k = 0; k < scan_n; k++) {
                        component& c = comp[scan_comp[k]];
                        for (int v = 0; v < c.v; v++)
                            for (int h = 0; h < c.h; h++) {
                                int16_t* data = c.coef.data() + (size_t(my * c.v + v) * c.blocks_w + (mx * c.h + h)) * 64;
                                if (!decode_block_prog_dc(c, data)) return false;
                            }
                    }
                    after_unit();
                }
        }
        reset_bits_to_marker();
        return true;
    }
    // After the last scan: dequantise (16-bit wrap, as in the sequential path) and transform every block.
    void finish_progressive() {
        for (int ci = 0; ci < n_comp; ci++) {
            component& c = comp[ci];
            const int w = (((img_w * c.h + h_max - 1) / h_max) + 7) >> 3, h = (((img_h * c.v + v_max - 1) / v_max) + 7) >> 3;
            const uint16_t* q = quant[c.tq];
            for (int by = 0; by < h; by++)
                for (int bx = 0; bx < w; bx++) {
                    int16_t* data = c.coef.data() + (size_t(by) * c.blocks_w + bx) * 64;
                    for (int k = 0; k < 64; k++) data[k] = int16_t(data[k] * q[k]);
                    idct_block(data, c.plane.data() + size_t(by) * 8 * c.plane_w + bx * 8, c.plane_w);
                }
        }
    }

    bool decode_scan() {
        if (progressive) return decode_scan_progressive();
        reset_bits();
        for (int c = 0; c < n_comp; c++) comp[c].dc_pred = 0;
        int16_t coef[64];
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        auto after_unit = [&]() -> bool {
            if (--todo > 0) return true;
            // a restart marker must follow: byte-align, check RSTn, reset predictors
            reset_bits_to_marker();
            if (src + 1 < end && src[0] == 0xFF && src[1] >= 0xD0 && src[1] <= 0xD7) {
                src += 2;
                reset_bits();
                for (int c = 0; c < n_comp; c++) comp[c].dc_pred = 0;
                todo = restart_interval ? restart_interval : 0x7fffffff;
                return true;
            }
            return true;  // no marker (end of scan, or a damaged file): keep what was decoded
        };
        if (scan_n == 1) {  // non-interleaved: the component's own blocks in raster order, (w+7)/8 x (h+7)/8 of them
            component& c = comp[scan_comp[0]];
            const int w = (((img_w * c.h + h_max - 1) / h_max) + 7) >> 3, h = (((img_h * c.v + v_max - 1) / v_max) + 7) >> 3;
            for (int by = 0; by < h; by++)
                for (int bx = 0; bx < w; bx++) {
                    if (!decode_block(c, coef)) return false;
                    idct_block(coef, c.plane.data() + size_t(by) * 8 * c.plane_w + bx * 8, c.plane_w);
                    if (!after_unit()) return false;
                }
        } else {
            const int mcus_x = comp[0].blocks_w / comp[0].h, mcus_y = comp[0].blocks_h / comp[0].v;
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    for (int k = 0; k < scan_n; k++) {
                        component& c = comp[scan_comp[k]];
                        for (int v = 0; v < c.v; v++)
                            for (int h = 0; h < c.h; h++) {
                                if (!decode_block(c, coef)) return false;
                                const int bx = mx * c.h + h, by = my * c.v + v;
                                idct_block(coef, c.plane.data() + size_t(by) * 8 * c.plane_w + bx * 8, c.plane_w);
                            }
                    }
                    if (!after_unit()) return false;
                }
        }
        reset_bits_to_marker();
        return true;
    }
    // Leave `src` at the marker that ended the entropy-coded data (bytes already pulled into the bit buffer are
    // never past it: fill_bits stops consuming at a marker).
    void reset_bits_to_marker() {
        if (!hit_marker) {
            // whole bytes still sitting in the buffer belong to the stream; scan forward to the next marker
            while (src + 1 < end && !(src[0] == 0xFF && src[1] != 0 && !(src[1] == 0xFF))) src++;
        }
        bit_buf = 0;
        bit_cnt = 0;
        hit_marker = false;
    }

    // ---- planes -> RGB
    // One output row of a component at full resolution: the triangle-filtered (h2 and/or v2) or replicated samples.
    void upsample_row(const component& c, int y, std::vector<uint8_t>& row, std::vector<uint8_t>& scratch) const {
        const int hs = h_max / c.h, vs = v_max / c.v;
        const int cw = (img_w * c.h + h_max - 1) / h_max;  // the component's own width in samples
        const int ch = (img_h * c.v + v_max - 1) / v_max;
        auto line = [&](int r) { return c.plane.data() + size_t(r < 0 ? 0 : (r >= ch ? ch - 1 : r)) * c.plane_w; };
        row.resize(size_t(cw) * hs + 8);
        if (hs == 1 && vs == 1) {
            std::memcpy(row.data(), line(y), size_t(cw));
        } else if (hs == 1 && vs == 2) {
            // vertical only: (3 near + far + 2) >> 2
            const int src_row = y >> 1;
            const uint8_t* near_ = line(src_row);
            const uint8_t* far_ = line((y & 1) ? src_row + 1 : src_row - 1);
            for (int i = 0; i < cw; i++) row[i] = uint8_t((3 * near_[i] + far_[i] + 2) >> 2);
        } else if (hs == 2 && vs == 1) {
            const uint8_t* in = line(y);
            h2_row(in, cw, row.data());
        } else if (hs == 2 && vs == 2) {
            const int src_row = y >> 1;
            const uint8_t* near_ = line(src_row);
            const uint8_t* far_ = line((y & 1) ? src_row + 1 : src_row - 1);
            if (cw == 1) {
                row[0] = row[1] = uint8_t((3 * near_[0] + far_[0] + 2) >> 2);
            } else {
                // vertical blend kept at 4x scale, horizontal blend on top: 16ths with +8 rounding
                int t1 = 3 * near_[0] + far_[0];
                row[0] = uint8_t((t1 + 2) >> 2);
                for (int i = 1; i < cw; i++) {
                    const int t0 = t1;
                    t1 = 3 * near_[i] + far_[i];
                    row[2 * i - 1] = uint8_t((3 * t0 + t1 + 8) >> 4);
                    row[2 * i] = uint8_t((3 * t1 + t0 + 8) >> 4);
                }
                row[2 * cw - 1] = uint8_t((t1 + 2) >> 2);
            }
        } else {
            // any other ratio: nearest sample
            const uint8_t* in = line(y / vs);
            for (int i = 0; i < cw; i++)
                for (int k = 0; k < hs; k++) row[size_t(i) * hs + k] = in[i];
        }
        (void)scratch;
    }
    static void h2_row(const uint8_t* in, int w, uint8_t* out) {
        if (w == 1) {
            out[0] = out[1] = in[0];
            return;
        }
        out[0] = in[0];
        out[1] = uint8_t((in[0] * 3 + in[1] + 2) >> 2);
        for (int i = 1; i < w - 1; i++) {
            const int n = 3 * in[i] + 2;
            out[2 * i] = uint8_t((n + in[i - 1]) >> 2);
            out[2 * i + 1] = uint8_t((n + in[i + 1]) >> 2);
        }
        out[2 * (w - 1)] = uint8_t((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[2 * (w - 1) + 1] = in[w - 1];
    }

    void to_rgb(std::vector<uint8_t>& rgb) const {
        rgb.assign(size_t(img_w) * img_h * 3, 0);
        std::vector<uint8_t> rows[3], scratch;
        for (int y = 0; y < img_h; y++) {
            for (int c = 0; c < n_comp; c++) upsample_row(comp[c], y, rows[c], scratch);
            uint8_t* out = rgb.data() + size_t(y) * img_w * 3;
            if (n_comp == 1) {
                for (int x = 0; x < img_w; x++) out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = rows[0][x];
                continue;
            }
            for (int x = 0; x < img_w; x++) {
                const int yf = (int(rows[0][x]) << 20) + (1 << 19);
                const int cb = int(rows[1][x]) - 128, cr = int(rows[2][x]) - 128;
                // 1.40200, 0.71414, 0.34414, 1.77200 as round(x * 4096) << 8
                int r = yf + cr * (5743 << 8);
                int g = yf + cr * -(2925 << 8) + int(uint32_t(cb * -(1410 << 8)) & 0xffff0000u);
                int b = yf + cb * (7258 << 8);
                r >>= 20;
                g >>= 20;
                b >>= 20;
                out[3 * x] = clamp8(r);
                out[3 * x + 1] = clamp8(g);
                out[3 * x + 2] = clamp8(b);
            }
        }
    }
};

}  // namespace rtk

#endif  // RTK_JPEG_H
