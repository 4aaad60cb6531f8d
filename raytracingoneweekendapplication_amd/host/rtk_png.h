// rtk_png.h -- PNG decoding for image_texture (second half).
//
// rtw_image (rtw_stb_image.h:23-66) loads whatever stb_image loads; besides the two
// JPEGs main.cpp names, the reference repository is full of PNGs (Images/*.png),
// and a user pointing image_texture at one of them gets it decoded.  PNG is
// lossless, so -- unlike JPEG (rtk_jpeg.h) -- any correct decoder yields the same
// samples; what has to follow stb_image is only how samples become the three
// 8-bit channels rtw_image asks for (stbi_load with req_comp = 3):
//   grey            -> replicated to R = G = B;   grey + alpha / RGBA -> alpha dropped
//   palette         -> looked up in PLTE (tRNS ignored once alpha is dropped)
//   1/2/4-bit grey  -> scaled by 255 / (2^depth - 1);   16-bit -> the high byte
//   interlaced      -> Adam7 passes de-interlaced;   gAMA / sRGB / iCCP -> ignored
// Written from the PNG specification (ISO/IEC 15948) and RFC 1950/1951; no code is
// taken from stb_image.  tests/test_image_decode.py pins the result against the
// reference's own loader (oracle/_ref `texels`) on generated fixtures of every
// colour type / depth / interlace combination and, where /root/reference exists,
// on PNG files of the reference itself.
#ifndef RTK_PNG_H
#define RTK_PNG_H

#include <cstdint>
#include <cstring>
#include <vector>

namespace rtk {

class png_decoder {
public:
    // Decodes to interleaved RGB8.  Returns false on malformed or unsupported input.
    bool decode(const uint8_t* data, size_t size, int& width, int& height, std::vector<uint8_t>& rgb) {
        static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
        if (size < 8 || std::memcmp(data, sig, 8) != 0) return false;
        size_t pos = 8;
        bool have_header = false;
        uint32_t w = 0, h = 0;
        int depth = 0, colour = 0, interlace = 0;
        std::vector<uint8_t> palette, idat;
        for (;;) {
            if (size - pos < 12) return false;
            const uint32_t len = be32(data + pos);
            const uint8_t* type = data + pos + 4;
            if (len > size - pos - 12) return false;
            const uint8_t* body = data + pos + 8;
            if (std::memcmp(type, "IHDR", 4) == 0) {
                if (have_header || len != 13) return false;
                w = be32(body);
                h = be32(body + 4);
                depth = body[8];
                colour = body[9];
                interlace = body[12];
                if (w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24) || body[10] != 0 || body[11] != 0 || interlace > 1) return false;
                if (!valid_format(colour, depth)) return false;
                have_header = true;
            } else if (std::memcmp(type, "PLTE", 4) == 0) {
                if (!have_header || len % 3 != 0 || len > 768) return false;
                palette.assign(body, body + len);
            } else if (std::memcmp(type, "IDAT", 4) == 0) {
                if (!have_header) return false;
                idat.insert(idat.end(), body, body + len);
            } else if (std::memcmp(type, "IEND", 4) == 0) {
                break;
            } else if (!(type[0] & 0x20)) {
                if (std::memcmp(type, "IHDR", 4) != 0) return false;  // an unknown critical chunk
            }
            pos += size_t(len) + 12;
        }
        if (!have_header || idat.empty()) return false;
        if (colour == 3 && palette.empty()) return false;

        const int channels = colour == 0 ? 1 : (colour == 2 ? 3 : (colour == 3 ? 1 : (colour == 4 ? 2 : 4)));
        std::vector<uint8_t> raw;
        if (!inflate_zlib(idat, raw)) return false;

        // The header's dimensions are only believed once the inflated data can pay for them: the filtered size of every
        // pass (one filter byte + the packed row, per row) must be present BEFORE anything is sized from w x h.  (A
        // 60-byte file may claim 2^24 x 2^24 pixels; deflate expands at most ~1032:1, so `raw` itself is bounded by the file.)
        if (uint64_t(w) * h > kMaxPixels) return false;
        {
            uint64_t need = 0;
            auto pass_bytes = [&](uint64_t pw, uint64_t ph) { return (pw == 0 || ph == 0) ? 0 : ph * (1 + (pw * uint64_t(channels) * depth + 7) / 8); };
            if (!interlace) {
                need = pass_bytes(w, h);
            } else {
                static const int ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
                for (int p = 0; p < 7; p++)
                    need += pass_bytes(w > uint32_t(ax0[p]) ? (w - ax0[p] + adx[p] - 1) / adx[p] : 0, h > uint32_t(ay0[p]) ? (h - ay0[p] + ady[p] - 1) / ady[p] : 0);
            }
            if (uint64_t(raw.size()) < need) return false;
        }

        // samples of the whole image as 8-bit channels (palette indices stay indices), then to RGB
        std::vector<uint8_t> samples(size_t(w) * h * channels);
        size_t offset = 0;
        if (!interlace) {
            if (!unfilter_pass(raw, offset, w, h, channels, depth, colour, samples.data(), w, 0, 0, 1, 1)) return false;
        } else {
            static const int x0[7] = {0, 4, 0, 2, 0, 1, 0}, y0[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
            for (int p = 0; p < 7; p++) {
                const uint32_t pw = (w > uint32_t(x0[p])) ? (w - x0[p] + dx[p] - 1) / dx[p] : 0;
                const uint32_t ph = (h > uint32_t(y0[p])) ? (h - y0[p] + dy[p] - 1) / dy[p] : 0;
                if (pw == 0 || ph == 0) continue;
                if (!unfilter_pass(raw, offset, pw, ph, channels, depth, colour, samples.data(), w, x0[p], y0[p], dx[p], dy[p])) return false;
            }
        }
        width = int(w);
        height = int(h);
        rgb.resize(size_t(w) * h * 3);
        for (size_t i = 0; i < size_t(w) * h; i++) {
            const uint8_t* s = &samples[i * channels];
            uint8_t* o = &rgb[i * 3];
            if (colour == 3) {
                const size_t idx = size_t(s[0]) * 3;
                if (idx + 2 < palette.size()) { o[0] = palette[idx]; o[1] = palette[idx + 1]; o[2] = palette[idx + 2]; }
                else { o[0] = o[1] = o[2] = 0; }
            } else if (channels <= 2) {
                o[0] = o[1] = o[2] = s[0];
            } else {
                o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
            }
        }
        return true;
    }

private:
    static constexpr uint64_t kMaxPixels = uint64_t(1) << 27;  // 134 Mpixel: larger images are refused (stb caps each dimension at 2^24)
    static uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
    static bool valid_format(int colour, int depth) {
        switch (colour) {
            case 0: return depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16;
            case 3: return depth == 1 || depth == 2 || depth == 4 || depth == 8;
            case 2: case 4: case 6: return depth == 8 || depth == 16;
        }
        return false;
    }

    // ---- RFC 1950 / 1951
    struct bit_reader {
        const uint8_t* p;
        const uint8_t* end;
        uint32_t buf = 0;
        int cnt = 0;
        bool need(int n) {
            while (cnt < n) {
                if (p >= end) return false;
                buf |= uint32_t(*p++) << cnt;
                cnt += 8;
            }
            return true;
        }
        bool bits(int n, uint32_t& v) {
            if (n == 0) { v = 0; return true; }
            if (!need(n)) return false;
            v = buf & ((1u << n) - 1);
            buf >>= n;
            cnt -= n;
            return true;
        }
    };
    struct huffman {
        uint16_t count[16];
        uint16_t symbol[288];
        bool build(const uint8_t* lengths, int n) {
            std::memset(count, 0, sizeof count);
            for (int i = 0; i < n; i++) count[lengths[i]]++;
            count[0] = 0;
            int left = 1;
            for (int len = 1; len < 16; len++) {
                left = (left << 1) - count[len];
                if (left < 0) return false;  // over-subscribed
            }
            uint16_t offs[16];
            offs[1] = 0;
            for (int len = 1; len < 15; len++) offs[len + 1] = uint16_t(offs[len] + count[len]);
            for (int i = 0; i < n; i++)
                if (lengths[i]) symbol[offs[lengths[i]]++] = uint16_t(i);
            return true;
        }
        int decode(bit_reader& br) const {
            int code = 0, first = 0, index = 0;
            for (int len = 1; len < 16; len++) {
                uint32_t b;
                if (!br.bits(1, b)) return -1;
                code |= int(b);
                const int c = count[len];
                if (code - c < first) return symbol[index + (code - first)];
                index += c;
                first += c;
                first <<= 1;
                code <<= 1;
            }
            return -1;
        }
    };

    static bool inflate_zlib(const std::vector<uint8_t>& in, std::vector<uint8_t>& out) {
        if (in.size() < 6) return false;
        if ((in[0] & 0x0F) != 8 || ((uint32_t(in[0]) << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) return false;
        bit_reader br{in.data() + 2, in.data() + in.size()};
        static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (;;) {
            uint32_t last, type;
            if (!br.bits(1, last) || !br.bits(2, type)) return false;
            if (type == 0) {
                br.buf = 0;
                br.cnt = 0;  // to the byte boundary
                if (br.end - br.p < 4) return false;
                const uint32_t len = br.p[0] | (uint32_t(br.p[1]) << 8), nlen = br.p[2] | (uint32_t(br.p[3]) << 8);
                br.p += 4;
                if ((len ^ 0xFFFFu) != nlen || uint32_t(br.end - br.p) < len) return false;
                out.insert(out.end(), br.p, br.p + len);
                br.p += len;
            } else if (type == 1 || type == 2) {
                huffman lit, dist;
                if (type == 1) {
                    uint8_t l[288];
                    for (int i = 0; i < 144; i++) l[i] = 8;
                    for (int i = 144; i < 256; i++) l[i] = 9;
                    for (int i = 256; i < 280; i++) l[i] = 7;
                    for (int i = 280; i < 288; i++) l[i] = 8;
                    uint8_t d[30];
                    for (int i = 0; i < 30; i++) d[i] = 5;
                    if (!lit.build(l, 288) || !dist.build(d, 30)) return false;
                } else {
                    uint32_t hlit, hdist, hclen;
                    if (!br.bits(5, hlit) || !br.bits(5, hdist) || !br.bits(4, hclen)) return false;
                    hlit += 257;
                    hdist += 1;
                    hclen += 4;
                    if (hlit > 286 || hdist > 30) return false;
                    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    uint8_t cl[19] = {0};
                    for (uint32_t i = 0; i < hclen; i++) {
                        uint32_t v;
                        if (!br.bits(3, v)) return false;
                        cl[order[i]] = uint8_t(v);
                    }
                    huffman clh;
                    if (!clh.build(cl, 19)) return false;
                    uint8_t lengths[320];
                    uint32_t n = 0;
                    while (n < hlit + hdist) {
                        int sym = clh.decode(br);
                        if (sym < 0) return false;
                        if (sym < 16) {
                            lengths[n++] = uint8_t(sym);
                        } else {
                            uint32_t rep, prev = 0;
                            if (sym == 16) {
                                if (n == 0 || !br.bits(2, rep)) return false;
                                prev = lengths[n - 1];
                                rep += 3;
                            } else if (sym == 17) {
                                if (!br.bits(3, rep)) return false;
                                rep += 3;
                            } else {
                                if (!br.bits(7, rep)) return false;
                                rep += 11;
                            }
                            if (n + rep > hlit + hdist) return false;
                            while (rep--) lengths[n++] = uint8_t(prev);
                        }
                    }
                    if (lengths[256] == 0) return false;
                    if (!lit.build(lengths, int(hlit)) || !dist.build(lengths + hlit, int(hdist))) return false;
                }
                for (;;) {
                    int sym = lit.decode(br);
                    if (sym < 0) return false;
                    if (sym < 256) {
                        out.push_back(uint8_t(sym));
                    } else if (sym == 256) {
                        break;
                    } else {
                        sym -= 257;
                        if (sym >= 29) return false;
                        uint32_t extra;
                        if (!br.bits(len_extra[sym], extra)) return false;
                        const uint32_t len = len_base[sym] + extra;
                        int ds = dist.decode(br);
                        if (ds < 0 || ds >= 30) return false;
                        if (!br.bits(dist_extra[ds], extra)) return false;
                        const size_t d = dist_base[ds] + extra;
                        if (d > out.size()) return false;
                        const size_t from = out.size() - d;
                        for (uint32_t k = 0; k < len; k++) out.push_back(out[from + k]);
                    }
                }
            } else {
                return false;
            }
            if (last) break;
        }
        return true;
    }

    static uint8_t paeth(int a, int b, int c) {
        const int p = a + b - c;
        const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
        return uint8_t((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
    }

    // One (sub-)image of pw x ph pixels: reverse the scanline filters, widen / narrow to 8-bit samples and scatter
    // them into `dst` (full-image sample buffer, `stride_px` pixels per row) at x0 + i*dx, y0 + j*dy.
    static bool unfilter_pass(const std::vector<uint8_t>& raw, size_t& offset, uint32_t pw, uint32_t ph, int channels, int depth, int colour,
                              uint8_t* dst, uint32_t stride_px, int x0, int y0, int dx, int dy) {
        const size_t bits_per_px = size_t(channels) * depth;
        const size_t row_bytes = (size_t(pw) * bits_per_px + 7) / 8;
        const size_t bpp = bits_per_px >= 8 ? bits_per_px / 8 : 1;  // filter distance in bytes
        if (raw.size() - offset < (row_bytes + 1) * ph) return false;
        std::vector<uint8_t> prev(row_bytes, 0), cur(row_bytes);
        for (uint32_t j = 0; j < ph; j++) {
            const uint8_t filter = raw[offset];
            const uint8_t* in = &raw[offset + 1];
            offset += row_bytes + 1;
            for (size_t i = 0; i < row_bytes; i++) {
                const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
                int v;
                switch (filter) {
                    case 0: v = in[i]; break;
                    case 1: v = in[i] + a; break;
                    case 2: v = in[i] + b; break;
                    case 3: v = in[i] + ((a + b) >> 1); break;
                    case 4: v = in[i] + paeth(a, b, c); break;
                    default: return false;
                }
                cur[i] = uint8_t(v);
            }
            uint8_t* row = dst + (size_t(y0 + int(j) * dy) * stride_px) * channels;
            for (uint32_t i = 0; i < pw; i++) {
                uint8_t* px = row + size_t(x0 + int(i) * dx) * channels;
                for (int ch = 0; ch < channels; ch++) {
                    const size_t s = size_t(i) * channels + ch;  // sample index in the row
                    uint8_t v;
                    if (depth == 8) {
                        v = cur[s];
                    } else if (depth == 16) {
                        v = cur[2 * s];  // the high byte
                    } else {
                        const size_t bit = s * depth;
                        const int shift = 8 - depth - int(bit & 7);
                        v = uint8_t((cur[bit >> 3] >> shift) & ((1 << depth) - 1));
                        if (colour == 0) v = uint8_t(v * (255 / ((1 << depth) - 1)));  // grey is scaled to 0..255; palette indices are not
                    }
                    px[ch] = v;
                }
            }
            prev.swap(cur);
        }
        return true;
    }
};

}  // namespace rtk

#endif  // RTK_PNG_H
