// Jpeg.cpp -- baseline (sequential Huffman, 8-bit) JPEG decoder for the colour frames of .klg logs
// (GUI/Tools/KlgLogReader.cpp:76-79 hands them to libjpeg through JPEGLoader).  libjpeg's headers are not part of this
// image, so the decoder is self-contained: SOF0/SOF1, 1 or 3 components, any sampling factors, restart intervals;
// progressive / arithmetic / 12-bit streams are rejected.
// IDCT (jidctint "islow"), chroma upsampling ("fancy" h2v1 / h2v2) and colour conversion follow libjpeg's defaults, so
// the pixels match what the reference's JPEGLoader produces (tests compare against Pillow's libjpeg).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace cofusion {

namespace {

struct Huff {
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    bool present = false;
    void build()
    {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k; mincode[l] = code;
            code += bits[l]; k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int n = 0; bool bad = false;
    int bit()
    {
        if (n == 0) {
            if (p >= end) { bad = true; return 0; }
            uint8_t b = *p++;
            if (b == 0xFF) {
                if (p < end && *p == 0x00) p++;          // stuffed zero
                else { bad = true; p--; return 0; }      // a marker inside entropy data: let the caller see it
            }
            acc = b; n = 8;
        }
        n--;
        return (acc >> n) & 1;
    }
    int receive(int s) { int v = 0; for (int i = 0; i < s; i++) v = (v << 1) | bit(); return v; }
    void reset() { n = 0; acc = 0; }
};

inline int extend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }

int decode_symbol(BitReader& br, const Huff& h)
{
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.bit();
        if (br.bad) return -1;
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}

const int kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                         35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// libjpeg's default "islow" integer IDCT (jidctint.c: 13-bit constants, 2 extra bits after the column pass), so that
// decoded pixels match what the reference's JPEGLoader (libjpeg) produces.
inline long descale(long x, int n) { return (x + (1L << (n - 1))) >> n; }
void idct8x8(const int* coef, uint8_t* out, int stride)
{
    constexpr int CB = 13, P1 = 2;
    constexpr long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
                   F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
    long ws[64];
    for (int c = 0; c < 8; c++) {  // pass 1: columns
        const int* in = coef + c;
        if (!(in[8] | in[16] | in[24] | in[32] | in[40] | in[48] | in[56])) {
            const long dc = (long)in[0] * (1 << P1);
            for (int r = 0; r < 8; r++) ws[r * 8 + c] = dc;
            continue;
        }
        long z2 = in[16], z3 = in[48];
        long z1 = (z2 + z3) * F0541;
        long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        z2 = in[0]; z3 = in[32];
        long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3; const long z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[0 * 8 + c] = descale(tmp10 + tmp3, CB - P1); ws[7 * 8 + c] = descale(tmp10 - tmp3, CB - P1);
        ws[1 * 8 + c] = descale(tmp11 + tmp2, CB - P1); ws[6 * 8 + c] = descale(tmp11 - tmp2, CB - P1);
        ws[2 * 8 + c] = descale(tmp12 + tmp1, CB - P1); ws[5 * 8 + c] = descale(tmp12 - tmp1, CB - P1);
        ws[3 * 8 + c] = descale(tmp13 + tmp0, CB - P1); ws[4 * 8 + c] = descale(tmp13 - tmp0, CB - P1);
    }
    auto put = [&](int r, int c, long v) {
        const long p = descale(v, CB + P1 + 3) + 128;
        out[r * stride + c] = (uint8_t)(p < 0 ? 0 : (p > 255 ? 255 : p));
    };
    for (int r = 0; r < 8; r++) {  // pass 2: rows
        const long* in = ws + r * 8;
        long z2 = in[2], z3 = in[6];
        long z1 = (z2 + z3) * F0541;
        long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        long tmp0 = (in[0] + in[4]) * (1L << CB), tmp1 = (in[0] - in[4]) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3; const long z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        put(r, 0, tmp10 + tmp3); put(r, 7, tmp10 - tmp3); put(r, 1, tmp11 + tmp2); put(r, 6, tmp11 - tmp2);
        put(r, 2, tmp12 + tmp1); put(r, 5, tmp12 - tmp1); put(r, 3, tmp13 + tmp0); put(r, 4, tmp13 - tmp0);
    }
}

struct Component { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0, stride = 0, rows = 0; std::vector<uint8_t> plane; };

}  // namespace

// Decodes into rgb [height*width*3]; returns "" on success or an error text.  The image must have the expected size.
std::string decodeJpegRGB(const uint8_t* data, size_t size, int width, int height, uint8_t* rgb)
{
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return "not a JPEG stream";
    uint16_t qt[4][64] = {{0}};
    Huff dc[4], ac[4];
    std::vector<Component> comps;
    int W = 0, H = 0, restart = 0, hmax = 1, vmax = 1;
    size_t pos = 2;
    auto be16 = [&](size_t p) { return (int)((data[p] << 8) | data[p + 1]); };
    while (pos + 4 <= size) {
        if (data[pos] != 0xFF) return "marker expected";
        const int m = data[pos + 1];
        if (m == 0xFF) { pos++; continue; }
        pos += 2;
        if (m == 0xD9) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > size) return "truncated segment";
        const int len = be16(pos);
        if (len < 2 || pos + len > size) return "truncated segment";
        const size_t seg = pos + 2, segend = pos + len;
        if (m == 0xDB) {  // DQT
            size_t p = seg;
            while (p < segend) {
                const int pq = data[p] >> 4, tq = data[p] & 15; p++;
                if (tq > 3) return "bad quantisation table id";
                if (p + (pq ? 128 : 64) > segend) return "truncated quantisation table";
                for (int i = 0; i < 64; i++) { qt[tq][kZigzag[i]] = pq ? (uint16_t)be16(p) : data[p]; p += pq ? 2 : 1; }
            }
        } else if (m == 0xC4) {  // DHT
            size_t p = seg;
            while (p < segend) {
                const int tc = data[p] >> 4, th = data[p] & 15; p++;
                if (th > 3 || tc > 1) return "bad Huffman table id";
                Huff& h = tc ? ac[th] : dc[th];
                int total = 0;
                if (p + 16 > segend) return "truncated Huffman table";
                for (int l = 1; l <= 16; l++) { h.bits[l] = data[p++]; total += h.bits[l]; }
                if (total > 256 || p + total > segend) return "bad Huffman table";
                memcpy(h.vals, data + p, (size_t)total); p += total;
                h.build();
            }
        } else if (m == 0xC0 || m == 0xC1) {  // SOF0 / SOF1
            if (len < 8) return "truncated frame header";
            if (data[seg] != 8) return "only 8-bit JPEG is supported";
            H = be16(seg + 1); W = be16(seg + 3);
            const int nc = data[seg + 5];
            if (nc != 1 && nc != 3) return "only 1- or 3-component JPEG is supported";
            if (len < 8 + 3 * nc) return "truncated frame header";
            comps.resize(nc);
            for (int i = 0; i < nc; i++) {
                comps[i].id = data[seg + 6 + i * 3]; comps[i].h = data[seg + 7 + i * 3] >> 4; comps[i].v = data[seg + 7 + i * 3] & 15;
                comps[i].tq = data[seg + 8 + i * 3];
                if (comps[i].h < 1 || comps[i].h > 4 || comps[i].v < 1 || comps[i].v > 4 || comps[i].tq > 3) return "bad frame header";
                if (comps[i].h > hmax) hmax = comps[i].h;
                if (comps[i].v > vmax) vmax = comps[i].v;
            }
        } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8)) {
            return "progressive / lossless / arithmetic JPEG is not supported";
        } else if (m == 0xDD) {
            if (len < 4) return "truncated restart interval";
            restart = be16(seg);
        } else if (m == 0xDA) {  // SOS: decode the (single, interleaved) scan
            if (comps.empty()) return "scan before frame header";
            if (W != width || H != height) return "JPEG size differs from the log's resolution";
            if (len < 3) return "truncated scan header";
            const int ns = data[seg];
            if (ns != (int)comps.size()) return "non-interleaved scans are not supported";
            if (len < 6 + 2 * ns) return "truncated scan header";
            for (int i = 0; i < ns; i++) {
                const int cid = data[seg + 1 + i * 2], t = data[seg + 2 + i * 2];
                bool found = false;
                for (auto& c : comps) if (c.id == cid) { c.td = t >> 4; c.ta = t & 15; found = true; }
                if (!found || (t >> 4) > 3 || (t & 15) > 3) return "bad scan header";
            }
            const int mcuw = 8 * hmax, mcuh = 8 * vmax;
            const int mx = (W + mcuw - 1) / mcuw, my = (H + mcuh - 1) / mcuh;
            for (auto& c : comps) {
                c.stride = mx * c.h * 8; c.rows = my * c.v * 8; c.pred = 0;
                c.plane.assign((size_t)c.stride * c.rows, 0);
                if (!dc[c.td].present || !ac[c.ta].present) return "missing Huffman table";
            }
            BitReader br{data + segend, data + size};
            int count = 0;
            for (int yy = 0; yy < my; yy++)
                for (int xx = 0; xx < mx; xx++) {
                    if (restart && count && count % restart == 0) {  // RSTn
                        br.reset();
                        while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) br.p++;
                        if (br.p + 1 >= br.end) return "missing restart marker";
                        br.p += 2; br.bad = false;
                        for (auto& c : comps) c.pred = 0;
                    }
                    count++;
                    for (auto& c : comps)
                        for (int by = 0; by < c.v; by++)
                            for (int bx = 0; bx < c.h; bx++) {
                                int coef[64] = {0};
                                const int t = decode_symbol(br, dc[c.td]);
                                if (t < 0 || t > 11) return "corrupt DC coefficient";
                                c.pred += extend(br.receive(t), t);
                                coef[0] = c.pred * qt[c.tq][0];
                                for (int k = 1; k < 64;) {
                                    const int rs = decode_symbol(br, ac[c.ta]);
                                    if (rs < 0) return "corrupt AC coefficient";
                                    const int r = rs >> 4, s = rs & 15;
                                    if (s == 0) { if (r == 15) { k += 16; continue; } break; }
                                    k += r;
                                    if (k > 63) return "corrupt AC run";
                                    coef[kZigzag[k]] = extend(br.receive(s), s) * qt[c.tq][kZigzag[k]];
                                    k++;
                                }
                                if (br.bad) return "truncated entropy data";
                                idct8x8(coef, &c.plane[(size_t)((yy * c.v + by) * 8) * c.stride + (xx * c.h + bx) * 8], c.stride);
                            }
                }
            // chroma upsampling as libjpeg does by default ("fancy" triangle filter for 2x1 and 2x2, jdsample.c), replication
            // for other factors; colour conversion with libjpeg's 16-bit fixed-point tables (jdcolor.c)
            std::vector<std::vector<uint8_t>> full(comps.size());
            for (size_t ci = 0; ci < comps.size(); ci++) {
                const Component& c = comps[ci];
                std::vector<uint8_t>& o = full[ci];
                o.assign((size_t)W * H, 0);
                const int cw = (W * c.h + hmax - 1) / hmax, ch = (H * c.v + vmax - 1) / vmax;  // downsampled size
                auto at = [&](int x, int y) { return (int)c.plane[(size_t)(y < 0 ? 0 : (y >= ch ? ch - 1 : y)) * c.stride + (x < 0 ? 0 : (x >= cw ? cw - 1 : x))]; };
                if (c.h == hmax && c.v == vmax) {
                    for (int y = 0; y < H; y++) memcpy(&o[(size_t)y * W], &c.plane[(size_t)y * c.stride], (size_t)W);
                } else if (c.h * 2 == hmax && c.v == vmax) {  // h2v1 fancy
                    for (int y = 0; y < H; y++)
                        for (int x = 0; x < W; x++) {
                            const int i = x >> 1;
                            int v;
                            if (x & 1) v = (i == cw - 1) ? at(i, y) : (3 * at(i, y) + at(i + 1, y) + 2) >> 2;
                            else v = (i == 0) ? at(0, y) : (3 * at(i, y) + at(i - 1, y) + 1) >> 2;
                            o[(size_t)y * W + x] = (uint8_t)v;
                        }
                } else if (c.h * 2 == hmax && c.v * 2 == vmax) {  // h2v2 fancy
                    for (int y = 0; y < H; y++) {
                        const int r = y >> 1, rn = (y & 1) ? r + 1 : r - 1;
                        for (int x = 0; x < W; x++) {
                            const int i = x >> 1;
                            const int cur = 3 * at(i, r) + at(i, rn);
                            int v;
                            if (x & 1) v = (i == cw - 1) ? (cur * 4 + 7) >> 4 : (cur * 3 + 3 * at(i + 1, r) + at(i + 1, rn) + 7) >> 4;
                            else v = (i == 0) ? (cur * 4 + 8) >> 4 : (cur * 3 + 3 * at(i - 1, r) + at(i - 1, rn) + 8) >> 4;
                            o[(size_t)y * W + x] = (uint8_t)v;
                        }
                    }
                } else {
                    for (int y = 0; y < H; y++)
                        for (int x = 0; x < W; x++) o[(size_t)y * W + x] = (uint8_t)at(x * c.h / hmax, y * c.v / vmax);
                }
            }
            auto fix = [](double v) { return (long)(v * 65536.0 + 0.5); };
            const long crr = fix(1.40200), cbb = fix(1.77200), crg = -fix(0.71414), cbg = -fix(0.34414), half = 32768;
            auto clamp = [](long r) { return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r)); };
            for (size_t q = 0; q < (size_t)W * H; q++) {
                uint8_t* o = rgb + q * 3;
                const long Y = full[0][q];
                if (comps.size() == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; continue; }
                const long cb = (long)full[1][q] - 128, cr = (long)full[2][q] - 128;
                o[0] = clamp(Y + ((crr * cr + half) >> 16));
                o[1] = clamp(Y + ((cbg * cb + half + crg * cr) >> 16));
                o[2] = clamp(Y + ((cbb * cb + half) >> 16));
            }
            return "";
        }
        pos = segend;
    }
    return "no scan found";
}

}  // namespace cofusion
