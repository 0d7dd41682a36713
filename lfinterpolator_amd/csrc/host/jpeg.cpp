// jpeg.cpp — JPEG input for LfLoader: baseline / extended-sequential and progressive Huffman JPEG, 8 bit, 1 / 3 / 4 components.
//
// The reference decodes its inputs with the stb_image it vendors (src/lfLoader.cpp:36, stbi_load(…, STBI_rgb_alpha)).  A JPEG
// decoder is only fixed by the standard up to its IDCT, chroma upsampling and colour conversion, so to hand the kernels the
// SAME bytes the reference's loader would, those three stages follow that decoder's arithmetic:
//   * IDCT: the "islow" integer algorithm of the IJG code with 12-bit constants, two extra bits kept between the passes,
//     +65536 + (128 << 17) before the final >> 17, coefficients × quantiser truncated to 16 bit first;
//   * upsampling: h2v1 / h1v2 (3·near + far + 2) >> 2, h2v2 the separable 3:1 triangle ((3a + b + 8) >> 4 of the vertically
//     filtered rows), everything else nearest; rows advance with the decoder's near / far line rule;
//   * YCbCr → RGB in 20-bit fixed point (1.402, 0.71414, 0.34414 — its product masked to the upper 16 bits —, 1.772).
// tests/test_host_io.py checks this file against oracle/_ref (the reference's codec built from the reference tree) pixel for pixel.
// Entropy decoding (Annex F / G of ITU T.81) is plain canonical-Huffman code, written for this file.
#include "image_io.h"

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace lfi {

namespace {

const uint8_t kZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                  6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                  39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  // a run past the end of a corrupt block lands here instead of outside the array
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman
{
    // canonical code: for each length 1..16 the first code, the index of its first symbol and the number of codes
    int first[17]{}, index[17]{}, count[17]{};
    uint8_t symbols[256]{};
    bool defined{false};
};

struct Component
{
    int id{0}, h{1}, v{1}, tq{0}, td{0}, ta{0};
    int x{0}, y{0};   // samples that carry image
    int w2{0}, h2{0}; // plane size: whole MCUs
    int pred{0};
    std::vector<uint8_t> plane;  // decoded samples
    std::vector<int16_t> coeffs; // progressive: all blocks' coefficients, natural order
    int blocksW{0}, blocksH{0};  // blocks per row / column of the coefficient array
};

class Decoder
{
  public:
    Decoder(const std::vector<uint8_t> &file, const std::string &path) : d(file), path(path) {}

    Image decode()
    {
        if(d.size() < 4 || d[0] != 0xff || d[1] != 0xd8)
            fail("not a JPEG");
        pos = 2;
        bool done = false;
        while(!done)
        {
            const int m = nextMarker();
            switch(m)
            {
                case 0xc0:
                case 0xc1:
                case 0xc2: frameHeader(m == 0xc2); break;
                case 0xc4: huffmanTables(); break;
                case 0xdb: quantTables(); break;
                case 0xdd: restartInterval(); break;
                case 0xda:
                    scan();
                    break;
                case 0xd9: done = true; break;
                case 0xe0: app0(); break;
                case 0xee: app14(); break;
                default:
                    if((m >= 0xc3 && m <= 0xcf) && m != 0xc4 && m != 0xc8 && m != 0xcc)
                        fail("unsupported JPEG process (lossless, hierarchical or arithmetic coding)");
                    skipSegment();
                    break;
            }
        }
        if(!haveFrame)
            fail("no frame header");
        if(progressive)
            finishProgressive();
        return assemble();
    }

  private:
    const std::vector<uint8_t> &d;
    const std::string &path;
    size_t pos{0};
    // frame
    bool haveFrame{false}, progressive{false}, jfif{false};
    int adobeTransform{-1};
    int width{0}, height{0}, ncomp{0}, hmax{1}, vmax{1}, mcusX{0}, mcusY{0}, restart{0};
    bool rgbIds{false};
    Component comp[4];
    uint16_t quant[4][64]{};
    Huffman dc[4], ac[4];
    // scan state
    int scanN{0}, order[4]{}, ss{0}, se{63}, ah{0}, al{0}, eobrun{0};
    // bit reader
    uint32_t bits{0};
    int nbits{0};
    bool hitMarker{false};

    [[noreturn]] void fail(const char *why) const { throw std::runtime_error("Cannot load image " + path + " (" + why + ")"); }

    int u8()
    {
        if(pos >= d.size())
            fail("truncated JPEG");
        return d[pos++];
    }
    int u16()
    {
        const int hi = u8();
        return hi << 8 | u8();
    }
    int nextMarker()
    {
        int c = u8();
        while(c != 0xff) // tolerate padding between segments
            c = u8();
        while(c == 0xff)
            c = u8();
        return c;
    }
    void skipSegment()
    {
        const int len = u16();
        if(len < 2 || pos + size_t(len - 2) > d.size())
            fail("bad segment length");
        pos += size_t(len - 2);
    }
    void app0()
    {
        const size_t start = pos;
        const int len = u16();
        if(len >= 7 && pos + 5 <= d.size() && !std::memcmp(&d[pos], "JFIF\0", 5))
            jfif = true;
        pos = start;
        skipSegment();
    }
    void app14()
    {
        const size_t start = pos;
        const int len = u16();
        if(len >= 14 && pos + 12 <= d.size() && !std::memcmp(&d[pos], "Adobe\0", 6))
            adobeTransform = d[pos + 11];
        pos = start;
        skipSegment();
    }
    void quantTables()
    {
        int len = u16() - 2;
        while(len > 0)
        {
            const int q = u8(), sixteen = q >> 4, t = q & 15;
            if((sixteen != 0 && sixteen != 1) || t > 3)
                fail("bad DQT");
            for(int i = 0; i < 64; i++)
                quant[t][kZigzag[i]] = uint16_t(sixteen ? u16() : u8());
            len -= sixteen ? 129 : 65;
        }
        if(len != 0)
            fail("bad DQT length");
    }
    void huffmanTables()
    {
        int len = u16() - 2;
        while(len > 0)
        {
            const int q = u8(), tc = q >> 4, th = q & 15;
            if(tc > 1 || th > 3)
                fail("bad DHT");
            Huffman &h = tc ? ac[th] : dc[th];
            int total = 0, code = 0;
            for(int l = 1; l <= 16; l++)
            {
                h.count[l] = u8();
                h.first[l] = code;
                h.index[l] = total;
                code = (code + h.count[l]) << 1;
                total += h.count[l];
            }
            if(total > 256)
                fail("bad DHT");
            for(int i = 0; i < total; i++)
                h.symbols[i] = uint8_t(u8());
            h.defined = true;
            len -= 17 + total;
        }
        if(len != 0)
            fail("bad DHT length");
    }
    void restartInterval()
    {
        if(u16() != 4)
            fail("bad DRI");
        restart = u16();
    }
    void frameHeader(bool isProgressive)
    {
        if(haveFrame)
            fail("several frames");
        const int len = u16();
        if(u8() != 8)
            fail("only 8-bit JPEG is supported");
        height = u16();
        width = u16();
        ncomp = u8();
        if(height == 0 || width == 0 || width > (1 << 15) || height > (1 << 15))
            fail("bad dimensions");
        if((ncomp != 1 && ncomp != 3 && ncomp != 4) || len != 8 + 3 * ncomp)
            fail("bad component count");
        static const char rgb[3] = {'R', 'G', 'B'};
        int matches = 0;
        for(int i = 0; i < ncomp; i++)
        {
            Component &c = comp[i];
            c.id = u8();
            if(ncomp == 3 && c.id == rgb[i])
                matches++;
            const int q = u8();
            c.h = q >> 4;
            c.v = q & 15;
            c.tq = u8();
            if(c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3)
                fail("bad component");
            hmax = std::max(hmax, c.h);
            vmax = std::max(vmax, c.v);
        }
        rgbIds = matches == 3;
        for(int i = 0; i < ncomp; i++)
            if(hmax % comp[i].h || vmax % comp[i].v)
                fail("bad sampling factors");
        mcusX = (width + 8 * hmax - 1) / (8 * hmax);
        mcusY = (height + 8 * vmax - 1) / (8 * vmax);
        for(int i = 0; i < ncomp; i++)
        {
            Component &c = comp[i];
            c.x = (width * c.h + hmax - 1) / hmax;
            c.y = (height * c.v + vmax - 1) / vmax;
            c.w2 = mcusX * c.h * 8;
            c.h2 = mcusY * c.v * 8;
            c.plane.assign(size_t(c.w2) * c.h2, 0);
            if(isProgressive)
            {
                c.blocksW = c.w2 / 8;
                c.blocksH = c.h2 / 8;
                c.coeffs.assign(size_t(c.w2) * c.h2, 0);
            }
        }
        progressive = isProgressive;
        haveFrame = true;
    }

    // ---- entropy-coded data: bits, Huffman symbols, receive + extend -----------------------------------------------------------
    void fill()
    {
        while(nbits <= 24)
        {
            int byte = 0;
            if(!hitMarker && pos < d.size())
            {
                byte = d[pos++];
                if(byte == 0xff)
                {
                    int next = pos < d.size() ? d[pos] : 0xd9;
                    while(next == 0xff && pos + 1 < d.size()) // fill bytes
                        next = d[++pos];
                    if(next == 0)
                        pos++; // stuffed zero
                    else
                    {
                        pos--; // leave the marker for the caller; feed zeros from here on
                        hitMarker = true;
                        byte = 0;
                    }
                }
            }
            bits |= uint32_t(byte) << (24 - nbits);
            nbits += 8;
        }
    }
    int getBits(int n)
    {
        if(n == 0)
            return 0;
        if(nbits < n)
            fill();
        const int v = int(bits >> (32 - n));
        bits <<= n;
        nbits -= n;
        return v;
    }
    int getBit() { return getBits(1); }
    int symbol(const Huffman &h)
    {
        if(!h.defined)
            fail("missing Huffman table");
        if(nbits < 16)
            fill();
        int code = 0;
        for(int l = 1; l <= 16; l++)
        {
            code = code << 1 | int(bits >> 31);
            bits <<= 1;
            nbits--;
            if(code - h.first[l] < h.count[l] && code >= h.first[l])
                return h.symbols[h.index[l] + code - h.first[l]];
        }
        fail("bad Huffman code");
    }
    int receiveExtend(int n)
    {
        if(n == 0)
            return 0;
        const int v = getBits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    void resetEntropy()
    {
        bits = 0;
        nbits = 0;
        hitMarker = false;
        eobrun = 0;
        for(int i = 0; i < 4; i++)
            comp[i].pred = 0;
    }

    // ---- one block ------------------------------------------------------------------------------------------------------------------
    void blockBaseline(Component &c, int16_t (&data)[64])
    {
        std::memset(data, 0, sizeof(data));
        const int t = symbol(dc[c.td]);
        if(t > 15)
            fail("bad DC code");
        c.pred += receiveExtend(t);
        data[0] = int16_t(c.pred * quant[c.tq][0]);
        for(int k = 1; k < 64;)
        {
            const int rs = symbol(ac[c.ta]), r = rs >> 4, s = rs & 15;
            if(s == 0)
            {
                if(rs != 0xf0)
                    break; // end of block
                k += 16;
            }
            else
            {
                k += r;
                const int z = kZigzag[k++];
                data[z] = int16_t(receiveExtend(s) * quant[c.tq][z]);
            }
        }
    }
    void blockProgressiveDC(Component &c, int16_t *data)
    {
        if(ah == 0)
        {
            const int t = symbol(dc[c.td]);
            if(t > 15)
                fail("bad DC code");
            c.pred += receiveExtend(t);
            data[0] = int16_t(c.pred * (1 << al));
        }
        else if(getBit())
            data[0] = int16_t(data[0] + (1 << al));
    }
    void blockProgressiveAC(Component &c, int16_t *data)
    {
        const Huffman &h = ac[c.ta];
        if(ah == 0)
        {
            if(eobrun)
            {
                eobrun--;
                return;
            }
            for(int k = ss; k <= se;)
            {
                const int rs = symbol(h), r = rs >> 4, s = rs & 15;
                if(s == 0)
                {
                    if(r < 15)
                    {
                        eobrun = (1 << r) - 1;
                        if(r)
                            eobrun += getBits(r);
                        break;
                    }
                    k += 16;
                }
                else
                {
                    k += r;
                    data[kZigzag[k++]] = int16_t(receiveExtend(s) * (1 << al));
                }
            }
            return;
        }
        // refinement: one more bit for the coefficients that are already non-zero, new ±1 coefficients in between
        const int16_t bit = int16_t(1 << al);
        auto refine = [&](int16_t *p) {
            if(getBit() && (*p & bit) == 0)
                *p = int16_t(*p > 0 ? *p + bit : *p - bit);
        };
        if(eobrun)
        {
            eobrun--;
            for(int k = ss; k <= se; k++)
            {
                int16_t *p = &data[kZigzag[k]];
                if(*p != 0)
                    refine(p);
            }
            return;
        }
        int k = ss;
        do
        {
            const int rs = symbol(h);
            int r = rs >> 4, s = rs & 15;
            if(s == 0)
            {
                if(r < 15)
                {
                    eobrun = (1 << r) - 1;
                    if(r)
                        eobrun += getBits(r);
                    r = 64; // run to the end of the band, refining on the way
                }
            }
            else
            {
                if(s != 1)
                    fail("bad refinement code");
                s = getBit() ? bit : -bit;
            }
            while(k <= se)
            {
                int16_t *p = &data[kZigzag[k++]];
                if(*p != 0)
                    refine(p);
                else
                {
                    if(r == 0)
                    {
                        *p = int16_t(s);
                        break;
                    }
                    r--;
                }
            }
        } while(k <= se);
    }

    // ---- inverse DCT ("islow", 12-bit constants) --------------------------------------------------------------------------------------
    static constexpr int fx(double x) { return int(x * 4096 + 0.5); }
    struct Butterfly
    {
        int e0, e1, e2, e3, o0, o1, o2, o3; // even part sums, odd part terms: out[k] = e_k ± o_(3-k)
    };
    static Butterfly idct1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7)
    {
        Butterfly b;
        const int z = (s2 + s6) * fx(0.5411961);
        const int ev2 = z + s6 * fx(-1.847759065), ev3 = z + s2 * fx(0.765366865);
        const int ev0 = (s0 + s4) * 4096, ev1 = (s0 - s4) * 4096;
        b.e0 = ev0 + ev3;
        b.e3 = ev0 - ev3;
        b.e1 = ev1 + ev2;
        b.e2 = ev1 - ev2;
        int t0 = s7, t1 = s5, t2 = s3, t3 = s1;
        const int p3 = t0 + t2, p4 = t1 + t3, p1 = t0 + t3, p2 = t1 + t2;
        const int p5 = (p3 + p4) * fx(1.175875602);
        t0 *= fx(0.298631336);
        t1 *= fx(2.053119869);
        t2 *= fx(3.072711026);
        t3 *= fx(1.501321110);
        const int q1 = p5 + p1 * fx(-0.899976223), q2 = p5 + p2 * fx(-2.562915447);
        const int q3 = p3 * fx(-1.961570560), q4 = p4 * fx(-0.390180644);
        b.o3 = t3 + q1 + q4;
        b.o2 = t2 + q2 + q3;
        b.o1 = t1 + q2 + q4;
        b.o0 = t0 + q1 + q3;
        return b;
    }
    static uint8_t clamp8(int v) { return uint8_t(v < 0 ? 0 : v > 255 ? 255 : v); }
    static void idct(uint8_t *out, int stride, const int16_t (&in)[64])
    {
        int mid[64];
        for(int c = 0; c < 8; c++)
        {
            const int16_t *s = in + c;
            if(!(s[8] | s[16] | s[24] | s[32] | s[40] | s[48] | s[56]))
            {
                const int v = s[0] * 4; // a DC-only column: the same value the full pass gives
                for(int r = 0; r < 8; r++)
                    mid[r * 8 + c] = v;
                continue;
            }
            const Butterfly b = idct1d(s[0], s[8], s[16], s[24], s[32], s[40], s[48], s[56]);
            const int e[4] = {b.e0 + 512, b.e1 + 512, b.e2 + 512, b.e3 + 512}, o[4] = {b.o3, b.o2, b.o1, b.o0};
            for(int k = 0; k < 4; k++)
            {
                mid[k * 8 + c] = (e[k] + o[k]) >> 10;
                mid[(7 - k) * 8 + c] = (e[k] - o[k]) >> 10;
            }
        }
        for(int r = 0; r < 8; r++)
        {
            const int *s = mid + r * 8;
            const Butterfly b = idct1d(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
            const int bias = 65536 + (128 << 17);
            const int e[4] = {b.e0 + bias, b.e1 + bias, b.e2 + bias, b.e3 + bias}, o[4] = {b.o3, b.o2, b.o1, b.o0};
            uint8_t *row = out + size_t(r) * stride;
            for(int k = 0; k < 4; k++)
            {
                row[k] = clamp8((e[k] + o[k]) >> 17);
                row[7 - k] = clamp8((e[k] - o[k]) >> 17);
            }
        }
    }

    // ---- scans ------------------------------------------------------------------------------------------------------------------------
    void scan()
    {
        if(!haveFrame)
            fail("scan before frame");
        const int len = u16();
        scanN = u8();
        if(scanN < 1 || scanN > ncomp || len != 6 + 2 * scanN)
            fail("bad SOS");
        for(int i = 0; i < scanN; i++)
        {
            const int id = u8(), q = u8();
            int which = 0;
            while(which < ncomp && comp[which].id != id)
                which++;
            if(which == ncomp || (q >> 4) > 3 || (q & 15) > 3)
                fail("bad SOS component");
            comp[which].td = q >> 4;
            comp[which].ta = q & 15;
            order[i] = which;
        }
        ss = u8();
        se = u8();
        const int a = u8();
        ah = a >> 4;
        al = a & 15;
        if(progressive)
        {
            if(ss > 63 || se > 63 || ss > se || ah > 13 || al > 13)
                fail("bad progressive scan");
        }
        else
        {
            if(ss != 0 || ah != 0 || al != 0)
                fail("bad baseline scan");
            se = 63;
        }
        resetEntropy();
        int todo = restart ? restart : 0x7fffffff;
        auto afterUnit = [&]() {
            if(--todo > 0)
                return true;
            // restart interval over: the next thing in the stream must be RSTn
            if(nbits < 24)
                fill();
            if(!hitMarker || pos + 1 >= d.size() || d[pos] != 0xff || d[pos + 1] < 0xd0 || d[pos + 1] > 0xd7)
                return false; // no restart marker: the scan ends here
            pos += 2;
            resetEntropy();
            todo = restart;
            return true;
        };
        int16_t data[64];
        if(scanN == 1)
        {
            // non-interleaved: the component's own blocks in raster order, only those that carry image
            Component &c = comp[order[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for(int by = 0; by < bh; by++)
                for(int bx = 0; bx < bw; bx++)
                {
                    if(progressive)
                    {
                        int16_t *blk = &c.coeffs[64 * (size_t(by) * c.blocksW + bx)];
                        if(ss == 0)
                            blockProgressiveDC(c, blk);
                        else
                            blockProgressiveAC(c, blk);
                    }
                    else
                    {
                        blockBaseline(c, data);
                        idct(&c.plane[size_t(by) * 8 * c.w2 + size_t(bx) * 8], c.w2, data);
                    }
                    if(!afterUnit())
                        return skipToMarker();
                }
        }
        else
        {
            if(progressive && ss != 0)
                fail("interleaved AC scan");
            for(int my = 0; my < mcusY; my++)
                for(int mx = 0; mx < mcusX; mx++)
                {
                    for(int i = 0; i < scanN; i++)
                    {
                        Component &c = comp[order[i]];
                        for(int v = 0; v < c.v; v++)
                            for(int h = 0; h < c.h; h++)
                            {
                                const int bx = mx * c.h + h, by = my * c.v + v;
                                if(progressive)
                                    blockProgressiveDC(c, &c.coeffs[64 * (size_t(by) * c.blocksW + bx)]);
                                else
                                {
                                    blockBaseline(c, data);
                                    idct(&c.plane[size_t(by) * 8 * c.w2 + size_t(bx) * 8], c.w2, data);
                                }
                            }
                    }
                    if(!afterUnit())
                        return skipToMarker();
                }
        }
        skipToMarker();
    }
    void skipToMarker()
    {
        // whatever the bit reader has not consumed is before pos; the next marker starts at or after pos
        if(hitMarker)
            return;
        while(pos + 1 < d.size() && !(d[pos] == 0xff && d[pos + 1] != 0 && d[pos + 1] != 0xff))
            pos++;
    }
    void finishProgressive()
    {
        int16_t data[64];
        for(int i = 0; i < ncomp; i++)
        {
            Component &c = comp[i];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for(int by = 0; by < bh; by++)
                for(int bx = 0; bx < bw; bx++)
                {
                    const int16_t *blk = &c.coeffs[64 * (size_t(by) * c.blocksW + bx)];
                    for(int k = 0; k < 64; k++)
                        data[k] = int16_t(blk[k] * quant[c.tq][k]);
                    idct(&c.plane[size_t(by) * 8 * c.w2 + size_t(bx) * 8], c.w2, data);
                }
        }
    }

    // ---- upsampling and colour ------------------------------------------------------------------------------------------------------
    static const uint8_t *upsampleRow(std::vector<uint8_t> &line, const uint8_t *nearRow, const uint8_t *farRow, int w, int hs, int vs)
    {
        uint8_t *out = line.data();
        if(hs == 1 && vs == 1)
            return nearRow;
        if(hs == 1 && vs == 2)
        {
            for(int i = 0; i < w; i++)
                out[i] = uint8_t((3 * nearRow[i] + farRow[i] + 2) >> 2);
            return out;
        }
        if(hs == 2 && vs == 1)
        {
            if(w == 1)
            {
                out[0] = out[1] = nearRow[0];
                return out;
            }
            out[0] = nearRow[0];
            out[1] = uint8_t((nearRow[0] * 3 + nearRow[1] + 2) >> 2);
            for(int i = 1; i < w - 1; i++)
            {
                const int n = 3 * nearRow[i] + 2;
                out[2 * i] = uint8_t((n + nearRow[i - 1]) >> 2);
                out[2 * i + 1] = uint8_t((n + nearRow[i + 1]) >> 2);
            }
            out[2 * (w - 1)] = uint8_t((nearRow[w - 2] * 3 + nearRow[w - 1] + 2) >> 2);
            out[2 * (w - 1) + 1] = nearRow[w - 1];
            return out;
        }
        if(hs == 2 && vs == 2)
        {
            int cur = 3 * nearRow[0] + farRow[0];
            if(w == 1)
            {
                out[0] = out[1] = uint8_t((cur + 2) >> 2);
                return out;
            }
            out[0] = uint8_t((cur + 2) >> 2);
            for(int i = 1; i < w; i++)
            {
                const int prev = cur;
                cur = 3 * nearRow[i] + farRow[i];
                out[2 * i - 1] = uint8_t((3 * prev + cur + 8) >> 4);
                out[2 * i] = uint8_t((3 * cur + prev + 8) >> 4);
            }
            out[2 * w - 1] = uint8_t((cur + 2) >> 2);
            return out;
        }
        for(int i = 0; i < w; i++)
            for(int j = 0; j < hs; j++)
                out[i * hs + j] = nearRow[i];
        return out;
    }
    static uint8_t mul255(int x, int y)
    {
        const unsigned t = unsigned(x * y + 128);
        return uint8_t((t + (t >> 8)) >> 8);
    }
    Image assemble()
    {
        Image img;
        img.width = width;
        img.height = height;
        img.pixels.resize(size_t(width) * height * 4);
        struct Resample
        {
            int hs, vs, step, row, wLow;
            const uint8_t *line0, *line1;
            std::vector<uint8_t> buffer;
        } rs[4];
        for(int k = 0; k < ncomp; k++)
        {
            Resample &r = rs[k];
            r.hs = hmax / comp[k].h;
            r.vs = vmax / comp[k].v;
            r.step = r.vs >> 1;
            r.row = 0;
            r.wLow = (width + r.hs - 1) / r.hs;
            r.line0 = r.line1 = comp[k].plane.data();
            r.buffer.resize(size_t(width) + 4 * 8 + 8);
        }
        const bool plainRgb = ncomp == 3 && (rgbIds || (adobeTransform == 0 && !jfif));
        const int cr_r = fx(1.40200) << 8, cr_g = -(fx(0.71414) << 8), cb_g = -(fx(0.34414) << 8), cb_b = fx(1.77200) << 8;
        for(int y = 0; y < height; y++)
        {
            const uint8_t *c[4] = {nullptr, nullptr, nullptr, nullptr};
            for(int k = 0; k < ncomp; k++)
            {
                Resample &r = rs[k];
                const bool bottom = r.step >= (r.vs >> 1);
                c[k] = upsampleRow(r.buffer, bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.wLow, r.hs, r.vs);
                if(++r.step >= r.vs)
                {
                    r.step = 0;
                    r.line0 = r.line1;
                    if(++r.row < comp[k].y)
                        r.line1 += comp[k].w2;
                }
            }
            uint8_t *out = &img.pixels[size_t(y) * width * 4];
            for(int x = 0; x < width; x++, out += 4)
            {
                out[3] = 255;
                if(ncomp == 1)
                {
                    out[0] = out[1] = out[2] = c[0][x];
                    continue;
                }
                if(plainRgb || (ncomp == 4 && adobeTransform == 0))
                {
                    out[0] = c[0][x];
                    out[1] = c[1][x];
                    out[2] = c[2][x];
                }
                else
                {
                    const int yf = (c[0][x] << 20) + (1 << 19), cb = c[1][x] - 128, cr = c[2][x] - 128;
                    int r = yf + cr * cr_r;
                    int g = yf + cr * cr_g + int(unsigned(cb * cb_g) & 0xffff0000u);
                    int b = yf + cb * cb_b;
                    out[0] = clamp8(r >> 20);
                    out[1] = clamp8(g >> 20);
                    out[2] = clamp8(b >> 20);
                }
                if(ncomp == 4)
                {
                    const int k = c[3][x];
                    if(adobeTransform == 0) // CMYK
                        for(int i = 0; i < 3; i++)
                            out[i] = mul255(out[i], k);
                    else if(adobeTransform == 2) // YCCK
                        for(int i = 0; i < 3; i++)
                            out[i] = mul255(255 - out[i], k);
                }
            }
        }
        return img;
    }
};

} // namespace

Image decodeJpeg(const std::vector<uint8_t> &file, const std::string &path)
{
    return Decoder(file, path).decode();
}

} // namespace lfi
