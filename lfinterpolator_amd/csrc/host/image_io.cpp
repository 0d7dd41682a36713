// image_io.cpp — RGBA8 image files for LfLoader / storeResults, on zlib (the reference vendors stb_image v2.27 and
// stb_image_write v1.16 for this: reference src/lfLoader.cpp:33-42, src/interpolator.cu:313).  Reads PNG (also Adam7-interlaced;
// grey, grey+alpha, RGB, RGBA, palette; 1–16 bit), JPEG (jpeg.cpp) and binary PPM/PGM; writes 8-bit RGBA/RGB PNG and PPM.
#include "image_io.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace lfi {

namespace {

uint32_t be32(const uint8_t *p)
{
    return uint32_t(p[0]) << 24 | uint32_t(p[1]) << 16 | uint32_t(p[2]) << 8 | uint32_t(p[3]);
}

void put32(std::vector<uint8_t> &out, uint32_t v)
{
    out.push_back(uint8_t(v >> 24));
    out.push_back(uint8_t(v >> 16));
    out.push_back(uint8_t(v >> 8));
    out.push_back(uint8_t(v));
}

std::vector<uint8_t> readFile(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if(!f)
        throw std::runtime_error("Cannot load image " + path);
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return data;
}

int paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if(pa <= pb && pa <= pc)
        return a;
    return pb <= pc ? b : c;
}

Image decodePng(const std::vector<uint8_t> &file, const std::string &path)
{
    static const uint8_t SIGNATURE[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    const auto bad = [&](const char *why) { return std::runtime_error("Cannot load image " + path + " (" + why + ")"); };
    if(file.size() < 8 + 25 || std::memcmp(file.data(), SIGNATURE, 8) != 0)
        throw bad("not a PNG");
    uint32_t width = 0, height = 0;
    int depth = 0, colorType = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, transparency;
    size_t pos = 8;
    bool end = false;
    while(!end && pos + 12 <= file.size())
    {
        const uint32_t length = be32(&file[pos]);
        const uint8_t *type = &file[pos + 4];
        const uint8_t *body = &file[pos + 8];
        if(pos + 12 + size_t(length) > file.size())
            throw bad("truncated chunk");
        if(!std::memcmp(type, "IHDR", 4))
        {
            if(length < 13)
                throw bad("bad IHDR");
            width = be32(body);
            height = be32(body + 4);
            depth = body[8];
            colorType = body[9];
            interlace = body[12];
        }
        else if(!std::memcmp(type, "PLTE", 4))
            palette.assign(body, body + length);
        else if(!std::memcmp(type, "tRNS", 4))
            transparency.assign(body, body + length);
        else if(!std::memcmp(type, "IDAT", 4))
            idat.insert(idat.end(), body, body + length);
        else if(!std::memcmp(type, "IEND", 4))
            end = true;
        pos += 12 + size_t(length);
    }
    if(width == 0 || height == 0 || width > (1u << 15) || height > (1u << 15))
        throw bad("bad dimensions");
    if(interlace > 1)
        throw bad("bad interlace method");
    int channels;
    switch(colorType)
    {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: throw bad("bad colour type");
    }
    if(!(depth == 8 || depth == 16 || ((colorType == 0 || colorType == 3) && (depth == 1 || depth == 2 || depth == 4))))
        throw bad("bad bit depth");
    const size_t bitsPerPixel = size_t(channels) * depth;
    const size_t bpp = std::max<size_t>(1, bitsPerPixel / 8);
    // the image is one pass, or the seven Adam7 passes one after the other (each a small image of its own)
    struct Pass
    {
        uint32_t x0, y0, dx, dy, w, h;
        size_t stride;
    };
    std::vector<Pass> passes;
    if(interlace == 0)
        passes.push_back({0, 0, 1, 1, width, height, 0});
    else
    {
        static const uint32_t X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1};
        static const uint32_t DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
        for(int i = 0; i < 7; i++)
        {
            const uint32_t w = (width + DX[i] - 1 - X0[i]) / DX[i], h = (height + DY[i] - 1 - Y0[i]) / DY[i];
            if(width > X0[i] && height > Y0[i] && w && h)
                passes.push_back({X0[i], Y0[i], DX[i], DY[i], w, h, 0});
        }
    }
    size_t rawBytes = 0;
    for(Pass &ps : passes)
    {
        ps.stride = (size_t(ps.w) * bitsPerPixel + 7) / 8;
        rawBytes += (ps.stride + 1) * ps.h;
    }
    std::vector<uint8_t> raw(rawBytes);
    uLongf rawSize = raw.size();
    if(uncompress(raw.data(), &rawSize, idat.data(), idat.size()) != Z_OK || rawSize != raw.size())
        throw bad("corrupt image data");

    Image img;
    img.width = int(width);
    img.height = int(height);
    img.pixels.resize(size_t(width) * height * 4);
    const auto sample = [&](const uint8_t *cur, size_t index) -> uint32_t { // sample `index` of the row, native depth
        if(depth == 8)
            return cur[index];
        if(depth == 16)
            return cur[2 * index]; // high byte
        const size_t bit = index * depth;
        return (cur[bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1u);
    };
    uint8_t *passData = raw.data();
    for(const Pass &ps : passes)
    {
        const size_t stride = ps.stride;
        // undo the per-row filters in place
        std::vector<uint8_t> prior(stride, 0);
        for(uint32_t y = 0; y < ps.h; y++)
        {
            uint8_t *row = passData + y * (stride + 1);
            const uint8_t filter = row[0];
            uint8_t *cur = row + 1;
            for(size_t i = 0; i < stride; i++)
            {
                const int a = i >= bpp ? cur[i - bpp] : 0, b = prior[i], c = i >= bpp ? prior[i - bpp] : 0;
                int v = cur[i];
                switch(filter)
                {
                    case 0: break;
                    case 1: v += a; break;
                    case 2: v += b; break;
                    case 3: v += (a + b) / 2; break;
                    case 4: v += paeth(a, b, c); break;
                    default: throw bad("bad filter");
                }
                cur[i] = uint8_t(v);
            }
            std::memcpy(prior.data(), cur, stride);
        }
        for(uint32_t y = 0; y < ps.h; y++)
        {
            const uint8_t *cur = passData + y * (stride + 1) + 1;
            for(uint32_t x = 0; x < ps.w; x++)
            {
                uint8_t *out = &img.pixels[(size_t(ps.y0 + y * ps.dy) * width + (ps.x0 + x * ps.dx)) * 4];
                switch(colorType)
                {
                    case 0:
                    {
                        uint32_t v = sample(cur, x);
                        const bool keyed = transparency.size() >= 2 && (depth == 16 ? cur[2 * x] == transparency[0] && cur[2 * x + 1] == transparency[1] : v == transparency[1]);
                        if(depth < 8)
                            v = v * 255 / ((1u << depth) - 1u);
                        out[0] = out[1] = out[2] = uint8_t(v);
                        out[3] = keyed ? 0 : 255;
                        break;
                    }
                    case 2:
                        out[0] = uint8_t(sample(cur, 3 * x));
                        out[1] = uint8_t(sample(cur, 3 * x + 1));
                        out[2] = uint8_t(sample(cur, 3 * x + 2));
                        out[3] = 255;
                        if(depth == 8 && transparency.size() >= 6 && out[0] == transparency[1] && out[1] == transparency[3] && out[2] == transparency[5])
                            out[3] = 0;
                        break;
                    case 3:
                    {
                        const uint32_t idx = sample(cur, x);
                        if(size_t(idx) * 3 + 2 >= palette.size())
                            throw bad("palette index out of range");
                        out[0] = palette[idx * 3];
                        out[1] = palette[idx * 3 + 1];
                        out[2] = palette[idx * 3 + 2];
                        out[3] = idx < transparency.size() ? transparency[idx] : 255;
                        break;
                    }
                    case 4:
                        out[0] = out[1] = out[2] = uint8_t(sample(cur, 2 * x));
                        out[3] = uint8_t(sample(cur, 2 * x + 1));
                        break;
                    default:
                        out[0] = uint8_t(sample(cur, 4 * x));
                        out[1] = uint8_t(sample(cur, 4 * x + 1));
                        out[2] = uint8_t(sample(cur, 4 * x + 2));
                        out[3] = uint8_t(sample(cur, 4 * x + 3));
                        break;
                }
            }
        }
        passData += (stride + 1) * ps.h;
    }
    return img;
}

Image decodePnm(const std::vector<uint8_t> &file, const std::string &path)
{
    const auto bad = [&](const char *why) { return std::runtime_error("Cannot load image " + path + " (" + why + ")"); };
    size_t pos = 2;
    const auto number = [&]() -> int {
        for(;;)
        {
            while(pos < file.size() && std::isspace(file[pos]))
                pos++;
            if(pos < file.size() && file[pos] == '#')
                while(pos < file.size() && file[pos] != '\n')
                    pos++;
            else
                break;
        }
        int v = 0;
        bool any = false;
        while(pos < file.size() && std::isdigit(file[pos]))
        {
            v = v * 10 + (file[pos++] - '0');
            any = true;
        }
        if(!any)
            throw bad("bad PNM header");
        return v;
    };
    const int channels = file[1] == '6' ? 3 : 1;
    const int width = number(), height = number(), maxval = number();
    pos++; // single whitespace after maxval
    if(width <= 0 || height <= 0 || maxval != 255 || pos + size_t(width) * height * channels > file.size())
        throw bad("unsupported PNM");
    Image img;
    img.width = width;
    img.height = height;
    img.pixels.resize(size_t(width) * height * 4);
    const uint8_t *in = &file[pos];
    for(size_t i = 0; i < size_t(width) * height; i++)
    {
        img.pixels[4 * i + 0] = in[channels * i];
        img.pixels[4 * i + 1] = in[channels * i + (channels == 3 ? 1 : 0)];
        img.pixels[4 * i + 2] = in[channels * i + (channels == 3 ? 2 : 0)];
        img.pixels[4 * i + 3] = 255;
    }
    return img;
}

void chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &body)
{
    put32(out, uint32_t(body.size()));
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    put32(out, uint32_t(crc32(0, &out[start], uInt(out.size() - start))));
}

} // namespace

Image decodeJpeg(const std::vector<uint8_t> &file, const std::string &path); // jpeg.cpp

Image loadImage(const std::string &path)
{
    const std::vector<uint8_t> file = readFile(path);
    if(file.size() >= 8 && file[0] == 0x89 && file[1] == 'P')
        return decodePng(file, path);
    if(file.size() >= 2 && file[0] == 'P' && (file[1] == '6' || file[1] == '5'))
        return decodePnm(file, path);
    if(file.size() >= 2 && file[0] == 0xff && file[1] == 0xd8)
        return decodeJpeg(file, path);
    throw std::runtime_error("Cannot load image " + path + " (only PNG, JPEG and binary PPM/PGM are supported)");
}

void writePng(const std::string &path, int width, int height, int channels, const uint8_t *data, size_t strideBytes)
{
    if(channels != 3 && channels != 4)
        throw std::runtime_error("writePng: 3 or 4 channels expected");
    const size_t rowBytes = size_t(width) * channels;
    std::vector<uint8_t> raw;
    raw.reserve((rowBytes + 1) * height);
    for(int y = 0; y < height; y++)
    {
        raw.push_back(0); // filter: none
        raw.insert(raw.end(), data + size_t(y) * strideBytes, data + size_t(y) * strideBytes + rowBytes);
    }
    uLongf bound = compressBound(uLong(raw.size()));
    std::vector<uint8_t> packed(bound);
    if(compress2(packed.data(), &bound, raw.data(), uLong(raw.size()), 1) != Z_OK) // level 1: speed over size
        throw std::runtime_error("Cannot compress image " + path);
    packed.resize(bound);

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> header;
    put32(header, uint32_t(width));
    put32(header, uint32_t(height));
    header.push_back(8);
    header.push_back(channels == 4 ? 6 : 2);
    header.push_back(0);
    header.push_back(0);
    header.push_back(0);
    chunk(out, "IHDR", header);
    chunk(out, "IDAT", packed);
    chunk(out, "IEND", {});
    std::ofstream f(path, std::ios::binary);
    if(!f.write(reinterpret_cast<const char *>(out.data()), std::streamsize(out.size())))
        throw std::runtime_error("Cannot write image " + path);
}

void writePpm(const std::string &path, int width, int height, const uint8_t *rgba, size_t strideBytes)
{
    std::ofstream f(path, std::ios::binary);
    f << "P6\n" << width << ' ' << height << "\n255\n";
    std::vector<uint8_t> row(size_t(width) * 3);
    for(int y = 0; y < height; y++)
    {
        const uint8_t *in = rgba + size_t(y) * strideBytes;
        for(int x = 0; x < width; x++)
        {
            row[3 * x] = in[4 * x];
            row[3 * x + 1] = in[4 * x + 1];
            row[3 * x + 2] = in[4 * x + 2];
        }
        f.write(reinterpret_cast<const char *>(row.data()), std::streamsize(row.size()));
    }
    if(!f)
        throw std::runtime_error("Cannot write image " + path);
}

} // namespace lfi
