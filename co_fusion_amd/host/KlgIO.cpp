// KlgIO.cpp -- see KlgIO.h.  zlib + the built-in baseline JPEG decoder (Jpeg.cpp); no OpenCV, no libjpeg.
#include "KlgIO.h"

#include <zlib.h>

#include <cmath>
#include <cstring>

namespace cofusion {

std::string decodeJpegRGB(const uint8_t* data, size_t size, int width, int height, uint8_t* rgb);  // Jpeg.cpp

KlgLogReader::KlgLogReader(const std::string& file, int w, int h, bool flipColors) : width(w), height(h), flip(flipColors)
{
    fp = fopen(file.c_str(), "rb");
    if (!fp) { err = "could not open log-file: " + file; return; }
    int32_t n = 0;
    if (fread(&n, sizeof(int32_t), 1, fp) != 1) { err = "could not read the frame count of " + file; fclose(fp); fp = nullptr; return; }
    numFrames = n;
    const size_t N = (size_t)width * height;
    depth.assign(N, 0.f); rgb.assign(N * 3, 0); depthMm.assign(N, 0);
    depthRaw.resize(N * 2 + 1024); rgbRaw.resize(N * 3 + 1024);
}

KlgLogReader::~KlgLogReader() { if (fp) fclose(fp); }

void KlgLogReader::rewind()
{
    if (!fp) return;
    fseek(fp, sizeof(int32_t), SEEK_SET);
    currentFrame = 0;
}

bool KlgLogReader::getNext()
{  // KlgLogReader::getCore, KlgLogReader.cpp:50-87
    if (!fp || !hasMore()) { err = "no more frames"; return false; }
    const size_t N = (size_t)width * height;
    int32_t depthSize = 0, rgbSize = 0;
    if (fread(&timestamp, sizeof(int64_t), 1, fp) != 1 || fread(&depthSize, sizeof(int32_t), 1, fp) != 1 ||
        fread(&rgbSize, sizeof(int32_t), 1, fp) != 1) { err = "truncated frame header"; return false; }
    if (depthSize < 0 || rgbSize < 0 || (size_t)depthSize > depthRaw.size() || (size_t)rgbSize > rgbRaw.size()) { err = "implausible frame sizes"; return false; }
    if (depthSize && fread(depthRaw.data(), (size_t)depthSize, 1, fp) != 1) { err = "truncated depth block"; return false; }
    if (rgbSize > 0 && fread(rgbRaw.data(), (size_t)rgbSize, 1, fp) != 1) { err = "truncated rgb block"; return false; }
    if ((size_t)depthSize != N * 2) {
        uLongf len = (uLongf)(N * 2);
        if (uncompress(reinterpret_cast<Bytef*>(depthMm.data()), &len, depthRaw.data(), (uLong)depthSize) != Z_OK || len != N * 2) {
            err = "zlib: depth block does not decompress to width*height u16"; return false;
        }
    } else {
        memcpy(depthMm.data(), depthRaw.data(), N * 2);
    }
    for (size_t i = 0; i < N; i++) depth[i] = (float)depthMm[i] * 0.001f;  // convertTo(CV_32FC1, 0.001): saturate_cast<float>(v * 0.001)
    if (rgbSize > 0) {
        if ((size_t)rgbSize != N * 3) {
            const std::string e = decodeJpegRGB(rgbRaw.data(), (size_t)rgbSize, width, height, rgb.data());
            if (!e.empty()) { err = "JPEG colour frame: " + e; return false; }
            // JPEGLoader::readData (JPEGLoader.h:70-78) stores the decoded triple reversed
            for (size_t i = 0; i < N; i++) { const uint8_t t = rgb[i * 3]; rgb[i * 3] = rgb[i * 3 + 2]; rgb[i * 3 + 2] = t; }
        } else {
            memcpy(rgb.data(), rgbRaw.data(), N * 3);
        }
    } else {
        memset(rgb.data(), 0, N * 3);
    }
    if (flip)
        for (size_t i = 0; i < N; i++) { const uint8_t t = rgb[i * 3]; rgb[i * 3] = rgb[i * 3 + 2]; rgb[i * 3 + 2] = t; }
    currentFrame++;
    return true;
}

KlgLogWriter::KlgLogWriter(const std::string& file, int w, int h, bool compressDepth) : width(w), height(h), compress(compressDepth)
{
    fp = fopen(file.c_str(), "wb");
    if (!fp) return;
    const int32_t zero = 0;
    fwrite(&zero, sizeof(int32_t), 1, fp);
    mm.resize((size_t)w * h);
    zbuf.resize(compressBound((uLong)((size_t)w * h * 2)));
}

KlgLogWriter::~KlgLogWriter() { close(); }

bool KlgLogWriter::write(int64_t timestamp, const float* d, const uint8_t* rgb)
{
    const size_t N = (size_t)width * height;
    for (size_t i = 0; i < N; i++) {
        const float m = d[i] * 1000.0f;
        mm[i] = (m > 0.f && m < 65535.f) ? (uint16_t)lroundf(m) : (uint16_t)0;
    }
    return writeRawMm(timestamp, mm.data(), rgb);
}

bool KlgLogWriter::writeRawMm(int64_t timestamp, const uint16_t* depthMm, const uint8_t* rgb)
{
    if (!fp) return false;
    const size_t N = (size_t)width * height;
    const void* dptr = depthMm; int32_t dsize = (int32_t)(N * 2);
    if (compress) {
        uLongf len = (uLongf)zbuf.size();
        if (compress2(zbuf.data(), &len, reinterpret_cast<const Bytef*>(depthMm), (uLong)(N * 2), Z_DEFAULT_COMPRESSION) != Z_OK) return false;
        if (len != N * 2) { dptr = zbuf.data(); dsize = (int32_t)len; }  // a block of exactly N*2 bytes would read back as "raw"
    }
    const int32_t rsize = rgb ? (int32_t)(N * 3) : 0;
    bool ok = fwrite(&timestamp, sizeof(int64_t), 1, fp) == 1 && fwrite(&dsize, sizeof(int32_t), 1, fp) == 1 &&
              fwrite(&rsize, sizeof(int32_t), 1, fp) == 1 && fwrite(dptr, (size_t)dsize, 1, fp) == 1;
    if (ok && rsize) ok = fwrite(rgb, (size_t)rsize, 1, fp) == 1;
    if (ok) numFrames++;
    return ok;
}

void KlgLogWriter::close()
{
    if (!fp) return;
    fseek(fp, 0, SEEK_SET);
    const int32_t n = numFrames;
    fwrite(&n, sizeof(int32_t), 1, fp);
    fclose(fp);
    fp = nullptr;
}

}  // namespace cofusion
