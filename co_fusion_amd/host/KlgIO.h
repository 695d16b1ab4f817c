// KlgIO.h -- reader / writer of the reference's .klg RGB-D log format (GUI/Tools/KlgLogReader.cpp:22-87):
//   int32 numFrames; per frame { int64 timestamp; int32 depthSize; int32 rgbSize; depth bytes; rgb bytes }
// depth: u16 millimetres, raw (depthSize == W*H*2) or zlib-compressed; rgb: raw 8-bit x3 (rgbSize == W*H*3),
// JPEG otherwise.  The reader converts depth to metres exactly as `convertTo(CV_32FC1, 0.001)` does
// (f32(u16) * f32(0.001), KlgLogReader.cpp:63-69).  JPEG colour frames go through the built-in baseline decoder (Jpeg.cpp;
// libjpeg's headers are not part of this image) and are stored channel-reversed like JPEGLoader::readData does;
// progressive JPEG is rejected.
#pragma once

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace cofusion {

class KlgLogReader {
  public:
    KlgLogReader(const std::string& file, int width, int height, bool flipColors = false);
    ~KlgLogReader();
    KlgLogReader(const KlgLogReader&) = delete;
    bool ok() const { return fp != nullptr; }
    const std::string& error() const { return err; }
    int getNumFrames() const { return numFrames; }
    int currentFrameIndex() const { return currentFrame; }
    // Every frame of the log is played.  The reference's KlgLogReader::hasMore() is `currentFrame + 1 < numFrames` (GUI/Tools/
    // KlgLogReader.cpp), i.e. it never plays the last frame: set referenceCompatible for frame-count parity with a reference run.
    bool referenceCompatible = false;
    bool hasMore() const { return referenceCompatible ? currentFrame + 1 < numFrames : currentFrame < numFrames; }
    // KlgLogReader::getNext/getCore: decodes the next frame into the members below; false on error
    bool getNext();
    void rewind();
    int64_t timestamp = 0;
    std::vector<float> depth;   // metres [H*W]
    std::vector<uint8_t> rgb;   // [H*W*3]

  private:
    FILE* fp = nullptr;
    std::string err;
    int width, height, numFrames = 0, currentFrame = 0;
    bool flip;
    std::vector<uint8_t> depthRaw, rgbRaw;
    std::vector<uint16_t> depthMm;
};

// Writes the same container (what the reference's recording tools produce); depth is quantised to millimetres
// like a sensor log: u16(lround(metres * 1000)), 0 for invalid.
class KlgLogWriter {
  public:
    KlgLogWriter(const std::string& file, int width, int height, bool compressDepth = true);
    ~KlgLogWriter();
    KlgLogWriter(const KlgLogWriter&) = delete;
    bool ok() const { return fp != nullptr; }
    bool write(int64_t timestamp, const float* depthMetres, const uint8_t* rgb);
    bool writeRawMm(int64_t timestamp, const uint16_t* depthMm, const uint8_t* rgb);
    void close();  // patches numFrames into the header

  private:
    FILE* fp = nullptr;
    int width, height, numFrames = 0;
    bool compress;
    std::vector<uint16_t> mm;
    std::vector<uint8_t> zbuf;
};

}  // namespace cofusion
