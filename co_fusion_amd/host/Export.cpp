// Export.cpp -- CoFusion::savePly (Core/CoFusion.cpp:646-754) and CoFusion::exportPoses (:756-783): the reference's
// externally checkable outputs, in the author's dataset-tools formats.
//   cloud-<id>.ply : binary little-endian, per confident surfel  x y z (f32)  r g b (u8)  nx ny nz (f32)  radius (f32),
//                    positions mapped by Tp = globalPose * modelPose^-1, normals negated after the transform
//   poses-<id>.txt : one line per logged frame:  timestamp x y z qx qy qz qw   (cam->world for the background
//                    model, object->world for the others)
// Deviation, on purpose: the reference builds the normal transform from an uninitialised matrix
// (`Eigen::Matrix4f Tn = Tn.inverse().transpose();`, CoFusion.cpp:697); here Tn = (Tp^-1)^T as evidently intended.
#include <cstdio>
#include <cstring>
#include <fstream>

#include <zlib.h>

#include <vector>

#include "CoFusion.h"

namespace cofusion {

// Minimal PNG writer: 8-bit greyscale, one zlib stream, filter type 0 on every row.
bool writePngGray8(const std::string& path, const uint8_t* data, int width, int height)
{
    std::vector<uint8_t> raw((size_t)(width + 1) * height);
    for (int y = 0; y < height; y++) {
        raw[(size_t)y * (width + 1)] = 0;
        memcpy(&raw[(size_t)y * (width + 1) + 1], data + (size_t)y * width, (size_t)width);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), Z_BEST_SPEED) != Z_OK) return false;
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    auto be32 = [](uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
    auto chunk = [&](const char* type, const uint8_t* body, uint32_t len) {
        uint8_t head[8]; be32(head, len); memcpy(head + 4, type, 4);
        fwrite(head, 1, 8, f);
        if (len) fwrite(body, 1, len, f);
        uLong c = crc32(0L, reinterpret_cast<const Bytef*>(type), 4);
        if (len) c = crc32(c, body, len);
        uint8_t tail[4]; be32(tail, (uint32_t)c);
        fwrite(tail, 1, 4, f);
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    fwrite(sig, 1, 8, f);
    uint8_t ihdr[13]; be32(ihdr, (uint32_t)width); be32(ihdr + 4, (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = 0; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;  // 8 bit, greyscale, deflate, adaptive filtering, no interlace
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)zlen);
    chunk("IEND", nullptr, 0);
    fclose(f);
    return true;
}

static void mul_point(const Mat4f& T, const float p[3], float o[3])
{
    for (int r = 0; r < 3; r++) o[r] = T.m[r * 4 + 0] * p[0] + T.m[r * 4 + 1] * p[1] + T.m[r * 4 + 2] * p[2] + T.m[r * 4 + 3];
}

int CoFusion::savePly(const std::string& exportDir)
{
    int written = 0;
    for (auto& model : models) {
        if (!model->isOwned()) continue;  // model-parallel mode: the owner rank writes the cloud
        const std::string filename = exportDir + "cloud-" + std::to_string(model->getID()) + ".ply";
        const std::vector<float> map = model->downloadMap();
        const size_t n = map.size() / 12;
        const float thr = model->getConfidenceThreshold();
        size_t valid = 0;
        for (size_t i = 0; i < n; i++) valid += map[i * 12 + 3] > thr ? 1 : 0;  // SurfelMap::countValid
        std::ofstream fs(filename.c_str(), std::ios::binary);
        if (!fs) return -1;
        fs << "ply\nformat binary_little_endian 1.0\nelement vertex " << valid
           << "\nproperty float x\nproperty float y\nproperty float z"
              "\nproperty uchar red\nproperty uchar green\nproperty uchar blue"
              "\nproperty float nx\nproperty float ny\nproperty float nz"
              "\nproperty float radius\nend_header\n";
        const Mat4f Tp = globalModel->getPose() * model->getPose().inverse();
        const Mat4f Ti = Tp.inverse();
        for (size_t i = 0; i < n; i++) {
            const float* s = &map[i * 12];
            if (!(s[3] > thr)) continue;
            float pos[3], nor[3];
            mul_point(Tp, s, pos);
            // (Tp^-1)^T * n, w = 0
            for (int r = 0; r < 3; r++) nor[r] = -(Ti.m[0 * 4 + r] * s[8] + Ti.m[1 * 4 + r] * s[9] + Ti.m[2 * 4 + r] * s[10]);
            const int c = (int)s[4];
            const unsigned char rgb[3] = {(unsigned char)(c >> 16 & 0xFF), (unsigned char)(c >> 8 & 0xFF), (unsigned char)(c & 0xFF)};
            fs.write(reinterpret_cast<const char*>(pos), 12);
            fs.write(reinterpret_cast<const char*>(rgb), 3);
            fs.write(reinterpret_cast<const char*>(nor), 12);
            fs.write(reinterpret_cast<const char*>(&s[11]), 4);
        }
        written++;
    }
    return written;
}

int CoFusion::exportPoses(const std::string& exportDir)
{
    int written = 0;
    auto exportModelPoses = [&](ModelList& list) {
        for (auto& m : list) {
            if (!m->isLoggingPoses()) continue;
            const std::string filename = exportDir + "poses-" + std::to_string(m->getID()) + ".txt";
            FILE* f = fopen(filename.c_str(), "w");
            if (!f) { written = -1; return; }
            for (const auto& p : m->poseLog) {
                fprintf(f, "%lld", (long long)p.ts);
                for (int i = 0; i < 7; i++) fprintf(f, " %g", (double)p.p[i]);  // == `fs << p.p(i)` (CoFusion.cpp:773): six significant digits
                fprintf(f, "\n");
            }
            fclose(f);
            written++;
        }
    };
    exportModelPoses(models);
    if (written >= 0) exportModelPoses(inactiveModels);
    return written;
}

}  // namespace cofusion
