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

#include "CoFusion.h"

namespace cofusion {

static void mul_point(const Mat4f& T, const float p[3], float o[3])
{
    for (int r = 0; r < 3; r++) o[r] = T.m[r * 4 + 0] * p[0] + T.m[r * 4 + 1] * p[1] + T.m[r * 4 + 2] * p[2] + T.m[r * 4 + 3];
}

int CoFusion::savePly(const std::string& exportDir)
{
    int written = 0;
    for (auto& model : models) {
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
                for (int i = 0; i < 7; i++) fprintf(f, " %.9g", (double)p.p[i]);
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
