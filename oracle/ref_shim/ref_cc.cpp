// ref_cc.cpp — the reference's own connectedLabels (Core/Segmentation/ConnectedLabels.hpp:50-172) behind a flat C entry point.
// TEST INFRASTRUCTURE ONLY.  stats rows: {label, top, right, bottom, left, size}; returns the number of components.
#include <opencv2/imgproc/imgproc.hpp>
#include <string.h>
#include "Segmentation/ConnectedLabels.hpp"

extern "C" int ref_connected_labels(const unsigned char* in, int cols, int rows, int* comp, int* stats6, int max_stats)
{
    cv::Mat input(rows, cols, CV_8UC1, (void*)in);
    std::vector<ComponentData> st;
    cv::Mat out = connectedLabels(input, &st);
    memcpy(comp, out.data, sizeof(int) * (size_t)cols * rows);
    for (size_t i = 0; i < st.size() && (int)i < max_stats; i++) {
        int* o = stats6 + i * 6;
        o[0] = st[i].label; o[1] = st[i].top; o[2] = st[i].right; o[3] = st[i].bottom; o[4] = st[i].left; o[5] = st[i].size;
    }
    return (int)st.size();
}
