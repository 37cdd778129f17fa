// contours.cpp — bounding boxes of the borders of a binary image, on the host.
//
// Replaces  cv2.findContours(img, cv2.RETR_LIST, cv2.CHAIN_APPROX_SIMPLE) + cv2.boundingRect
// at /root/reference/mass/nn/applications/semantic_projection_layer.py:323-328 (one call per
// class on a [map_height, map_width] uint8 image).  OpenCV is a third-party dependency of the
// reference (unpinned, absent from this image), whose findContours implements
//   S. Suzuki, K. Abe, "Topological structural analysis of digitized binary images by border
//   following", CVGIP 30(1), 1985 (Algorithm 1)
// with 8-connected foreground.  This file restates that published algorithm: raster scan, an
// outer border starts at a 1-pixel whose left neighbour is 0, a hole border at a pixel >= 1
// whose right neighbour is 0; followed borders are labelled NBD / -NBD exactly as in the paper
// so no border is followed twice.  RETR_LIST keeps no hierarchy, so step (2) of the paper
// (parent lookup) is skipped.  Only the bounding box of each border is kept (CHAIN_APPROX_SIMPLE
// does not change a bounding box).  OpenCV returns the list in reverse order of discovery
// (last border found first); `reverse_order` selects that.  PARITY UNPINNED: no reference test
// or fixture pins these outputs and cv2 cannot be imported here.
#include <algorithm>
#include <vector>
#include "common.h"

namespace {

// 8-neighbourhood in clockwise order starting from "west" (image coordinates: i down, j right)
const int DI[8] = {0, -1, -1, -1, 0, 1, 1, 1};
const int DJ[8] = {-1, -1, 0, 1, 1, 1, 0, -1};

inline int dir_of(int di, int dj)
{
    for (int d = 0; d < 8; ++d)
        if (DI[d] == di && DJ[d] == dj) return d;
    return -1;
}

}  // namespace

extern "C" int mf_contour_boxes(const uint8_t *img, int32_t height, int32_t width, int32_t reverse_order,
                                int32_t *boxes /* [max_boxes][4] = x, y, w, h */, int32_t max_boxes)
{
    if (!img || height < 0 || width < 0 || (max_boxes > 0 && !boxes))
        return mf::fail(MF_ERR_INVALID, "bad argument");
    const int H = height + 2, W = width + 2;                 // frame of zeros around the picture
    std::vector<int> f((size_t)H * W, 0);
    for (int i = 0; i < height; ++i)
        for (int j = 0; j < width; ++j) f[(size_t)(i + 1) * W + j + 1] = img[(size_t)i * width + j] ? 1 : 0;
    auto at = [&](int i, int j) -> int & { return f[(size_t)i * W + j]; };

    std::vector<int> found;                                   // x, y, w, h per border, discovery order
    int nbd = 1;
    for (int i = 1; i <= height; ++i) {
        for (int j = 1; j <= width; ++j) {
            const int fij = at(i, j);
            if (fij == 0) continue;
            int i2, j2;
            if (fij == 1 && at(i, j - 1) == 0) { i2 = i; j2 = j - 1; }            // outer border
            else if (fij >= 1 && at(i, j + 1) == 0) { i2 = i; j2 = j + 1; }       // hole border
            else continue;
            ++nbd;
            int min_i = i, max_i = i, min_j = j, max_j = j;
            // (3.1) clockwise from (i2, j2) around (i, j): first non-zero pixel (i1, j1)
            int d0 = dir_of(i2 - i, j2 - j), d1 = -1;
            for (int k = 0; k < 8; ++k) {
                const int d = (d0 + k) & 7;
                if (at(i + DI[d], j + DJ[d]) != 0) { d1 = d; break; }
            }
            if (d1 < 0) {
                at(i, j) = -nbd;                                                   // isolated pixel
            } else {
                const int i1 = i + DI[d1], j1 = j + DJ[d1];
                int pi2 = i1, pj2 = j1, i3 = i, j3 = j;                            // (3.2)
                for (;;) {
                    // (3.3) counter-clockwise from the element after (i2, j2) around (i3, j3)
                    const int ds = dir_of(pi2 - i3, pj2 - j3);
                    bool east_zero_examined = false;
                    int i4 = i3, j4 = j3;
                    for (int k = 1; k <= 8; ++k) {
                        const int d = (ds - k) & 7;                               // counter-clockwise
                        const int ni = i3 + DI[d], nj = j3 + DJ[d];
                        if (at(ni, nj) != 0) { i4 = ni; j4 = nj; break; }
                        if (d == 4) east_zero_examined = true;                    // (i3, j3 + 1) seen as 0
                    }
                    // (3.4)
                    if (east_zero_examined) at(i3, j3) = -nbd;
                    else if (at(i3, j3) == 1) at(i3, j3) = nbd;
                    min_i = std::min(min_i, i3); max_i = std::max(max_i, i3);
                    min_j = std::min(min_j, j3); max_j = std::max(max_j, j3);
                    // (3.5)
                    if (i4 == i && j4 == j && i3 == i1 && j3 == j1) break;
                    pi2 = i3; pj2 = j3; i3 = i4; j3 = j4;
                }
            }
            found.push_back(min_j - 1); found.push_back(min_i - 1);
            found.push_back(max_j - min_j + 1); found.push_back(max_i - min_i + 1);
        }
    }
    const int n = (int)(found.size() / 4);
    for (int k = 0; k < n && k < max_boxes; ++k) {
        const int src = reverse_order ? n - 1 - k : k;
        for (int q = 0; q < 4; ++q) boxes[k * 4 + q] = found[(size_t)src * 4 + q];
    }
    return n;
}
