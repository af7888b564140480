// ORACLE — TEST INFRASTRUCTURE ONLY. "parity unpinned" (the reference has no tests for this path).
// lidar_features.cpp — CPU restatement of the LOAM-style feature extraction in front of scan-to-map:
//   featureExtraction::{getLaserCloud, featureEdge_Surf, featureExtractionFromSector, extractFeature}
//   (feature_tracker/include/featureExtraction.hpp:54-232; DistanceXY common.h:59-62).
// Kept quirks: ring assignment from the vertical angle with the reference's float / double mix; curvature summed in FLOAT, left to
// right; every sector drops its last element (iterator range excludes sector_end, :208); the 21st pick of a sector is marked as
// picked but emitted neither as edge nor as surf (:128-137); neighbour suppression is local to a sector.
// One deliberate definition: std::sort's order of EXACTLY equal curvatures is unspecified in the reference; here ties are broken
// by the point index (ascending), and the HIP path does the same.
#include "oracle_api.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace {
struct P4 { float x, y, z, i; };

int ring_of(const P4 &p, int n_scans, double min_r, double max_r) {
    const float dxy = std::sqrt(p.x * p.x + p.y * p.y);          // DistanceXY: float
    const double distance = dxy;
    if (distance < min_r || distance > max_r) return -1;
    const double angle = std::atan(p.z / distance) * 180 / M_PI;
    if (!(angle == angle)) return -1;                             // NaN points fall out of every branch of the reference
    int id;
    if (n_scans == 16) { id = int((angle + 15) / 2 + 0.5); if (id > n_scans - 1 || id < 0) return -1; }
    else if (n_scans == 32) { id = int((angle + 92.0 / 3.0) * 3.0 / 4.0); if (id > n_scans - 1 || id < 0) return -1; }
    else if (n_scans == 64) {
        if (angle >= -8.83) id = int((2 - angle) * 3.0 + 0.5);
        else id = n_scans / 2 + int((-8.83 - angle) * 2.0 + 0.5);
        if (angle > 2 || angle < -24.33 || id > 63 || id < 0) return -1;
    } else return -1;
    return id;
}
}  // namespace

extern "C" int vilo_extract_features(const float *xyzi, int n, int n_scans, double min_range, double max_range, double edge_threshold,
                                     float *edge_out, int cap_edge, int *n_edge, float *surf_out, int cap_surf, int *n_surf) {
    if (n < 0 || (n_scans != 16 && n_scans != 32 && n_scans != 64) || !n_edge || !n_surf) return VILF_ERR_INVALID_ARGUMENT;
    std::vector<std::vector<P4>> rings(n_scans);
    for (int i = 0; i < n; i++) {
        const P4 p{xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2], xyzi[4 * i + 3]};
        const int id = ring_of(p, n_scans, min_range, max_range);
        if (id >= 0) rings[id].push_back(p);
    }
    std::vector<P4> edge, surf;
    for (int r = 0; r < n_scans; r++) {
        const std::vector<P4> &c = rings[r];
        if (c.size() < 131) continue;
        const size_t smooth_size = c.size() - 5;
        std::vector<std::pair<double, int>> curv;                // (value, ind)
        for (size_t j = 5; j < smooth_size; j++) {
            const float dx = c[j - 5].x + c[j - 4].x + c[j - 3].x + c[j - 2].x + c[j - 1].x - 10 * c[j].x + c[j + 1].x + c[j + 2].x + c[j + 3].x + c[j + 4].x + c[j + 5].x;
            const float dy = c[j - 5].y + c[j - 4].y + c[j - 3].y + c[j - 2].y + c[j - 1].y - 10 * c[j].y + c[j + 1].y + c[j + 2].y + c[j + 3].y + c[j + 4].y + c[j + 5].y;
            const float dz = c[j - 5].z + c[j - 4].z + c[j - 3].z + c[j - 2].z + c[j - 1].z - 10 * c[j].z + c[j + 1].z + c[j + 2].z + c[j + 3].z + c[j + 4].z + c[j + 5].z;
            const double X = dx, Y = dy, Z = dz;
            curv.emplace_back(X * X + Y * Y + Z * Z, (int)j);
        }
        const int cloud_size = (int)smooth_size - 5;
        for (int s = 0; s < 6; s++) {
            const int len = cloud_size / 6, start = len * s;
            const int end = (s == 5) ? cloud_size - 1 : len * (s + 1) - 1;
            if (end <= start) continue;
            std::vector<std::pair<double, int>> sub(curv.begin() + start, curv.begin() + end);
            std::sort(sub.begin(), sub.end());                    // ascending value, ties by index
            std::vector<char> picked(c.size(), 0);
            int largest = 0;
            auto d2 = [&](int a, int b) { const double ex = c[a].x - c[b].x, ey = c[a].y - c[b].y, ez = c[a].z - c[b].z; return ex * ex + ey * ey + ez * ez; };
            for (int i = (int)sub.size() - 1; i >= 0; i--) {
                const int ind = sub[i].second;
                if (picked[ind]) continue;
                if (sub[i].first <= edge_threshold) break;
                largest++;
                picked[ind] = 1;
                if (largest <= 20) edge.push_back(c[ind]); else break;
                for (int k = 1; k <= 5; k++) { if (d2(ind + k, ind + k - 1) > 0.05) break; picked[ind + k] = 1; }
                for (int l = -1; l >= -5; l--) { if (d2(ind + l, ind + l + 1) > 0.05) break; picked[ind + l] = 1; }
            }
            for (size_t i = 0; i < sub.size(); i++) if (!picked[sub[i].second]) surf.push_back(c[sub[i].second]);
        }
    }
    *n_edge = (int)edge.size(); *n_surf = (int)surf.size();
    for (size_t i = 0; i < edge.size() && (int)i < cap_edge; i++) { edge_out[4 * i] = edge[i].x; edge_out[4 * i + 1] = edge[i].y; edge_out[4 * i + 2] = edge[i].z; edge_out[4 * i + 3] = edge[i].i; }
    for (size_t i = 0; i < surf.size() && (int)i < cap_surf; i++) { surf_out[4 * i] = surf[i].x; surf_out[4 * i + 1] = surf[i].y; surf_out[4 * i + 2] = surf[i].z; surf_out[4 * i + 3] = surf[i].i; }
    return VILF_OK;
}
