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

// getFeatureDepth (feature_tracker/feature_tracker_node.cpp:54-163): LiDAR depth for the tracked visual features. Features and the
// depth cloud (camera frame) are projected onto the unit sphere; per feature the 3 nearest cloud points (kd-tree in the reference,
// exact brute force here: float squared distance, ties by index) span a plane that the feature ray is intersected with.
// All arithmetic is FLOAT as in the reference (Eigen::Vector3f, PointXYZI); the threshold is pow(sin(0.5 deg) * 5, 2) cast to float.
extern "C" int vilo_feature_depth(const float *cloud_xyzi, int n, const float *feat_xyz, int m, float *depth_out) {
    if (n < 0 || m < 0 || (m && (!feat_xyz || !depth_out))) return VILF_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < m; i++) depth_out[i] = -1.0f;
    std::vector<P4> unit(n);
    for (int i = 0; i < n; i++) {
        P4 p{cloud_xyzi[4 * i], cloud_xyzi[4 * i + 1], cloud_xyzi[4 * i + 2], cloud_xyzi[4 * i + 3]};
        const float range = std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z);
        p.x /= range; p.y /= range; p.z /= range; p.i = range;
        unit[i] = p;
    }
    if (n < 10) return VILF_OK;
    const float bin_res = 180.0 / (float)360;
    const float thr = (float)std::pow(std::sin(bin_res / 180.0 * M_PI) * 5.0, 2);
    for (int f = 0; f < m; f++) {
        float vx = feat_xyz[3 * f], vy = feat_xyz[3 * f + 1], vz = feat_xyz[3 * f + 2];
        const float nrm = std::sqrt(vx * vx + vy * vy + vz * vz);
        vx /= nrm; vy /= nrm; vz /= nrm;
        int idx[3] = {-1, -1, -1}; float d2[3] = {3.0e38f, 3.0e38f, 3.0e38f};
        for (int i = 0; i < n; i++) {
            const float ex = unit[i].x - vx, ey = unit[i].y - vy, ez = unit[i].z - vz;
            const float d = ex * ex + ey * ey + ez * ez;
            if (d < d2[2]) {
                int k = 2;
                while (k > 0 && d < d2[k - 1]) { d2[k] = d2[k - 1]; idx[k] = idx[k - 1]; k--; }
                d2[k] = d; idx[k] = i;
            }
        }
        if (idx[2] < 0 || !(d2[2] < thr)) continue;
        const float r1 = unit[idx[0]].i, r2 = unit[idx[1]].i, r3 = unit[idx[2]].i;
        const float A[3] = {unit[idx[0]].x * r1, unit[idx[0]].y * r1, unit[idx[0]].z * r1};
        const float B[3] = {unit[idx[1]].x * r2, unit[idx[1]].y * r2, unit[idx[1]].z * r2};
        const float Cc[3] = {unit[idx[2]].x * r3, unit[idx[2]].y * r3, unit[idx[2]].z * r3};
        const float a[3] = {A[0] - B[0], A[1] - B[1], A[2] - B[2]}, b[3] = {B[0] - Cc[0], B[1] - Cc[1], B[2] - Cc[2]};
        const float N[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        float s = (N[0] * A[0] + N[1] * A[1] + N[2] * A[2]) / (N[0] * vx + N[1] * vy + N[2] * vz);
        const float mn = std::min(r1, std::min(r2, r3)), mx = std::max(r1, std::max(r2, r3));
        if (mx - mn > 2 || s <= 0.5) continue;
        else if (s - mx > 0) s = mx;
        else if (s - mn < 0) s = mn;
        const float inten = vz * s;                               // the estimator wants the depth of the z = 1 normalised feature
        if (inten > 2.0) depth_out[f] = inten;
    }
    return VILF_OK;
}
