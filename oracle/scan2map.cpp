// ORACLE — TEST INFRASTRUCTURE ONLY. "parity unpinned" vs PCL 1.7.2 / Ceres 2.0.0 (neither is on disk; the reference has
// no tests for this path). scan2map.cpp — CPU restatement of the F-LOAM scan-to-local-map step:
//   EstimationMapping::{localMapInited, optimation_processing, EdgeCostFactor, SurfCostFactor, createSubMap, pointAssociaToMap}
//   (feature_tracker/include/EstimationMapping.hpp:105-363) with
//   pcl::VoxelGrid (centroid per leaf, output in ascending leaf index), pcl::CropBox (inclusive box), and
//   pcl::KdTreeFLANN::nearestKSearch(k = 5) replaced by an exact 5-NN (squared float distances, ascending): brute force is the
//   definition (knn5); the step uses a 1 m cell grid (GridIndex, built once per map per step like the reference's kd-tree) that
//   returns the same neighbours whenever the 5th squared distance is < 1 — the only case the reference uses (:129,189).
#include "oracle_api.h"
#include "solver.hpp"
#include <cstdint>
#include <numeric>

using namespace ora;

namespace {

struct P4 { float x, y, z, i; };
typedef std::vector<P4> Cloud;

// pcl::VoxelGrid<PointXYZI>::applyFilter (PCL 1.7.2 voxel_grid.hpp): leaf indices from floor(p * inverse_leaf) relative to the
// cloud's min corner, points sorted by leaf index, one centroid (all fields, float accumulators) per occupied leaf.
void voxel_grid(const Cloud &in, float leaf, Cloud &out) {
    out.clear();
    if (in.empty()) return;
    float mn[3] = {in[0].x, in[0].y, in[0].z}, mx[3] = {in[0].x, in[0].y, in[0].z};
    for (const P4 &p : in) { mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z); mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z); }
    const float inv = 1.0f / leaf;
    int minb[3], maxb[3], divb[3];
    for (int k = 0; k < 3; k++) { minb[k] = (int)std::floor(mn[k] * inv); maxb[k] = (int)std::floor(mx[k] * inv); divb[k] = maxb[k] - minb[k] + 1; }
    const int64_t mul1 = divb[0], mul2 = (int64_t)divb[0] * divb[1];
    std::vector<std::pair<int64_t, int>> idx(in.size());
    for (size_t i = 0; i < in.size(); i++) {
        const P4 &p = in[i];
        int64_t ijk0 = (int64_t)std::floor(p.x * inv) - minb[0], ijk1 = (int64_t)std::floor(p.y * inv) - minb[1], ijk2 = (int64_t)std::floor(p.z * inv) - minb[2];
        idx[i] = {ijk0 + ijk1 * mul1 + ijk2 * mul2, (int)i};
    }
    std::stable_sort(idx.begin(), idx.end(), [](const std::pair<int64_t, int> &a, const std::pair<int64_t, int> &b) { return a.first < b.first; });
    size_t s = 0;
    while (s < idx.size()) {
        size_t e = s;
        float cx = 0, cy = 0, cz = 0, ci = 0;
        while (e < idx.size() && idx[e].first == idx[s].first) { const P4 &p = in[idx[e].second]; cx += p.x; cy += p.y; cz += p.z; ci += p.i; e++; }
        const float nn = (float)(e - s);
        out.push_back(P4{cx / nn, cy / nn, cz / nn, ci / nn});
        s = e;
    }
}

// pcl::CropBox with setNegative(false): keep min <= p <= max (Vector4f bounds)
void crop_box(const Cloud &in, const float mn[3], const float mx[3], Cloud &out) {
    out.clear();
    for (const P4 &p : in) if (!(p.x < mn[0] || p.y < mn[1] || p.z < mn[2] || p.x > mx[0] || p.y > mx[1] || p.z > mx[2])) out.push_back(p);
}

// exact 5 nearest neighbours, squared float distance (FLANN L2_Simple accumulates in float), ascending, ties -> lower index
void knn5(const Cloud &map, float qx, float qy, float qz, int idx[5], float d2[5]) {
    for (int k = 0; k < 5; k++) { idx[k] = -1; d2[k] = std::numeric_limits<float>::max(); }
    for (size_t i = 0; i < map.size(); i++) {
        const float dx = map[i].x - qx, dy = map[i].y - qy, dz = map[i].z - qz;
        const float d = dx * dx + dy * dy + dz * dz;
        if (d < d2[4]) {
            int k = 4;
            while (k > 0 && d < d2[k - 1]) { d2[k] = d2[k - 1]; idx[k] = idx[k - 1]; k--; }
            d2[k] = d; idx[k] = (int)i;
        }
    }
}

// 1 m cell grid over the map: (cell key, point index) sorted by key; a query scans the 27 cells around it. Every point closer
// than 1 m lies in that block, so whenever d2[4] < 1 the result equals knn5(); otherwise the caller rejects the query anyway.
struct GridIndex {
    const Cloud *map = nullptr;
    std::vector<std::pair<uint64_t, int>> cells;
    static uint64_t key(int x, int y, int z) { return ((uint64_t)(uint32_t)(x + (1 << 20)) << 42) | ((uint64_t)(uint32_t)(y + (1 << 20)) << 21) | (uint64_t)(uint32_t)(z + (1 << 20)); }
    void build(const Cloud &m) {
        map = &m;
        cells.resize(m.size());
        for (size_t i = 0; i < m.size(); i++) cells[i] = {key((int)std::floor(m[i].x), (int)std::floor(m[i].y), (int)std::floor(m[i].z)), (int)i};
        std::sort(cells.begin(), cells.end());
    }
    void knn5(float qx, float qy, float qz, int idx[5], float d2[5]) const {
        for (int k = 0; k < 5; k++) { idx[k] = 0x7fffffff; d2[k] = std::numeric_limits<float>::max(); }
        const int cx = (int)std::floor(qx), cy = (int)std::floor(qy), cz = (int)std::floor(qz);
        for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            const uint64_t k = key(cx + dx, cy + dy, cz + dz);
            auto it = std::lower_bound(cells.begin(), cells.end(), std::make_pair(k, 0));
            for (; it != cells.end() && it->first == k; ++it) {
                const P4 &m = (*map)[it->second];
                const float ex = m.x - qx, ey = m.y - qy, ez = m.z - qz;
                const float d = ex * ex + ey * ey + ez * ez;
                const int oi = it->second;
                if (d < d2[4] || (d == d2[4] && oi < idx[4])) {
                    int kk = 4;
                    while (kk > 0 && (d < d2[kk - 1] || (d == d2[kk - 1] && oi < idx[kk - 1]))) { d2[kk] = d2[kk - 1]; idx[kk] = idx[kk - 1]; kk--; }
                    d2[kk] = d; idx[kk] = oi;
                }
            }
        }
    }
};

inline void associate_point(const double pose[7], const P4 &p, float out[3]) {   // pointAssociaToMap (:354-362)
    Q4 q = Q4::from_xyzw(pose);
    V3 pw = q * V3(p.x, p.y, p.z) + V3(pose + 4);
    out[0] = (float)pw.x; out[1] = (float)pw.y; out[2] = (float)pw.z;
}

// EdgeCostFactor (:117-172): returns true and the line end points if the 5-NN pass the 1 m gate and the PCA line test
bool edge_association(const Cloud &map, const GridIndex *gi, const double pose[7], const P4 &p, V3 &pa, V3 &pb) {
    if (map.size() < 5) return false;
    float c[3];
    associate_point(pose, p, c);
    int idx[5]; float d2[5];
    if (gi) gi->knn5(c[0], c[1], c[2], idx, d2); else knn5(map, c[0], c[1], c[2], idx, d2);
    if (!(d2[4] < 1.0f)) return false;
    V3 near[5], center;
    for (int j = 0; j < 5; j++) { near[j] = V3(map[idx[j]].x, map[idx[j]].y, map[idx[j]].z); center = center + near[j]; }
    center = center / 5.0;
    M3 cov;
    for (int j = 0; j < 5; j++) { V3 d = near[j] - center; for (int a = 0; a < 3; a++) for (int b2 = 0; b2 < 3; b2++) cov(a, b2) += d[a] * d[b2]; }
    double w[3]; M3 V;
    sym_eigen3(cov, w, V);
    if (!(w[2] > 3 * w[1])) return false;
    V3 dir(V(0, 2), V(1, 2), V(2, 2));
    pa = 0.1 * dir + center;
    pb = -0.1 * dir + center;
    return true;
}

// SurfCostFactor (:174-232)
bool surf_association(const Cloud &map, const GridIndex *gi, const double pose[7], const P4 &p, V3 &nrm, double &d) {
    if (map.size() < 5) return false;
    float c[3];
    associate_point(pose, p, c);
    int idx[5]; float d2[5];
    if (gi) gi->knn5(c[0], c[1], c[2], idx, d2); else knn5(map, c[0], c[1], c[2], idx, d2);
    if (!(d2[4] < 1.0f)) return false;
    double A[15], B[5] = {-1, -1, -1, -1, -1};
    for (int j = 0; j < 5; j++) { A[3 * j] = map[idx[j]].x; A[3 * j + 1] = map[idx[j]].y; A[3 * j + 2] = map[idx[j]].z; }
    V3 nn = colpiv_qr_solve_5x3(A, B);
    d = 1.0 / norm(nn);
    nrm = nn / norm(nn);
    for (int j = 0; j < 5; j++)
        if (std::fabs(nrm.x * map[idx[j]].x + nrm.y * map[idx[j]].y + nrm.z * map[idx[j]].z + d) > 0.2) return false;
    return true;
}

}  // namespace

struct vilo_s2m {
    vilf_options o;
    Cloud mapEdge, mapSurf;
    double pose[7] = {0, 0, 0, 1, 0, 0, 0};        // parameter_opti: q (xyzw), t  == globalOdom
    double pose_last[7] = {0, 0, 0, 1, 0, 0, 0};   // globalOdom_last
};

extern "C" vilo_s2m *vilo_s2m_create(const vilf_options *o) { vilo_s2m *s = new vilo_s2m(); s->o = *o; return s; }
extern "C" void vilo_s2m_destroy(vilo_s2m *s) { delete s; }
extern "C" vilo_s2m *vilo_s2m_clone(const vilo_s2m *s) { return new vilo_s2m(*s); }      // copy of maps + poses (replays from a prepared state)
extern "C" int vilo_s2m_set_pose(vilo_s2m *s, const double p[7], const double pl[7]) { std::memcpy(s->pose, p, 56); std::memcpy(s->pose_last, pl, 56); return VILF_OK; }

static Cloud to_cloud(const float *xyzi, int n) { Cloud c(n); for (int i = 0; i < n; i++) c[i] = P4{xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2], xyzi[4 * i + 3]}; return c; }

extern "C" int vilo_s2m_init(vilo_s2m *s, const float *e, int ne, const float *f, int nf) {   // localMapInited (:105-115)
    Cloud ce = to_cloud(e, ne), cf = to_cloud(f, nf);
    s->mapEdge.insert(s->mapEdge.end(), ce.begin(), ce.end());
    s->mapSurf.insert(s->mapSurf.end(), cf.begin(), cf.end());
    return VILF_OK;
}

extern "C" int vilo_s2m_get_map(vilo_s2m *s, int which, float *out, int cap, int *n_out) {
    const Cloud &c = which == 0 ? s->mapEdge : s->mapSurf;
    *n_out = (int)c.size();
    for (int i = 0; i < (int)c.size() && i < cap; i++) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = c[i].i; }
    return VILF_OK;
}

extern "C" int vilo_s2m_step(vilo_s2m *s, const float *e, int ne, const float *f, int nf, vilf_scan2map_result *res) {
    std::memset(res, 0, sizeof(*res));
    // constant-velocity prediction: globalOdom_est = globalOdom * (globalOdom_last^-1 * globalOdom)   (:238-243)
    M3 R = toR(Q4::from_xyzw(s->pose)), Rl = toR(Q4::from_xyzw(s->pose_last));
    V3 t(s->pose + 4), tl(s->pose_last + 4);
    M3 Rrel = transpose(Rl) * R;
    V3 trel = transpose(Rl) * (t - tl);
    M3 Re = R * Rrel;
    V3 te = R * trel + t;
    double prev[7];
    std::memcpy(prev, s->pose, 56);
    std::memcpy(s->pose_last, s->pose, 56);
    fromR(Re).to_xyzw(s->pose);
    s->pose[4] = te.x; s->pose[5] = te.y; s->pose[6] = te.z;
    // down-sampling (:246-251)
    Cloud ve, vs;
    voxel_grid(to_cloud(e, ne), (float)s->o.edge_leaf_size, ve);
    voxel_grid(to_cloud(f, nf), (float)s->o.surf_leaf_size, vs);
    res->n_edge_ds = (int)ve.size(); res->n_surf_ds = (int)vs.size();
    if (s->mapEdge.size() > 10 && s->mapSurf.size() > 50) {
        GridIndex ge, gs;                              // ≙ kdtreeEdgeMap / kdtreeSurfMap ->setInputCloud (:256-257)
        ge.build(s->mapEdge); gs.build(s->mapSurf);
        for (int iter = 0; iter < s->o.s2m_outer_iterations && iter < 2; iter++) {
            HuberLoss loss(s->o.huber_a);
            Problem pb;
            int blk = pb.add_parameter_block(s->pose, 7, PARAM_SE3);
            std::vector<std::unique_ptr<CostFunction>> costs;
            int nef = 0, nsf = 0;
            for (const P4 &p : ve) { V3 a, b; if (edge_association(s->mapEdge, &ge, s->pose, p, a, b)) { costs.emplace_back(new EdgeCostFunction(V3(p.x, p.y, p.z), a, b)); pb.add_residual_block(costs.back().get(), &loss, {blk}); nef++; } }
            for (const P4 &p : vs) { V3 nn; double d; if (surf_association(s->mapSurf, &gs, s->pose, p, nn, d)) { costs.emplace_back(new SurfCostFunction(V3(p.x, p.y, p.z), nn, d)); pb.add_residual_block(costs.back().get(), &loss, {blk}); nsf++; } }
            res->n_edge_factors[iter] = nef; res->n_surf_factors[iter] = nsf;
            SolverOptions so;
            so.strategy = STRATEGY_LM;                 // Ceres default trust-region strategy; DENSE_QR on one 6-dof block
            so.max_num_iterations = s->o.s2m_max_iterations;
            SolveSummary sum;
            if (nef + nsf > 0) solve(so, pb, sum);
            res->iterations[iter] = sum.num_iterations;
            res->final_cost[iter] = sum.final_cost;
        }
    }
    // createSubMap (:298-352): append the registered points, crop to +-100 m around the pose, voxel down-sample
    for (const P4 &p : ve) { float c[3]; associate_point(s->pose, p, c); s->mapEdge.push_back(P4{c[0], c[1], c[2], p.i}); }
    for (const P4 &p : vs) { float c[3]; associate_point(s->pose, p, c); s->mapSurf.push_back(P4{c[0], c[1], c[2], p.i}); }
    const double h = s->o.s2m_crop_half;
    float mn[3] = {(float)(s->pose[4] - h), (float)(s->pose[5] - h), (float)(s->pose[6] - h)}, mx[3] = {(float)(s->pose[4] + h), (float)(s->pose[5] + h), (float)(s->pose[6] + h)};
    Cloud ce, cs, oe, os;
    crop_box(s->mapEdge, mn, mx, ce); crop_box(s->mapSurf, mn, mx, cs);
    voxel_grid(ce, (float)s->o.edge_leaf_size, oe); voxel_grid(cs, (float)s->o.surf_leaf_size, os);
    s->mapEdge.swap(oe); s->mapSurf.swap(os);
    res->map_edge_size = (int)s->mapEdge.size(); res->map_surf_size = (int)s->mapSurf.size();
    std::memcpy(res->pose_qt, s->pose, 56);
    // /Odometry: q_last^-1 * q, q_last^-1 * (t - t_last)  (feature_tracker_node.cpp:392-394)
    Q4 ql = Q4::from_xyzw(prev), qn = Q4::from_xyzw(s->pose);
    Q4 qr = inverse(ql) * qn;
    V3 tr = inverse(ql) * (V3(s->pose + 4) - V3(prev + 4));
    qr.to_xyzw(res->rel_q);
    res->rel_t[0] = tr.x; res->rel_t[1] = tr.y; res->rel_t[2] = tr.z;
    return VILF_OK;
}

extern "C" int vilo_knn5_bruteforce(const float *map, int nm, const float *q, int nq, int *idx5, float *d5) {
    Cloud c = to_cloud(map, nm);
    for (int i = 0; i < nq; i++) knn5(c, q[3 * i], q[3 * i + 1], q[3 * i + 2], idx5 + 5 * i, d5 + 5 * i);
    return VILF_OK;
}
extern "C" int vilo_knn5_grid(const float *map, int nm, const float *q, int nq, int *idx5, float *d5) {
    Cloud c = to_cloud(map, nm);
    GridIndex g; g.build(c);
    for (int i = 0; i < nq; i++) g.knn5(q[3 * i], q[3 * i + 1], q[3 * i + 2], idx5 + 5 * i, d5 + 5 * i);
    return VILF_OK;
}
extern "C" int vilo_voxel_grid(const float *xyzi, int n, float leaf, float *out, int cap, int *n_out) {
    Cloud o; voxel_grid(to_cloud(xyzi, n), leaf, o);
    *n_out = (int)o.size();
    for (int i = 0; i < (int)o.size() && i < cap; i++) { out[4 * i] = o[i].x; out[4 * i + 1] = o[i].y; out[4 * i + 2] = o[i].z; out[4 * i + 3] = o[i].i; }
    return VILF_OK;
}
extern "C" int vilo_s2m_associate_edge(const float *map, int nm, const float *pts, int np, const double pose[7], unsigned char *valid, double *pa, double *pb) {
    Cloud c = to_cloud(map, nm), p = to_cloud(pts, np);
    for (int i = 0; i < np; i++) { V3 a, b; valid[i] = edge_association(c, nullptr, pose, p[i], a, b) ? 1 : 0; if (valid[i]) { pa[3 * i] = a.x; pa[3 * i + 1] = a.y; pa[3 * i + 2] = a.z; pb[3 * i] = b.x; pb[3 * i + 1] = b.y; pb[3 * i + 2] = b.z; } }
    return VILF_OK;
}
extern "C" int vilo_s2m_associate_surf(const float *map, int nm, const float *pts, int np, const double pose[7], unsigned char *valid, double *nrm, double *d) {
    Cloud c = to_cloud(map, nm), p = to_cloud(pts, np);
    for (int i = 0; i < np; i++) { V3 n; double dd; valid[i] = surf_association(c, nullptr, pose, p[i], n, dd) ? 1 : 0; if (valid[i]) { nrm[3 * i] = n.x; nrm[3 * i + 1] = n.y; nrm[3 * i + 2] = n.z; d[i] = dd; } }
    return VILF_OK;
}
