// ORACLE — TEST INFRASTRUCTURE ONLY. "parity unpinned" vs the reference binary (cannot be built here).
// window.cpp — restatement of Estimator::optimization() (vins_estimator/estimator.cpp:689-1050),
// vector2double / double2vector (:505-638) and MarginalizationInfo (factor/marginalization_factor.cpp:89-319).
#include "oracle_api.h"
#include "solver.hpp"
#include <map>
#include <chrono>
#include <set>

using namespace ora;

static thread_local SolveSummary g_last_summary;

extern "C" void vilo_default_options(vilf_options *o) {
    std::memset(o, 0, sizeof(*o));
    o->window_size = 10;
    o->max_num_iterations = 8;       // kitti_config.yaml:74
    o->max_solver_time = -1.0;       // parity runs: wall-clock limit disabled (SURVEY finding 3)
    o->focal_length = 460.0;         // parameters.h:23
    o->cauchy_a = 1.0;               // estimator.cpp:694
    o->G[0] = 0; o->G[1] = 0; o->G[2] = 9.81007;  // kitti_config.yaml:82
    o->estimate_extrinsic = 0; o->estimate_td = 0; o->use_lidar_const = 1;
    // kitti_config.yaml:48-61 (RIC is re-orthonormalised through a quaternion in parameters.cpp:108-110)
    const double ric[9] = {0.00781297, -0.0042792, 0.99996, -0.999859, -0.014868, 0.00774856, 0.0148343, -0.99988, -0.00439476};
    const double tic[3] = {1.1439, -0.312718, 0.726546};
    const double rcl[9] = {7.027555e-03, -9.999753e-01, 2.599616e-05, -2.254837e-03, -4.184312e-05, -9.999975e-01, 9.999728e-01, 7.027479e-03, -2.255075e-03};
    const double tcl[3] = {-7.137748e-03, -7.482656e-02, -3.336324e-01};
    M3 R1 = toR(normalized(fromR(M3::from(ric))));
    M3 R2 = toR(normalized(fromR(M3::from(rcl))));
    for (int i = 0; i < 9; i++) { o->RIC[i] = R1.m[i]; o->RCL[i] = R2.m[i]; }
    for (int i = 0; i < 3; i++) { o->TIC[i] = tic[i]; o->TCL[i] = tcl[i]; }
    o->TR = 0; o->ROW = 370;
    o->init_depth = 5.0;
    o->edge_leaf_size = 0.4; o->surf_leaf_size = 0.8; o->huber_a = 0.1;
    o->s2m_outer_iterations = 2; o->s2m_max_iterations = 4; o->s2m_crop_half = 100.0;
}

namespace {

struct WindowState {
    int NF = 0, F = 0;
    std::vector<double> pose, sb, feat;
    double ex[7]; double td[1];
};

struct FactorSet {  // owns the cost functions of one problem / marginalization
    std::vector<std::unique_ptr<CostFunction>> costs;
    std::unique_ptr<LossFunction> loss;
};

static inline int id_pose(int NF, int i) { (void)NF; return i; }
static inline int id_sb(int NF, int i) { return NF + i; }
static inline int id_ex(int NF) { return 2 * NF; }
static inline int id_td(int NF) { return 2 * NF + 1; }
static inline int id_feat(int NF, int k) { return 2 * NF + 2 + k; }

double *block_ptr(WindowState &s, int id) {
    const int NF = s.NF;
    if (id < NF) return &s.pose[7 * id];
    if (id < 2 * NF) return &s.sb[9 * (id - NF)];
    if (id == 2 * NF) return s.ex;
    if (id == 2 * NF + 1) return s.td;
    return &s.feat[id - (2 * NF + 2)];
}
int block_size(const WindowState &s, int id) {
    const int NF = s.NF;
    if (id < NF) return 7;
    if (id < 2 * NF) return 9;
    if (id == 2 * NF) return 7;
    return 1;
}

}  // namespace

extern "C" int vilo_window_solve(const vilf_options *o, const vilf_window_in *in, const vilf_prior *prior, vilf_window_out *out) {
    if (!o || !in || !out) return VILF_ERR_INVALID_ARGUMENT;
    const int NF = in->n_frames, F = in->n_features;
    if (NF != o->window_size + 1) return VILF_ERR_INVALID_ARGUMENT;
    WindowState s; s.NF = NF; s.F = F;
    s.pose.assign(in->para_pose, in->para_pose + 7 * NF);
    s.sb.assign(in->para_speed_bias, in->para_speed_bias + 9 * NF);
    s.feat.assign(in->para_feature, in->para_feature + F);
    std::memcpy(s.ex, in->para_ex_pose, sizeof(s.ex));
    s.td[0] = in->para_td;

    // gauge reference (double2vector :551-552): Rs[0], Ps[0] before the solve
    M3 R0_before = in->gauge_R0 ? M3::from(in->gauge_R0) : toR(Q4::from_xyzw(&s.pose[3]));
    V3 P0_before = in->gauge_P0 ? V3(in->gauge_P0) : V3(&s.pose[0]);

    Problem pb;
    FactorSet fs;
    fs.loss.reset(new CauchyLoss(o->cauchy_a));
    std::vector<int> bid_pose(NF), bid_sb(NF), bid_feat(F);
    for (int i = 0; i < NF; i++) {                                       // estimator.cpp:695-700
        bid_pose[i] = pb.add_parameter_block(&s.pose[7 * i], 7, PARAM_POSE);
        bid_sb[i] = pb.add_parameter_block(&s.sb[9 * i], 9, PARAM_EUCLID);
    }
    int bid_ex = pb.add_parameter_block(s.ex, 7, PARAM_POSE);            // :701-712
    if (!o->estimate_extrinsic) pb.set_constant(bid_ex);
    int bid_td = -1;
    if (o->estimate_td) bid_td = pb.add_parameter_block(s.td, 1, PARAM_EUCLID);  // :713-717
    for (int k = 0; k < F; k++) {
        bid_feat[k] = pb.add_parameter_block(&s.feat[k], 1, PARAM_EUCLID);
        pb.blocks[bid_feat[k]].eliminate = true;
        if (in->feature_const[k]) pb.set_constant(bid_feat[k]);           // :780-781,789-790
    }
    auto id_to_bid = [&](int id) -> int {
        if (id < NF) return bid_pose[id];
        if (id < 2 * NF) return bid_sb[id - NF];
        if (id == 2 * NF) return bid_ex;
        if (id == 2 * NF + 1) return bid_td;
        return bid_feat[id - (2 * NF + 2)];
    };
    if (prior && prior->valid) {                                          // :722-728
        fs.costs.emplace_back(new MarginalizationFactor(prior));
        std::vector<int> ps;
        for (int i = 0; i < prior->n_blocks; i++) ps.push_back(id_to_bid(prior->block_id[i]));
        pb.add_residual_block(fs.costs.back().get(), nullptr, ps);
    }
    if (o->use_lidar_const) {                                             // :730-740
        for (int i = 0; i < NF - 1; i++) {
            fs.costs.emplace_back(new LidarFactor(&in->lidar[i + 1], o));
            pb.add_residual_block(fs.costs.back().get(), nullptr, {bid_pose[i], bid_pose[i + 1]});
        }
    }
    V3 G(o->G);
    for (int i = 0; i < NF - 1; i++) {                                    // :742-749
        int j = i + 1;
        if (in->imu[j].sum_dt > 10.0) continue;
        fs.costs.emplace_back(new IMUFactor(&in->imu[j], G));
        pb.add_residual_block(fs.costs.back().get(), nullptr, {bid_pose[i], bid_sb[i], bid_pose[j], bid_sb[j]});
    }
    const double sqrt_info = o->focal_length / 1.5;                       // estimator.cpp:17
    for (int k = 0; k < F; k++) {                                         // :750-794
        int o0 = in->feature_obs_offset[k], o1 = in->feature_obs_offset[k + 1];
        int imu_i = in->feature_start_frame[k];
        V3 pts_i(in->obs_point + 3 * o0);
        for (int t = o0 + 1; t < o1; t++) {
            int imu_j = imu_i + (t - o0);
            V3 pts_j(in->obs_point + 3 * t);
            if (o->estimate_td) {
                fs.costs.emplace_back(new ProjectionTdFactor(pts_i, pts_j, in->obs_velocity + 2 * o0, in->obs_velocity + 2 * t,
                                                             in->obs_cur_td[o0], in->obs_cur_td[t], in->obs_row[o0], in->obs_row[t],
                                                             sqrt_info, o->TR, o->ROW));
                pb.add_residual_block(fs.costs.back().get(), fs.loss.get(), {bid_pose[imu_i], bid_pose[imu_j], bid_ex, bid_feat[k], bid_td});
            } else {
                fs.costs.emplace_back(new ProjectionFactor(pts_i, pts_j, sqrt_info));
                pb.add_residual_block(fs.costs.back().get(), fs.loss.get(), {bid_pose[imu_i], bid_pose[imu_j], bid_ex, bid_feat[k]});
            }
        }
    }
    SolverOptions so;                                                     // :838-850
    so.strategy = STRATEGY_DOGLEG;
    so.max_num_iterations = o->max_num_iterations;
    so.max_solver_time = o->max_solver_time;
    if (so.max_solver_time > 0 && in->marginalization_flag == VILF_MARGIN_OLD) so.max_solver_time *= 4.0 / 5.0;
    SolveSummary sum;
    auto t0 = std::chrono::steady_clock::now();
    solve(so, pb, sum);
    double usec = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    g_last_summary = sum;

    if (out->para_pose) std::memcpy(out->para_pose, s.pose.data(), sizeof(double) * 7 * NF);
    if (out->para_speed_bias) std::memcpy(out->para_speed_bias, s.sb.data(), sizeof(double) * 9 * NF);
    if (out->para_feature) std::memcpy(out->para_feature, s.feat.data(), sizeof(double) * F);

    // double2vector (estimator.cpp:549-638)
    V3 origin_R0 = R2ypr(R0_before);
    V3 origin_P0 = P0_before;
    M3 R00 = toR(Q4::from_xyzw(&s.pose[3]));
    V3 origin_R00 = R2ypr(R00);
    double y_diff = origin_R0.x - origin_R00.x;
    M3 rot_diff = ypr2R(V3(y_diff, 0, 0));
    if (std::fabs(std::fabs(origin_R0.y) - 90) < 1.0 || std::fabs(std::fabs(origin_R00.y) - 90) < 1.0)
        rot_diff = R0_before * transpose(R00);
    for (int i = 0; i < NF; i++) {
        M3 Ri = rot_diff * toR(normalized(Q4::from_xyzw(&s.pose[7 * i + 3])));
        V3 Pi = rot_diff * V3(s.pose[7 * i] - s.pose[0], s.pose[7 * i + 1] - s.pose[1], s.pose[7 * i + 2] - s.pose[2]) + origin_P0;
        V3 Vi = rot_diff * V3(&s.sb[9 * i]);
        for (int k = 0; k < 9; k++) out->Rs[9 * i + k] = Ri.m[k];
        out->Ps[3 * i] = Pi.x; out->Ps[3 * i + 1] = Pi.y; out->Ps[3 * i + 2] = Pi.z;
        out->Vs[3 * i] = Vi.x; out->Vs[3 * i + 1] = Vi.y; out->Vs[3 * i + 2] = Vi.z;
        for (int k = 0; k < 3; k++) { out->Bas[3 * i + k] = s.sb[9 * i + 3 + k]; out->Bgs[3 * i + k] = s.sb[9 * i + 6 + k]; }
    }
    for (int k = 0; k < 3; k++) out->tic[k] = s.ex[k];
    M3 ricm = toR(Q4::from_xyzw(s.ex + 3));
    for (int k = 0; k < 9; k++) out->ric[k] = ricm.m[k];
    out->td = o->estimate_td ? s.td[0] : in->para_td;
    out->summary.num_iterations = sum.num_iterations;
    out->summary.num_successful_steps = sum.num_successful_steps;
    out->summary.num_linear_solves = sum.num_linear_solves;
    out->summary.termination = sum.termination;
    out->summary.initial_cost = sum.initial_cost;
    out->summary.final_cost = sum.final_cost;
    out->summary.final_radius = sum.final_radius;
    out->summary.usec_solve = usec;
    return VILF_OK;
}

extern "C" int vilo_last_trace(double *rows, int capacity) {
    int n = std::min((int)g_last_summary.iterations.size(), capacity);
    for (int i = 0; i < n; i++) {
        auto &it = g_last_summary.iterations[i];
        double *r = rows + 9 * i;
        r[0] = it.iteration; r[1] = it.cost; r[2] = it.cost_change; r[3] = it.gradient_max_norm; r[4] = it.step_norm;
        r[5] = it.relative_decrease; r[6] = it.trust_region_radius; r[7] = it.step_is_valid; r[8] = it.step_is_successful;
    }
    return n;
}

// ---------------------------------------------------------------------------------------------------------
// Marginalization
namespace {

struct ResidualBlockInfo {      // marginalization_factor.h:15-36
    const CostFunction *cost;
    const LossFunction *loss;
    std::vector<int> ids;       // parameter block ids (replace the reference's addresses)
    std::vector<int> drop_set;
    std::vector<std::vector<double>> jacobians;  // global-size row-major
    std::vector<double> residuals;
};

}  // namespace

extern "C" int vilo_window_marginalize(const vilf_options *o, const vilf_window_in *in, const vilf_window_out *solved,
                                       const vilf_prior *prior, vilf_prior *pout) {
    if (!o || !in || !solved || !pout) return VILF_ERR_INVALID_ARGUMENT;
    const int NF = in->n_frames, F = in->n_features, W = o->window_size;
    // vector2double() from the post-gauge state (estimator.cpp:866 / :986)
    WindowState s; s.NF = NF; s.F = F;
    s.pose.resize(7 * NF); s.sb.resize(9 * NF); s.feat.resize(F);
    for (int i = 0; i < NF; i++) {
        for (int k = 0; k < 3; k++) s.pose[7 * i + k] = solved->Ps[3 * i + k];
        Q4 q = fromR(M3::from(solved->Rs + 9 * i));
        q.to_xyzw(&s.pose[7 * i + 3]);
        for (int k = 0; k < 3; k++) { s.sb[9 * i + k] = solved->Vs[3 * i + k]; s.sb[9 * i + 3 + k] = solved->Bas[3 * i + k]; s.sb[9 * i + 6 + k] = solved->Bgs[3 * i + k]; }
    }
    for (int k = 0; k < 3; k++) s.ex[k] = solved->tic[k];
    fromR(M3::from(solved->ric)).to_xyzw(s.ex + 3);
    s.td[0] = solved->td;
    for (int k = 0; k < F; k++) {
        // setDepth (feature_manager.cpp:150-168) then getDepthVector (:194-216)
        double est = 1.0 / solved->para_feature[k];
        s.feat[k] = est > 0 ? 1. / est : 1. / o->init_depth;
    }

    FactorSet fs;
    fs.loss.reset(new CauchyLoss(o->cauchy_a));
    std::vector<ResidualBlockInfo> factors;
    const bool have_prior = prior && prior->valid;
    std::map<int, int> addr_shift;  // id -> shifted id

    if (in->marginalization_flag == VILF_MARGIN_OLD) {
        if (have_prior) {                                                  // :868-884
            ResidualBlockInfo rb;
            fs.costs.emplace_back(new MarginalizationFactor(prior));
            rb.cost = fs.costs.back().get(); rb.loss = nullptr;
            for (int i = 0; i < prior->n_blocks; i++) {
                rb.ids.push_back(prior->block_id[i]);
                if (prior->block_id[i] == id_pose(NF, 0) || prior->block_id[i] == id_sb(NF, 0)) rb.drop_set.push_back(i);
            }
            factors.push_back(std::move(rb));
        }
        if (o->use_lidar_const) {                                          // :886-895
            ResidualBlockInfo rb;
            fs.costs.emplace_back(new LidarFactor(&in->lidar[1], o));
            rb.cost = fs.costs.back().get(); rb.loss = nullptr;
            rb.ids = {id_pose(NF, 0), id_pose(NF, 1)};
            rb.drop_set = {0, 1};
            factors.push_back(std::move(rb));
        }
        if (in->imu[1].sum_dt < 10.0) {                                    // :896-905
            ResidualBlockInfo rb;
            fs.costs.emplace_back(new IMUFactor(&in->imu[1], V3(o->G)));
            rb.cost = fs.costs.back().get(); rb.loss = nullptr;
            rb.ids = {id_pose(NF, 0), id_sb(NF, 0), id_pose(NF, 1), id_sb(NF, 1)};
            rb.drop_set = {0, 1};
            factors.push_back(std::move(rb));
        }
        const double sqrt_info = o->focal_length / 1.5;
        for (int k = 0; k < F; k++) {                                      // :907-950
            if (in->feature_start_frame[k] != 0) continue;
            int o0 = in->feature_obs_offset[k], o1 = in->feature_obs_offset[k + 1];
            V3 pts_i(in->obs_point + 3 * o0);
            for (int t = o0 + 1; t < o1; t++) {
                int imu_j = t - o0;
                V3 pts_j(in->obs_point + 3 * t);
                ResidualBlockInfo rb;
                if (o->estimate_td) {
                    fs.costs.emplace_back(new ProjectionTdFactor(pts_i, pts_j, in->obs_velocity + 2 * o0, in->obs_velocity + 2 * t,
                                                                 in->obs_cur_td[o0], in->obs_cur_td[t], in->obs_row[o0], in->obs_row[t],
                                                                 sqrt_info, o->TR, o->ROW));
                    rb.ids = {id_pose(NF, 0), id_pose(NF, imu_j), id_ex(NF), id_feat(NF, k), id_td(NF)};
                } else {
                    fs.costs.emplace_back(new ProjectionFactor(pts_i, pts_j, sqrt_info));
                    rb.ids = {id_pose(NF, 0), id_pose(NF, imu_j), id_ex(NF), id_feat(NF, k)};
                }
                rb.cost = fs.costs.back().get(); rb.loss = fs.loss.get();
                rb.drop_set = {0, 3};
                factors.push_back(std::move(rb));
            }
        }
        for (int i = 1; i <= W; i++) { addr_shift[id_pose(NF, i)] = id_pose(NF, i - 1); addr_shift[id_sb(NF, i)] = id_sb(NF, i - 1); }  // :960-971
        addr_shift[id_ex(NF)] = id_ex(NF);
        if (o->estimate_td) addr_shift[id_td(NF)] = id_td(NF);
    } else {
        bool touches = false;                                              // :982-983
        if (have_prior) for (int i = 0; i < prior->n_blocks; i++) if (prior->block_id[i] == id_pose(NF, W - 1)) touches = true;
        if (!touches) {
            if (have_prior) *pout = *prior; else { std::memset(pout, 0, sizeof(*pout)); }
            return VILF_OK;
        }
        ResidualBlockInfo rb;
        fs.costs.emplace_back(new MarginalizationFactor(prior));
        rb.cost = fs.costs.back().get(); rb.loss = nullptr;
        for (int i = 0; i < prior->n_blocks; i++) {
            rb.ids.push_back(prior->block_id[i]);
            if (prior->block_id[i] == id_pose(NF, W - 1)) rb.drop_set.push_back(i);
        }
        factors.push_back(std::move(rb));
        for (int i = 0; i <= W; i++) {                                     // :1016-1037
            if (i == W - 1) continue;
            if (i == W) { addr_shift[id_pose(NF, i)] = id_pose(NF, i - 1); addr_shift[id_sb(NF, i)] = id_sb(NF, i - 1); }
            else { addr_shift[id_pose(NF, i)] = id_pose(NF, i); addr_shift[id_sb(NF, i)] = id_sb(NF, i); }
        }
        addr_shift[id_ex(NF)] = id_ex(NF);
        if (o->estimate_td) addr_shift[id_td(NF)] = id_td(NF);
    }

    // addResidualBlockInfo (:89-108): sizes of every touched block; dropped blocks get an idx entry
    std::map<int, int> parameter_block_size;  // ordered by id: the build's deterministic replacement for the
    std::set<int> dropped;                    // reference's address-keyed unordered_map iteration order
    for (auto &rb : factors) {
        for (size_t i = 0; i < rb.ids.size(); i++) parameter_block_size[rb.ids[i]] = rb.cost->block_sizes[i];
        for (int d : rb.drop_set) dropped.insert(rb.ids[d]);
    }
    // preMarginalize (:110-129): Evaluate with corrector + snapshot x0
    std::map<int, std::vector<double>> parameter_block_data;
    for (auto &rb : factors) {
        const int np = (int)rb.ids.size(), nr = rb.cost->num_residuals;
        std::vector<const double *> pp(np);
        std::vector<double *> jp(np);
        rb.jacobians.resize(np);
        rb.residuals.assign(nr, 0.0);
        for (int i = 0; i < np; i++) { pp[i] = block_ptr(s, rb.ids[i]); rb.jacobians[i].assign((size_t)nr * rb.cost->block_sizes[i], 0.0); jp[i] = rb.jacobians[i].data(); }
        rb.cost->Evaluate(pp.data(), rb.residuals.data(), jp.data());
        if (rb.loss) apply_corrector(rb.loss, nr, rb.residuals.data(), np, rb.cost->block_sizes.data(), jp.data(), nullptr);
        for (int i = 0; i < np; i++)
            if (!parameter_block_data.count(rb.ids[i])) parameter_block_data[rb.ids[i]] = std::vector<double>(pp[i], pp[i] + rb.cost->block_sizes[i]);
    }
    // marginalize (:174-297)
    auto localSize = [](int size) { return size == 7 ? 6 : size; };
    std::map<int, int> parameter_block_idx;
    int pos = 0;
    for (int id : dropped) { parameter_block_idx[id] = pos; pos += localSize(parameter_block_size[id]); }
    const int m = pos;
    for (auto &kv : parameter_block_size) if (!dropped.count(kv.first)) { parameter_block_idx[kv.first] = pos; pos += localSize(kv.second); }
    const int n = pos - m;
    if (n > VILF_PRIOR_MAX_DIM) return VILF_ERR_UNSUPPORTED;
    // A, b: 4 round-robin partial sums reduced in the order 3,2,1,0 (:233-261)
    const int NT = 4;
    std::vector<Mat> At(NT, Mat(pos, pos));
    std::vector<std::vector<double>> bt(NT, std::vector<double>(pos, 0.0));
    for (size_t f = 0; f < factors.size(); f++) {
        auto &rb = factors[f];
        Mat &A = At[f % NT]; std::vector<double> &b = bt[f % NT];
        const int np = (int)rb.ids.size(), nr = rb.cost->num_residuals;
        for (int i = 0; i < np; i++) {
            int idx_i = parameter_block_idx[rb.ids[i]], gi = rb.cost->block_sizes[i], size_i = localSize(gi);
            const double *Ji = rb.jacobians[i].data();
            for (int j = i; j < np; j++) {
                int idx_j = parameter_block_idx[rb.ids[j]], gj = rb.cost->block_sizes[j], size_j = localSize(gj);
                const double *Jj = rb.jacobians[j].data();
                for (int a = 0; a < size_i; a++)
                    for (int c = 0; c < size_j; c++) {
                        double sacc = 0;
                        for (int r = 0; r < nr; r++) sacc += Ji[r * gi + a] * Jj[r * gj + c];
                        A(idx_i + a, idx_j + c) += sacc;
                        if (i != j) A(idx_j + c, idx_i + a) = A(idx_i + a, idx_j + c);
                    }
            }
            for (int a = 0; a < size_i; a++) { double sacc = 0; for (int r = 0; r < nr; r++) sacc += Ji[r * gi + a] * rb.residuals[r]; b[idx_i + a] += sacc; }
        }
    }
    Mat A(pos, pos); std::vector<double> b(pos, 0.0);
    for (int t = NT - 1; t >= 0; t--) { for (size_t k = 0; k < A.d.size(); k++) A.d[k] += At[t].d[k]; for (int k = 0; k < pos; k++) b[k] += bt[t][k]; }

    const double eps = 1e-8;
    Mat Amm(m, m);
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Amm(i, j) = 0.5 * (A(i, j) + A(j, i));
    std::vector<double> w; Mat V;
    sym_eigen(Amm, w, V);
    Mat Amm_inv(m, m);   // V diag(w>eps ? 1/w : 0) V^T
    {
        Mat VS(m, m);
        for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) VS(i, j) = V(i, j) * (w[j] > eps ? 1.0 / w[j] : 0.0);
        Amm_inv = matmul(VS, transpose(V));
    }
    Mat Arm(n, m), Amr(m, n), Arr(n, n);
    std::vector<double> bmm(b.begin(), b.begin() + m), brr(b.begin() + m, b.end());
    for (int i = 0; i < n; i++) for (int j = 0; j < m; j++) { Arm(i, j) = A(m + i, j); Amr(j, i) = A(j, m + i); }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Arr(i, j) = A(m + i, m + j);
    Mat T = matmul(Arm, Amm_inv);
    Mat TA = matmul(T, Amr);
    Mat Ar(n, n); std::vector<double> br(n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) Ar(i, j) = Arr(i, j) - TA(i, j);
        double sacc = 0; for (int k = 0; k < m; k++) sacc += T(i, k) * bmm[k];
        br[i] = brr[i] - sacc;
    }
    std::vector<double> w2; Mat V2;
    sym_eigen(Ar, w2, V2);
    std::memset(pout, 0, sizeof(*pout));
    pout->valid = 1; pout->n = n; pout->m = m;
    for (int i = 0; i < n; i++) {
        double S = w2[i] > eps ? w2[i] : 0.0;
        double S_inv = w2[i] > eps ? 1.0 / w2[i] : 0.0;
        double S_sqrt = std::sqrt(S), S_inv_sqrt = std::sqrt(S_inv);
        double sacc = 0;
        for (int k = 0; k < n; k++) { pout->linearized_jacobians[(size_t)i * n + k] = S_sqrt * V2(k, i); sacc += V2(k, i) * br[k]; }
        pout->linearized_residuals[i] = S_inv_sqrt * sacc;
    }
    // getParameterBlocks (:299-319)
    int nb = 0;
    for (auto &kv : parameter_block_idx) {
        if (kv.second < m) continue;
        if (nb >= VILF_PRIOR_MAX_BLOCKS) return VILF_ERR_UNSUPPORTED;
        auto it = addr_shift.find(kv.first);
        if (it == addr_shift.end()) return VILF_ERR_INVALID_ARGUMENT;
        pout->block_id[nb] = it->second;
        pout->block_size[nb] = parameter_block_size[kv.first];
        pout->block_idx[nb] = kv.second - m;
        auto &d = parameter_block_data[kv.first];
        for (size_t k = 0; k < d.size(); k++) pout->block_x0[nb][k] = d[k];
        nb++;
    }
    pout->n_blocks = nb;
    return VILF_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Ceres-layout hooks
extern "C" int vilo_eval_projection(const vilf_options *o, const double *const *p, const double pi[3], const double pj[3], double *r, double **J) {
    ProjectionFactor f(V3(pi), V3(pj), o->focal_length / 1.5);
    f.Evaluate(p, r, J);
    return VILF_OK;
}
extern "C" int vilo_eval_projection_td(const vilf_options *o, const double *const *p, const double pi[3], const double pj[3], const double vi[2],
                                       const double vj[2], double tdi, double tdj, double rowi, double rowj, double *r, double **J) {
    ProjectionTdFactor f(V3(pi), V3(pj), vi, vj, tdi, tdj, rowi, rowj, o->focal_length / 1.5, o->TR, o->ROW);
    f.Evaluate(p, r, J);
    return VILF_OK;
}
extern "C" int vilo_eval_imu(const vilf_options *o, const double *const *p, const vilf_imu_preint *pre, double *r, double **J) {
    IMUFactor f(pre, V3(o->G));
    f.Evaluate(p, r, J);
    return VILF_OK;
}
extern "C" int vilo_eval_imu_raw(const vilf_options *o, const double *const *p, const vilf_imu_preint *pre, double *r, double **J) {
    IMUFactor f(pre, V3(o->G));
    f.whiten = false;
    f.Evaluate(p, r, J);
    return VILF_OK;
}
extern "C" int vilo_eval_lidar_between(const vilf_options *o, const double *const *p, const vilf_lidar_constraint *c, double *r, double **J) {
    LidarFactor f(c, o);
    f.Evaluate(p, r, J);
    return VILF_OK;
}
extern "C" int vilo_eval_prior(const vilf_prior *prior, const double *const *p, double *r, double **J) {
    MarginalizationFactor f(prior);
    f.Evaluate(p, r, J);
    return VILF_OK;
}
extern "C" int vilo_eval_edge(const double pose[7], const double c[3], const double a[3], const double b[3], double r[3], double *J) {
    EdgeCostFunction f{V3(c), V3(a), V3(b)};
    const double *pp[1] = {pose}; double *jj[1] = {J};
    f.Evaluate(pp, r, J ? jj : nullptr);
    return VILF_OK;
}
extern "C" int vilo_eval_surf(const double pose[7], const double c[3], const double n[3], double d, double r[1], double *J) {
    SurfCostFunction f{V3(c), V3(n), d};
    const double *pp[1] = {pose}; double *jj[1] = {J};
    f.Evaluate(pp, r, J ? jj : nullptr);
    return VILF_OK;
}
extern "C" int vilo_pose_plus(const double x[7], const double d[6], double xp[7]) { pose_plus(x, d, xp); return VILF_OK; }
extern "C" int vilo_se3_plus(const double x[7], const double d[6], double xp[7]) { se3_plus(x, d, xp); return VILF_OK; }
extern "C" int vilo_imu_sqrt_info(const vilf_imu_preint *pre, double out[225]) { IMUFactor::sqrt_info(pre, out); return VILF_OK; }
extern "C" int vilo_imu_preintegrate(const vilf_imu_noise *nz, const double a0[3], const double g0[3], const double ba[3], const double bg[3],
                                     int n, const double *dt, const double *acc, const double *gyr, vilf_imu_preint *out) {
    imu_preintegrate(nz, a0, g0, ba, bg, n, dt, acc, gyr, out);
    return VILF_OK;
}
extern "C" int vilo_corrector(int loss, double a, int nres, double *residuals, int ncols, double *jacobian, double rho_out[3]) {
    std::unique_ptr<LossFunction> l;
    if (loss == 0) l.reset(new CauchyLoss(a)); else l.reset(new HuberLoss(a));
    double *jj[1] = {jacobian};
    int sizes[1] = {ncols};
    apply_corrector(l.get(), nres, residuals, 1, sizes, jacobian ? jj : nullptr, rho_out);
    return VILF_OK;
}
extern "C" int vilo_sym_eigen(int n, const double *A, double *w, double *V) {
    Mat a(n, n), v; std::vector<double> ww;
    std::memcpy(a.d.data(), A, sizeof(double) * n * n);
    sym_eigen(a, ww, v);
    std::memcpy(w, ww.data(), sizeof(double) * n);
    std::memcpy(V, v.d.data(), sizeof(double) * n * n);
    return VILF_OK;
}
extern "C" int vilo_quat_from_R(const double R[9], double q[4]) { fromR(M3::from(R)).to_xyzw(q); return VILF_OK; }
extern "C" int vilo_R2ypr(const double R[9], double ypr[3]) { V3 v = R2ypr(M3::from(R)); ypr[0] = v.x; ypr[1] = v.y; ypr[2] = v.z; return VILF_OK; }
extern "C" int vilo_ypr2R(const double ypr[3], double R[9]) { M3 m = ypr2R(V3(ypr)); std::memcpy(R, m.m, sizeof(m.m)); return VILF_OK; }
