/* ORACLE — TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle_vilf.so; the product path (vil_fusion_amd/, libvilfusion_hip.so) never does.
 * "parity unpinned": the reference ships no golden vectors and cannot be built here.
 *
 * The oracle exports the same POD structs as include/vilfusion.h under the vilo_ prefix so that tests feed
 * byte-identical inputs to the HIP path and to this CPU restatement. */
#ifndef ORACLE_API_H
#define ORACLE_API_H
#include "../include/vilfusion.h"
#ifdef __cplusplus
extern "C" {
#endif

void vilo_default_options(vilf_options *o);

/* Estimator::optimization() lines 689-860 (estimator.cpp): build, Solve (8 iterations, no time limit), double2vector. */
int vilo_window_solve(const vilf_options *o, const vilf_window_in *in, const vilf_prior *prior_or_null, vilf_window_out *out);
/* lines 863-1046: marginalization from the post-gauge state in `solved`. prior_out->valid = 0 if nothing was produced
 * (SECOND_NEW without Pose[WINDOW_SIZE-1] in the prior keeps the old prior: prior_out = *prior_in). */
int vilo_window_marginalize(const vilf_options *o, const vilf_window_in *in, const vilf_window_out *solved,
                            const vilf_prior *prior_or_null, vilf_prior *prior_out);
/* per-iteration trace of the last vilo_window_solve on this thread: rows of
 * [iteration cost cost_change gradient_max_norm step_norm relative_decrease radius valid successful] */
int vilo_last_trace(double *rows9, int capacity);

/* Ceres-layout hooks */
int vilo_eval_projection(const vilf_options *o, const double *const *parameters, const double pts_i[3], const double pts_j[3],
                         double *residuals, double **jacobians);
int vilo_eval_projection_td(const vilf_options *o, const double *const *parameters, const double pts_i[3], const double pts_j[3],
                            const double vel_i[2], const double vel_j[2], double td_i, double td_j, double row_i, double row_j,
                            double *residuals, double **jacobians);
int vilo_eval_imu(const vilf_options *o, const double *const *parameters, const vilf_imu_preint *pre, double *residuals, double **jacobians);
/* the same before the multiplication by sqrt_info (residual [15], Jacobians in the same layout): compared on its own, tightly (tests) */
int vilo_eval_imu_raw(const vilf_options *o, const double *const *parameters, const vilf_imu_preint *pre, double *residuals, double **jacobians);
int vilo_eval_lidar_between(const vilf_options *o, const double *const *parameters, const vilf_lidar_constraint *c,
                            double *residuals, double **jacobians);
int vilo_eval_prior(const vilf_prior *prior, const double *const *parameters, double *residuals, double **jacobians);
int vilo_eval_edge(const double pose_qt[7], const double curr_point[3], const double a[3], const double b[3], double residuals[3], double *jacobian);
int vilo_eval_surf(const double pose_qt[7], const double curr_point[3], const double n[3], double d, double residuals[1], double *jacobian);
int vilo_pose_plus(const double x[7], const double delta[6], double xp[7]);
int vilo_se3_plus(const double x[7], const double delta[6], double xp[7]);
int vilo_imu_sqrt_info(const vilf_imu_preint *pre, double out225[225]);
int vilo_imu_preintegrate(const vilf_imu_noise *noise, const double acc_0[3], const double gyr_0[3], const double ba[3], const double bg[3],
                          int n, const double *dt, const double *acc, const double *gyr, vilf_imu_preint *out);
/* VisualIMUAlignment (initial_aligment.cpp:199): n frames (R = c0_R_bk, T = c0_T_ck up to scale), n - 1 raw IMU intervals (interval k joins
 * frames k and k + 1; [n-1][max_samples] sample slots), first integrated at lin_ba / lin_bg. Out: delta_bg (added to every Bgs[i] by the
 * caller), the intervals re-integrated at (0, bgs0 + delta_bg), refined gravity g in the c0 frame, x = [v_0 .. v_{n-1} (body frames), 2 tangent
 * coefficients, s] (n_x = 3 n + 3; 3 n + 4 with [.., g, 100 s] when the first gate fails), ok = the reference's bool result. */
/* A.ldlt().solve(b) as restated in initial_alignment.cpp (n x n row-major, lower triangle read) */
int vilo_ldlt_solve(int n, const double *A, const double *b, double *x);
int vilo_visual_imu_alignment(const vilf_options *o, const vilf_imu_noise *noise, int n_frames, const double *frame_R, const double *frame_T,
                              const double *acc_0, const double *gyr_0, const double *lin_ba, const double *lin_bg, const int *n_samples,
                              int max_samples, const double *dt, const double *acc, const double *gyr, const double bgs0[3],
                              double delta_bg[3], double g[3], double *x, int *n_x, vilf_imu_preint *pre_out, int *ok);
/* global_fusion pose graph (poseGraphOptimization.cpp): batch Gauss-Newton over PriorFactor + BetweenFactor<Pose3> (see pose_graph.cpp) */
int vilo_posegraph_optimize(int n_nodes, double *poses_qt, const double prior_sigma[6], int n_edges, const vilf_pg_edge *edges, int max_iterations, double tol,
                            int *iterations_out, double *final_cost);
/* one whitened (robust: re-weighted) BetweenFactor: e[6], A = de/d(delta_i), B = de/d(delta_j) (row-major 6x6), cost = rho / 2 */
int vilo_pg_between(const double pi_qt[7], const double pj_qt[7], const double meas_qt[7], const double sigma[6], int robust, double e[6], double A36[36], double B36[36], double *cost);
int vilo_pg_retract(const double p_qt[7], const double delta[6], double out_qt[7]);   /* Pose3::retract = p * Expmap(delta) */
/* robust corrector on one residual block: loss 0 = Cauchy(a), 1 = Huber(a) */
int vilo_corrector(int loss, double a, int nres, double *residuals, int ncols, double *jacobian, double rho_out[3]);
/* small linear algebra used by the path (for numpy cross-checks) */
int vilo_sym_eigen(int n, const double *A, double *w, double *V);
int vilo_quat_from_R(const double R[9], double q_xyzw[4]);
int vilo_R2ypr(const double R[9], double ypr[3]);
int vilo_ypr2R(const double ypr[3], double R[9]);

/* scan-to-map (EstimationMapping) — stateful CPU mirror */
typedef struct vilo_s2m vilo_s2m;
vilo_s2m *vilo_s2m_create(const vilf_options *o);
void vilo_s2m_destroy(vilo_s2m *s);
vilo_s2m *vilo_s2m_clone(const vilo_s2m *s);
int vilo_s2m_init(vilo_s2m *s, const float *edge_xyzi, int n_edge, const float *surf_xyzi, int n_surf);
int vilo_s2m_step(vilo_s2m *s, const float *edge_xyzi, int n_edge, const float *surf_xyzi, int n_surf, vilf_scan2map_result *res);
int vilo_s2m_get_map(vilo_s2m *s, int which, float *xyzi_out, int capacity, int *n_out);
int vilo_s2m_set_pose(vilo_s2m *s, const double pose_qt[7], const double pose_last_qt[7]);
/* brute-force exact 5-NN (stands in for pcl::KdTreeFLANN::nearestKSearch k=5): squared distances, ascending */
int vilo_knn5_bruteforce(const float *map_xyzi, int n_map, const float *query_xyz3, int n_query, int *idx5, float *sqdist5);
/* PCL VoxelGrid (centroid per leaf) restatement */
int vilo_voxel_grid(const float *xyzi, int n, float leaf, float *out_xyzi, int capacity, int *n_out);
/* featureExtraction::extractFeature (featureExtraction.hpp:54-232): raw scan -> edge / surf feature clouds */
/* getFeatureDepth (feature_tracker_node.cpp:54-163): LiDAR depth of the visual features (camera-frame cloud, features with z = 1); -1 = none */
int vilo_feature_depth(const float *cloud_xyzi, int n, const float *feat_xyz, int m, float *depth_out);
int vilo_extract_features(const float *xyzi, int n, int n_scans, double min_range, double max_range, double edge_threshold,
                          float *edge_out, int cap_edge, int *n_edge, float *surf_out, int cap_surf, int *n_surf);
/* association products for one query set at a given pose (EdgeCostFactor / SurfCostFactor :117-232) */
int vilo_s2m_associate_edge(const float *map_xyzi, int n_map, const float *pts_xyzi, int n_pts, const double pose_qt[7],
                            unsigned char *valid, double *point_a /*[n][3]*/, double *point_b /*[n][3]*/);
int vilo_s2m_associate_surf(const float *map_xyzi, int n_map, const float *pts_xyzi, int n_pts, const double pose_qt[7],
                            unsigned char *valid, double *norm /*[n][3]*/, double *d /*[n]*/);

#ifdef __cplusplus
}
#endif
#endif
