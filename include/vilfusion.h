/*
 * vilfusion.h — C ABI of the MI355X-native sliding-window back-end for VIL_Fusion.
 *
 * Drop-in boundary for ONE hot path of RichExplor/VIL_Fusion (reference paths are relative to
 * src/visual_inertial_lidar/):
 *   - Estimator::optimization()            vins_estimator/estimator.cpp:689-1050
 *   - MarginalizationInfo                  vins_estimator/factor/marginalization_factor.{h,cpp}
 *   - EstimationMapping::optimation_processing   feature_tracker/include/EstimationMapping.hpp:235-296
 *
 * Plain C: POD structs, caller-owned host buffers, library-owned device state, int status codes,
 * never throws / never aborts (reference error convention: Evaluate() always returns true, solver
 * failures are not checked — estimator.cpp:852-855).
 *
 * Memory-layout conventions are the reference's:
 *   pose block   [tx ty tz qx qy qz qw]            (estimator.cpp:509-516)
 *   speed-bias   [vx vy vz bax bay baz bgx bgy bgz] (estimator.cpp:518-528)
 *   quaternions in structs: x y z w   (Eigen coeffs order)
 *   matrices: row-major
 *   scan-to-map pose [qx qy qz qw tx ty tz]        (EstimationMapping.hpp:383-385)
 *
 * Parameter-block ids replace the reference's address keys (marginalization_factor.h:59-62):
 *   Pose[i] -> i,  SpeedBias[i] -> NF+i,  Ex_Pose -> 2NF,  Td -> 2NF+1,  Feature[k] -> 2NF+2+k
 * with NF = window_size+1 frames.
 */
#ifndef VILFUSION_H
#define VILFUSION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VILF_MAX_FRAMES 11         /* WINDOW_SIZE + 1, vins_estimator/parameters.h:24 */
#define VILF_MAX_FEATURES 1000     /* NUM_OF_F, parameters.h:26 */
#define VILF_PRIOR_MAX_DIM 160     /* 10 poses*6 + 10 speedbias*9 + ex 6 + td 1 = 157 */
#define VILF_PRIOR_MAX_BLOCKS 24

/* status codes */
#define VILF_OK 0
#define VILF_ERR_INVALID_ARGUMENT (-1)
#define VILF_ERR_DEVICE (-2)
#define VILF_ERR_UNSUPPORTED (-3)
#define VILF_ERR_NO_GPU (-4)
#define VILF_SOLVER_ABNORMAL 1     /* >0: solver terminated abnormally (summary tells why) */

/* marginalization_flag, estimator.h:63-67 */
#define VILF_MARGIN_OLD 0
#define VILF_MARGIN_SECOND_NEW 1

/* termination types (mirrors ceres::TerminationType semantics for the configured minimizer) */
#define VILF_TERM_NO_CONVERGENCE 0    /* max_num_iterations reached */
#define VILF_TERM_CONVERGENCE_FUNCTION 1
#define VILF_TERM_CONVERGENCE_PARAMETER 2
#define VILF_TERM_CONVERGENCE_GRADIENT 3
#define VILF_TERM_FAILURE 4           /* too many invalid steps / min radius */

typedef struct vilf_handle vilf_handle;

/* Options ≙ the globals read by readParameters() (vins_estimator/parameters.cpp:45-155) plus the
 * solver options set in estimator.cpp:838-850. */
typedef struct vilf_options {
    int window_size;          /* WINDOW_SIZE = 10 */
    int max_num_iterations;   /* NUM_ITERATIONS = 8 (kitti_config.yaml:74) */
    double max_solver_time;   /* SOLVER_TIME seconds (estimator.cpp:847-850); <= 0 disables the wall-clock limit (parity runs). When > 0 the
                               * solve tests the host clock at the top of every iteration, as Ceres does; windows whose marginalization_flag is
                               * VILF_MARGIN_OLD get 4/5 of it (the factor the reference applies). A stopped window reports VILF_TERM_NO_CONVERGENCE
                               * and keeps its last accepted state. The batched solve then waits for the stream once per iteration. */
    double focal_length;      /* FOCAL_LENGTH = 460; sqrt_info = focal/1.5 * I2 (estimator.cpp:17) */
    double cauchy_a;          /* CauchyLoss(1.0) (estimator.cpp:694) */
    double G[3];              /* global G (parameters.cpp:14,77) */
    int estimate_extrinsic;   /* ESTIMATE_EXTRINSIC */
    int estimate_td;          /* ESTIMATE_TD */
    int use_lidar_const;      /* #define USE_LIDAR_CONST (parameters.h:56) */
    double RIC[9], TIC[3];    /* imu^R_cam, imu^T_cam   (globals used by lidarFactor) */
    double RCL[9], TCL[3];    /* cam^R_lidar, cam^T_lidar */
    double TR, ROW;           /* rolling shutter read-out time, image rows */
    double init_depth;        /* INIT_DEPTH = 5.0 (parameters.cpp:132), feature_manager.cpp:205-208 */
    /* scan-to-map (velodyne_param_64.yaml:22-23, EstimationMapping.hpp:263,277,327) */
    double edge_leaf_size;    /* 0.4 */
    double surf_leaf_size;    /* 0.8 */
    double huber_a;           /* 0.1 */
    int s2m_outer_iterations; /* 2 */
    int s2m_max_iterations;   /* 4 */
    double s2m_crop_half;     /* 100.0 */
} vilf_options;

/* IntegrationBase state consumed by IMUFactor (factor/integration_base.h:188-207) */
typedef struct vilf_imu_preint {
    double sum_dt;
    double delta_p[3];
    double delta_q[4];        /* x y z w */
    double delta_v[3];
    double linearized_ba[3];
    double linearized_bg[3];
    double jacobian[225];     /* 15x15 row-major */
    double covariance[225];   /* 15x15 row-major */
} vilf_imu_preint;

/* lidarConstraintsBase (factor/lidarConstraint_base.h:24-25) */
typedef struct vilf_lidar_constraint {
    double q[4];              /* x y z w */
    double t[3];
} vilf_lidar_constraint;

/* One window ≙ the members optimization() reads (estimator.h:70-146). */
typedef struct vilf_window_in {
    int n_frames;                         /* options.window_size + 1. 11 = the reference's WINDOW_SIZE: batched LDS kernels; any other size: the general
                                             path (vilf_window_solve, or several at once: vilf_window_solve_group; no prior, no marginalization) —
                                             BASELINE configs[4].
                                             options.estimate_extrinsic / estimate_td (para_ex_pose / para_td become variables; obs_velocity,
                                             obs_cur_td, obs_row required for td): solved through the same general path at any window size, by
                                             vilf_window_solve and by vilf_batch_solve (all slots side by side as one group); an 11-frame window keeps its device
                                             prior, and vilf_window_marginalize() / vilf_batch_marginalize() carry Ex_Pose and Td as kept blocks
                                             of it (ProjectionTdFactor rows when estimate_td is set). */
    const double *para_pose;              /* [n_frames][7] */
    const double *para_speed_bias;        /* [n_frames][9] */
    double para_ex_pose[7];
    double para_td;
    int n_features;                       /* f_manager.getFeatureCount() */
    const double *para_feature;           /* [n_features] inverse depth (feature_manager.cpp:194-216) */
    const uint8_t *feature_const;         /* [n_features] lidar_depth_flag (estimator.cpp:780,789) */
    const int32_t *feature_start_frame;   /* [n_features] */
    const int32_t *feature_obs_offset;    /* [n_features+1] CSR into obs_*; obs k of a feature is in frame start+k */
    int n_obs;
    const double *obs_point;              /* [n_obs][3] feature_per_frame[k].point */
    const double *obs_velocity;           /* [n_obs][2] or NULL (td factor only) */
    const double *obs_cur_td;             /* [n_obs]    or NULL */
    const double *obs_row;                /* [n_obs] uv.y() or NULL */
    const vilf_imu_preint *imu;           /* [n_frames]; entry j = pre_integrations[j], j >= 1 used */
    const vilf_lidar_constraint *lidar;   /* [n_frames]; entry j = lidarConstraints[j], j >= 1 used; may be NULL if !use_lidar_const */
    int marginalization_flag;
    const double *gauge_R0;               /* optional Rs[0] (9, row-major) override for double2vector (failure_occur path); NULL: from para_pose[0] */
    const double *gauge_P0;               /* optional Ps[0] override; NULL: from para_pose[0] */
} vilf_window_in;

typedef struct vilf_summary {
    int num_iterations;        /* trust-region iterations executed (excluding iteration 0) */
    int num_successful_steps;
    int num_linear_solves;     /* dense Schur solves (rejected steps re-use the previous one) */
    int termination;           /* VILF_TERM_* */
    double initial_cost;
    double final_cost;
    double final_radius;
    double usec_solve;         /* wall time spent in the solve, microseconds */
} vilf_summary;

/* State after Solve + double2vector() (estimator.cpp:549-638). Caller-owned buffers. */
typedef struct vilf_window_out {
    double *para_pose;         /* [n_frames][7]   raw solver output (before gauge fix); may be NULL */
    double *para_speed_bias;   /* [n_frames][9]   may be NULL */
    double *para_feature;      /* [n_features]    may be NULL */
    double *Ps;                /* [n_frames][3] */
    double *Rs;                /* [n_frames][9] */
    double *Vs;                /* [n_frames][3] */
    double *Bas;               /* [n_frames][3] */
    double *Bgs;               /* [n_frames][3] */
    double tic[3], ric[9];
    double td;
    vilf_summary summary;
} vilf_window_out;

/* Marginalization prior ≙ MarginalizationInfo after marginalize()+getParameterBlocks()
 * (marginalization_factor.h:58-70). Block ids are already shifted (estimator.cpp:960-971). */
typedef struct vilf_prior {
    int valid;
    int n;                                       /* rows = kept local dimension */
    int m;                                       /* marginalized local dimension (informative) */
    int n_blocks;
    int block_id[VILF_PRIOR_MAX_BLOCKS];
    int block_size[VILF_PRIOR_MAX_BLOCKS];       /* global size (7 / 9 / 1) */
    int block_idx[VILF_PRIOR_MAX_BLOCKS];        /* local column offset = keep_block_idx - m */
    double block_x0[VILF_PRIOR_MAX_BLOCKS][9];   /* keep_block_data */
    double linearized_residuals[VILF_PRIOR_MAX_DIM];
    double linearized_jacobians[VILF_PRIOR_MAX_DIM * VILF_PRIOR_MAX_DIM]; /* n x n row-major, leading dim n */
} vilf_prior;

/* ---- lifecycle ------------------------------------------------------------------------- */
void vilf_default_options(vilf_options *opts);                 /* KITTI config values */
/* hip_stream: a hipStream_t (as void*) all work is enqueued on, or NULL for the library's own stream. */
int vilf_create(const vilf_options *opts, int device, void *hip_stream, vilf_handle **out);
void vilf_destroy(vilf_handle *h);
int vilf_reset(vilf_handle *h);                                /* drop priors ≙ clearState(), estimator.cpp:72-77 */
const char *vilf_last_error(const vilf_handle *h);
const char *vilf_version(void);

/* ---- single-window drop-in (≙ Estimator::optimization()) ------------------------------- */
/* estimator.cpp:689-860: build problem, Solve, double2vector. Uses/keeps the prior of slot 0. */
int vilf_window_solve(vilf_handle *h, const vilf_window_in *in, vilf_window_out *out);
/* n independent windows of sizes other than 11 frames (the general path), solved side by side in ONE chain of launches: a window's chain is ~32 small dependent
 * launches per iteration, so a group fills the chip where a single window (or one handle / stream per window) cannot. in / out: arrays of n; every out[i] needs
 * Ps / Rs / Vs / Bas / Bgs. Same results as n calls of vilf_window_solve, bit for bit (no reduction on this path depends on the order in which workgroups finish; with
 * options.estimate_extrinsic / estimate_td: to the rounding of the remaining atomics). 11-frame windows:
 * VILF_ERR_UNSUPPORTED (use vilf_batch_*). Returns VILF_SOLVER_ABNORMAL if any window terminated abnormally (its summary tells). */
int vilf_window_solve_group(vilf_handle *h, int n, const vilf_window_in *in, vilf_window_out *out);
/* estimator.cpp:863-1046: marginalization of the just-solved window (slot 0); new prior stays on device. */
int vilf_window_marginalize(vilf_handle *h);

/* ---- batched windows (independent window snapshots resident in HBM) -------------------- */
/* Pack + upload n windows into slots 0..n-1 (inputs stay resident until the next upload). */
int vilf_batch_upload(vilf_handle *h, int n_windows, const vilf_window_in *wins);
/* Solve every resident window from its resident state (all kernels enqueued on the handle's stream). sync!=0 waits; with sync == 0 the usec_solve of the summaries
 * is filled by the next call that waits for the stream (vilf_batch_summaries / vilf_batch_download_states). Returns VILF_OK also when single windows terminated
 * abnormally: their summaries tell (termination = VILF_TERM_FAILURE). options.estimate_extrinsic / estimate_td: all slots run through the general path as ONE group
 * of launches (vilf_window_solve_group's machinery, each slot with its device-resident prior); that path reads its results back, so the call is always synchronous.
 * options.max_solver_time > 0: ONE host clock for the whole batch, started by this call (the windows of a batch run in lockstep), and a stream wait per iteration. */
int vilf_batch_solve(vilf_handle *h, int sync);
/* Re-arm the resident windows with their uploaded initial state (bench loop: repeated identical steps). */
int vilf_batch_rewind(vilf_handle *h);
int vilf_batch_marginalize(vilf_handle *h, int sync);
int vilf_batch_download(vilf_handle *h, int first, int n_windows, vilf_window_out *outs);
int vilf_batch_summaries(vilf_handle *h, int first, int n_windows, vilf_summary *sums);
/* the estimator's outputs only (double2vector(): Ps [n][33], Rs [n][99] row-major, Vs, Bas, Bgs [n][33]) and the summaries into contiguous caller arrays;
 * any pointer may be NULL. The per-frame caller's download: the parameter arrays stay on the device for the marginalization. */
int vilf_batch_download_states(vilf_handle *h, int first, int n_windows, double *Ps, double *Rs, double *Vs, double *Bas, double *Bgs, vilf_summary *sums);
int vilf_synchronize(vilf_handle *h);
/* Work enqueued on h's stream after this call starts once everything enqueued on other's stream so far has finished — a dependency on the device
 * (hipEventRecord + hipStreamWaitEvent), the host does not wait. The reference runs the LiDAR node and the estimator node side by side and hands
 * lidarConstraints across (feature_tracker_node.cpp:384,524 -> estimator.cpp:689-860); with one handle per stage this orders a frame's two stages
 * without a host round trip between them. Both handles must be on the same device. */
int vilf_wait_for(vilf_handle *h, vilf_handle *other);
/* on != 0: vilf_batch_upload returns as soon as its copies and set-up launches are enqueued instead of waiting for them (the caller's window structs are
 * consumed while the call packs them, as before; the library's pinned staging is protected by a wait at the start of the handle's NEXT upload). Everything
 * that follows on the handle is ordered behind the copies on its stream. For streams of batches over two handles: while one handle's copies and solve
 * (vilf_batch_solve with sync == 0) are on the device, the host packs the other handle's next batch — bench.py: pcie_inclusive.stream_of_batches. Default: off. */
int vilf_set_async_upload(vilf_handle *h, int on);
/* per-kernel timing by HIP events on the handle's stream (kind 0 linearize incl. the trust-region step, 1 reduce+solve, 2 the step-only launch that ends a solve, 3 other).
 * With sync == 0 calls the spans stay pending and are read by the next call that waits for the stream (a sync call, vilf_batch_summaries, vilf_get_profile*). */
int vilf_set_profiling(vilf_handle *h, int on);
int vilf_get_profile(vilf_handle *h, double ms_out[4], long launches_out[4]);
/* same switch, scan-to-map launches by group: 0 voxel grid of the scan clouds, 1 radix sort, 2 neighbour index, 3 associate (5-NN + fits),
 * 4 LM solve, 5 sub-map maintenance, 6 other, 7 fused local-map update (crop + merge + voxel grid + directory) */
int vilf_get_profile_scan2map(vilf_handle *h, double ms_out[8], long launches_out[8]);
/* same switch, marginalization kernels: 0 prepare (factor re-evaluation at the linearisation point), 1 Schur complement, 2 eigen + prior, 3 prior H/g */
int vilf_get_profile_marginalize(vilf_handle *h, double ms_out[4], long launches_out[4]);
/* Which form the last vilf_batch_marginalize() / vilf_window_marginalize() took per window. marginalization_factor.cpp:267-291 eigen-decomposes the dropped
 * block Amm and the kept block with a 1e-8 truncation; where that truncation provably removes nothing (positive Cholesky pivots and trace(A^-1) < 1e8, i.e.
 * lambda_min > 1e-8) the library uses Cholesky factors instead — same J0^T J0, J0^T r0, |r0|^2 — and falls back to the eigen-decompositions otherwise.
 * counts[0] windows that produced a new prior; [1] of those: Amm by the arrow Cholesky; [2] of those: kept block by Cholesky (J0 = L^T, r0 = L^-1 b);
 * [3] windows whose prior was left as it was. */
int vilf_batch_marginalize_stats(vilf_handle *h, int counts[4]);
/* the general (window_size != 10) path of vilf_window_solve: factor scatter (linearisations), Schur SYRK, Cholesky, unused */
int vilf_get_profile_large_window(vilf_handle *h, double ms_out[4], long launches_out[4]);
/* newest-frame pose per resident window: [stamp x y z qx qy qz qw] (8 doubles each) into a DEVICE buffer
 * (feeds the RCCL gather for global_fusion, poseGraphOptimization.cpp:116-121). Enqueued on the handle's stream; the call does NOT wait for it (stamps_host is
 * copied to a pinned staging buffer of the handle before the call returns): order a consumer behind it by using the same stream (vilf_gather_poses takes one) or
 * vilf_synchronize(). */
int vilf_batch_newest_poses_device(vilf_handle *h, const double *stamps_host, void *device_out8);

/* ---- multi-GPU: the pose gather over RCCL (SURVEY.md §8(b),(e)) --------------------------- */
/* Independent windows / sequence segments shard over the GPUs of a node, one process and one vilf_handle per GPU, no intra-solve
 * communication. The only exchange is an all-gather of the newest-frame pose rows above — what global_fusion subscribes to as
 * nav_msgs/Odometry (src/global_fusion/poseGraphOptimization.cpp:116-121: position, quaternion, stamp). The communicator is RCCL
 * (ncclCommInitRank / ncclAllGather over xGMI); the launcher distributes rank 0's id (MPI, a file, a ROS parameter ...). */
#define VILF_COMM_ID_BYTES 128
typedef struct vilf_comm vilf_comm;
int vilf_comm_unique_id(unsigned char id[VILF_COMM_ID_BYTES]);                       /* rank 0: ncclGetUniqueId */
int vilf_comm_create(const unsigned char id[VILF_COMM_ID_BYTES], int world_size, int rank, int device, vilf_comm **out);
int vilf_comm_destroy(vilf_comm *comm);
/* local_dev8: n_local rows of 8 doubles on this rank's device; out_dev8: world_size * n_local rows, rank-major (= the global unit order
 * under contiguous sharding). Enqueued on hip_stream (NULL: default stream); the caller synchronises. Every rank passes the same n_local. */
int vilf_gather_poses(vilf_comm *comm, void *hip_stream, const double *local_dev8, int n_local, double *out_dev8);
/* The same gather enqueued on the stream h works on — whichever that is: the one handed to vilf_create, or the library's own. The library's own stream is
 * NON-BLOCKING: it does not synchronise with the default (NULL) stream, so a gather enqueued on NULL is NOT ordered behind vilf_batch_newest_poses_device /
 * an asynchronous vilf_batch_solve of such a handle. This entry point is: it runs behind everything enqueued on h so far. local_dev8 / out_dev8 must be
 * ready (allocated and, for local_dev8, written by work on h's stream or synchronised) when the call is made. */
int vilf_gather_poses_handle(vilf_comm *comm, vilf_handle *h, const double *local_dev8, int n_local, double *out_dev8);
/* ranks of the communicator as RCCL reports them (ncclCommCount) and this process's rank in it (ncclCommUserRank) */
int vilf_comm_ranks(vilf_comm *comm, int *world_size_out, int *rank_out);
/* the hipStream_t (as void*) h enqueues its work on */
int vilf_get_stream(vilf_handle *h, void **hip_stream_out);
const char *vilf_comm_last_error(void);

/* ---- prior import / export (tests, snapshots) ------------------------------------------ */
int vilf_prior_export(vilf_handle *h, int slot, vilf_prior *out);
int vilf_prior_import(vilf_handle *h, int slot, const vilf_prior *prior);

/* ---- fine-grained hooks in the reference's Ceres layout -------------------------------- */
/* bool Evaluate(double const *const *parameters, double *residuals, double **jacobians): row-major
 * jacobians in GLOBAL size (pose: 7 columns, last = 0); jacobians / jacobians[i] may be NULL. All run on
 * the device through the same device functions the solve kernels use. */
int vilf_eval_projection(vilf_handle *h, const double *const *parameters, const double pts_i[3],
                         const double pts_j[3], double *residuals, double **jacobians);   /* projection_factor.cpp:21 */
/* ProjectionTdFactor (5 blocks: Pose_i, Pose_j, Ex_Pose, inverse depth, td); row_* = uv.y of the two observations, TR / ROW from the options.
 * In the solve: options.estimate_td = 1 makes vilf_window_solve use this factor for every visual observation and td a variable
 * (estimator.cpp:713-717, 765-777); see vilf_window_solve. */
int vilf_eval_projection_td(vilf_handle *h, const double *const *parameters, const double pts_i[3], const double pts_j[3],
                            const double velocity_i[2], const double velocity_j[2], double td_i, double td_j, double row_i, double row_j,
                            double *residuals, double **jacobians);                          /* projection_td_factor.cpp:34 */
int vilf_eval_imu(vilf_handle *h, const double *const *parameters, const vilf_imu_preint *pre,
                  double *residuals, double **jacobians);                                   /* imu_factor.h:19 */
/* the two parts of that product on their own (test hook): residuals / jacobians BEFORE the multiplication by sqrt_info (imu_factor.h:60-62, 86-173 without the
 * sqrt_info * ... lines), same layouts, and sqrt_info = LLT(covariance^-1).matrixL()^T (:64) as the device computes it once per upload; any output may be NULL. */
int vilf_eval_imu_raw(vilf_handle *h, const double *const *parameters, const vilf_imu_preint *pre,
                      double *residuals, double **jacobians, double *sqrt_info225);
int vilf_eval_lidar_between(vilf_handle *h, const double *const *parameters,
                            const vilf_lidar_constraint *c, double *residuals, double **jacobians); /* lidar_factor.h:19 */
int vilf_eval_prior(vilf_handle *h, const vilf_prior *prior, const double *const *parameters,
                    double *residuals, double **jacobians);                                 /* marginalization_factor.cpp:333 */
int vilf_eval_edge(vilf_handle *h, const double pose_qt[7], const double curr_point[3], const double point_a[3],
                   const double point_b[3], double residuals[3], double *jacobian /*3x7 or NULL*/);   /* lidarFactor.hpp:21 */
int vilf_eval_surf(vilf_handle *h, const double pose_qt[7], const double curr_point[3], const double norm[3],
                   double negative_OA_dot_norm, double residuals[1], double *jacobian /*1x7 or NULL*/); /* lidarFactor.hpp:79 */
int vilf_pose_plus(vilf_handle *h, const double x[7], const double delta[6], double x_plus_delta[7]); /* pose_local_parameterization.cpp:3 */
int vilf_se3_plus(vilf_handle *h, const double x[7], const double delta[6], double x_plus_delta[7]);  /* EstimationMapping.hpp:34 */

/* ---- IMU pre-integration (host; ≙ IntegrationBase, integration_base.h:30-158) ---------- */
typedef struct vilf_imu_noise { double acc_n, gyr_n, acc_w, gyr_w; } vilf_imu_noise;
int vilf_imu_preintegrate(const vilf_imu_noise *noise, const double acc_0[3], const double gyr_0[3],
                          const double linearized_ba[3], const double linearized_bg[3], int n_samples,
                          const double *dt, const double *acc /*[n][3]*/, const double *gyr /*[n][3]*/,
                          vilf_imu_preint *out);

/* the same integration for n intervals at once on the device (one lane per interval): all inputs are [n][...] host arrays with
 * max_samples slots per interval (n_samples[i] of them used); out[n] */
int vilf_imu_preintegrate_batch(vilf_handle *h, int n, const vilf_imu_noise *noise, const double *acc_0 /*[n][3]*/, const double *gyr_0,
                                const double *linearized_ba, const double *linearized_bg, const int *n_samples, int max_samples,
                                const double *dt /*[n][max]*/, const double *acc /*[n][max][3]*/, const double *gyr, vilf_imu_preint *out);
/* ---- visual-inertial alignment before the first window solve (≙ VisualIMUAlignment, initial/initial_aligment.cpp:199-207; device) ----
 * n_frames frames of all_image_frame in time order: frame_R[k] = c0_R_bk (ImageFrame::R), frame_T[k] = c0_T_ck up to scale (ImageFrame::T),
 * and the n_frames - 1 raw IMU intervals between them (interval k joins frames k and k + 1 = frame k + 1's pre_integration; arrays as in
 * vilf_imu_preintegrate_batch), first integrated at lin_ba / lin_bg. TIC and G come from the handle's options.
 *   solveGyroscopeBias (:3)  -> delta_bg (the caller adds it to every Bgs[i]); every interval is re-integrated at (0, bgs0 + delta_bg) -> pre_out[n-1]
 *   LinearAlignment (:125) + RefineGravity (:55) -> g (c0 frame), x = [v_0 .. v_{n-1} in the body frames, 2 tangent coefficients, s], *n_x = 3 n + 3
 *   (3 n + 4 = [v.., g, 100 s] when the |g| / s gate at :184 fails), *ok = the reference's bool result. */
int vilf_visual_imu_alignment(vilf_handle *h, int n_frames, const double *frame_R /*[n][9]*/, const double *frame_T /*[n][3]*/,
                              const vilf_imu_noise *noise, const double *acc_0 /*[n-1][3]*/, const double *gyr_0, const double *lin_ba, const double *lin_bg,
                              const int *n_samples /*[n-1]*/, int max_samples, const double *dt /*[n-1][max]*/, const double *acc /*[n-1][max][3]*/, const double *gyr,
                              const double bgs0[3], double delta_bg[3], double g[3], double *x /*[3 n + 4]*/, int *n_x, vilf_imu_preint *pre_out /*[n-1] or NULL*/,
                              int *ok);

/* ---- pose-graph back-end of global_fusion (≙ poseGraphOptimization.cpp: the gtsam graph + isam update, :349-374, :560-587, :433-436; device) ----
 * Nodes = key-frame poses [qx qy qz qw tx ty tz] (gtsam::Pose3), node 0 carries the PriorFactor (its pose as handed in, sigma = prior_sigma).
 * Edges = BetweenFactor<Pose3>(i, j, measured = T_i^-1 T_j) with a Diagonal noise model given as sigmas in gtsam's tangent order
 * [rot x y z, trans x y z] (the reference: odometry variances 1e-6 / 1e-4, loop variances 0.5) and robust = 1 for
 * noiseModel::Robust(Cauchy(1), ...) (the ICP loop edges). Every consecutive pair (k, k + 1) needs at least one edge (the odometry
 * chain); all other edges are loop closures. ISAM2's incremental Gauss-Newton is run as batch Gauss-Newton to convergence:
 * at most max_iterations steps, stop when max |delta| < tol. poses_qt is updated in place. */
typedef struct vilf_pg_edge {
    int i, j;
    double q[4], t[3];        /* measured relative pose, x y z w */
    double sigma[6];
    int robust;
    int pad_;
} vilf_pg_edge;
/* Limit: at most 2048 loop edges (edges between non-consecutive key frames): their 6 L x 6 L capacitance block is solved by the library's own dense Cholesky, whose
 * back substitution keeps the solution in LDS. More loop edges: VILF_ERR_UNSUPPORTED, before any work is enqueued. */
int vilf_posegraph_optimize(vilf_handle *h, int n_nodes, double *poses_qt /*[n][7] in/out*/, const double prior_sigma[6], int n_edges, const vilf_pg_edge *edges,
                            int max_iterations, double tol, int *iterations_out, double *final_cost);

/* ---- scan-to-map (≙ EstimationMapping) -------------------------------------------------- */
/* points are float xyzi (pcl::PointXYZI without padding): [n][4] */
int vilf_scan2map_init(vilf_handle *h, const float *edge_xyzi, int n_edge, const float *surf_xyzi, int n_surf);   /* localMapInited, :105 */
/* optimation_processing(:235): in/out pose_qt = parameter_opti [qx qy qz qw tx ty tz] is kept in the handle
 * (globalOdom / globalOdom_last); returns the new global pose and the frame-to-frame relative pose T_ij. */
typedef struct vilf_scan2map_result {
    double pose_qt[7];          /* globalOdom after the step */
    double rel_q[4], rel_t[3];  /* globalOdom_last^-1 * globalOdom: the /Odometry message (feature_tracker_node.cpp:400-415) */
    int n_edge_ds, n_surf_ds;   /* after voxel down-sampling */
    int n_edge_factors[2];      /* accepted edge factors in association pass 0/1 */
    int n_surf_factors[2];
    int iterations[2];          /* solver iterations per pass */
    double final_cost[2];
    int map_edge_size, map_surf_size;
} vilf_scan2map_result;
int vilf_scan2map_step(vilf_handle *h, const float *edge_xyzi, int n_edge, const float *surf_xyzi, int n_surf,
                       vilf_scan2map_result *res);
int vilf_scan2map_get_map(vilf_handle *h, int which /*0 edge, 1 surf*/, float *xyzi_out, int capacity, int *n_out);
int vilf_scan2map_set_pose(vilf_handle *h, const double pose_qt[7], const double pose_last_qt[7]);

/* ---- batched scan-to-map: n_streams independent EstimationMapping objects (one per LiDAR stream / replayed segment)
 * stepped together with no host round trip. Capacities are per stream and fixed (a step that would overflow a local map
 * reports VILF_ERR_UNSUPPORTED through vilf_scan2map_batch_results). Stream i's result equals what a single-stream
 * handle fed with the same clouds returns. */
int vilf_scan2map_batch_create(vilf_handle *h, int n_streams, int cap_scan_edge, int cap_scan_surf, int cap_map_edge, int cap_map_surf);
/* localMapInited (:105) of one stream: its local map := the clouds; pose_qt (NULL = identity) -> globalOdom,
 * pose_last_qt (NULL = pose_qt) -> globalOdom_last (the constant-velocity prediction of the first step, :238-243) */
int vilf_scan2map_batch_init(vilf_handle *h, int stream, const float *edge_xyzi, int n_edge, const float *surf_xyzi, int n_surf,
                             const double *pose_qt, const double *pose_last_qt);
/* the scan the next vilf_scan2map_batch_step consumes for this stream (stays resident in HBM until replaced) */
int vilf_scan2map_batch_set_scan(vilf_handle *h, int stream, const float *edge_xyzi, int n_edge, const float *surf_xyzi, int n_surf);
/* stream dst := stream src (local maps, poses, resident scan) by device copies: replicas of a few distinct streams without one upload per stream */
int vilf_scan2map_batch_copy_stream(vilf_handle *h, int src, int dst);
int vilf_scan2map_batch_step(vilf_handle *h, int sync);                 /* optimation_processing (:235) for every stream */
int vilf_scan2map_batch_snapshot(vilf_handle *h);                       /* remember maps + poses ... */
int vilf_scan2map_batch_rewind(vilf_handle *h);                         /* ... and restore them (bench loop: repeated identical steps) */
int vilf_scan2map_batch_results(vilf_handle *h, int first, int n, vilf_scan2map_result *res);
int vilf_scan2map_batch_get_map(vilf_handle *h, int stream, int which, float *xyzi_out, int capacity, int *n_out);

/* ---- LiDAR feature extraction (≙ featureExtraction::extractFeature, feature_tracker/include/featureExtraction.hpp:54-232) ----
 * raw scan (xyzi, firing order) -> edge / surf feature clouds, the inputs of vilf_scan2map_*: ring assignment from the vertical
 * angle (n_scans 16 / 32 / 64), per-ring 10-neighbour curvature, six sectors per ring, <= 20 edge picks per sector with +-5
 * neighbour suppression, the remaining points as surf. Outputs are truncated to the capacities; the counts are always complete. */
int vilf_lidar_extract_features(vilf_handle *h, const float *xyzi, int n_points, int n_scans, double min_range, double max_range,
                                double edge_threshold, float *edge_xyzi_out, int cap_edge, int *n_edge,
                                float *surf_xyzi_out, int cap_surf, int *n_surf);
/* LiDAR depth of the tracked visual features ≙ getFeatureDepth (feature_tracker/feature_tracker_node.cpp:54-163): depth cloud in the
 * camera frame (xyzi), features as normalised image points (x, y, z = 1); depth_out[i] = depth of feature i (what the estimator
 * receives as point(7), estimator_node.cpp) or -1 when the 3 nearest returns do not support one. */
int vilf_feature_depth(vilf_handle *h, const float *depth_cloud_xyzi, int n_points, const float *features_xyz, int n_features, float *depth_out);
#ifdef __cplusplus
}
#endif
#endif /* VILFUSION_H */
