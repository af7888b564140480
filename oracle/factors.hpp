// ORACLE — TEST INFRASTRUCTURE ONLY. "parity unpinned" vs Ceres (see omath.hpp).
// factors.hpp — CPU restatement of every cost function on the hot path, in the reference's Ceres layout:
//   bool Evaluate(double const *const *parameters, double *residuals, double **jacobians)
// row-major jacobians in GLOBAL parameter size (pose blocks: 7 columns, 7th = 0).
#pragma once
#include "omath.hpp"
#include "../include/vilfusion.h"
#include <vector>
#include <memory>

namespace ora {

struct CostFunction {
    int num_residuals = 0;
    std::vector<int> block_sizes;
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
};

// ceres::LossFunction::Evaluate(s, rho[3])
struct LossFunction {
    virtual ~LossFunction() {}
    virtual void Evaluate(double s, double rho[3]) const = 0;
};
struct CauchyLoss : LossFunction {  // ceres CauchyLoss(a): rho(s) = b log(1 + s/b), b = a^2
    double b, c;
    explicit CauchyLoss(double a) : b(a * a), c(1.0 / (a * a)) {}
    void Evaluate(double s, double rho[3]) const override;
};
struct HuberLoss : LossFunction {   // ceres HuberLoss(a)
    double a, b;
    explicit HuberLoss(double a_) : a(a_), b(a_ * a_) {}
    void Evaluate(double s, double rho[3]) const override;
};

// Robust-loss corrector: marginalization_factor.cpp:37-68 (identical to ceres::internal::Corrector).
// Scales residuals[nres] and the row-major jacobians in place.
void apply_corrector(const LossFunction *loss, int nres, double *residuals, int nblocks, const int *block_sizes,
                     double **jacobians, double *rho_out /*[3] or null*/);

// projection_factor.cpp:21-121
struct ProjectionFactor : CostFunction {
    V3 pts_i, pts_j;
    double sqrt_info;  // focal/1.5 (estimator.cpp:17), times I2
    ProjectionFactor(V3 pi, V3 pj, double sqrt_info_);
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override;
};
// projection_td_factor.cpp:6-141
struct ProjectionTdFactor : CostFunction {
    V3 pts_i, pts_j, velocity_i, velocity_j;
    double td_i, td_j, row_i, row_j, sqrt_info, TR, ROW;
    ProjectionTdFactor(V3 pi, V3 pj, const double vel_i[2], const double vel_j[2], double td_i_, double td_j_, double row_i_,
                       double row_j_, double sqrt_info_, double TR_, double ROW_);
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override;
};
// imu_factor.h:19-179 + integration_base.h:160-186
struct IMUFactor : CostFunction {
    const vilf_imu_preint *pre;
    V3 G;
    bool whiten = true;      // false (tests only): residual and Jacobians before the multiplication by sqrt_info — the part of imu_factor.h:60-178 that does not depend on how sqrt_info is computed
    IMUFactor(const vilf_imu_preint *p, V3 G_);
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override;
    // sqrt_info = LLT(covariance^-1).matrixL().transpose()  (imu_factor.h:64)
    static void sqrt_info(const vilf_imu_preint *p, double out[225]);
};
// lidar_factor.h:19-78
struct LidarFactor : CostFunction {
    Q4 lidar_q; V3 lidar_t;
    M3 RIC, RCL; V3 TIC, TCL;
    LidarFactor(const vilf_lidar_constraint *c, const vilf_options *o);
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override;
};
// marginalization_factor.cpp:321-381
struct MarginalizationFactor : CostFunction {
    const vilf_prior *prior;
    explicit MarginalizationFactor(const vilf_prior *p);
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override;
};
// feature_tracker/include/lidarFactor.hpp:6-52
struct EdgeCostFunction : CostFunction {
    V3 curr_point, point_a, point_b;
    EdgeCostFunction(V3 c, V3 a, V3 b);
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override;
};
// feature_tracker/include/lidarFactor.hpp:64-102
struct SurfCostFunction : CostFunction {
    V3 curr_point, nrm; double negative_OA_dot_norm;
    SurfCostFunction(V3 c, V3 n, double d);
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override;
};

// pose_local_parameterization.cpp:3-19
void pose_plus(const double *x, const double *delta, double *x_plus_delta);
// EstimationMapping.hpp:34-49 + common.h:137-176
void se3_plus(const double *x, const double *delta, double *x_plus_delta);
void getTransformFromSe3(const double se3[6], Q4 &q, V3 &t);

// IntegrationBase (integration_base.h:13-158): mid-point pre-integration with 15x15 jacobian / covariance.
void imu_preintegrate(const vilf_imu_noise *noise, const double acc_0[3], const double gyr_0[3], const double ba[3],
                      const double bg[3], int n, const double *dt, const double *acc, const double *gyr, vilf_imu_preint *out);

}  // namespace ora
