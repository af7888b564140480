// ORACLE — TEST INFRASTRUCTURE ONLY. "parity unpinned" vs Ceres 2.0.0 (source not under /root/reference; restated
// from its published algorithm: trust_region_minimizer.cc, dogleg_strategy.cc, levenberg_marquardt_strategy.cc,
// schur_complement_solver.cc, corrector.cc, loss_function.cc — README.md:23-31 pins the version).
// Call sites that fix the configuration: vins_estimator/estimator.cpp:691-728,838-853 (DENSE_SCHUR + DOGLEG,
// CauchyLoss, 8 iterations) and feature_tracker/include/EstimationMapping.hpp:263-283 (DENSE_QR + default LM,
// HuberLoss, 4 iterations).
#pragma once
#include "factors.hpp"

namespace ora {

enum ParamType { PARAM_EUCLID = 0, PARAM_POSE = 1 /*PoseLocalParameterization*/, PARAM_SE3 = 2 /*LocalSE3Parameterization*/ };
enum StrategyType { STRATEGY_DOGLEG = 0, STRATEGY_LM = 1 };

struct ParamBlock {
    double *data;
    int size;
    int local_size;
    ParamType type;
    bool constant = false;
    bool eliminate = false;  // e-block of the Schur complement (dim-1 feature blocks)
    int col = -1;            // tangent-space column in the reduced program
    int xoff = -1;           // global-space offset in the reduced program state vector
};

struct ResidualBlock {
    const CostFunction *cost;
    const LossFunction *loss;  // may be null
    std::vector<int> params;   // indices into Problem::blocks
};

struct Problem {
    std::vector<ParamBlock> blocks;
    std::vector<ResidualBlock> residuals;
    int add_parameter_block(double *data, int size, ParamType type);
    void set_constant(int id) { blocks[id].constant = true; }
    void add_residual_block(const CostFunction *c, const LossFunction *l, const std::vector<int> &params);
};

struct SolverOptions {
    StrategyType strategy = STRATEGY_DOGLEG;
    int max_num_iterations = 8;
    double max_solver_time = -1;          // <= 0: disabled
    bool jacobi_scaling = true;
    double initial_trust_region_radius = 1e4;
    double max_trust_region_radius = 1e16;
    double min_trust_region_radius = 1e-32;
    double min_relative_decrease = 1e-3;
    double min_lm_diagonal = 1e-6;
    double max_lm_diagonal = 1e32;
    double function_tolerance = 1e-6;
    double gradient_tolerance = 1e-10;
    double parameter_tolerance = 1e-8;
    int max_num_consecutive_invalid_steps = 5;
};

struct IterationRecord {
    int iteration; double cost; double cost_change; double gradient_max_norm; double step_norm;
    double relative_decrease; double trust_region_radius; int step_is_valid; int step_is_successful;
};

struct SolveSummary {
    int num_iterations = 0;          // excluding iteration 0
    int num_successful_steps = 0;
    int num_linear_solves = 0;
    int termination = VILF_TERM_NO_CONVERGENCE;
    double initial_cost = 0, final_cost = 0, final_radius = 0;
    std::vector<IterationRecord> iterations;
};

void solve(const SolverOptions &opt, Problem &problem, SolveSummary &summary);

}  // namespace ora
