// ORACLE — TEST INFRASTRUCTURE ONLY. "parity unpinned" vs Ceres 2.0.0 (see solver.hpp).
#include "solver.hpp"
#include <chrono>
#include <cassert>

namespace ora {

int Problem::add_parameter_block(double *data, int size, ParamType type) {
    ParamBlock b;
    b.data = data; b.size = size; b.type = type;
    b.local_size = (type == PARAM_EUCLID) ? size : 6;
    blocks.push_back(b);
    return (int)blocks.size() - 1;
}
void Problem::add_residual_block(const CostFunction *c, const LossFunction *l, const std::vector<int> &params) {
    ResidualBlock r; r.cost = c; r.loss = l; r.params = params;
    residuals.push_back(r);
}

namespace {

struct RBJac {                      // block-sparse Jacobian row block (local parameterisation applied)
    int row0, nres;
    std::vector<int> pcol;          // tangent column per parameter (-1: constant)
    std::vector<int> plocal;        // local size
    std::vector<int> pelim;         // 1 if e-block
    std::vector<std::vector<double>> J;  // nres x local, row-major
};

struct Program {
    Problem &pb;
    std::vector<int> active_res;    // residual blocks with >= 1 non-constant parameter
    int nf = 0, ne = 0, ncols = 0, nx = 0, nrows = 0;
    double fixed_cost = 0;
    std::vector<RBJac> jac;
    explicit Program(Problem &p) : pb(p) {}

    void build() {
        for (auto &b : pb.blocks) { b.col = -1; b.xoff = -1; }
        // only blocks that appear in some residual block and are not constant take part
        std::vector<char> used(pb.blocks.size(), 0);
        for (auto &r : pb.residuals) for (int id : r.params) used[id] = 1;
        for (size_t i = 0; i < pb.blocks.size(); i++) {
            auto &b = pb.blocks[i];
            if (b.constant || !used[i] || b.eliminate) continue;
            b.col = nf; nf += b.local_size; b.xoff = nx; nx += b.size;
        }
        for (size_t i = 0; i < pb.blocks.size(); i++) {
            auto &b = pb.blocks[i];
            if (b.constant || !used[i] || !b.eliminate) continue;
            assert(b.local_size == 1);
            b.col = nf + ne; ne += 1; b.xoff = nx; nx += b.size;
        }
        ncols = nf + ne;
        for (size_t k = 0; k < pb.residuals.size(); k++) {
            auto &r = pb.residuals[k];
            bool any = false;
            for (int id : r.params) if (pb.blocks[id].col >= 0) any = true;
            if (!any) {  // ceres removes it and books its cost as fixed_cost
                std::vector<const double *> pp;
                for (int id : r.params) pp.push_back(pb.blocks[id].data);
                std::vector<double> res(r.cost->num_residuals);
                r.cost->Evaluate(pp.data(), res.data(), nullptr);
                double s = 0; for (double v : res) s += v * v;
                if (r.loss) { double rho[3]; r.loss->Evaluate(s, rho); fixed_cost += 0.5 * rho[0]; } else fixed_cost += 0.5 * s;
                continue;
            }
            active_res.push_back((int)k);
            RBJac jb; jb.row0 = nrows; jb.nres = r.cost->num_residuals;
            nrows += jb.nres;
            for (int id : r.params) {
                auto &b = pb.blocks[id];
                jb.pcol.push_back(b.col);
                jb.plocal.push_back(b.local_size);
                jb.pelim.push_back(b.eliminate ? 1 : 0);
                jb.J.emplace_back(b.col >= 0 ? (size_t)jb.nres * b.local_size : 0, 0.0);
            }
            jac.push_back(std::move(jb));
        }
    }
    void state_from_user(std::vector<double> &x) const {
        x.assign(nx, 0.0);
        for (auto &b : pb.blocks) if (b.xoff >= 0) std::memcpy(&x[b.xoff], b.data, sizeof(double) * b.size);
    }
    void state_to_user(const std::vector<double> &x) const {
        for (auto &b : pb.blocks) if (b.xoff >= 0) std::memcpy(b.data, &x[b.xoff], sizeof(double) * b.size);
    }
    void plus(const std::vector<double> &x, const std::vector<double> &delta, std::vector<double> &xp) const {
        xp = x;
        for (auto &b : pb.blocks) {
            if (b.xoff < 0) continue;
            const double *xb = &x[b.xoff]; const double *d = &delta[b.col]; double *o = &xp[b.xoff];
            if (b.type == PARAM_EUCLID) for (int i = 0; i < b.size; i++) o[i] = xb[i] + d[i];
            else if (b.type == PARAM_POSE) pose_plus(xb, d, o);
            else se3_plus(xb, d, o);
        }
    }
    // Evaluate cost (+ residuals, gradient, jacobian when want_jac) at state x. Cost excludes fixed_cost.
    void evaluate(const std::vector<double> &x, double &cost, std::vector<double> *residuals, std::vector<double> *gradient,
                  bool want_jac) {
        cost = 0;
        if (residuals) residuals->assign(nrows, 0.0);
        if (gradient) gradient->assign(ncols, 0.0);
        std::vector<const double *> pp;
        std::vector<double> res;
        std::vector<std::vector<double>> gj;
        std::vector<double *> gjp;
        for (size_t a = 0; a < active_res.size(); a++) {
            auto &r = pb.residuals[active_res[a]];
            auto &jb = jac[a];
            const int np = (int)r.params.size();
            pp.resize(np);
            for (int i = 0; i < np; i++) { auto &b = pb.blocks[r.params[i]]; pp[i] = b.xoff >= 0 ? &x[b.xoff] : b.data; }
            res.assign(jb.nres, 0.0);
            if (want_jac) {
                gj.resize(np); gjp.resize(np);
                for (int i = 0; i < np; i++) {
                    auto &b = pb.blocks[r.params[i]];
                    if (b.col >= 0) { gj[i].assign((size_t)jb.nres * b.size, 0.0); gjp[i] = gj[i].data(); } else gjp[i] = nullptr;
                }
                r.cost->Evaluate(pp.data(), res.data(), gjp.data());
            } else {
                r.cost->Evaluate(pp.data(), res.data(), nullptr);
            }
            double sq = 0; for (double v : res) sq += v * v;
            if (r.loss) {
                double rho[3];
                std::vector<int> sizes(np);
                for (int i = 0; i < np; i++) sizes[i] = pb.blocks[r.params[i]].size;
                apply_corrector(r.loss, jb.nres, res.data(), np, sizes.data(), want_jac ? gjp.data() : nullptr, rho);
                cost += 0.5 * rho[0];
            } else {
                cost += 0.5 * sq;
            }
            if (residuals) for (int i = 0; i < jb.nres; i++) (*residuals)[jb.row0 + i] = res[i];
            if (want_jac) {
                for (int i = 0; i < np; i++) {
                    auto &b = pb.blocks[r.params[i]];
                    if (b.col < 0) continue;
                    // local jacobian = global jacobian * d(Plus)/d(delta); both reference parameterisations use [I6; 0]
                    for (int rr = 0; rr < jb.nres; rr++)
                        for (int c = 0; c < b.local_size; c++) jb.J[i][rr * b.local_size + c] = gj[i][rr * b.size + c];
                    if (gradient)
                        for (int c = 0; c < b.local_size; c++) {
                            double s = 0;
                            for (int rr = 0; rr < jb.nres; rr++) s += jb.J[i][rr * b.local_size + c] * res[rr];
                            (*gradient)[b.col + c] += s;
                        }
                }
            }
        }
    }
    void scale_columns(const std::vector<double> &scale) {
        for (auto &jb : jac)
            for (size_t i = 0; i < jb.pcol.size(); i++) {
                if (jb.pcol[i] < 0) continue;
                int L = jb.plocal[i];
                for (int rr = 0; rr < jb.nres; rr++) for (int c = 0; c < L; c++) jb.J[i][rr * L + c] *= scale[jb.pcol[i] + c];
            }
    }
    void squared_column_norm(std::vector<double> &out) const {
        out.assign(ncols, 0.0);
        for (auto &jb : jac)
            for (size_t i = 0; i < jb.pcol.size(); i++) {
                if (jb.pcol[i] < 0) continue;
                int L = jb.plocal[i];
                for (int rr = 0; rr < jb.nres; rr++) for (int c = 0; c < L; c++) { double v = jb.J[i][rr * L + c]; out[jb.pcol[i] + c] += v * v; }
            }
    }
    void right_multiply(const std::vector<double> &v, std::vector<double> &out) const {  // out = J v
        out.assign(nrows, 0.0);
        for (auto &jb : jac)
            for (size_t i = 0; i < jb.pcol.size(); i++) {
                if (jb.pcol[i] < 0) continue;
                int L = jb.plocal[i];
                for (int rr = 0; rr < jb.nres; rr++) {
                    double s = 0;
                    for (int c = 0; c < L; c++) s += jb.J[i][rr * L + c] * v[jb.pcol[i] + c];
                    out[jb.row0 + rr] += s;
                }
            }
    }
    // Solve (J^T J + diag(D)^2) y = J^T r. Dense Schur: e-blocks (dim 1) are eliminated, the reduced f-block system is
    // factorised by dense Cholesky (Eigen LLT in Ceres 2.0), then back-substituted. Returns false on factorisation failure.
    bool linear_solve(const std::vector<double> &r, const std::vector<double> &D, std::vector<double> &y) const {
        Mat H(nf, nf);
        std::vector<double> g(nf, 0.0), he(ne, 0.0), ge(ne, 0.0);
        std::vector<std::vector<double>> we(ne, std::vector<double>());  // H_fe column per e-block (dense nf)
        for (int e = 0; e < ne; e++) we[e].assign(nf, 0.0);
        for (auto &jb : jac) {
            const int np = (int)jb.pcol.size();
            const double *rb = &r[jb.row0];
            for (int i = 0; i < np; i++) {
                if (jb.pcol[i] < 0) continue;
                const int Li = jb.plocal[i];
                const double *Ji = jb.J[i].data();
                if (jb.pelim[i]) {
                    int e = jb.pcol[i] - nf;
                    for (int rr = 0; rr < jb.nres; rr++) { he[e] += Ji[rr] * Ji[rr]; ge[e] += Ji[rr] * rb[rr]; }
                    for (int j = 0; j < np; j++) {
                        if (jb.pcol[j] < 0 || jb.pelim[j]) continue;
                        const int Lj = jb.plocal[j];
                        const double *Jj = jb.J[j].data();
                        for (int c = 0; c < Lj; c++) { double s = 0; for (int rr = 0; rr < jb.nres; rr++) s += Jj[rr * Lj + c] * Ji[rr]; we[e][jb.pcol[j] + c] += s; }
                    }
                    continue;
                }
                for (int c = 0; c < Li; c++) { double s = 0; for (int rr = 0; rr < jb.nres; rr++) s += Ji[rr * Li + c] * rb[rr]; g[jb.pcol[i] + c] += s; }
                for (int j = 0; j < np; j++) {
                    if (jb.pcol[j] < 0 || jb.pelim[j]) continue;
                    const int Lj = jb.plocal[j];
                    const double *Jj = jb.J[j].data();
                    for (int a = 0; a < Li; a++)
                        for (int b = 0; b < Lj; b++) {
                            double s = 0;
                            for (int rr = 0; rr < jb.nres; rr++) s += Ji[rr * Li + a] * Jj[rr * Lj + b];
                            H(jb.pcol[i] + a, jb.pcol[j] + b) += s;
                        }
                }
            }
        }
        for (int i = 0; i < nf; i++) H(i, i) += D[i] * D[i];
        for (int e = 0; e < ne; e++) {
            double h = he[e] + D[nf + e] * D[nf + e];
            if (!(h > 0.0)) return false;
            const double inv = 1.0 / h;
            const std::vector<double> &w = we[e];
            int lo = nf, hi = -1;
            for (int i = 0; i < nf; i++) if (w[i] != 0.0) { lo = std::min(lo, i); hi = std::max(hi, i); }
            for (int i = lo; i <= hi; i++) {
                if (w[i] == 0.0) continue;
                double wi = w[i] * inv;
                for (int j = lo; j <= hi; j++) H(i, j) -= wi * w[j];
                g[i] -= wi * ge[e];
            }
        }
        if (!cholesky_lower(H)) return false;
        y.assign(ncols, 0.0);
        for (int i = 0; i < nf; i++) y[i] = g[i];
        chol_solve(H, y.data());
        for (int e = 0; e < ne; e++) {
            double h = he[e] + D[nf + e] * D[nf + e];
            double s = ge[e];
            for (int i = 0; i < nf; i++) s -= we[e][i] * y[i];
            y[nf + e] = s / h;
        }
        for (double v : y) if (!std::isfinite(v)) return false;
        return true;
    }
};

inline double vnorm(const std::vector<double> &v) { double s = 0; for (double a : v) s += a * a; return std::sqrt(s); }
inline double vdot(const std::vector<double> &a, const std::vector<double> &b) { double s = 0; for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i]; return s; }

// ---- trust-region strategies --------------------------------------------------------------------------
struct Strategy {
    const SolverOptions &opt;
    Program &prog;
    double radius;
    int num_linear_solves = 0;
    Strategy(const SolverOptions &o, Program &p) : opt(o), prog(p), radius(o.initial_trust_region_radius) {}
    virtual ~Strategy() {}
    virtual bool compute_step(const std::vector<double> &residuals, std::vector<double> &step) = 0;  // false: LINEAR_SOLVER_FAILURE
    virtual void step_accepted(double q) = 0;
    virtual void step_rejected(double q) = 0;
    virtual void step_is_invalid() = 0;
};

struct Dogleg : Strategy {  // dogleg_strategy.cc, TRADITIONAL_DOGLEG
    double mu, min_mu = 1e-8, max_mu = 1.0, mu_increase_factor = 10.0;
    double increase_threshold = 0.75, decrease_threshold = 0.25;
    double dogleg_step_norm = 0, alpha = 0;
    bool reuse = false;
    std::vector<double> diagonal, gradient, gauss_newton_step;
    Dogleg(const SolverOptions &o, Program &p) : Strategy(o, p), mu(1e-8) {}
    void traditional(std::vector<double> &step) {
        const int n = prog.ncols;
        step.assign(n, 0.0);
        const double gradient_norm = vnorm(gradient);
        const double gn_norm = vnorm(gauss_newton_step);
        if (gn_norm <= radius) {
            for (int i = 0; i < n; i++) step[i] = gauss_newton_step[i] / diagonal[i];
            dogleg_step_norm = gn_norm;
            return;
        }
        if (gradient_norm * alpha >= radius) {
            for (int i = 0; i < n; i++) step[i] = -(radius / gradient_norm) * gradient[i] / diagonal[i];
            dogleg_step_norm = radius;
            return;
        }
        const double b_dot_a = -alpha * vdot(gradient, gauss_newton_step);
        const double a_squared_norm = std::pow(alpha * gradient_norm, 2.0);
        const double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + std::pow(gn_norm, 2);
        const double c = b_dot_a - a_squared_norm;
        const double d = std::sqrt(c * c + b_minus_a_squared_norm * (std::pow(radius, 2.0) - a_squared_norm));
        double beta = (c <= 0) ? (d - c) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d + c);
        double nn = 0;
        for (int i = 0; i < n; i++) { step[i] = (-alpha * (1.0 - beta)) * gradient[i] + beta * gauss_newton_step[i]; nn += step[i] * step[i]; }
        dogleg_step_norm = std::sqrt(nn);
        for (int i = 0; i < n; i++) step[i] /= diagonal[i];
    }
    bool compute_step(const std::vector<double> &residuals, std::vector<double> &step) override {
        const int n = prog.ncols;
        if (reuse) { traditional(step); return true; }
        reuse = true;
        prog.squared_column_norm(diagonal);
        for (int i = 0; i < n; i++) diagonal[i] = std::sqrt(std::min(std::max(diagonal[i], opt.min_lm_diagonal), opt.max_lm_diagonal));
        // gradient_ = (J^T r) ./ diagonal
        gradient.assign(n, 0.0);
        for (auto &jb : prog.jac)
            for (size_t i = 0; i < jb.pcol.size(); i++) {
                if (jb.pcol[i] < 0) continue;
                int L = jb.plocal[i];
                for (int c = 0; c < L; c++) { double s = 0; for (int rr = 0; rr < jb.nres; rr++) s += jb.J[i][rr * L + c] * residuals[jb.row0 + rr]; gradient[jb.pcol[i] + c] += s; }
            }
        for (int i = 0; i < n; i++) gradient[i] /= diagonal[i];
        // Cauchy point
        std::vector<double> sg(n), Jg;
        for (int i = 0; i < n; i++) sg[i] = gradient[i] / diagonal[i];
        prog.right_multiply(sg, Jg);
        alpha = vdot(gradient, gradient) / vdot(Jg, Jg);
        // Gauss-Newton step with mu-regularisation retry loop
        bool ok = false;
        std::vector<double> lm(n), y;
        while (mu < max_mu) {
            for (int i = 0; i < n; i++) lm[i] = diagonal[i] * std::sqrt(mu);
            num_linear_solves++;
            if (!prog.linear_solve(residuals, lm, y)) { mu *= mu_increase_factor; continue; }
            ok = true;
            break;
        }
        if (!ok) return false;
        gauss_newton_step.assign(n, 0.0);
        for (int i = 0; i < n; i++) gauss_newton_step[i] = y[i] * -diagonal[i];
        traditional(step);
        return true;
    }
    void step_accepted(double q) override {
        if (q < decrease_threshold) radius *= 0.5;
        if (q > increase_threshold) radius = std::max(radius, 3.0 * dogleg_step_norm);
        mu = std::max(min_mu, 2.0 * mu / mu_increase_factor);
        reuse = false;
    }
    void step_rejected(double) override { radius *= 0.5; reuse = true; }
    void step_is_invalid() override { mu *= mu_increase_factor; reuse = false; }
};

struct LevenbergMarquardt : Strategy {  // levenberg_marquardt_strategy.cc
    double decrease_factor = 2.0;
    bool reuse_diagonal = false;
    std::vector<double> diagonal;
    LevenbergMarquardt(const SolverOptions &o, Program &p) : Strategy(o, p) {}
    bool compute_step(const std::vector<double> &residuals, std::vector<double> &step) override {
        const int n = prog.ncols;
        if (!reuse_diagonal) {
            prog.squared_column_norm(diagonal);
            for (int i = 0; i < n; i++) diagonal[i] = std::min(std::max(diagonal[i], opt.min_lm_diagonal), opt.max_lm_diagonal);
        }
        std::vector<double> lm(n), y;
        for (int i = 0; i < n; i++) lm[i] = std::sqrt(diagonal[i] / radius);
        num_linear_solves++;
        bool ok = prog.linear_solve(residuals, lm, y);
        reuse_diagonal = true;
        if (!ok) return false;
        step.assign(n, 0.0);
        for (int i = 0; i < n; i++) step[i] = -y[i];
        return true;
    }
    void step_accepted(double q) override {
        radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * q - 1.0, 3));
        radius = std::min(opt.max_trust_region_radius, radius);
        decrease_factor = 2.0;
        reuse_diagonal = false;
    }
    void step_rejected(double) override { radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true; }
    void step_is_invalid() override { step_rejected(0.0); reuse_diagonal = true; }
};

}  // namespace

// trust_region_minimizer.cc (Ceres 2.0): monotonic steps, no inner iterations, unconstrained.
void solve(const SolverOptions &opt, Problem &problem, SolveSummary &sum) {
    auto t0 = std::chrono::steady_clock::now();
    Program prog(problem);
    prog.build();
    sum = SolveSummary();
    if (prog.ncols == 0) { sum.termination = VILF_TERM_CONVERGENCE_GRADIENT; return; }
    std::unique_ptr<Strategy> strategy;
    if (opt.strategy == STRATEGY_DOGLEG) strategy.reset(new Dogleg(opt, prog)); else strategy.reset(new LevenbergMarquardt(opt, prog));

    std::vector<double> x, candidate_x, residuals, gradient, scale, step, delta(prog.ncols), model_residuals, neg_g, proj;
    prog.state_from_user(x);
    double x_norm = vnorm(x);
    double x_cost = 0, candidate_cost = 0;
    bool scaling_ready = false;
    double gradient_max_norm = 0;

    auto eval_grad_jac = [&]() {
        prog.evaluate(x, x_cost, &residuals, &gradient, true);
        if (opt.jacobi_scaling) {
            if (!scaling_ready) {
                prog.squared_column_norm(scale);
                for (auto &s : scale) s = 1.0 / (1.0 + std::sqrt(s));
                scaling_ready = true;
            }
            prog.scale_columns(scale);
        } else if (!scaling_ready) { scale.assign(prog.ncols, 1.0); scaling_ready = true; }
        neg_g.resize(prog.ncols);
        for (int i = 0; i < prog.ncols; i++) neg_g[i] = -gradient[i];
        prog.plus(x, neg_g, proj);
        gradient_max_norm = 0;
        for (int i = 0; i < prog.nx; i++) gradient_max_norm = std::max(gradient_max_norm, std::fabs(x[i] - proj[i]));
    };

    // iteration zero
    eval_grad_jac();
    sum.initial_cost = x_cost + prog.fixed_cost;
    {
        IterationRecord it{}; it.iteration = 0; it.cost = sum.initial_cost; it.gradient_max_norm = gradient_max_norm;
        it.trust_region_radius = strategy->radius; it.step_is_valid = 1; it.step_is_successful = 1;
        sum.iterations.push_back(it);
    }
    int termination = VILF_TERM_NO_CONVERGENCE;
    bool done = false;
    if (gradient_max_norm <= opt.gradient_tolerance) { termination = VILF_TERM_CONVERGENCE_GRADIENT; done = true; }

    int iteration = 0, consecutive_invalid = 0;
    while (!done) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue() of the previous iteration
        if (opt.max_solver_time > 0) {
            double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el >= opt.max_solver_time) { termination = VILF_TERM_NO_CONVERGENCE; break; }
        }
        if (iteration >= opt.max_num_iterations) { termination = VILF_TERM_NO_CONVERGENCE; break; }
        if (gradient_max_norm <= opt.gradient_tolerance) { termination = VILF_TERM_CONVERGENCE_GRADIENT; break; }
        if (strategy->radius <= opt.min_trust_region_radius) { termination = VILF_TERM_FAILURE; break; }

        iteration++;
        IterationRecord it{}; it.iteration = iteration; it.gradient_max_norm = gradient_max_norm;
        // ComputeTrustRegionStep
        bool valid = strategy->compute_step(residuals, step);
        double model_cost_change = 0;
        if (valid) {
            prog.right_multiply(step, model_residuals);
            double s = 0;
            for (int i = 0; i < prog.nrows; i++) s += model_residuals[i] * (residuals[i] + model_residuals[i] / 2.0);
            model_cost_change = -s;
            if (model_cost_change <= 0.0) valid = false;
        }
        it.step_is_valid = valid;
        if (!valid) {
            consecutive_invalid++;
            strategy->step_is_invalid();
            it.cost = x_cost + prog.fixed_cost; it.trust_region_radius = strategy->radius;
            sum.iterations.push_back(it);
            if (consecutive_invalid >= opt.max_num_consecutive_invalid_steps) { termination = VILF_TERM_FAILURE; break; }
            continue;
        }
        consecutive_invalid = 0;
        for (int i = 0; i < prog.ncols; i++) delta[i] = step[i] * scale[i];
        prog.plus(x, delta, candidate_x);
        prog.evaluate(candidate_x, candidate_cost, nullptr, nullptr, false);
        // ParameterToleranceReached
        double sn = 0; for (int i = 0; i < prog.nx; i++) sn += (x[i] - candidate_x[i]) * (x[i] - candidate_x[i]);
        it.step_norm = std::sqrt(sn);
        if (it.step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) {
            termination = VILF_TERM_CONVERGENCE_PARAMETER; it.cost = x_cost + prog.fixed_cost; sum.iterations.push_back(it); break;
        }
        // FunctionToleranceReached
        it.cost_change = x_cost - candidate_cost;
        if (std::fabs(it.cost_change) <= opt.function_tolerance * x_cost) {
            termination = VILF_TERM_CONVERGENCE_FUNCTION; it.cost = x_cost + prog.fixed_cost; sum.iterations.push_back(it); break;
        }
        it.relative_decrease = (x_cost - candidate_cost) / model_cost_change;
        if (it.relative_decrease > opt.min_relative_decrease) {
            x = candidate_x; x_norm = vnorm(x);
            eval_grad_jac();
            it.step_is_successful = 1; it.gradient_max_norm = gradient_max_norm;
            strategy->step_accepted(it.relative_decrease);
            sum.num_successful_steps++;
            it.cost = x_cost + prog.fixed_cost;
        } else {
            it.step_is_successful = 0;
            it.cost = candidate_cost + prog.fixed_cost;
            strategy->step_rejected(it.relative_decrease);
        }
        it.trust_region_radius = strategy->radius;
        sum.iterations.push_back(it);
    }
    prog.state_to_user(x);
    sum.num_iterations = iteration;
    sum.num_linear_solves = strategy->num_linear_solves;
    sum.termination = termination;
    sum.final_cost = x_cost + prog.fixed_cost;
    sum.final_radius = strategy->radius;
}

}  // namespace ora
