"""GPU-side development check of k_solve_sb: (1) same results as the dense-Cholesky kernel k_solve (VILF_SOLVE_DENSE=1) and as the oracle,
(2) per-kernel timing at B windows."""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
import oracle_lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = BackendSolver()
opts = s.options
cfg = synth.SynthConfig(n_features=230)
wins, priors = synth.make_batch(1000, 32, opts, cfg, distinct=32)


def run(dense):
    if dense: os.environ["VILF_SOLVE_DENSE"] = "1"
    else: os.environ.pop("VILF_SOLVE_DENSE", None)
    s.batch_upload(wins, priors)
    s.batch_solve()
    return s.batch_download(), s.batch_summaries()

o_sb, s_sb = run(False)
o_de, s_de = run(True)
os.environ.pop("VILF_SOLVE_DENSE", None)
worst = 0
for i in range(len(wins)):
    a, b = s_sb[i], s_de[i]
    same = (a.num_iterations, a.num_successful_steps, a.num_linear_solves, a.termination) == (b.num_iterations, b.num_successful_steps, b.num_linear_solves, b.termination)
    dP = np.abs(o_sb[i].Ps - o_de[i].Ps).max(); dV = np.abs(o_sb[i].Vs - o_de[i].Vs).max()
    worst = max(worst, dP)
    if not same or dP > 1e-7:
        print("MISMATCH window", i, (a.num_iterations, a.num_successful_steps, a.num_linear_solves, a.termination, a.final_cost), (b.num_iterations, b.num_successful_steps, b.num_linear_solves, b.termination, b.final_cost), dP, dV)
print("sb vs dense: worst |dP| over 32 windows", worst)
ref = oracle_lib.window_solve(oracle_lib.default_options(), wins[0], priors[0])
print("vs oracle window 0: its", s_sb[0].num_iterations, ref.summary["num_iterations"], "dP", np.abs(o_sb[0].Ps - ref.Ps).max(), "cost", s_sb[0].final_cost, ref.summary["final_cost"])
# no-prior windows
cfg2 = synth.SynthConfig(n_features=150, with_prior=False)
w2, p2 = synth.make_batch(77, 8, opts, cfg2, distinct=8)
s.batch_upload(w2, p2); s.batch_solve(); o2 = s.batch_download(); s2 = s.batch_summaries()
for i in range(2):
    ref = oracle_lib.window_solve(oracle_lib.default_options(), w2[i], p2[i])
    print("no prior", i, "its", s2[i].num_iterations, ref.summary["num_iterations"], "dP", np.abs(o2[i].Ps - ref.Ps).max())

wins, priors = synth.make_batch(7, B, opts, cfg, distinct=64)
for dense in (False, True):
    if dense: os.environ["VILF_SOLVE_DENSE"] = "1"
    else: os.environ.pop("VILF_SOLVE_DENSE", None)
    s.batch_upload(wins, priors)
    s.batch_solve()
    s.set_profiling(True)
    for rep in range(3):
        s.batch_rewind(); t = time.time(); s.batch_solve(); dt = time.time() - t
    its = sum(x.num_iterations for x in s.batch_summaries())
    prof = s.get_profile()
    print("dense" if dense else "sb", "B", B, "solve ms", dt * 1e3, "iter/s", its / dt, {k: round(v["ms"] / max(v["launches"], 1), 4) for k, v in prof.items()})
    s.set_profiling(False)
