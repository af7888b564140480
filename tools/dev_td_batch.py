"""estimate_td batches (ProjectionTdFactor, td a variable: the general path, all slots as ONE group) against the plain batch of the same windows (LDS kernels)"""
import time, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vil_fusion_amd import synth, abi
from vil_fusion_amd.estimator import BackendSolver
from vil_fusion_amd.lib import default_options


def with_td_inputs(win, seed, td_true=0.004):
    rng = np.random.default_rng(seed)
    vel = rng.normal(0.0, 0.4, (win.n_obs, 2))
    pts = win.obs_point.copy(); pts[:, :2] += td_true * vel
    return abi.Window(win.para_pose, win.para_speed_bias, win.para_ex_pose, win.para_feature, win.feature_const, win.feature_start_frame,
                      win.feature_obs_offset, pts, win.imu, win.lidar, para_td=0.0, marginalization_flag=win.marginalization_flag,
                      obs_velocity=vel, obs_cur_td=np.zeros(win.n_obs), obs_row=rng.uniform(0.0, 370.0, win.n_obs))


for B in ([int(os.environ["VILF_TD_B"])] if "VILF_TD_B" in os.environ else (8, 64, 256)):
    res = {}
    o0 = default_options()
    made = [synth.make_window(300 + k, o0, synth.SynthConfig(n_features=120, with_prior=False)) for k in range(min(B, 16))]
    base = [with_td_inputs(m[0], 10 + k) for k, m in enumerate(made)]
    wins = [base[k % len(base)] for k in range(B)]
    for td in (0, 1):
        o = default_options(); o.estimate_td = td
        s = BackendSolver(o)
        s.batch_upload(wins, None)
        for _ in range(2):
            s.batch_rewind(); s.batch_solve()
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 5
        for _ in range(n):
            s.batch_rewind(); s.batch_solve()
        torch.cuda.synchronize(); res[td] = (time.perf_counter() - t0) / n
        its = sum(x.num_iterations for x in s.batch_summaries()) if hasattr(s, "batch_summaries") else -1
        s.close()
        res[(td, "its")] = its
    print("B", B, "plain ms", round(1e3 * res[0], 3), "estimate_td ms", round(1e3 * res[1], 3), "ratio", round(res[1] / res[0], 2), "iterations", res[(0, "its")], res[(1, "its")])
