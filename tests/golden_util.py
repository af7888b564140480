"""Load the committed golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py)."""
import glob
import os
import numpy as np
from vil_fusion_amd import abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
WINDOW_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "window_*.npz")))


def _prior(d, prefix):
    if not int(d[prefix + "valid"]):
        return None
    blocks = [dict(id=int(i), size=int(s), idx=int(x), x0=d[prefix + "x0"][k, :int(s)])
              for k, (i, s, x) in enumerate(zip(d[prefix + "ids"], d[prefix + "sizes"], d[prefix + "idx"]))]
    return abi.make_prior(d[prefix + "J"], d[prefix + "r"], blocks, m=int(d[prefix + "m"]))


def load_window(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    win = abi.Window(d["para_pose"], d["para_speed_bias"], d["para_ex_pose"], d["para_feature"], d["feature_const"],
                     d["feature_start_frame"], d["feature_obs_offset"], d["obs_point"], d["imu"],
                     lidar=d["lidar"] if "lidar" in d.files else None, marginalization_flag=int(d["marginalization_flag"]))
    return win, _prior(d, "prior_"), d


def check_solve(res, d, tol_p, tol_v):
    assert [res.summary[k] for k in ("num_iterations", "num_successful_steps", "num_linear_solves", "termination")] == list(d["out_summary"])
    assert np.isclose(res.summary["initial_cost"], d["out_cost"][0], rtol=1e-9)
    assert np.isclose(res.summary["final_cost"], d["out_cost"][1], rtol=1e-7)
    assert np.abs(res.Ps - d["out_Ps"]).max() < tol_p and np.abs(res.Rs - d["out_Rs"]).max() < tol_p
    assert np.abs(res.Vs - d["out_Vs"]).max() < tol_v
    assert np.abs(res.Bas - d["out_Bas"]).max() < tol_v and np.abs(res.Bgs - d["out_Bgs"]).max() < tol_v
    assert np.abs(res.para_feature - d["out_para_feature"]).max() < tol_v


def check_prior(p, d, rtol):
    J, r, blocks = abi.prior_to_numpy(p)
    assert [b["id"] for b in blocks] == list(d["newprior_ids"]) and [b["size"] for b in blocks] == list(d["newprior_sizes"])
    assert [b["idx"] for b in blocks] == list(d["newprior_idx"]) and p.m == int(d["newprior_m"])
    for k, b in enumerate(blocks):
        assert np.abs(b["x0"] - d["newprior_x0"][k, :b["size"]]).max() < 1e-6
    # J0 is unique only up to the sign / order of eigenvectors: compare the information form J0^T J0 and J0^T r0
    H, H0 = J.T @ J, d["newprior_J"].T @ d["newprior_J"]
    g, g0 = J.T @ r, d["newprior_J"].T @ d["newprior_r"]
    assert np.abs(H - H0).max() <= rtol * np.abs(H0).max()
    assert np.abs(g - g0).max() <= rtol * np.abs(g0).max()
