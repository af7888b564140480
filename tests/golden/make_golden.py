#!/usr/bin/env python3
"""Generate the committed golden fixtures (inputs + expected outputs) for the hot path.

The reference ships no golden vectors for this path and cannot be built or run here, so these vectors come
from the CPU restatement in oracle/ ("parity unpinned": they certify HIP == restatement and guard the restatement against
regressions; they do not certify restatement == Ceres/PCL). Fixtures are DATA: window inputs in the ABI's layouts and the
expected post-solve state / prior / scan-to-map results.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import os
import sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, os.path.join(HERE, ".."))
import oracle_lib as ol
from vil_fusion_amd import abi, synth

WINDOW_FIELDS = ["para_pose", "para_speed_bias", "para_ex_pose", "para_feature", "feature_const", "feature_start_frame",
                 "feature_obs_offset", "obs_point", "imu", "lidar"]

CASES = {   # name: (seed, SynthConfig kwargs, with_prior, use_lidar, marginalization_flag)
    "window_prior_lidar": (101, dict(n_features=60), True, True, abi.MARGIN_OLD),
    "window_noprior_nolidar": (102, dict(n_features=40), False, False, abi.MARGIN_OLD),
    "window_second_new": (103, dict(n_features=50), True, True, abi.MARGIN_SECOND_NEW),
    "window_all_const_depth": (104, dict(n_features=30, const_fraction=1.0), True, True, abi.MARGIN_OLD),
}


def pack_prior(prefix, p, d):
    if p is None or not p.valid:
        d[prefix + "valid"] = np.array(0)
        return
    J, r, blocks = abi.prior_to_numpy(p)
    d[prefix + "valid"] = np.array(1); d[prefix + "m"] = np.array(p.m)
    d[prefix + "J"] = J; d[prefix + "r"] = r
    d[prefix + "ids"] = np.array([b["id"] for b in blocks]); d[prefix + "sizes"] = np.array([b["size"] for b in blocks])
    d[prefix + "idx"] = np.array([b["idx"] for b in blocks])
    x0 = np.zeros((len(blocks), 9))
    for i, b in enumerate(blocks):
        x0[i, :b["size"]] = b["x0"]
    d[prefix + "x0"] = x0


def main():
    o = ol.default_options()
    for name, (seed, kw, with_prior, use_lidar, flag) in CASES.items():
        cfg = synth.SynthConfig(with_prior=with_prior, use_lidar=use_lidar, **kw)
        win, prior, _ = synth.make_window(seed, o, cfg)
        win.marginalization_flag = flag
        res = ol.window_solve(o, win, prior)
        newp = ol.window_marginalize(o, win, res, prior)
        d = {f: getattr(win, f) for f in WINDOW_FIELDS if getattr(win, f) is not None}
        d["marginalization_flag"] = np.array(flag); d["use_lidar"] = np.array(int(use_lidar))
        pack_prior("prior_", prior, d)
        for f in ("para_pose", "para_speed_bias", "para_feature", "Ps", "Rs", "Vs", "Bas", "Bgs"):
            d["out_" + f] = getattr(res, f)
        s = res.summary
        d["out_summary"] = np.array([s["num_iterations"], s["num_successful_steps"], s["num_linear_solves"], s["termination"]])
        d["out_cost"] = np.array([s["initial_cost"], s["final_cost"], s["final_radius"]])
        pack_prior("newprior_", newp, d)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, {k: v.shape for k, v in d.items() if k.startswith("out_P") or k == "newprior_J"}, s)
    # scan-to-map: 6 scans (32 rings x 900 azimuths), expected per-step results and final maps
    scans, poses = synth.make_lidar_sequence(3, 6, rings=32, azimuths=900)
    m = ol.OracleS2M(o); m.init(*scans[0])
    d = {}
    for k, (e, s_) in enumerate(scans):
        d[f"edge{k}"] = e; d[f"surf{k}"] = s_
    for k in range(1, 6):
        r = m.step(*scans[k])
        d[f"res{k}_pose"] = np.array(r.pose_qt[:]); d[f"res{k}_rel"] = np.array(list(r.rel_q[:]) + list(r.rel_t[:]))
        d[f"res{k}_ints"] = np.array([r.n_edge_ds, r.n_surf_ds, *r.n_edge_factors, *r.n_surf_factors, *r.iterations, r.map_edge_size, r.map_surf_size])
        d[f"res{k}_cost"] = np.array(r.final_cost[:])
    d["map_edge"] = m.get_map(0); d["map_surf"] = m.get_map(1)
    np.savez_compressed(os.path.join(HERE, "scan2map_seq.npz"), **d)
    print("scan2map_seq", d["map_edge"].shape, d["map_surf"].shape, d["res5_ints"])


if __name__ == "__main__":
    main()
