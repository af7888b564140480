"""Long-horizon trajectory parity — the stand-in for BASELINE configs[3] (KITTI-08, ~4070 frames, 8 shards; KITTI itself cannot be read here).

Eight independent sequence segments x >= 500 solved frames (different seeds, yaw profiles and speeds; slow and stop-and-go drives give
MARGIN_SECOND_NEW frames; one segment has a 0.7 s sensor gap that trips failureDetection() -> clearState() reboot, estimator.cpp:212-220,640-686):

  * HIP side: vil_fusion_amd.sequence.run_sequences_lockstep — the Python host loop, the 8 segments stepped frame by frame as ONE batch of 8 on the
    device (vilf_batch_upload / _solve / _marginalize per frame, the priors chained in their slots).
  * oracle side: oracle/sequence.cpp — an independent C++ statement of the same host loop (feature manager, processIMU / processImage, failure
    detection, slideWindow) around the oracle's own window solve and marginalization, one thread per segment.

What can and cannot be equal. The reference's marginalization (marginalization_factor.cpp:267-291) eigen-decomposes a Schur complement whose gauge
directions are null up to rounding and cuts at 1e-8: a perturbation of 1e-13 m in the start-up positions moves the ORACLE'S OWN trajectory by
1e-6 m after one frame and by millimetres to centimetres after 500 (test_free_running: `self_divergence`, measured in the same run). Free-running
trajectories of two correct implementations therefore agree only within that envelope, and the iteration count of a window changes when the
function-tolerance test (|d cost| < 1e-6 cost) falls on the other side of its threshold. The tight statement is the teacher-forced one: every one
of the ~4000 windows, given identical inputs (window + prior), is solved by both sides with identical iteration / step counts and equal poses.
"""
import threading
import numpy as np
import pytest
from vil_fusion_amd import sequence
import seq_backends

N_SEG, N_FRAMES = 8, 531
GAP_SEGMENT, GAP_AT, GAP_LEN = 3, 265, 7


def make_segments(opts, n_seg=N_SEG, n_frames=N_FRAMES):
    """seeded segments: fast drives with different yaw profiles, two slow / stop-and-go ones (SECOND_NEW frames), one with a sensor gap"""
    profiles = [dict(), dict(yaw_amplitude=0.15), dict(mean_speed=3.0), dict(), dict(yaw_amplitude=0.3), dict(mean_speed=4.0, speed_modulation=0.8),
                dict(yaw_amplitude=0.45), dict(mean_speed=6.0, speed_modulation=0.6, yaw_amplitude=0.2)]
    segs = []
    for i in range(n_seg):
        q = sequence.make_sequence(100 + i, n_frames, opts, **profiles[i % len(profiles)])
        if i == GAP_SEGMENT:
            q = sequence.drop_frames(q, min(GAP_AT, n_frames // 2), GAP_LEN)
        segs.append(q)
    return segs


def perturbed(seq, eps=1e-13, seed=0):
    """the same segment with the start-up positions moved by eps metres (about 14 ulp of a 50 m coordinate)"""
    rng = np.random.default_rng(seed)
    out = dict(seq)
    out["init"] = [(P + eps * rng.uniform(-1, 1, 3), R, V, ba, bg) for (P, R, V, ba, bg) in seq["init"]]
    return out


def run_oracle_loops(oracle, opts, seqs, n):
    refs = [oracle.OracleSequence(opts) for _ in seqs]
    th = [threading.Thread(target=r.run, args=(q, n)) for r, q in zip(refs, seqs)]
    [t.start() for t in th]; [t.join() for t in th]
    return refs


def divergence(a, b):
    """max |dP| over the common part of two trajectories, and the same for the frame-to-frame increments (free of accumulated drift)"""
    m = min(len(a.trajectory), len(b.trajectory))
    Pa = np.array([a.trajectory[k][1] for k in range(m)]); Pb = np.array([b.trajectory[k][1] for k in range(m)])
    dP = np.abs(Pa - Pb).max(axis=1)
    dInc = np.abs(np.diff(Pa, axis=0) - np.diff(Pb, axis=0)).max(axis=1)
    return dP, dInc


def iterations(e):
    return [x["num_iterations"] for x in e.summaries]


def test_python_host_loop_matches_the_cpp_restatement_frame_by_frame(oracle, opts):
    """sequence.py's bookkeeping (addFeatureCheckParallax, triangulate, setDepth / removeFailures, removeBackShiftDepth, removeFront, slideWindow,
    failureDetection + reboot) against oracle/sequence.cpp, both around the ORACLE's solve: the feature lists are compared after every frame —
    ids, start frames, track lengths, solve flags, LiDAR-depth flags identical, depths and states equal to 1e-2 (the two triangulations use different
    SVDs, the two IMU pre-integrations are different codes; the marginalization amplifies their 1e-13 m, see the module docstring)."""
    seq = sequence.drop_frames(sequence.make_sequence(3, 125, opts, mean_speed=6.0, speed_modulation=0.6), 81, 12)      # the gap falls on a fast stretch: > 5 m
    est = sequence.SlidingWindowEstimator(opts, seq_backends.OracleBackend(opts))
    ref = oracle.OracleSequence(opts)
    est.process_imu(0.0, *seq["imu0"]); ref.process_imu(0.0, *seq["imu0"])
    worst = 0.0
    for k in range(len(seq["images"])):
        sequence._feed_measurements(est, seq, k)
        if k >= 1:
            dt, acc, gyr = seq["imu"][k]
            for a, w in zip(acc, gyr):
                ref.process_imu(dt, a, w)
            ref.process_odometry(*seq["lidar"][k])
        init = seq["init"][k] if est.solver_flag == est.INITIAL else None
        assert (init is None) == (ref.solver_flag != 0)
        est.process_image(seq["images"][k], seq["stamps"][k], init)
        out = ref.process_image(seq["images"][k], seq["stamps"][k], init)
        assert est.events[-1] == ref.events[-1], k
        fa = [(it.feature_id, it.start_frame, len(it.feature_per_frame), it.solve_flag, int(it.lidar_depth_flag), it.estimated_depth) for it in est.f.feature]
        fb = ref.features()
        assert [x[:5] for x in fa] == [x[:5] for x in fb], k
        worst = max([worst] + [abs(x[5] - y[5]) / max(1.0, abs(y[5])) for x, y in zip(fa, fb)])
        if out.status == 1:
            Ps, Rs, Vs, Bas, Bgs = ref.state()
            assert np.abs(Ps - est.Ps).max() < 1e-2 and np.abs(Rs - est.Rs).max() < 1e-3 and np.abs(Vs - est.Vs).max() < 1e-2, k
    assert worst < 1e-2, worst
    assert est.n_reboots == 1 and est.events.count("reboot") == 1                  # the gap tripped failureDetection() in both
    assert 0 < sum(est.flags) < len(est.flags)                                     # both MARGIN_OLD and MARGIN_SECOND_NEW frames
    assert est.flags == ref.flags and iterations(est) == iterations(ref)
    dP, dInc = divergence(est, ref)
    assert dP.max() < 1e-2, dP.max()


@pytest.mark.gpu
def test_free_running_trajectories_stay_within_the_oracles_own_divergence(oracle, opts, tmp_path):
    """8 x >= 500 solved frames, free running (every side feeds on its own results). Asserted: window events (fill / solved / reboot) and key-frame
    flags identical over all ~4100 frames; the reboot, both marginalization flags and both prior forms (Cholesky / eigen) occur; per segment
    max |dP| <= 10 x the largest divergence the oracle shows against ITSELF under a 1e-13 m perturbation (and <= 0.5 m absolute — the stated
    tolerance of a 500-frame / ~500 m free run); iteration counts identical up to the first divergence, which is printed with both summaries."""
    from vil_fusion_amd.estimator import BackendSolver
    seqs = make_segments(opts)
    n = min(len(q["images"]) for q in seqs)
    both = run_oracle_loops(oracle, opts, seqs + [perturbed(q) for q in seqs], n)          # 16 threads: the oracle and its perturbed twin
    refs, twins = both[:len(seqs)], both[len(seqs):]
    s = BackendSolver(opts)
    ests = sequence.run_sequences_lockstep(seqs, opts, s, n)
    stats = s.lockstep_stats
    s.close()
    self_div = [divergence(a, b)[0].max() for a, b in zip(refs, twins)]
    envelope = max(self_div)
    solved = 0
    print()
    for i, (a, b) in enumerate(zip(ests, refs)):
        assert a.events == b.events, i
        assert a.flags == b.flags, i
        solved += len(a.trajectory)
        dP, dInc = divergence(a, b)
        ia, ib = iterations(a), iterations(b)
        first = next((k for k in range(len(ia)) if ia[k] != ib[k]), None)
        print(f"segment {i}: {len(a.trajectory)} solved frames, {sum(a.flags)} SECOND_NEW, {a.n_reboots} reboot(s); max|dP| {dP.max():.2e} m (frame-to-frame {dInc.max():.2e}); "
              f"oracle vs its 1e-13-perturbed twin {self_div[i]:.2e} m; iteration counts identical for the first {first if first is not None else len(ia)} frames")
        if first is not None:
            print(f"    first difference at frame {first}: HIP {a.summaries[first]['num_iterations']} iterations, termination {a.summaries[first]['termination']}, cost "
                  f"{a.summaries[first]['final_cost']:.6f}; oracle {b.summaries[first]['num_iterations']}, termination {b.summaries[first]['termination']}, cost "
                  f"{b.summaries[first]['final_cost']:.6f}; |dP| there {dP[first]:.2e} m")
            assert first >= 20, (i, first)                # nothing but accumulated divergence moves a count
        # measured (rounds 4 and 5): 1e-3 ... 2.7e-2 m per segment after ~500 m, the oracle's own 1e-13-perturbed twin 1e-3 ... 3e-2 m; the absolute cap is the largest
        # measured value x 3 (it was 0.5 m)
        assert dP.max() <= max(10 * envelope, 1e-4) and dP.max() < 0.1, f"segment {i}: max|dP| {dP.max():.3e} m after {len(a.trajectory)} frames; the oracle's own divergence under a 1e-13 m perturbation: {envelope:.3e} m (cap: 10 x that, and 0.1 m)"
        sequence.write_tum(str(tmp_path / f"vins_result_no_loop_{i}.txt"), a.trajectory)
        rows = np.loadtxt(str(tmp_path / f"vins_result_no_loop_{i}.txt"))
        assert rows.shape == (len(a.trajectory), 8)
    print(f"marginalization paths over {stats['frames']} lock-step frames: new priors {stats['new_prior']}, Amm by Cholesky {stats['amm_cholesky']}, "
          f"kept block by Cholesky {stats['kept_cholesky']} ({100.0 * stats['kept_cholesky'] / max(stats['new_prior'], 1):.0f} %), by the eigen path "
          f"{stats['new_prior'] - stats['kept_cholesky']}, unchanged {stats['unchanged']}")
    assert solved >= 4000
    assert ests[GAP_SEGMENT].n_reboots == 1 and refs[GAP_SEGMENT].events.count("reboot") == 1
    assert any(0 < sum(e.flags) for e in ests) and any(sum(e.flags) < len(e.flags) for e in ests)
    assert 0 < stats["kept_cholesky"] < stats["new_prior"]


@pytest.mark.gpu
def test_every_window_of_the_long_run_agrees_when_both_sides_get_the_same_inputs(oracle, opts):
    """Teacher-forced replay of the same 8 segments along the ORACLE's trajectory: at every frame the HIP path receives exactly the window and the
    prior the oracle solves (priors imported per slot), so nothing accumulates. Asserted for every window of the ~4100: identical iteration,
    successful-step and linear-solve counts and termination; poses within 1e-6 m / rotations 1e-7 of the oracle (measured: 9.6e-10 m / 1.1e-10);
    the new prior's J0^T J0 within 2e-5 and J0^T r0 within 1e-4 of their largest entries (two fp64 algorithms on a Schur complement conditioned ~1e13)."""
    from concurrent.futures import ThreadPoolExecutor
    from vil_fusion_amd.estimator import BackendSolver
    seqs = make_segments(opts)
    n = min(len(q["images"]) for q in seqs)
    S = len(seqs)
    backs = [seq_backends.OracleBackend(opts) for _ in range(S)]
    ests = [sequence.SlidingWindowEstimator(opts, b) for b in backs]
    for est, q in zip(ests, seqs):
        est.process_imu(0.0, *q["imu0"])
    s = BackendSolver(opts)
    pool = ThreadPoolExecutor(S)
    placeholder = [None] * S
    worst = dict(dP=0.0, dR=0.0, dV=0.0, JtJ=0.0, Jtr=0.0)
    windows = 0

    def oracle_frame(i, win):
        res = oracle.window_solve(opts, win, backs[i].prior)
        new = oracle.window_marginalize(opts, win, res, backs[i].prior)
        return res, new

    for k in range(n):
        wins = []
        for est, q in zip(ests, seqs):
            sequence._feed_measurements(est, q, k)
            wins.append(est.begin_image(q["images"][k], q["stamps"][k], q["init"][k] if est.solver_flag == est.INITIAL else None))
        live = [i for i in range(S) if wins[i] is not None]
        if not live:
            continue
        futures = {i: pool.submit(oracle_frame, i, wins[i]) for i in live}
        for i in range(S):
            s.set_prior(backs[i].prior if i in futures else None, i)
            if wins[i] is not None:
                placeholder[i] = wins[i]
        fill = placeholder[live[0]]
        s.batch_upload([wins[i] if wins[i] is not None else (placeholder[i] if placeholder[i] is not None else fill) for i in range(S)])
        s.batch_solve(sync=False)
        s.batch_marginalize(sync=True)
        got = s.batch_download()
        for i in live:
            ref, new = futures[i].result()
            g = got[i]
            for key in ("num_iterations", "num_successful_steps", "num_linear_solves", "termination"):
                assert g.summary[key] == ref.summary[key], (i, k, key, g.summary, ref.summary)
            worst["dP"] = max(worst["dP"], np.abs(g.Ps - ref.Ps).max()); worst["dR"] = max(worst["dR"], np.abs(g.Rs - ref.Rs).max())
            worst["dV"] = max(worst["dV"], np.abs(g.Vs - ref.Vs).max())
            hp = s.get_prior(i)
            assert bool(hp.valid) == bool(new.valid), (i, k)
            if new.valid:
                assert hp.n == new.n and [hp.block_id[b] for b in range(hp.n_blocks)] == [new.block_id[b] for b in range(new.n_blocks)], (i, k)
                Jh = np.frombuffer(hp.linearized_jacobians, dtype=np.float64, count=hp.n * hp.n).reshape(hp.n, hp.n); rh = np.frombuffer(hp.linearized_residuals, dtype=np.float64, count=hp.n)
                Jo = np.frombuffer(new.linearized_jacobians, dtype=np.float64, count=new.n * new.n).reshape(new.n, new.n); ro = np.frombuffer(new.linearized_residuals, dtype=np.float64, count=new.n)
                A, Ao = Jh.T @ Jh, Jo.T @ Jo
                worst["JtJ"] = max(worst["JtJ"], np.abs(A - Ao).max() / np.abs(Ao).max())
                bh, bo = Jh.T @ rh, Jo.T @ ro
                worst["Jtr"] = max(worst["Jtr"], np.abs(bh - bo).max() / np.abs(bo).max())
            backs[i].prior = new if new.valid else None
            ests[i].end_image(ref)
            windows += 1
    s.close()
    pool.shutdown()
    print(f"\n{windows} windows, identical counts in all of them; max |dP| {worst['dP']:.2e} m, |dR| {worst['dR']:.2e}, |dV| {worst['dV']:.2e} m/s; "
          f"prior J0^T J0 {worst['JtJ']:.2e}, J0^T r0 {worst['Jtr']:.2e} (relative to the largest entry)")
    assert windows >= 4000
    assert worst["dP"] < 1e-6 and worst["dR"] < 1e-7 and worst["dV"] < 1e-5, worst
    assert worst["JtJ"] < 2e-5 and worst["Jtr"] < 1e-4, worst          # measured over the 4102 windows: 3.8e-8 and 3.7e-5
    assert ests[GAP_SEGMENT].n_reboots == 1
