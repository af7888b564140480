"""Committed golden fixtures: the oracle must reproduce them on CPU (guards the restatement), the HIP path on the GPU."""
import numpy as np
import pytest
import golden_util as gu


@pytest.mark.parametrize("name", gu.WINDOW_CASES)
def test_oracle_reproduces_window_fixture(oracle, opts, name):
    win, prior, d = gu.load_window(name)
    res = oracle.window_solve(opts, win, prior)
    gu.check_solve(res, d, 1e-10, 1e-9)
    gu.check_prior(oracle.window_marginalize(opts, win, res, prior), d, 1e-9)


def _s2m_check(r, d, k, tol):
    ints = [r.n_edge_ds, r.n_surf_ds, *r.n_edge_factors, *r.n_surf_factors, *r.iterations, r.map_edge_size, r.map_surf_size]
    assert ints == list(d[f"res{k}_ints"])
    assert np.abs(np.array(r.pose_qt[:]) - d[f"res{k}_pose"]).max() < tol
    assert np.abs(np.array(list(r.rel_q[:]) + list(r.rel_t[:])) - d[f"res{k}_rel"]).max() < tol
    assert np.allclose(list(r.final_cost[:]), d[f"res{k}_cost"], rtol=1e-8)


def test_oracle_reproduces_scan2map_fixture(oracle, opts):
    d = np.load(gu.GOLDEN + "/scan2map_seq.npz")
    m = oracle.OracleS2M(opts); m.init(d["edge0"], d["surf0"])
    for k in range(1, 6):
        _s2m_check(m.step(d[f"edge{k}"], d[f"surf{k}"]), d, k, 1e-12)
    assert np.array_equal(m.get_map(0), d["map_edge"]) and np.array_equal(m.get_map(1), d["map_surf"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", gu.WINDOW_CASES)
def test_hip_reproduces_window_fixture(opts, name):
    from vil_fusion_amd.estimator import BackendSolver
    win, prior, d = gu.load_window(name)
    s = BackendSolver(opts)
    s.set_prior(prior)
    res = s.optimization(win)
    gu.check_solve(res, d, 1e-7, 1e-6)
    s.marginalize()
    gu.check_prior(s.get_prior(), d, 2e-5)     # Amm spans ~14 decades: two fp64 eigen-solvers agree to ~1e-6..1e-5 relative
    s.close()


@pytest.mark.gpu
def test_hip_reproduces_scan2map_fixture(opts):
    from vil_fusion_amd.estimator import BackendSolver, Scan2Map
    d = np.load(gu.GOLDEN + "/scan2map_seq.npz")
    s = BackendSolver(opts); m = Scan2Map(s); m.localMapInited(d["edge0"], d["surf0"])
    for k in range(1, 6):
        _s2m_check(m.optimation_processing(d[f"edge{k}"], d[f"surf{k}"]), d, k, 1e-9)
    assert np.array_equal(m.getMapCloud(0), d["map_edge"]) and np.array_equal(m.getMapCloud(1), d["map_surf"])
    s.close()
