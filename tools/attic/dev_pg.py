import sys, time, os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np, oracle_lib
from vil_fusion_amd import posegraph
from vil_fusion_amd.estimator import BackendSolver, posegraph_optimize
PS = np.full(6, 1e-6)
s = BackendSolver()
for K, L in ((350, 10), (2000, 100), (4500, 300)):
    loops = [(int(i), int(K - 1 - i)) for i in np.linspace(0, K // 3, L).astype(int)]
    loops = sorted(set(l for l in loops if abs(l[0] - l[1]) > 1))
    truth, x0, edges = posegraph.make_synthetic_graph(K, K, loops=loops)
    t0 = time.perf_counter(); ref, itr, cr = oracle_lib.posegraph_optimize(x0, PS, edges, 30, 1e-9); t1 = time.perf_counter()
    posegraph_optimize(s, x0, PS, edges, 30, 1e-9)
    t2 = time.perf_counter(); got, it, c = posegraph_optimize(s, x0, PS, edges, 30, 1e-9); t3 = time.perf_counter()
    print(f"K {K} loops {len(loops)}: oracle {1e3*(t1-t0):.1f} ms ({itr} it)  hip {1e3*(t3-t2):.1f} ms ({it} it)  dP {np.abs(got[:,4:]-ref[:,4:]).max():.2e}")
