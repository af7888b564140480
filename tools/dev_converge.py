# how much of a solve's time do windows that terminate early give back? (diagnostic for the converging-batch line of bench.py)
import sys, time, copy, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = BackendSolver(); opts = s.options
wins, priors = synth.make_batch(1000, 32, opts, synth.SynthConfig(n_features=230), distinct=32)
o2 = type(opts).from_buffer_copy(opts); o2.max_num_iterations = 1000
pre = BackendSolver(o2); pre.batch_upload(wins, priors); pre.batch_solve(); res = pre.batch_download(); pre.close()
conv = []
for w, r in zip(wins, res):
    w2 = copy.deepcopy(w)
    w2.para_pose = np.ascontiguousarray(np.asarray(r.para_pose).reshape(-1, 7)); w2.para_speed_bias = np.ascontiguousarray(np.asarray(r.para_speed_bias).reshape(-1, 9)); w2.para_feature = np.ascontiguousarray(r.para_feature)
    conv.append(w2)
for tag, pick in (("all regular", lambda i: wins[i % 32]), ("half converged (alternating 16)", lambda i: conv[i % 32] if (i % 32) < 16 else wins[i % 32]),
                  ("half converged (random slots)", (lambda tbl: (lambda i: conv[i % 32] if tbl[i] else wins[i % 32]))(np.random.default_rng(5).integers(0, 2, 1 << 20))),
                  ("half converged (first half of the batch)", lambda i: conv[i % 32] if i < B // 2 else wins[i % 32]), ("all converged", lambda i: conv[i % 32])):
    s.batch_upload([pick(i) for i in range(B)], [priors[i % 32] for i in range(B)])
    s.batch_solve()
    s.set_profiling(True)
    t = time.perf_counter()
    for _ in range(3):
        s.batch_rewind(); s.batch_solve(sync=True)
    dt = (time.perf_counter() - t) / 3
    sm = s.batch_summaries(); prof = s.get_profile(); s.set_profiling(False)
    print(f"{tag:45s} ms/solve {dt*1e3:7.2f} mean its {np.mean([x.num_iterations for x in sm]):.2f}", {k: round(v['ms'] / 3, 2) for k, v in prof.items()})
