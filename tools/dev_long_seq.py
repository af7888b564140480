"""long-horizon replay: S segments x N frames through the HIP path (lock-step batch) and through the C++ oracle loop (threads); prints timings + agreement"""
import sys, time, os, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from vil_fusion_amd import sequence
from vil_fusion_amd.estimator import BackendSolver

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 531
oracle_lib.build(); oracle_lib.lib()
opts = oracle_lib.default_options()
t = time.time()
seqs = []
for i in range(S):
    q = sequence.make_sequence(100 + i, N, opts, yaw_amplitude=None if i % 2 == 0 else 0.1 + 0.05 * i)
    if i == 3:
        q = sequence.drop_frames(q, N // 2, 7)
    seqs.append(q)
n = min(len(q["images"]) for q in seqs)
print("sequences", time.time() - t, "s; frames per segment", n, flush=True)

t = time.time()
refs = [oracle_lib.OracleSequence(opts) for _ in range(S)]
th = [threading.Thread(target=r.run, args=(q, n)) for r, q in zip(refs, seqs)]
[x.start() for x in th]; [x.join() for x in th]
print("oracle", time.time() - t, "s", [len(r.trajectory) for r in refs], flush=True)

t = time.time()
s = BackendSolver(opts)
ests = sequence.run_sequences_lockstep(seqs, opts, s, n)
print("hip lockstep", time.time() - t, "s", s.lockstep_stats, flush=True)
s.close()
for i, (a, b) in enumerate(zip(ests, refs)):
    ev = a.events == b.events; fl = a.flags == b.flags
    ia = [x["num_iterations"] for x in a.summaries]; ib = [x["num_iterations"] for x in b.summaries]
    m = min(len(a.trajectory), len(b.trajectory))
    dP = np.array([np.abs(a.trajectory[k][1] - b.trajectory[k][1]).max() for k in range(m)])
    inc = np.array([np.abs((a.trajectory[k + 1][1] - a.trajectory[k][1]) - (b.trajectory[k + 1][1] - b.trajectory[k][1])).max() for k in range(m - 1)])
    first = next((k for k in range(min(len(ia), len(ib))) if ia[k] != ib[k]), None)
    print(f"seg {i}: events {ev} flags {fl} iters {ia == ib} (first diff {first}) solved {len(a.trajectory)}/{len(b.trajectory)} reboots {a.n_reboots} "
          f"max dP {dP.max():.2e} at {dP.argmax()} max dInc {inc.max():.2e} sec_new {sum(a.flags)}", flush=True)
    print("    dP at", {k: "%.1e" % dP[k] for k in (1, 5, 20, 50, 100, 200, 300, 400, 500) if k < m})
    if first is not None:
        for k in (first - 1, first):
            print("    frame", k, "hip", {x: a.summaries[k][x] for x in ("num_iterations", "num_successful_steps", "termination", "initial_cost", "final_cost")},
                  "oracle", {x: b.summaries[k][x] for x in ("num_iterations", "num_successful_steps", "termination", "initial_cost", "final_cost")}, "dP", "%.2e" % dP[k])
