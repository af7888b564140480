"""phase timing of vilf_batch_upload (VILF_DEBUG_TIMING=1) for 2048 windows, repeated uploads into the same handle (the bench's pcie_inclusive leg)"""
import os, sys, time
os.environ["VILF_DEBUG_TIMING"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vil_fusion_amd import synth, abi
from vil_fusion_amd.estimator import BackendSolver
s = BackendSolver()
wins, priors = synth.make_batch(1000, 2048, s.options, synth.SynthConfig(n_features=230), distinct=64)
s.batch_upload(wins, priors)
parr = (abi.WindowIn * 2048)()
for i in range(2048): parr[i] = wins[i].as_struct()
for k in range(3):
    print("---- upload", k, file=sys.stderr)
    t0 = time.perf_counter(); s._check(s._L.vilf_batch_upload(s._h, 2048, parr), "up"); t1 = time.perf_counter()
    print("total ms", 1e3 * (t1 - t0), file=sys.stderr)
    s.batch_solve()
