"""Back-end adapters for vil_fusion_amd.sequence.SlidingWindowEstimator: the CPU oracle and the HIP path."""
import oracle_lib


class OracleBackend:
    def __init__(self, opts):
        self.o, self.prior = opts, None

    def solve(self, win):
        return oracle_lib.window_solve(self.o, win, self.prior)

    def align(self, inputs):
        from vil_fusion_amd import abi, synth
        return oracle_lib.visual_imu_alignment(self.o, abi.ImuNoise(synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W), **inputs)

    def marginalize(self, win, res):
        p = oracle_lib.window_marginalize(self.o, win, res, self.prior)
        self.prior = p if p.valid else None

    def reset(self):
        self.prior = None


class HipBackend:
    def __init__(self, solver):
        self.s = solver
        self.s.set_prior(None)

    def solve(self, win):
        return self.s.optimization(win)         # uses / keeps the device-resident prior of slot 0

    def align(self, inputs):
        from vil_fusion_amd import abi, synth
        from vil_fusion_amd.estimator import visual_imu_alignment
        return visual_imu_alignment(self.s, abi.ImuNoise(synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W), **inputs)

    def marginalize(self, win, res):
        self.s.marginalize()

    def reset(self):
        self.s.set_prior(None)
