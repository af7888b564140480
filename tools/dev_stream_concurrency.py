import torch, time
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(two):
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.cuda.stream(s1): torch.cuda._sleep(200_000_000)
    with torch.cuda.stream(s2 if two else s1): torch.cuda._sleep(200_000_000)
    torch.cuda.synchronize(); return time.perf_counter() - t
run(True)
print("same stream: %.3f s, two streams: %.3f s" % (run(False), run(True)))
import os; print({k: v for k, v in os.environ.items() if 'QUEUE' in k or 'HIP_' in k or 'GPU_' in k or 'HSA_' in k})
