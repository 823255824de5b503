"""dev: as dev_soak_probe.py, but the foreign kernels come from THIS process (torch, its own stream, a host thread): same-process sharing of the chip"""
import sys, os, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
import bench
from cogaps_amd import _capi
data = bench.synthetic_dense(20000, 2000)
stop = False
def foreign():
    st = torch.cuda.Stream()
    x = torch.ones(1 << 30, dtype=torch.float32, device="cuda")
    n = 0
    with torch.cuda.stream(st):
        while not stop:
            for _ in range(64): x.mul_(1.0000001)
            st.synchronize(); n += 64
    print("foreign kernels", n, flush=True)
th = threading.Thread(target=foreign); th.start()
time.sleep(2.0)
S = _capi.Session(data, nIterations=100, nPatterns=50, seed=42, outputFrequency=10)
for it in range(0, 120, 4):
    t0 = time.time()
    S.run_iterations(1 if it < 100 else 2, it if it < 100 else it - 100, 4)
    print(it, "%.2f s" % (time.time() - t0), {w: (S.chained(w), S.chain_recoveries(w), S.natoms(w)) for w in "AP"}, flush=True)
stop = True; th.join()
