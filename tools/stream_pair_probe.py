import ctypes, os, sys, time
REPO = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import numpy as np, torch
import starkcore as sc, synth
from workloads import nth_root
n = 1 << 20
sc.init(0); lib = sc.lib(); dev = torch.device("cuda", 0)
root = sc.fe_bytes(nth_root(n))
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
junk = [torch.cuda.Stream(device=dev) for _ in range(pre)]      # streams created before ours (like the bench's)
streams = [torch.cuda.Stream(device=dev) for _ in range(6)]
print("stream handles", [hex(s.cuda_stream) for s in streams])
bufs = []
for k in range(2):
    x = torch.from_numpy(synth.synth_packed(1 + k, n).view(np.int64)).to(dev)
    bufs.append((x, torch.empty_like(x), torch.empty_like(x)))
torch.cuda.synchronize()
def pair(k, s):
    x, y, z = bufs[k]
    p = ctypes.c_void_p(s.cuda_stream)
    sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, p))
    sc._check(lib.sc_ntt_dev(y.data_ptr(), z.data_ptr(), n, root, 1, p))
def run(a, b):
    for i in range(40): pair(i & 1, streams[a if i & 1 else b])
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(300): pair(i & 1, streams[a if i & 1 else b])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 300
        best = dt if best is None or dt < best else best
    return best * 1e6
for a in range(6):
    print("pair with stream", a, ["%.1f" % run(a, b) for b in range(6)])
