import os, sys, time
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd")); sys.path.insert(0, REPO)
import starkcore as sc
import workloads
import fast_stark
from fast_stark import DeviceTrace, FastStark
log_fri = int(sys.argv[1]); mode = sys.argv[2]
if mode == "noprefetch":
    fast_stark.prefetch_random_polynomial = lambda *a, **k: None
sc.init(0)
field, T, packed, air, boundary = workloads.synthetic_stark_instance(log_fri, 40)
stark = FastStark(field, 4, 40, 80, 2, T)
trace = DeviceTrace.from_packed(packed, field)
tz, tzc, root = stark.preprocess(device_resident=True)
times = []
for _ in range(12):
    sc.synchronize(); t0 = time.perf_counter()
    proof = stark.prove(trace, air, boundary, tz, tzc)
    times.append((time.perf_counter() - t0) * 1e3)
print(f"fri 2^{log_fri} {mode} RAND_THREADS={os.environ.get('STARKCORE_RAND_THREADS','-')} nproc={os.cpu_count()} affinity={len(os.sched_getaffinity(0))}: best {min(times):.3f} median {sorted(times)[6]:.3f}")
