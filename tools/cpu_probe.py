import os, time, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("OMP_PROC_BIND", os.environ.get("BIND", "spread"))
os.environ.setdefault("OMP_PLACES", "cores")
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA node\\(s\\)' ; nproc")
from ilqr_iterative_tasks_amd import workloads
from oracle import oracle as orc
cfg = workloads.config_for("config2", "f64")
host = workloads.make_batch(cfg, 65536)
def run(n):
    sl = slice(0, n)
    t0 = time.perf_counter()
    orc.ilqr_batch(cfg, host["X"][sl], host["U"][sl], host["x_term"][sl], host["lamb"][sl], host["obs"][sl], max_iter=10, early_exit=False, want_gains=True)
    return time.perf_counter() - t0
for nt in (1, 8, 16, 32, 64, 128, 256):
    orc.set_threads(nt)
    n = min(65536, 2048 * nt)
    run(n)
    t = min(run(n) for _ in range(2))
    print(f"threads {nt:4d}: {n*10/t/1e6:8.3f} M it/s  ({n} problems, {t:.3f} s)", flush=True)
