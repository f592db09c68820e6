"""How many CPUs this process may actually use.

A one-GPU box of the pool shows 256 CPUs (`os.cpu_count()`, the affinity mask) but its cgroup grants 16 (`cpu.max` =
"1600000 100000"); PyTorch sizes its intra-op pool from the former (128 threads), and 128 threads on a 16-core quota spend
their time being throttled: the oracle-bound GPU tests ran 5.8x slower than with 16 threads (85 s -> 14.6 s for
tests/test_gpu_smallunet.py + test_gpu_interunet.py).  tests/conftest.py, bench.py's cpu_baseline / dice_vs_oracle legs and
__graft_entry__.smoke() size the pool with this number.  No torch import here: the environment variables must be set first."""
import os


def usable_cpus():
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:                                                   # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    try:                                                   # cgroup v1
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0 and period > 0:
            n = min(n, max(1, quota // period))
    except (OSError, ValueError):
        pass
    return n


def size_thread_pools(env=os.environ):
    """Set OMP / MKL thread counts for this process and its children BEFORE torch is imported (no-op where the user set them);
    the ranks of one node (torchrun's LOCAL_WORLD_SIZE) share the quota."""
    n = usable_cpus()
    try:
        n = max(1, n // max(1, int(env.get("LOCAL_WORLD_SIZE", "1"))))
    except ValueError:
        pass
    env.setdefault("OMP_NUM_THREADS", str(n))
    env.setdefault("MKL_NUM_THREADS", str(n))
    return n
