"""Host-side hygiene: keep the CPU thread pools inside the container's CPU quota.

On a many-core host (the MI355X boxes show 256 CPUs) torch sizes its intra-op pool by the CPU count; a cgroup quota far below
that (cpu.max = 16 CPUs here) then throttles the WHOLE process -- the thread that launches kernels and polls the device
included -- for the rest of the 100 ms period whenever the pool's workers spin after a parallel region: 20-50 ms freezes that
showed up as "device stalls" in every latency measurement of the decode paths (cpu.stat: nr_throttled / throttled_usec).
"""
import os


def cpu_quota():
    """CPUs the cgroup lets this process use (cpu.max quota / period, v1 cfs files as a fallback), or None if unlimited"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return max(1, int(quota) // max(1, int(period)))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0:
            return max(1, quota // max(1, period))
    except (OSError, ValueError):
        pass
    return None


_done = False


def respect_cpu_quota(reserve=4):
    """cap torch's intra-op threads at (quota - reserve) CPUs (the runtime's own threads need the rest); idempotent.
    EMOASR_CPU_THREADS=<n> overrides, 0 leaves torch alone. -> the thread count in effect"""
    global _done
    import torch
    if _done:
        return torch.get_num_threads()
    _done = True
    env = os.environ.get("EMOASR_CPU_THREADS")
    if env is not None:
        if int(env) > 0:
            torch.set_num_threads(int(env))
        return torch.get_num_threads()
    q = cpu_quota()
    have = min(torch.get_num_threads(), len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())
    if q is not None:
        # one process per GPU: the ranks of this node share the quota
        local = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
        share = max(1, q // max(1, local))
        want = max(1, min(have, share - reserve if share > 2 * reserve else max(1, share // 2)))
        if want < torch.get_num_threads():
            torch.set_num_threads(want)
    return torch.get_num_threads()
