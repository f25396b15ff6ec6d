"""Host-side facts a benchmark run depends on: CPU model, the CPU budget of the container, physical cores, and which CPUs
are local to a GPU (what `rocm-smi --showtoponuma` prints, read from sysfs) - used by bench.py to pin each rank of a
multi-GPU run to its GPU's NUMA node and to say in the bench line what it ran on.  No GPU work, no oracle."""
from __future__ import annotations

import os


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_budget():
    """CPUs this process may use: the affinity mask, cut by the container's CPU quota (cgroup v2 cpu.max / v1 cfs quota)"""
    try:
        n = float(len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        n = float(os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse is None:
                q = float(txt)
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip())
                quota = q / per if q > 0 else None
            else:
                quota = parse(txt)
            if quota:
                n = min(n, quota)
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return n


def _parse_cpulist(txt):
    cpus = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def _format_cpulist(cpus):
    cpus, out, i = sorted(cpus), [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append("%d" % cpus[i] if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def gpu_local_cpus(pci_bdf, sysfs="/sys/bus/pci/devices"):
    """(NUMA node, CPUs local to it) of the PCI device `pci_bdf` ("0000:c1:00.0"), from sysfs - the data
    `rocm-smi --showtoponuma` prints; (None, empty set) when sysfs does not say."""
    base = os.path.join(sysfs, pci_bdf)
    try:
        node = int(open(os.path.join(base, "numa_node")).read().strip())
    except (OSError, ValueError):
        node = None
    try:
        cpus = _parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
    except (OSError, ValueError):
        cpus = set()
    return (node if node is not None and node >= 0 else None), cpus


def rank_cpu_slice(local_cpus, allowed, ranks_sharing, index):
    """CPUs for one of `ranks_sharing` ranks whose GPUs sit on the same NUMA node: an equal contiguous share of the
    node's CPUs this process may use (whole cores stay together when SMT siblings are numbered n and n + N/2 only by
    luck - the share is by CPU number).  Empty when there is nothing to slice."""
    cpus = sorted(set(local_cpus) & set(allowed))
    if not cpus or ranks_sharing < 1:
        return set()
    per = len(cpus) // ranks_sharing
    if per < 2:                       # fewer than 2 CPUs per rank: leave the scheduler alone
        return set()
    return set(cpus[index * per:(index + 1) * per])


def pin_rank_to_gpu_numa_node(local_rank, world):
    """N > 1: keep this rank's threads (8 proof drivers + their pollers; ~3.4 busy CPUs at full rate) on the CPUs of its
    GPU's NUMA node, an equal share per rank of that node - the host work of a proof touches page-locked staging memory
    the GPU reads.  Returns what was done, for the bench line.  LMN_BENCH_AFFINITY=0 switches it off."""
    info = {"pinned": False}
    if world < 2 or os.environ.get("LMN_BENCH_AFFINITY", "1") == "0":
        info["reason"] = "off" if world > 1 else "single rank"
        return info
    try:
        import torch
        allowed = os.sched_getaffinity(0)
        n_gpus = torch.cuda.device_count()
        nodes = {}
        for g in range(n_gpus):
            pr = torch.cuda.get_device_properties(g)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            nodes[g] = gpu_local_cpus(bdf) + (bdf,)
        if local_rank not in nodes:
            info["reason"] = "no such GPU"
            return info
        node, cpus, bdf = nodes[local_rank]
        info.update({"gpu_pci": bdf, "numa_node": node})
        if not cpus:
            info["reason"] = "sysfs has no local_cpulist for the GPU"
            return info
        sharing = sorted(g for g in range(min(world, n_gpus)) if nodes[g][1] == cpus)
        mine = rank_cpu_slice(cpus, allowed, len(sharing), sharing.index(local_rank))
        if not mine:
            info["reason"] = "fewer than 2 usable CPUs per rank on the GPU's node"
            return info
        os.sched_setaffinity(0, mine)
        info.update({"pinned": True, "cpus": _format_cpulist(mine), "ranks_on_node": len(sharing)})
    except Exception as e:  # noqa: BLE001 - affinity is an optimisation, never a reason to fail the run
        info["reason"] = "%s: %s" % (type(e).__name__, e)
    return info


def physical_cores():
    """(physical id, core id) pairs of /proc/cpuinfo; half the logical CPUs if that cannot be read"""
    try:
        seen, phys = set(), None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("physical id"):
                    phys = ln.split(":", 1)[1].strip()
                elif ln.startswith("core id"):
                    seen.add((phys, ln.split(":", 1)[1].strip()))
        if seen:
            return len(seen)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)
