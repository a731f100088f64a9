"""Host-side placement for the process that drives a GPU.

The launching thread, the pinned staging buffers and the loader threads of the matching path talk to the GPU all the time
(a thousand launches, a few small copies in each direction and several event waits per batch).  On a two-socket host
it can matter which socket they run on: on the MI355X test boxes (2 x EPYC 9575F, the GPU behind node 1; shared hosts,
load average 11-17) runs started on the far socket measured 588-662 pairs/s against 677-680 on the near one in one visit
and no difference in another (profiles/r03_host_effects.txt).  `pin_process_to_gpu` moves
every thread of the process onto the CPUs that sysfs lists as local to the GPU's PCIe function.  Explicit, never done
behind the caller's back: bench.py and the tools call it; an application decides for itself.
"""
import os

import torch


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(index=0):
    """CPUs on the NUMA node of GPU `index` (sysfs local_cpulist of its PCIe function); empty set if unknown."""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        return _parse_cpulist(open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read())
    except (OSError, AttributeError, RuntimeError, ValueError):
        return set()


def pin_process_to_gpu(index=0):
    """Restrict every existing thread of this process (new ones inherit) to the CPUs local to GPU `index`, within the
    affinity the process already has.  Returns the CPU set used, or None when nothing was changed."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    local = gpu_local_cpus(index)
    cpus = (local & os.sched_getaffinity(0)) or local          # (an affinity mask that excludes the GPU's node is overridden if the
    if not cpus:                                               #  cpuset allows it; otherwise the calls below fail and nothing changes)
        return None
    changed = False
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), cpus)
            changed = True
        except (OSError, ValueError):
            pass
    return cpus if changed else None
