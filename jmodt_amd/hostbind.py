"""Host-side placement of one-process-per-GPU ranks: each rank on cores of the NUMA node its GPU hangs off.

The reference runs ONE process for all GPUs (`nn.DataParallel`, tools/train.py:86-88).  The MI355X form is one process per GPU
(dist.py), and every rank's host thread enqueues 4 - 20 ms of launches per step: eight such threads, each with an intra-op pool,
must not share cores, and a rank whose cores sit on the other socket pays a cross-socket hop on every doorbell write and every
pinned-buffer copy.  An MI355X node has two sockets with four GPUs each; which four is read from sysfs, not assumed:

    /sys/bus/pci/devices/<dddd:bb:dd.f>/numa_node       the GPU's NUMA node (-1 = the platform does not say)
    /sys/devices/system/node/node<k>/cpulist            the node's logical CPUs
    /sys/devices/system/cpu/cpu<c>/topology/thread_siblings_list    SMT siblings (a rank gets WHOLE physical cores)

Everything takes a `sysfs` root so that tests/test_hostbind_cpu.py can run it on a fake tree.
"""
import glob
import os
from typing import Dict, List, Optional, Sequence


def parse_cpulist(text: str) -> List[int]:
    """"0-3,8,10-11" -> [0, 1, 2, 3, 8, 10, 11]"""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def format_bus_id(domain: int, bus: int, device: int, function: int = 0) -> str:
    return f"{domain:04x}:{bus:02x}:{device:02x}.{function:x}"


def torch_bus_ids(n: int) -> Optional[List[str]]:
    """PCI addresses of HIP devices 0 .. n-1 in HIP's enumeration order (what LOCAL_RANK indexes), or None without a GPU runtime"""
    try:
        import torch
        if not torch.cuda.is_available() or torch.cuda.device_count() < n:
            return None
        out = []
        for i in range(n):
            p = torch.cuda.get_device_properties(i)
            out.append(format_bus_id(int(getattr(p, "pci_domain_id", 0)), int(p.pci_bus_id), int(p.pci_device_id)))
        return out
    except Exception:        # noqa: BLE001 — a property this torch build does not carry: fall back to the drm enumeration
        return None


def drm_bus_ids(sysfs: str = "/sys") -> List[str]:
    """PCI addresses of the AMD (vendor 0x1002) display-class devices under /sys/class/drm, in bus order — HIP's default order;
    used only when torch cannot be asked"""
    ids = set()
    for dev in glob.glob(os.path.join(sysfs, "class", "drm", "card*", "device")):
        if os.path.basename(os.path.dirname(dev)).count("-"):        # card0-DP-1 connectors
            continue
        if (_read(os.path.join(dev, "vendor")) or "").lower() != "0x1002":
            continue
        real = os.path.realpath(dev)
        ids.add(os.path.basename(real))
    return sorted(ids)


def gpu_numa_nodes(bus_ids: Sequence[str], sysfs: str = "/sys") -> List[Optional[int]]:
    """NUMA node of each PCI device (None: unknown / -1)"""
    out: List[Optional[int]] = []
    for b in bus_ids:
        t = _read(os.path.join(sysfs, "bus", "pci", "devices", b, "numa_node"))
        try:
            v = int(t) if t is not None else -1
        except ValueError:
            v = -1
        out.append(v if v >= 0 else None)
    return out


def node_cpus(sysfs: str = "/sys") -> Dict[int, List[int]]:
    out: Dict[int, List[int]] = {}
    for d in glob.glob(os.path.join(sysfs, "devices", "system", "node", "node[0-9]*")):
        t = _read(os.path.join(d, "cpulist"))
        if t:
            out[int(os.path.basename(d)[4:])] = parse_cpulist(t)
    return out


def physical_order(cpus: Sequence[int], sysfs: str = "/sys") -> List[int]:
    """the CPUs ordered so that SMT siblings are neighbours (key: the lowest sibling): a contiguous split then hands out whole
    physical cores instead of putting rank 2 on the hyper-threads of rank 0's cores (cpulist "0-63,128-191")"""
    def key(c):
        t = _read(os.path.join(sysfs, "devices", "system", "cpu", f"cpu{c}", "topology", "thread_siblings_list"))
        try:
            first = min(parse_cpulist(t)) if t else c
        except ValueError:
            first = c
        return (first, c)
    return sorted(cpus, key=key)


def even_split(allowed: Sequence[int], local_rank: int, ranks_on_node: int) -> List[int]:
    allowed = list(allowed)
    per = max(1, len(allowed) // max(1, ranks_on_node))
    return allowed[local_rank * per:(local_rank + 1) * per] or allowed


def rank_cores(local_rank: int, ranks_on_node: int, allowed: Sequence[int], sysfs: str = "/sys",
               bus_ids: Optional[Sequence[str]] = None) -> Dict[str, object]:
    """{"cores": [...], "numa_node": k | None, "source": "numa" | "even-split", ...} — the cores of rank `local_rank`: the allowed
    cores of its GPU's NUMA node, split (whole physical cores, rank order) among the local ranks whose GPUs share that node; an
    even split of all allowed cores in rank order when the platform does not tell (no numa_node, a container without sysfs nodes)"""
    allowed = sorted(allowed)
    if bus_ids is None:
        bus_ids = torch_bus_ids(ranks_on_node)
    if bus_ids is None:
        ids = drm_bus_ids(sysfs)
        bus_ids = ids if len(ids) >= ranks_on_node else None
    fallback = {"cores": even_split(physical_order(allowed, sysfs), local_rank, ranks_on_node), "numa_node": None, "source": "even-split"}
    if bus_ids is None or local_rank >= len(bus_ids):
        return fallback
    nodes = gpu_numa_nodes(list(bus_ids)[:ranks_on_node], sysfs)
    cpus = node_cpus(sysfs)
    ok = set(allowed)
    # all or nothing: a node where only SOME GPUs report their NUMA node would mix the two schemes and hand a core to two ranks
    if any(n is None or not [c for c in cpus.get(n, ()) if c in ok] for n in nodes):
        return fallback
    node = nodes[local_rank]
    mine = [c for c in cpus[node] if c in ok]
    peers = [r for r in range(len(nodes)) if nodes[r] == node]
    cores = even_split(physical_order(mine, sysfs), peers.index(local_rank), len(peers))
    return {"cores": cores, "numa_node": node, "source": "numa", "gpu": bus_ids[local_rank], "ranks_on_this_node": len(peers)}
