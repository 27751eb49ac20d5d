"""CPU tier: rank -> host-core placement of the one-process-per-GPU launch (jmodt_amd/hostbind.py, bench.pin_rank) on FAKE sysfs
trees — the 8-GPU node is not available to the builder, its topology is: two sockets, four GPUs each, SMT on."""
import os

import pytest

from jmodt_amd import hostbind


def _write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text + "\n")


def fake_node(root, gpu_nodes, node_cpulists, smt_offset=None, drm=True):
    """a sysfs tree: GPUs at 0000:(10 + 16 i):00.0 with the given numa_node values, NUMA nodes with the given cpulists, SMT siblings
    (c, c + smt_offset)"""
    ids = []
    for i, node in enumerate(gpu_nodes):
        bid = hostbind.format_bus_id(0, 0x10 + 0x10 * i, 0)
        ids.append(bid)
        dev = os.path.join(root, "bus", "pci", "devices", bid)
        _write(os.path.join(dev, "numa_node"), str(node))
        _write(os.path.join(dev, "vendor"), "0x1002")
        if drm:
            os.makedirs(os.path.join(root, "class", "drm", f"card{i}"), exist_ok=True)
            os.symlink(dev, os.path.join(root, "class", "drm", f"card{i}", "device"))
    all_cpus = []
    for k, text in node_cpulists.items():
        _write(os.path.join(root, "devices", "system", "node", f"node{k}", "cpulist"), text)
        all_cpus += hostbind.parse_cpulist(text)
    if smt_offset:
        phys = [c for c in all_cpus if c < smt_offset]
        for c in phys:
            for t in (c, c + smt_offset):
                _write(os.path.join(root, "devices", "system", "cpu", f"cpu{t}", "topology", "thread_siblings_list"), f"{c},{c + smt_offset}")
    return ids, sorted(all_cpus)


def test_cpulist_parsing():
    assert hostbind.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    assert hostbind.parse_cpulist("") == [] and hostbind.parse_cpulist("5") == [5]


def test_two_sockets_four_gpus_each_whole_physical_cores(tmp_path):
    """the MI355X node: node0 = CPUs 0-63 + their hyper-threads 128-191 with GPUs 0-3, node1 = 64-127 + 192-255 with GPUs 4-7.
    Every rank gets 16 physical cores of ITS socket with both hyper-threads, no CPU belongs to two ranks, all 256 are used"""
    root = str(tmp_path)
    ids, cpus = fake_node(root, [0, 0, 0, 0, 1, 1, 1, 1], {0: "0-63,128-191", 1: "64-127,192-255"}, smt_offset=128)
    seen = set()
    for r in range(8):
        got = hostbind.rank_cores(r, 8, cpus, root, ids)
        assert got["source"] == "numa" and got["numa_node"] == r // 4 and got["ranks_on_this_node"] == 4
        cores = got["cores"]
        assert len(cores) == 32 and not (set(cores) & seen)
        seen |= set(cores)
        phys = sorted(c for c in cores if c < 128)
        assert phys == list(range(16 * r, 16 * r + 16))                     # its socket's cores, a contiguous block
        assert sorted(c - 128 for c in cores if c >= 128) == phys           # with BOTH hyper-threads of each
    assert seen == set(cpus)


def test_gpu_order_that_does_not_follow_the_sockets(tmp_path):
    """HIP device i is not on socket i // 4 everywhere: interleaved GPUs still land on their own socket"""
    root = str(tmp_path)
    ids, cpus = fake_node(root, [1, 0, 1, 0], {0: "0-7", 1: "8-15"})
    got = [hostbind.rank_cores(r, 4, cpus, root, ids) for r in range(4)]
    assert [g["numa_node"] for g in got] == [1, 0, 1, 0]
    assert [g["cores"] for g in got] == [[8, 9, 10, 11], [0, 1, 2, 3], [12, 13, 14, 15], [4, 5, 6, 7]]


def test_restricted_affinity_and_unknown_numa_fall_back_consistently(tmp_path):
    root = str(tmp_path)
    ids, cpus = fake_node(root, [0, 1], {0: "0-7", 1: "8-15"})
    # a cgroup that allows part of each node: the split stays inside what is allowed
    got = [hostbind.rank_cores(r, 2, [2, 3, 4, 5, 10, 11], root, ids) for r in range(2)]
    assert got[0]["cores"] == [2, 3, 4, 5] and got[1]["cores"] == [10, 11]
    # ... none of node 1: all-or-nothing, an even split of the allowed cores in rank order (no core twice)
    got = [hostbind.rank_cores(r, 2, [0, 1, 2, 3], root, ids) for r in range(2)]
    assert [g["source"] for g in got] == ["even-split"] * 2 and got[0]["cores"] == [0, 1] and got[1]["cores"] == [2, 3]
    # numa_node = -1 (one-socket boxes, VMs): even split
    root2 = str(tmp_path / "b")
    ids2, cpus2 = fake_node(root2, [-1, -1], {0: "0-15"})
    got = [hostbind.rank_cores(r, 2, cpus2, root2, ids2) for r in range(2)]
    assert [g["source"] for g in got] == ["even-split"] * 2 and not (set(got[0]["cores"]) & set(got[1]["cores"]))
    # no sysfs at all
    got = hostbind.rank_cores(1, 2, list(range(8)), str(tmp_path / "nothing"), [])
    assert got["source"] == "even-split" and got["cores"] == [4, 5, 6, 7]


def test_drm_enumeration_is_the_fallback_for_the_bus_ids(tmp_path):
    root = str(tmp_path)
    ids, _ = fake_node(root, [0, 1], {0: "0-3", 1: "4-7"})
    assert hostbind.drm_bus_ids(root) == ids
    assert hostbind.gpu_numa_nodes(ids, root) == [0, 1]


def test_bench_pin_rank_binds_to_the_gpus_numa_node(tmp_path):
    """bench.pin_rank over a fake tree built from THIS host's allowed cores: rank 1 of 2 lands on the second half, the line says why"""
    import bench
    import torch
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity on this platform")
    before, threads = os.sched_getaffinity(0), torch.get_num_threads()
    allowed = sorted(before)
    if len(allowed) < 2:
        pytest.skip("one core")
    half = len(allowed) // 2
    root = str(tmp_path)
    lists = {0: ",".join(map(str, allowed[:half])), 1: ",".join(map(str, allowed[half:]))}
    ids, _ = fake_node(root, [1, 0], lists)
    try:
        got = bench.pin_rank(0, 2, sysfs=root, bus_ids=ids)
        assert got["pinned"] and got["source"] == "numa" and got["numa_node"] == 1 and got["n_cores"] == len(allowed) - half
        assert sorted(os.sched_getaffinity(0)) == allowed[half:]
        assert got["cores"] == bench.hostbind_ranges(allowed[half:])
    finally:
        os.sched_setaffinity(0, before)
        torch.set_num_threads(threads)
    assert bench.hostbind_ranges([0, 1, 2, 3, 64, 65, 70]) == "0-3,64-65,70"
