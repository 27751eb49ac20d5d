"""CPU tier: the stdout contract of bench.py.  Round 3's line had grown to 24 KB and the driver, which keeps the TAIL of stdout,
could not parse it (BENCH_r03.parsed == null).  bench.compact_line() is applied here to the full records of round 3
(profiles/bench_r03/*.json = what bench.py used to print) and must give one strict-JSON line of at most 4 KB that still carries
the contract's keys, `roofline` (+ rocprof cross-check), `roofline_by_time`, `cpu_baseline` and the per-cloud values."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "bench_r03", "*.json")))


def _strict(text):
    def bad(tok):
        raise ValueError(tok)
    return json.loads(text, parse_constant=bad)


@pytest.mark.parametrize("path", RECORDS, ids=[os.path.basename(p) for p in RECORDS])
def test_compact_line_of_a_full_record(path):
    import bench
    full = json.load(open(path))
    ms = full["ms_per_step"]
    full["roofline_by_time"] = bench.roofline_by_time(full["kernels"], ms)
    full["fps"] = bench.fps_summary(full["kernels"], ms)
    line = bench.compact_line(full, "bench_out/x.json")
    assert "\n" not in line and len(line.encode()) <= bench.COMPACT_LIMIT <= 4096
    r = _strict(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "roofline_by_time", "full_record"):
        assert key in r, key
    assert r["value"] == full["value"] and r["config"]["workload"]
    if full["roofline"] is not None:
        for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in r["roofline"], key
        assert r["roofline"]["frac"] == full["roofline"]["frac"]
    assert r["roofline_by_time"]["kernel"] in [k["kernel"] for k in full["kernels"]]
    name = os.path.basename(path)
    if name == "default.json":
        assert r["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and r["cpu_baseline"]["cores"] >= 1
        assert set(r["clouds"]) >= {"uniform", "kitti", "packed"}
        assert r["roofline"]["rocprof"]["avg_us"] > 0
        # where the TIME is, as opposed to where the work is: an RPN set-abstraction entry in round 3's record
        assert r["roofline_by_time"]["kernel"].startswith("rpn_sa")
    if name == "sa.json":
        # configs[1] is a step of FPS: the compact line alone must say so
        assert r["roofline_by_time"]["kernel"].startswith("fps_pyramid/") and r["roofline_by_time"]["us_per_fps_iteration"] > 0
        assert r["fps"]["chain_share_of_step"] > 0.5
    if name.startswith("train"):
        assert "grad_allreduce" in r


def test_fraction_sanitiser_removes_what_is_not_a_fraction():
    """VERDICT r4 (measurement #8): `bench_r04/ops.line.json` printed `roofline.rocprof.frac` 7.85.  Every `frac` / `*_frac` of a
    record passes through bench.sanitise_fractions before anything is printed: values outside (0, 1] are removed and listed"""
    import bench
    rec = {"roofline": {"frac": 0.73, "rocprof": {"frac": 7.85, "avg_us": 445.0}, "isolated": {"frac": 0.82}},
           "kernels": [{"kernel": "a", "hbm_frac": 0.4, "mfma_frac": 1.0}, {"kernel": "b", "hbm_frac": 0.0}, {"kernel": "c", "valu_frac": -0.1}],
           "step_mfma_frac": 0.5, "steps": 20}
    dropped = bench.sanitise_fractions(rec)
    assert sorted(p for p, _ in dropped) == ["/kernels[1]/hbm_frac", "/kernels[2]/valu_frac", "/roofline/rocprof/frac"]
    assert "frac" not in rec["roofline"]["rocprof"] and rec["roofline"]["rocprof"]["avg_us"] == 445.0
    assert rec["roofline"]["frac"] == 0.73 and rec["kernels"][0] == {"kernel": "a", "hbm_frac": 0.4, "mfma_frac": 1.0} and rec["steps"] == 20
    assert bench.sanitise_fractions(rec) == []
    # the round-4 `ops` record itself: its out-of-range cross-check is the one removed
    path = os.path.join(ROOT, "profiles", "bench_r04", "ops.json")
    if os.path.exists(path):
        full = json.load(open(path))
        gone = bench.sanitise_fractions(full)
        assert any(p.endswith("rocprof/frac") and v > 1 for p, v in gone), gone


def test_rocprof_cross_check_is_keyed_on_the_workload_and_the_live_time():
    """the committed per-shape table of the SAME workload only, and only a shape whose duration is within 3x of the live time"""
    import bench
    row = "affinity_8x128x128/affinity_forward_batched"
    hit = bench.rocprof_average(row, "detect")
    assert hit is not None and hit["workload"] == "detect" and hit["round"] in bench.PROFILE_ROUNDS and hit["avg_us"] > 0
    assert len(hit["shapes"]) == len(hit["kernels"]) and "x" in hit["shapes"][0]
    assert bench.rocprof_average(row, "detect", live_us=hit["avg_us"]) is not None
    assert bench.rocprof_average(row, "detect", live_us=hit["avg_us"] * 10) is None        # another dispatch: no cross-check
    assert bench.rocprof_average(row, "train") is None and bench.rocprof_average("no such row", "detect") is None
    # the dense RCNN SA1 entry of the `ops` workload never reads the detect profile's compacted dispatch
    ops = bench.rocprof_average("rcnn_sa1/sa_mlp_pm_forward", "ops", live_us=5100.0)
    assert ops is None or (ops["workload"] == "ops" and 1700.0 <= ops["avg_us"] <= 15300.0)


def test_rank_binding_splits_the_allowed_cores():
    import bench
    import torch
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity on this platform")
    before, threads = os.sched_getaffinity(0), torch.get_num_threads()
    try:
        n = len(before)
        got = bench.pin_rank(1 if n >= 2 else 0, 2 if n >= 2 else 1)
        assert got["pinned"] and got["n_cores"] == max(1, n // (2 if n >= 2 else 1)) and 1 <= got["torch_threads"] <= 16
        assert os.sched_getaffinity(0) <= before and len(os.sched_getaffinity(0)) == got["n_cores"]
        os.environ["JM_BENCH_NO_PIN"] = "1"
        assert bench.pin_rank(0, 2) == {"pinned": False}
    finally:
        os.environ.pop("JM_BENCH_NO_PIN", None)
        os.sched_setaffinity(0, before)
        torch.set_num_threads(threads)


def test_training_lines_say_what_they_time():
    """VERDICT r5 weak #4 / ADVICE r5 #1: every --workload train line names its mode, route, BatchNorm handling, loss and where its
    RoIs come from, and the workload text matches the route that runs"""
    import argparse
    import bench
    A = lambda **k: argparse.Namespace(**{"workload": "train", "joint": False, "rcnn": False, **k})     # noqa: E731
    os.environ.pop("JM_JOINT_ROUTE", None)
    j = bench.train_mode_keys(A(joint=True))
    assert j["route"] == "rows" and j["batchnorm"].startswith("frozen") and j["loss"].startswith("proxy") and "first-K" in j["proposals"]
    assert bench.workload_key(A(joint=True)) == "train_joint" and "row kernels" in bench.WORKLOAD_TEXT["train_joint"]
    assert "un-fused operator route" not in bench.WORKLOAD_TEXT["train_joint"].lower()
    os.environ["JM_JOINT_ROUTE"] = "operators"
    try:
        o = bench.train_mode_keys(A(joint=True))
        assert o["route"] == "operators" and o["batchnorm"].startswith("train mode")
        assert bench.workload_key(A(joint=True)) == "train_joint_operators"
    finally:
        os.environ.pop("JM_JOINT_ROUTE", None)
    r = bench.train_mode_keys(A(rcnn=True))
    assert "RPN.FIXED = True" in r["mode"] and r["loss"].startswith("proxy") and "frozen fused engine" in r["route"]
    assert bench.workload_key(A(rcnn=True)) == "train_rcnn"
    f = bench.train_mode_keys(A())
    assert f["mode"].startswith("finetune") and "proxy" not in f["loss"]
    assert bench.train_mode_keys(argparse.Namespace(workload="detect", joint=False, rcnn=False)) == {}
