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
