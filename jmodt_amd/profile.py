"""Per-entry-point timing with HIP events + the ALGORITHMIC work of every C-ABI call.

`_lib.load()` hands out a proxy of libjmodt_hip.so; while the profiler is enabled every `jm_*` call is
bracketed by two events recorded on the stream the call launches on (torch's current stream — the ops pass
exactly that stream to the library), and its algorithmic bytes / flops are computed from the call's own
integer arguments with the formulas of SURVEY.md §8(d).  Nothing else changes: same library, same kernels,
same stream.  Disabled (the default) the proxy forwards straight to ctypes.

bench.py reads `summary()`; `scope("rpn_sa1")` prefixes the records made inside it so the same entry point
at different pyramid levels stays distinguishable; `region()` times a span of non-library work (MIOpen /
rocBLAS calls of the caller) the same way so the step's time is fully accounted for.
"""
import contextlib
import ctypes
from typing import Callable, Dict, List, Tuple

import torch


def _i(a, k):
    v = a[k]
    return int(v.value) if hasattr(v, "value") else int(v)


def _mlp3(arg):
    s = getattr(arg, "_obj", None)
    return (int(s.c), int(s.h1), int(s.h2)) if s is not None else None


def _mlp3_flops(rows, dims):
    c, h1, h2 = dims
    return rows * (2 * c * h1 + 2 * h1 * h2 + 2 * h2)


def _train_flops(rows, dims):
    c, h1, h2 = dims
    return rows * (2 * c * h1 + 2 * h1 * h2 + 2 * h2) + rows * (2 * h2 * h1 + 2 * h2 * h1 + 2 * h1 * c)


def _fps(a):
    b, n, m = _i(a, 0), _i(a, 1), _i(a, 2)
    # streaming-equivalent bytes (what the reference re-reads per iteration); the compulsory
    # bytes b*(12n+4m) and the iteration count ride along in `extra`
    return b * m * 20 * n, 0, dict(compulsory_bytes=b * (12 * n + 4 * m), evals=b * m * n, iterations=max(m - 1, 1))


def _sa_mlp(a):
    b, n, m, c, ns, nl = _i(a, 0), _i(a, 1), _i(a, 2), _i(a, 3), _i(a, 4), _i(a, 9)
    w = [int(a[10][k]) for k in range(nl + 1)]
    flops = 2 * b * m * ns * sum(w[k] * w[k + 1] for k in range(nl))
    return b * (12 * n + 12 * m + 4 * c * n + 4 * m * ns + 4 * w[-1] * m), flops, {}


def _sa_mlp_listed(a):
    """the dense block's algorithmic work + what one executed row costs: the rows executed (2^q per group of
    class q) live in device memory (fused.ListedStats; bench.py multiplies)"""
    nl = _i(a, 9)
    w = [int(a[10][k]) for k in range(nl + 1)]
    # no flops / bytes from the arguments: the dense block's figure is rows_dense x flops_per_row and the executed one
    # rows_executed x flops_per_row, both from the plan's counters (bench.py) — the arguments alone only know the capacity
    return 0, 0, dict(flops_per_row=2 * sum(w[k] * w[k + 1] for k in range(nl)), bytes_per_row=4 * (w[0] + 1), listed=True)


def _sa_mlp_pre(a):
    b, n, m, c, ns, nl = _i(a, 0), _i(a, 1), _i(a, 2), _i(a, 3), _i(a, 4), _i(a, 9)
    w = [int(a[10][k]) for k in range(nl + 1)]
    flops = 2 * b * m * ns * sum(w[k] * w[k + 1] for k in range(nl))
    return b * (4 * c * n + 12 * m + 4 * m * ns + 4 * w[-1] * m), flops, {}


def _roipool(a):
    B, N, M, C, S = (_i(a, k) for k in range(5))
    return B * (12 * N + 28 * M + 4 * C * N) + B * M * S * (3 + C) * 4 + 4 * B * M, 0, dict(evals=B * M * N)


def _nms_bytes(n):
    return n * 20 + n * ((n + 63) // 64) * 8


def _affinity(a):
    p, d = _i(a, 0), _i(a, 1)
    link, se = _mlp3(a[4]), _mlp3(a[5]) if a[5] is not None else None
    flops = _mlp3_flops(p * d, link) + (_mlp3_flops(p + d, se) if se else 0)
    nbytes = (p + d) * link[0] * 4 + 4 * (link[0] * link[1] + link[1] * link[2] + link[2]) * (2 if se else 1) + 4 * p * d
    return nbytes, flops, {}


# symbol -> f(args) -> (algorithmic bytes, flops, extra)
ALGO: Dict[str, Callable] = {
    "jm_furthest_point_sampling": _fps,
    "jm_furthest_point_sampling_xyz": _fps,
    "jm_furthest_point_sampling_ws": _fps,
    "jm_gather_points": lambda a: (_i(a, 0) * (4 * _i(a, 3) + 4 * _i(a, 1) * _i(a, 2) + 4 * _i(a, 1) * _i(a, 3)), 0, {}),
    "jm_ball_query": lambda a: (_i(a, 0) * (12 * _i(a, 1) + 12 * _i(a, 2) + 4 * _i(a, 2) * _i(a, 4)), 0,
                                dict(evals=_i(a, 0) * _i(a, 1) * _i(a, 2))),
    "jm_ball_query_dual": lambda a: (_i(a, 0) * (12 * _i(a, 1) + 12 * _i(a, 2) + 4 * _i(a, 2) * (_i(a, 4) + _i(a, 6))), 0,
                                     dict(evals=_i(a, 0) * _i(a, 1) * _i(a, 2))),
    # grid search: same compulsory bytes; the number of evaluated candidates is data dependent (not known on the host)
    "jm_ball_query_ws": lambda a: (_i(a, 0) * (12 * _i(a, 1) + 12 * _i(a, 2) + 4 * _i(a, 2) * _i(a, 4)), 0,
                                   dict(brute_force_evals=_i(a, 0) * _i(a, 1) * _i(a, 2))),
    "jm_ball_query_dual_ws": lambda a: (_i(a, 0) * (12 * _i(a, 1) + 12 * _i(a, 2) + 4 * _i(a, 2) * (_i(a, 4) + _i(a, 6))), 0,
                                        dict(brute_force_evals=_i(a, 0) * _i(a, 1) * _i(a, 2))),
    "jm_group_points": lambda a: (_i(a, 0) * (4 * _i(a, 3) * _i(a, 4) + 4 * _i(a, 1) * _i(a, 2)
                                              + 4 * _i(a, 1) * _i(a, 3) * _i(a, 4)), 0, {}),
    "jm_three_nn": lambda a: (_i(a, 0) * (12 * _i(a, 1) + 12 * _i(a, 2) + 24 * _i(a, 1)), 0,
                              dict(evals=_i(a, 0) * _i(a, 1) * _i(a, 2))),
    "jm_three_nn_ws": lambda a: (_i(a, 0) * (12 * _i(a, 1) + 12 * _i(a, 2) + 24 * _i(a, 1)), 0,
                                 dict(brute_force_evals=_i(a, 0) * _i(a, 1) * _i(a, 2))),
    "jm_three_interpolate": lambda a: (_i(a, 0) * (4 * _i(a, 1) * _i(a, 2) + 24 * _i(a, 3) + 4 * _i(a, 1) * _i(a, 3)), 0, {}),
    "jm_sa_mlp_forward": _sa_mlp,
    "jm_sa_mlp_forward_pre": _sa_mlp_pre,
    "jm_sa_mlp_forward_listed": _sa_mlp_listed,
    "jm_roipool3d_forward": _roipool,
    "jm_roipool3d_canonical": _roipool,
    "jm_nms": lambda a: (_nms_bytes(_i(a, 0)), 0, dict(evals=_i(a, 0) * _i(a, 0) // 2)),
    "jm_nms_batched": lambda a: (_i(a, 0) * _nms_bytes(_i(a, 1)), 0, dict(evals=_i(a, 0) * _i(a, 1) * _i(a, 1) // 2)),
    "jm_proposal_select": lambda a: (_i(a, 0) * (_i(a, 1) * 40 + _nms_bytes(int(_i(a, 6) * 0.7)) + _nms_bytes(_i(a, 6) - int(_i(a, 6) * 0.7))),
                                     0, {}),
    "jm_decode_rpn_proposals": lambda a: (_i(a, 0) * (_i(a, 1) + 3 + 7) * 4, 0, {}),
    "jm_decode_rcnn_boxes": lambda a: (_i(a, 0) * (_i(a, 1) + 7 + 7) * 4, 0, {}),
    "jm_feature_gather": lambda a: (_i(a, 0) * _i(a, 4) * 4 * _i(a, 1) * 4 + _i(a, 0) * _i(a, 1) * _i(a, 4) * 4, 0, {}),
    "jm_attention_fusion_forward": lambda a: (
        _i(a, 0) * _i(a, 1) * 4 * (_i(a, 2) + _i(a, 3) + _i(a, 5)),
        2 * _i(a, 0) * _i(a, 1) * (_i(a, 4) * (_i(a, 2) + _i(a, 3) + 1) + _i(a, 3) * _i(a, 2) + 2 * _i(a, 3) * _i(a, 5)), {}),
    "jm_image_fusion_gather": lambda a: (
        _i(a, 0) * _i(a, 1) * 4 * 4 * sum(int(a[6][k]) for k in range(_i(a, 5))) + _i(a, 0) * _i(a, 1) * _i(a, 4) * 4,
        2 * _i(a, 0) * _i(a, 1) * 4 * 32 * sum(int(a[6][k]) for k in range(_i(a, 5))), {}),
    "jm_rcnn_lift_forward": lambda a: (
        _i(a, 0) * _i(a, 1) * 4 * (_i(a, 2) + _i(a, 3) + (_i(a, 7) or _i(a, 6))),
        2 * _i(a, 0) * _i(a, 1) * (_i(a, 2) * _i(a, 4) + _i(a, 4) * _i(a, 5) + (_i(a, 5) + _i(a, 3)) * _i(a, 6)
                                   + (_i(a, 6) + 3) * _i(a, 7)), {}),
    "jm_bias_relu_channels_last": lambda a: (8 * _i(a, 0), 0, {}),
    "jm_affinity_forward": _affinity,
    "jm_affinity_forward_batched": lambda a: (
        _i(a, 0) * ((_i(a, 1) + _i(a, 2)) * _mlp3(a[5])[0] * 4 + 4 * _i(a, 1) * _i(a, 2)),
        _mlp3_flops(_i(a, 0) * _i(a, 1) * _i(a, 2), _mlp3(a[5])), {}),
    "jm_affinity_start_end_batched": lambda a: (0, _mlp3_flops(_i(a, 0) * (_i(a, 1) + _i(a, 2)), _mlp3(a[5])), {}),
    "jm_affinity_start_end": lambda a: (0, _mlp3_flops(_i(a, 0) + _i(a, 1), _mlp3(a[4])), {}),
    "jm_conv1d_stack_forward": lambda a: (
        4 * _i(a, 0) * _i(a, 1) * (_i(a, 2) + _i(a, 4) + int(a[8][_i(a, 7) - 1])),
        2 * _i(a, 0) * _i(a, 1) * sum(k * int(a[8][l]) for l, k in enumerate([_i(a, 2) + _i(a, 4)] + [int(a[8][j]) for j in range(_i(a, 7) - 1)])), {}),
    "jm_argsort_desc_stable": lambda a: (12 * _i(a, 0) * _i(a, 1), 0, {}),
    "jm_conv3x3_rgb_bias_relu": lambda a: (4 * _i(a, 0) * _i(a, 1) * _i(a, 2) * (3 + _i(a, 3)), 0, {}),
    # algorithmic = the direct form's 2 * 9 * cin * cout per pixel; executed on the matrix cores = 16 / 36 of it (F(2x2, 3x3))
    "jm_conv3x3_wino_bias_relu": lambda a: (
        4 * _i(a, 0) * _i(a, 1) * _i(a, 2) * (_i(a, 3) + _i(a, 4)) + 64 * _i(a, 3) * _i(a, 4),
        18 * _i(a, 0) * _i(a, 1) * _i(a, 2) * _i(a, 3) * _i(a, 4),
        dict(executed_flops=8 * _i(a, 0) * _i(a, 1) * _i(a, 2) * _i(a, 3) * _i(a, 4))),
    "jm_sa_mlp_pm_forward": lambda a: (
        4 * _i(a, 0) * (_i(a, 2) * _i(a, 4) * (_i(a, 3) + 1) + _i(a, 2) * _i(a, 6)),
        2 * _i(a, 0) * _i(a, 2) * _i(a, 4) * (_i(a, 3) * _i(a, 5) + _i(a, 5) * _i(a, 6)), {}),
    "jm_sa_mlp_pm_forward_listed": lambda a: (
        0, 0, dict(flops_per_row=2 * (_i(a, 3) * _i(a, 5) + _i(a, 5) * _i(a, 6)), bytes_per_row=4 * (_i(a, 3) + 1), listed=True)),
    # duplicate-compacted form: the row count lives in device memory (bench.py multiplies by the rows it reads back)
    "jm_sa_mlp_pm_forward_dyn": lambda a: (0, 0, dict(flops_per_row=2 * (_i(a, 2) * _i(a, 4) + _i(a, 4) * _i(a, 5)),
                                                      bytes_per_row=4 * (_i(a, 2) + 1))),
    "jm_linear_rows": lambda a: (4 * (_i(a, 0) * (_i(a, 1) + _i(a, 2)) + _i(a, 1) * _i(a, 2)), 2 * _i(a, 0) * _i(a, 1) * _i(a, 2), {}),
    "jm_mlp3_forward": lambda a: (0, _mlp3_flops(_i(a, 0), _mlp3(a[2])), {}),
    # a16: forward (3 layers) + backward (dH1, dW2, dW1) GEMMs of a head on M rows
    "jm_affinity_train_link_step": lambda a: (0, _train_flops(_i(a, 0) * _i(a, 1) * _i(a, 1), _mlp3(a[9])), {}),
    "jm_affinity_train_se_step": lambda a: (0, _train_flops(_i(a, 0) * 2 * _i(a, 1), _mlp3(a[11])), {}),
    "jm_association_cost": lambda a: ((_i(a, 0) + _i(a, 2)) * 28 + 4 * _i(a, 0) * _i(a, 2), 0, {}),
    "jm_boxes_overlap_bev": lambda a: ((_i(a, 0) + _i(a, 2)) * 20 + 4 * _i(a, 0) * _i(a, 2), 0, {}),
    "jm_boxes_iou_bev": lambda a: ((_i(a, 0) + _i(a, 2)) * 20 + 4 * _i(a, 0) * _i(a, 2), 0, {}),
    "jm_boxes_iou3d_batched": lambda a: (_i(a, 0) * ((_i(a, 1) + _i(a, 3)) * 28 + 4 * _i(a, 1) * _i(a, 3)), 0, {}),
}


VALU_F32_PEAK_TF = 157.3      # fp32 vector peak (256 CUs x 4 SIMD x 2.4 GHz x 64 flops per cycle)


class Profiler:
    def __init__(self):
        self.enabled = False
        self.records: Dict[str, List[Tuple]] = {}   # name -> [(start, end, bytes, flops, extra)]
        self._scope: List[str] = []
        self._hoisted = 0
        self.only = None     # set of record names: time ONLY these (two HIP event records per timed call cost GPU time:
                             # ~0.9 ms of a 21 ms composed step when every entry is timed)

    def reset(self):
        self.records = {}

    @contextlib.contextmanager
    def scope(self, name: str):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def hoisted_flops(self, flops: int):
        """the next jm_* call evaluates an operator part of whose ALGORITHMIC work (2*rows*sum c_in*c_out of
        the whole SA block) was restructured away (first layer hoisted in front of the gather): count it as algorithmic
        work of that call, and keep what the kernel really executes in `executed_flops`"""
        if self.enabled:
            self._hoisted = int(flops)

    def _key(self, name: str) -> str:
        return "/".join(self._scope + [name]) if self._scope else name

    def call(self, sym: str, fn, args):
        """a jm_* entry point under the profiler"""
        if sym.endswith("_into"):        # the same operator writing into a channel slice: listed under the operator's name
            sym = sym[:-5]
        if self.only is not None and self._key(sym[3:]) not in self.only:
            self._hoisted = 0
            return fn(*args)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        algo = ALGO.get(sym)
        nbytes, flops, extra = algo(args) if algo else (0, 0, {})
        if self._hoisted:
            extra = dict(extra, executed_flops=flops)
            flops += self._hoisted
            self._hoisted = 0
        self.records.setdefault(self._key(sym[3:]), []).append((s, e, nbytes, flops, extra))
        return rc

    def region(self, name: str, fn, algo_bytes: int = 0, flops: int = 0):
        """time a span of caller-side work (torch / MIOpen / rocBLAS) on the current stream"""
        if not self.enabled or (self.only is not None and self._key(name) not in self.only):
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.records.setdefault(self._key(name), []).append((s, e, algo_bytes, flops, {}))
        return out

    def stall(self, name: str, wait: Callable[[], None]):
        """time how long the CURRENT stream is held up by `wait()` (a wait_event / wait_stream): the exposed part
        of work running on another stream"""
        if not self.enabled or (self.only is not None and self._key(name) not in self.only):
            return wait()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        wait()
        e.record()
        self.records.setdefault(self._key(name), []).append((s, e, 0, 0, dict(stall=True)))

    def summary(self, steps: int, hbm_peak_gbs: float, mfma_peak_tf: float) -> List[dict]:
        rows = []
        for name, evs in self.records.items():
            times = [s.elapsed_time(e) for s, e, *_ in evs]
            ms = sum(times) / steps
            nbytes = sum(r[2] for r in evs) / steps
            flops = sum(r[3] for r in evs) / steps
            row = dict(kernel=name, ms_per_step=round(ms, 5), launches_per_step=len(evs) / steps,
                       max_launch_ms=round(max(times), 5))
            if evs[0][4].get("stall"):
                row["stall"] = True
            if nbytes:
                gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                row.update(algo_bytes_per_step=int(nbytes), achieved_gbs=round(gbs, 2), hbm_frac=round(gbs / hbm_peak_gbs, 5))
            if flops:
                tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                row.update(algo_flops_per_step=int(flops), achieved_tflops=round(tf, 2), mfma_frac=round(tf / mfma_peak_tf, 4))
            ex = evs[0][4]
            if "executed_flops" in ex:
                xf = sum(r[4].get("executed_flops", 0) for r in evs) / steps
                row["executed_flops_per_step"] = int(xf)
                row["executed_mfma_frac"] = round(xf / (ms * 1e-3) / 1e12 / mfma_peak_tf, 4) if ms > 0 else 0.0
            if "evals" in ex:
                # pairwise point evaluations (distance + compare / select): the brute-force searches are VALU-bound, not
                # HBM-bound — priced at 8 fp32 operations per evaluation (3 sub, 3 mul/fma, compare, select) against the
                # vector peak (= the 157.3 TFLOP/s of MI355X_MICROARCH.md's fp32 row, 2 flops per lane-FMA)
                ev = sum(r[4].get("evals", 0) for r in evs) / steps
                row["evals_per_s"] = round(ev / (ms * 1e-3), 1) if ms > 0 else 0.0
                row["valu_frac"] = round(8.0 * ev / (ms * 1e-3) / 1e12 / VALU_F32_PEAK_TF, 4) if ms > 0 else 0.0
            if "brute_force_evals" in ex:
                row["brute_force_evals_per_step"] = int(sum(r[4].get("brute_force_evals", 0) for r in evs) / steps)
            if "flops_per_row" in ex:
                row["flops_per_row"] = ex["flops_per_row"]
                row["bytes_per_row"] = ex["bytes_per_row"]
                if ex.get("listed"):
                    row["listed"] = True
            if "iterations" in ex:
                # FPS keeps its cloud in registers: the bytes above are the reference's per-iteration re-reads (SURVEY.md §8d
                # "streaming-equivalent"), never HBM traffic — no HBM fraction is formed on them (it would exceed 1); the
                # figure of merit is the time per iteration, the HBM fraction is the one on the compulsory bytes
                it = sum(r[4]["iterations"] for r in evs) / steps
                row["us_per_fps_iteration"] = round(ms * 1e3 / it, 4)
                comp = sum(r[4]["compulsory_bytes"] for r in evs) / steps
                row["compulsory_bytes_per_step"] = int(comp)
                row["streaming_equivalent_bytes_per_step"] = row.pop("algo_bytes_per_step")
                row["streaming_equivalent_gbs"] = row.pop("achieved_gbs")
                row.pop("hbm_frac")
                row["algo_bytes_per_step"] = int(comp)
                gbs = comp / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                row["achieved_gbs"], row["hbm_frac"] = round(gbs, 2), round(gbs / hbm_peak_gbs, 5)
            rows.append(row)
        rows.sort(key=lambda r: -r["ms_per_step"])
        return rows


prof = Profiler()


class LibProxy:
    """attribute access returns the bound ctypes function, routed through the profiler when it is enabled"""

    def __init__(self, cdll: ctypes.CDLL):
        object.__setattr__(self, "_cdll", cdll)
        object.__setattr__(self, "_cache", {})

    def __getattr__(self, sym: str):
        cache = object.__getattribute__(self, "_cache")
        hit = cache.get(sym)
        if hit is not None:
            return hit
        fn = getattr(object.__getattribute__(self, "_cdll"), sym)
        if (not sym.startswith("jm_") or sym.endswith("_bytes") or sym.endswith("_elems") or sym.endswith("_supported")
                or sym.endswith("_offset") or sym.endswith("_capacity") or sym.endswith("_qmin")) or sym in (
                "jm_version", "jm_last_error", "jm_sa_mlp_pack", "jm_image_fusion_pack", "jm_pts_in_boxes3d_cpu", "jm_roipool3d_cpu"):
            cache[sym] = fn
            return fn

        def routed(*args, _fn=fn, _sym=sym):
            if prof.enabled:
                return prof.call(_sym, _fn, args)
            return _fn(*args)
        cache[sym] = routed
        return routed
