"""`pointnet2_cuda` extension-module shim (pointnet2_api.cpp:10-24) over the C ABI."""
import ctypes

import torch

from .. import _lib as L

f32, i32 = torch.float32, torch.int32


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    lib = L.load()
    ws_bytes = lib.jm_ball_query_workspace_bytes(b, n)          # hash-grid search for n >= 2048 (same output)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=xyz.device) if ws_bytes else None
    L.check(lib.jm_ball_query_ws(b, n, m, float(radius), nsample, L.dev(new_xyz, f32, "new_xyz"), L.dev(xyz, f32, "xyz"),
                                 L.dev(idx, i32, "idx"), ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, ws_bytes,
                                 L.stream_ptr()), "ball_query_wrapper")
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    lib = L.load()
    L.check(lib.jm_group_points(b, c, n, npoints, nsample, L.dev(points, f32, "points"), L.dev(idx, i32, "idx"),
                                L.dev(out, f32, "out"), L.stream_ptr()), "group_points_wrapper")
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    lib = L.load()
    L.check(lib.jm_group_points_grad(b, c, n, npoints, nsample, L.dev(grad_out, f32, "grad_out"),
                                     L.dev(idx, i32, "idx"), L.dev(grad_points, f32, "grad_points"), L.stream_ptr()),
            "group_points_grad_wrapper")
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    lib = L.load()
    L.check(lib.jm_gather_points(b, c, n, npoints, L.dev(points, f32, "points"), L.dev(idx, i32, "idx"),
                                 L.dev(out, f32, "out"), L.stream_ptr()), "gather_points_wrapper")
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    lib = L.load()
    L.check(lib.jm_gather_points_grad(b, c, n, npoints, L.dev(grad_out, f32, "grad_out"), L.dev(idx, i32, "idx"),
                                      L.dev(grad_points, f32, "grad_points"), L.stream_ptr()),
            "gather_points_grad_wrapper")
    return 1


_coop_fps_done = {}   # device index -> event recorded after the last co-operative FPS launch


def farthest_point_sampling_wrapper(b, n, m, points, temp, idx, new_xyz=None):
    """reference signature (pointnet2_api.cpp:22) + optional new_xyz (B, m, 3): the sampled coordinates,
    gathered inside the same call"""
    lib = L.load()
    ws_bytes = lib.jm_fps_workspace_bytes(b, n)
    nx = L.dev(new_xyz, f32, "new_xyz") if new_xyz is not None else None
    tmp = L.dev(temp, f32, "temp") if temp is not None else None      # None: allowed with new_xyz (n <= 131072)
    if ws_bytes == 0:
        if nx is None:
            L.check(lib.jm_furthest_point_sampling(b, n, m, L.dev(points, f32, "points"), L.dev(temp, f32, "temp"),
                                                   L.dev(idx, i32, "idx"), L.stream_ptr()), "farthest_point_sampling_wrapper")
        else:
            L.check(lib.jm_furthest_point_sampling_xyz(b, n, m, L.dev(points, f32, "points"), tmp,
                                                       L.dev(idx, i32, "idx"), nx, None, 0, L.stream_ptr()),
                    "farthest_point_sampling_wrapper")
        return 1
    # clouds larger than one register file (n > 16384): several workgroups per cloud exchange candidates
    # through this workspace; such launches must not overlap on a device (their workgroups wait for each
    # other), so each one is ordered after the previous one, whatever stream that ran on
    dev_i = points.device.index if points.device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(points.device)
    prev = _coop_fps_done.get(dev_i)
    if prev is not None:
        stream.wait_event(prev)
    ws = torch.empty((ws_bytes + 64,), dtype=torch.uint8, device=points.device)
    base = ws.data_ptr()
    aligned = (base + 63) // 64 * 64
    if nx is None:
        L.check(lib.jm_furthest_point_sampling_ws(b, n, m, L.dev(points, f32, "points"), L.dev(temp, f32, "temp"),
                                                  L.dev(idx, i32, "idx"), ctypes.c_void_p(aligned), ws_bytes,
                                                  L.stream_ptr()), "farthest_point_sampling_wrapper")
    else:
        L.check(lib.jm_furthest_point_sampling_xyz(b, n, m, L.dev(points, f32, "points"), tmp,
                                                   L.dev(idx, i32, "idx"), nx, ctypes.c_void_p(aligned), ws_bytes,
                                                   L.stream_ptr()), "farthest_point_sampling_wrapper")
    ev = torch.cuda.Event()
    ev.record(stream)
    _coop_fps_done[dev_i] = ev
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    lib = L.load()
    ws_bytes = lib.jm_three_nn_workspace_bytes(b, n, m)         # hash-grid search for 1024 <= m <= 16384 (same output)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=unknown.device) if ws_bytes else None
    L.check(lib.jm_three_nn_ws(b, n, m, L.dev(unknown, f32, "unknown"), L.dev(known, f32, "known"),
                               L.dev(dist2, f32, "dist2"), L.dev(idx, i32, "idx"),
                               ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, ws_bytes, L.stream_ptr()), "three_nn_wrapper")


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    lib = L.load()
    L.check(lib.jm_three_interpolate(b, c, m, n, L.dev(points, f32, "points"), L.dev(idx, i32, "idx"),
                                     L.dev(weight, f32, "weight"), L.dev(out, f32, "out"), L.stream_ptr()),
            "three_interpolate_wrapper")


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    lib = L.load()
    L.check(lib.jm_three_interpolate_grad(b, c, n, m, L.dev(grad_out, f32, "grad_out"), L.dev(idx, i32, "idx"),
                                          L.dev(weight, f32, "weight"), L.dev(grad_points, f32, "grad_points"),
                                          L.stream_ptr()), "three_interpolate_grad_wrapper")
