"""`iou3d_cuda` extension-module shim (iou3d.cpp:170-175) over the C ABI."""
import ctypes

import torch

from .. import _lib as L

f32 = torch.float32


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    lib = L.load()
    L.check(lib.jm_boxes_overlap_bev(boxes_a.size(0), L.dev(boxes_a, f32, "boxes_a"), boxes_b.size(0),
                                     L.dev(boxes_b, f32, "boxes_b"), L.dev(ans_overlap, f32, "ans_overlap"),
                                     L.stream_ptr()), "boxes_overlap_bev_gpu")
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    lib = L.load()
    L.check(lib.jm_boxes_iou_bev(boxes_a.size(0), L.dev(boxes_a, f32, "boxes_a"), boxes_b.size(0),
                                 L.dev(boxes_b, f32, "boxes_b"), L.dev(ans_iou, f32, "ans_iou"), L.stream_ptr()),
            "boxes_iou_bev_gpu")
    return 1


def nms_device(boxes, thresh, normal):
    """MI355X-native form: everything on the device; returns (keep int64 (N) device, num_keep int32 (1) device)."""
    lib = L.load()
    n = boxes.size(0)
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=boxes.device)
    num = torch.empty((1,), dtype=torch.int32, device=boxes.device)
    ws_bytes = lib.jm_nms_workspace_bytes(n)
    ws = torch.empty((max(ws_bytes, 8),), dtype=torch.uint8, device=boxes.device)
    L.check(lib.jm_nms(n, L.dev(boxes, f32, "boxes"), float(thresh), int(normal), ctypes.c_void_p(keep.data_ptr()),
                       ctypes.c_void_p(num.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws_bytes, L.stream_ptr()),
            "nms")
    return keep, num


def nms_batched_device(boxes, counts, thresh, normal):
    """P independent NMS problems in one mask launch + one reduce launch (jm_nms_batched).
    boxes (P, Nmax, 5) score-sorted per problem, counts (P) int32 device
    -> keep (P, Nmax) int64 (first num_keep[p] entries valid), num_keep (P) int32; nothing syncs."""
    lib = L.load()
    nprob, nmax = boxes.size(0), boxes.size(1)
    keep = torch.empty((nprob, max(nmax, 1)), dtype=torch.int64, device=boxes.device)
    num = torch.empty((max(nprob, 1),), dtype=torch.int32, device=boxes.device)
    ws_bytes = lib.jm_nms_workspace_bytes(nmax) * nprob
    ws = torch.empty((max(ws_bytes, 8),), dtype=torch.uint8, device=boxes.device)
    L.check(lib.jm_nms_batched(nprob, nmax, L.dev(counts, torch.int32, "counts"), L.dev(boxes, f32, "boxes"),
                               float(thresh), int(normal), ctypes.c_void_p(keep.data_ptr()),
                               ctypes.c_void_p(num.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws_bytes,
                               L.stream_ptr()), "nms_batched")
    return keep, num[:nprob]


def nms_normal_first_k_device(boxes, counts, thresh, first_k):
    """the first `first_k` survivors of P axis-aligned NMS problems without the pair mask (jm_nms_normal_first_k_batched):
    boxes (P, Nmax, 5) score-sorted per problem, counts (P) int32 device -> keep (P, Nmax) int64 (first num_keep[p] entries
    valid), num_keep (P) int32 = min(first_k, survivors); nothing syncs, no workspace."""
    lib = L.load()
    nprob, nmax = boxes.size(0), boxes.size(1)
    keep = torch.empty((nprob, max(nmax, 1)), dtype=torch.int64, device=boxes.device)
    num = torch.empty((max(nprob, 1),), dtype=torch.int32, device=boxes.device)
    L.check(lib.jm_nms_normal_first_k_batched(nprob, nmax, L.dev(counts, torch.int32, "counts"), L.dev(boxes, f32, "boxes"),
                                              float(thresh), int(first_k), ctypes.c_void_p(keep.data_ptr()),
                                              ctypes.c_void_p(num.data_ptr()), L.stream_ptr()), "nms_normal_first_k_batched")
    return keep, num[:nprob]


def _nms_to_cpu_keep(boxes, keep, thresh, normal):
    if keep.is_cuda or keep.dtype != torch.int64 or not keep.is_contiguous():
        raise RuntimeError("keep must be a contiguous CPU int64 tensor (iou3d.cpp:76-82)")
    dkeep, dnum = nms_device(boxes, thresh, normal)
    num = int(dnum.item())  # the reference API returns the count on the host: one 4-byte D2H
    keep[:num].copy_(dkeep[:num])
    return num


def nms_gpu(boxes, keep, nms_overlap_thresh):
    return _nms_to_cpu_keep(boxes, keep, nms_overlap_thresh, 0)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    return _nms_to_cpu_keep(boxes, keep, nms_overlap_thresh, 1)
