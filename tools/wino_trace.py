"""where a wave of conv3x3_wino_kernel spends its cycles: shader-clock stamps at every stage of sampled workgroups (tools build)
    gpurun -- 'PYTHONPATH=. python tools/wino_trace.py'"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd import _lib
from jmodt_amd.csrc import build as _hip_build
_lib.LIB_PATH = _hip_build.TOOLS_LIB
from jmodt_amd.ops.fusion import conv3x3_wino_bias_relu, pack_wino_weight

raw = ctypes.CDLL(_hip_build.TOOLS_LIB)
stride = raw.jm_tools_wino_trace_stride()
TR = (stride - 4) // 4
torch.manual_seed(0)
for cin, cout, H, W in ((64, 128, 192, 640), (128, 256, 96, 320), (256, 512, 48, 160)):
    x = torch.relu(torch.randn(8, cin, H, W, device="cuda")).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device="cuda") * 0.1
    packed = pack_wino_weight(w)
    for _ in range(3): conv3x3_wino_bias_relu(x, packed, b, cout)
    nwg = 8 * (H // 8) * (W // 16) * (cout // 64)
    ns = 512 // 61 + 1
    buf = torch.zeros(ns * stride, dtype=torch.int64, device="cuda")
    raw.jm_tools_wino_trace(ctypes.c_void_p(buf.data_ptr()))
    conv3x3_wino_bias_relu(x, packed, b, cout)
    torch.cuda.synchronize()
    raw.jm_tools_wino_trace(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(ns, stride)
    t = t[t[:, 4 * TR + 1] > 0]
    nch = int(t[0, 4 * TR + 1])
    tr = t[:, :4 * TR].reshape(-1, 4, TR).astype(np.float64)
    start, ep0, ep1, fin = tr[:, :, TR - 4], tr[:, :, TR - 3], tr[:, :, TR - 2], tr[:, :, TR - 1]
    ch = tr[:, :, :11 * min(nch, 16)].reshape(len(tr), 4, -1, 11)
    stages = np.diff(ch[..., :9], axis=-1)                  # 8 stage durations
    loads = ch[..., 9] - ch[..., 8]
    barrier = ch[..., 10] - ch[..., 9]
    chunk = ch[..., 10] - ch[..., 0]
    print(f"{cin}->{cout}: {len(tr)} workgroups sampled, {nch} chunks; cycles (mean over workgroups, waves, chunks)")
    items = nwg / 512
    print(f"  last item of a workgroup: k-loop {np.mean(ch[:, :, -1, 10] - ch[:, :, 0, 0]):8.0f}   epilogue {np.mean(ep1 - ep0):8.0f}   whole workgroup {np.mean(fin - start):9.0f} = {np.mean(fin - start) / items:8.0f} per item ({items:.0f} items)")
    print("  stage    " + " ".join(f"{v:7.0f}" for v in stages.mean(axis=(0, 1, 2))) + f"   window loads {loads.mean():6.0f}  barrier {barrier.mean():6.0f}  chunk {chunk.mean():7.0f} (128 MFMAs = 4096 pipe cycles)")
    print("  by chunk " + " ".join(f"{v:7.0f}" for v in chunk.mean(axis=(0, 1))))
    print("  p10/p50/p90 of a stage: " + " ".join(f"{np.percentile(stages, q):7.0f}" for q in (10, 50, 90)), flush=True)
