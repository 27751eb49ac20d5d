"""what does recording an event between two kernels of one stream cost on this stack?  A: a kernel writing 64 MB,
B: a small kernel; gap = start(B) - end(A) from the kernel timestamps of two timing events... measured as the total time
of N (A, [record], B) pairs minus the same without the record."""
import ctypes, time, torch
hip = ctypes.CDLL("libamdhip64.so")
x = torch.empty(16 * 1024 * 1024, device="cuda")
y = torch.empty(1024, device="cuda")
def run(mode, n=200):
    evs = []
    if mode == "torch":
        evs = [torch.cuda.Event() for _ in range(n)]
    elif mode == "torch_timing":
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    elif mode in ("nofence", "raw"):
        flags = 0x2 | (0x20000000 if mode == "nofence" else 0)          # hipEventDisableTiming | hipEventDisableSystemFence
        for _ in range(n):
            e = ctypes.c_void_p()
            assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0
            evs.append(e)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        x.fill_(1.0)
        if mode in ("torch", "torch_timing"):
            evs[i].record()
        elif mode in ("nofence", "raw"):
            hip.hipEventRecord(evs[i], st)
        y.add_(1.0)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for mode in ("none", "torch", "torch_timing", "raw", "nofence", "none"):
    run(mode, 20)
    print(f"{mode:14s} {run(mode):8.1f} us per (64 MB fill, record, small kernel)")
