"""is a hipMemsetAsync captured into a HIP graph ordered against the kernel nodes around it on replay?  Graph: memset(buf) ->
buf += 1 (kernel) -> big = big * 1.0001 (kernel chain) -> buf2 memset -> atomics-like accumulate ... replayed with and without a
device synchronisation between replays."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
n = 1 << 20
buf = torch.full((n,), 7.0, device=dev)
src = torch.rand(n, device=dev)
out = torch.zeros(n, device=dev)
def body(use_memset):
    s = torch.cuda.current_stream().cuda_stream
    t = src
    for _ in range(20):
        t = t * 1.0001 + 0.5            # work in front of the memset
    if use_memset:
        hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, n * 4, ctypes.c_void_p(s))
    else:
        buf.fill_(0.0)
    buf.add_(t)                          # accumulate into the zeroed buffer
    out.copy_(buf)
cap = torch.cuda.Stream()
for use_memset in (True, False):
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        body(use_memset); body(use_memset)
    torch.cuda.synchronize()
    want = out.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        body(use_memset)
    torch.cuda.synchronize()
    for mode in ("back to back", "sync between", "eager work between"):
        bad = 0
        for it in range(30):
            g.replay()
            if mode == "sync between":
                torch.cuda.synchronize()
            elif mode == "eager work between":
                x = torch.rand(1 << 22, device=dev); x = x * 2 + 1; del x
            torch.cuda.synchronize()
            bad += int(not torch.equal(out, want))
        print(f"{'hipMemsetAsync' if use_memset else 'fill kernel   '} {mode:20s}: wrong {bad} of 30", flush=True)
