"""GPU time per call from HIP-graph replays (the host's ~20 us per launch through Python / ctypes stays out of the number)"""
import torch


def timeit(fn, reps=20):
    """GPU time per call: the calls are captured into a HIP graph (10 per graph) and replayed, so the host's ~20 us per launch
    through Python / ctypes is not in the number"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (10 * reps) * 1e3


