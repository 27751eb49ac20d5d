"""do graph replays launched on DIFFERENT streams overlap on the device?  Two graphs of 200 small dependent kernels each (a few
workgroups: no contention for CUs), replayed on two streams: both together against one alone, and the same eagerly."""
import os, subprocess, sys, time
def body():
    import torch
    dev = torch.device("cuda:0")
    xs = [torch.zeros(4096, device=dev) for _ in range(2)]
    def chain(x):
        for _ in range(200):
            x.mul_(1.0001).add_(1.0)
    cap = torch.cuda.Stream()
    graphs = []
    for x in xs:
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            chain(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            chain(x)
        graphs.append(g)
    torch.cuda.synchronize()
    A, B = torch.cuda.Stream(), torch.cuda.Stream()
    def timed(fn, n=10):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    def one_graph():
        with torch.cuda.stream(A): graphs[0].replay()
    def two_graphs():
        with torch.cuda.stream(A): graphs[0].replay()
        with torch.cuda.stream(B): graphs[1].replay()
    def one_eager():
        with torch.cuda.stream(A): chain(xs[0])
    def two_eager():
        with torch.cuda.stream(A): chain(xs[0])
        with torch.cuda.stream(B): chain(xs[1])
    def graph_and_eager():
        with torch.cuda.stream(A): graphs[0].replay()
        with torch.cuda.stream(B): chain(xs[1])
    print(f"graph: one {timed(one_graph):.3f} ms, two streams {timed(two_graphs):.3f} ms | eager: one {timed(one_eager):.3f} ms, two streams "
          f"{timed(two_eager):.3f} ms | graph + eager {timed(graph_and_eager):.3f} ms", flush=True)
if __name__ == "__main__":
    if len(sys.argv) > 1:
        body()
    else:
        for env in ({}, {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}, {"GPU_MAX_HW_QUEUES": "8"}, {"DEBUG_HIP_FORCE_GRAPH_QUEUES": "8", "GPU_MAX_HW_QUEUES": "8"},
                    {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0", "GPU_MAX_HW_QUEUES": "8"}, {"DEBUG_HIP_GRAPH_BATCH_SIZE": "1"}):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "x"], capture_output=True, text=True, env=dict(os.environ, **env))
            out = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith("graph:")]
            print(env, "->", out[-1] if out else (r.stdout + r.stderr)[-300:], flush=True)
