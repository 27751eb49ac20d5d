"""cross-stream / graph-to-graph ordering around replays: producer graph on stream A -> event -> consumer (eager kernel or another
graph) on stream B; and two graphs back to back on one stream"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
n = 1 << 22
a = torch.zeros(n, device=dev); mid = torch.zeros(n, device=dev); out = torch.zeros(n, device=dev)
def producer():
    t = a
    for _ in range(200):
        t = t * 1.0001 + 1.0
    mid.copy_(t)
def consumer():
    out.copy_(mid * 2.0 + 1.0)
cap = torch.cuda.Stream()
def capture(fn):
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        fn()
    torch.cuda.synchronize()
    return g
gp, gc_ = capture(producer), capture(consumer)
A, B = torch.cuda.Stream(), torch.cuda.Stream()
def reference(v):
    a.fill_(v); producer(); consumer(); torch.cuda.synchronize(); return out.clone()
for name in ("graph->event->eager", "graph->event->graph", "graph,graph same stream", "eager->event->graph", "graph->wait_stream->graph"):
    bad = 0
    for it in range(20):
        ref = reference(float(it))
        mid.zero_(); out.zero_(); a.fill_(float(it)); torch.cuda.synchronize()
        if name == "graph->event->eager":
            with torch.cuda.stream(A):
                gp.replay(); ev = torch.cuda.Event(); ev.record(A)
            with torch.cuda.stream(B):
                B.wait_event(ev); consumer()
        elif name == "graph->event->graph":
            with torch.cuda.stream(A):
                gp.replay(); ev = torch.cuda.Event(); ev.record(A)
            with torch.cuda.stream(B):
                B.wait_event(ev); gc_.replay()
        elif name == "graph,graph same stream":
            with torch.cuda.stream(A):
                gp.replay(); gc_.replay()
        elif name == "eager->event->graph":
            with torch.cuda.stream(A):
                producer(); ev = torch.cuda.Event(); ev.record(A)
            with torch.cuda.stream(B):
                B.wait_event(ev); gc_.replay()
        else:
            with torch.cuda.stream(A):
                gp.replay()
            B.wait_stream(A)
            with torch.cuda.stream(B):
                gc_.replay()
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, ref))
    print(f"{name:28s} wrong {bad} of 20", flush=True)
