"""HIP-graph SECTIONS: a single-stream piece of a step, captured once (a forward graph and — when something in it requires
grad — a backward graph) and replayed with ONE launch per direction.

Why sections and not one graph for the step: on this stack a replayed graph runs at the eager kernels' speed but SERIALISES its
branches (tools/graph_probe.py: the composed inference step 16.0 ms replayed against 10.5 ms eager with four streams, 16.6 ms
against 16.7 ms with one) — so the unit of capture is what runs on ONE stream between two cross-stream dependencies anyway (a
set-abstraction + LI-Fusion level of the point branch, an image block, the RCNN), and the streams, events and the autograd
engine's stream hand-over between the sections stay as they are.  What a section removes is the HOST: the joint-mode training step
enqueues ~1050 launches through ~60 autograd Functions, 18 ms of Python / dispatcher time for a step whose device work is shorter
than that.

    sec = GraphedSection(fn, "level1")          # fn(*tensors) -> tensor or tuple of tensors; one stream, no host synchronisation
    out = sec(a, b, w)                          # first call with this signature: 2 eager warm-up runs, capture, replay

Rules (the ones torch.cuda.make_graphed_callables lives by, plus address keying):
  * an input that lives at a STABLE address — an nn.Parameter, the output of another section, anything passed through
    `mark_static` — is captured in place and is part of the cache key (another address = another capture: the two parities of a
    double-buffered producer get one graph each); any other input is copied into a buffer of the section's own at every call;
  * every tensor that requires grad inside fn must come in through the arguments (parameters included): the backward graph is
    torch.autograd.grad(outputs, those arguments) under capture;
  * outputs are static buffers, overwritten by the next replay: consume them within the step;
  * gradients of parameter arguments are handed over as `param.grad` directly (the static buffer itself when .grad is None, else
    accumulated), not through AccumulateGrad — which would clone every one of them (they are still referenced here);
  * no host synchronisation, no cross-stream wait on work outside the section, no allocation-dependent control flow inside fn;
  * at capture time no autograd graph of an EARLIER eager step through the same parameters may still be alive (a loss or output
    dictionary kept around): its AccumulateGrad nodes remember the streams they were created on, the engine orders the capture
    stream against those, and the capture forks into a stream that never joins (observed as a crash in capture_end).
"""
from typing import Callable, Dict, Sequence

import torch

_STATIC_STORAGES = set()          # data pointers of storages whose address is stable for the life of the process


def mark_static(t: torch.Tensor) -> torch.Tensor:
    """declare that `t`'s storage never moves and is never recycled while sections that captured it are alive (the caller keeps
    it referenced)"""
    _STATIC_STORAGES.add(t.untyped_storage().data_ptr())
    return t


def is_static(t: torch.Tensor) -> bool:
    return isinstance(t, torch.nn.Parameter) or t.untyped_storage().data_ptr() in _STATIC_STORAGES


_cap_streams = {}
import os as _os
TIMELINE = None                                          # a list while a GPU-side timeline is being collected (train_graphs trace)


def _stamp(name: str, what: str):
    if TIMELINE is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        TIMELINE.append((name, what, ev))


_SYNC = _os.environ.get("JM_GRAPH_SYNC", "")             # debugging: a device synchronisation behind every replay ("f", "b", "fb")


def _capture_stream(dev) -> torch.cuda.Stream:
    d = torch.device(dev)
    k = d.index if d.index is not None else torch.cuda.current_device()
    if k not in _cap_streams:
        _cap_streams[k] = torch.cuda.Stream(device=k)
    return _cap_streams[k]


def _dense_span(t: torch.Tensor) -> int:
    """elements a dense (non-overlapping, possibly permuted) tensor spans; others are given a contiguous buffer's worth"""
    if t.numel() == 0:
        return 0
    span = 1 + sum((s - 1) * st for s, st in zip(t.size(), t.stride()))
    if span != t.numel():
        raise RuntimeError(f"graphed section: output of shape {tuple(t.shape)} / strides {t.stride()} is not dense")
    return span


class _Entry:
    __slots__ = ("s_in", "copy_in", "outs", "outs_graph", "g_f", "g_b", "s_gout", "gin", "grad_in", "diff_out", "single", "pool",
                 "held", "replays", "gout_flat", "pool_b", "name")


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, entry: _Entry, *inputs):
        _stamp(entry.name, "fwd<")
        for i in entry.copy_in:
            entry.s_in[i].copy_(inputs[i])
        entry.g_f.replay()
        _stamp(entry.name, "fwd>")
        if "f" in _SYNC:
            torch.cuda.synchronize()
        entry.replays += 1
        ctx.entry = entry
        outs = tuple(o.detach() for o in entry.outs)
        nd = [o for o, d in zip(outs, entry.diff_out) if not d]
        if nd:
            ctx.mark_non_differentiable(*nd)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        e = ctx.entry
        _stamp(e.name, "bwd<")
        live = [g for g, d in zip(grads, e.diff_out) if d]
        if any(g is None for g in live):
            e.gout_flat.zero_()
        for g, s in zip(live, e.s_gout):
            if g is not None and (g.data_ptr() != s.data_ptr() or g.stride() != s.stride()):
                s.copy_(g)
        # a parameter whose .grad still IS this section's buffer (a second backward pass before zero_grad): the replay would
        # overwrite the first pass's gradient — keep it and add it back
        again = [(gi, gi.clone()) for i, gi in zip(e.grad_in, e.gin)
                 if gi is not None and isinstance(e.s_in[i], torch.nn.Parameter) and e.s_in[i].grad is gi]
        e.g_b.replay()
        _stamp(e.name, "bwd>")
        if "b" in _SYNC:
            torch.cuda.synchronize()
        for gi, prev in again:
            gi.add_(prev)
        res = [None] * len(e.s_in)
        for i, gi in zip(e.grad_in, e.gin):
            if gi is None:
                continue
            s = e.s_in[i]
            if isinstance(s, torch.nn.Parameter):
                if s.grad is None:
                    s.grad = gi
                elif s.grad is not gi:
                    s.grad.add_(gi)
            else:
                res[i] = gi
        return (None, *res)


class GraphedSection:
    def __init__(self, fn: Callable, name: str, warmup: int = 2, enabled: bool = True):
        """enabled=False: a pass-through (fn runs eagerly under plain autograd) — the caller's composition stays the same"""
        self.fn, self.name, self.warmup, self.enabled = fn, name, warmup, enabled
        self._cache: Dict[tuple, _Entry] = {}
        self.captures = 0

    # ------------------------------------------------------------------------------------------------------------------------
    def _key(self, args: Sequence[torch.Tensor], grad: bool) -> tuple:
        k = [grad]
        for a in args:
            k.append((tuple(a.shape), a.dtype, bool(grad and a.requires_grad), a.data_ptr() if is_static(a) else None))
        return tuple(k)

    def __call__(self, *args: torch.Tensor):
        if not self.enabled:
            return self.fn(*args)
        for a in args:
            if not (isinstance(a, torch.Tensor) and a.is_cuda):
                raise RuntimeError(f"section {self.name}: every argument must be a GPU tensor (got {type(a).__name__})")
        grad = torch.is_grad_enabled() and any(a.requires_grad for a in args)
        key = self._key(args, grad)
        e = self._cache.get(key)
        if e is None:
            e = self._cache[key] = self._capture(args, grad)
        if grad:
            outs = _Replay.apply(e, *args)
        else:
            with torch.no_grad():
                _stamp(e.name, "fwd<")
                for i in e.copy_in:
                    e.s_in[i].copy_(args[i])
                e.g_f.replay()
                _stamp(e.name, "fwd>")
                e.replays += 1
                outs = tuple(o.detach() for o in e.outs)
        return outs[0] if e.single else outs

    # ------------------------------------------------------------------------------------------------------------------------
    def _capture(self, args: Sequence[torch.Tensor], grad: bool) -> _Entry:
        import os
        from .profile import prof
        trace = bool(os.environ.get("JM_GRAPH_TRACE"))
        if trace:
            print(f"[graphed] capturing {self.name}: {len(args)} inputs, grad={grad}", flush=True)
        e = _Entry()
        e.name = self.name
        e.replays = 0
        e.held = list(args)              # static inputs stay referenced: their addresses are inside the graphs
        s_in, copy_in = [], []
        with torch.no_grad():
            for i, a in enumerate(args):
                need = grad and a.requires_grad
                if isinstance(a, torch.nn.Parameter):
                    s = a                                    # a leaf at a stable address: captured as it is
                elif is_static(a):
                    s = a.detach()
                    if need:
                        s.requires_grad_(True)
                else:
                    s = a.detach().clone(memory_format=torch.preserve_format)
                    if need:
                        s.requires_grad_(True)
                    copy_in.append(i)
                s_in.append(s)
        e.s_in, e.copy_in = s_in, copy_in
        e.grad_in = [i for i, s in enumerate(s_in) if grad and s.requires_grad]
        dev = args[0].device
        was_prof, prof.enabled = prof.enabled, False          # no event pairs inside a capture
        cur = torch.cuda.current_stream(dev)
        cap = _capture_stream(dev)
        try:
            def run():
                out = self.fn(*s_in)
                single = isinstance(out, torch.Tensor)
                return ((out,) if single else tuple(out)), single

            # eager warm-up on the capture stream: lazily created workspaces, MIOpen's solver choice, the autograd engine's
            # stream bookkeeping — everything that may not happen under capture happens here
            cap.wait_stream(cur)
            with torch.cuda.stream(cap):
                for _ in range(max(1, self.warmup)):
                    with torch.set_grad_enabled(grad):
                        outs, single = run()
                    if grad:
                        diff = [o for o in outs if o.requires_grad]
                        if diff:
                            torch.autograd.grad(diff, [s_in[i] for i in e.grad_in], [torch.ones_like(o) for o in diff], allow_unused=True)
                    del outs
            torch.cuda.synchronize(dev)
            e.pool = torch.cuda.graph_pool_handle()
            e.g_f = torch.cuda.CUDAGraph()
            # (capture_begin / capture_end directly: torch.cuda.graph() also empties the allocator's cache at every capture)
            with torch.cuda.stream(cap):
                e.g_f.capture_begin(pool=e.pool)
                try:
                    with torch.set_grad_enabled(grad):
                        outs, single = run()
                finally:
                    e.g_f.capture_end()
            if trace:
                print(f"[graphed]   {self.name}: forward captured", flush=True)
            e.single = single
            e.outs_graph = outs
            e.diff_out = [bool(grad and o.requires_grad) for o in outs]
            e.g_b, e.s_gout, e.gin, e.gout_flat = None, [], [], None
            if grad and any(e.diff_out):
                diff = [o for o, d in zip(outs, e.diff_out) if d]
                # the incoming gradients' buffers: views of ONE flat tensor (a missing gradient = one fill, not one per output)
                with torch.no_grad():
                    sizes = [_dense_span(o) for o in diff]
                    e.gout_flat = torch.zeros((sum(sizes),), dtype=diff[0].dtype, device=dev)
                    off = 0
                    for o, n in zip(diff, sizes):
                        e.s_gout.append(e.gout_flat[off:off + n].as_strided(o.size(), o.stride()))
                        off += n
                # a pool of its own: allocated from the forward graph's pool, the gradient buffers would reuse blocks the forward
                # capture freed (its temporaries) and the NEXT forward replay would scribble over a gradient that is still
                # somebody's .grad (a second pass before zero_grad: tests/test_gpu_graphs.py)
                e.pool_b = e.pool if _os.environ.get("JM_GRAPH_ONE_POOL") else torch.cuda.graph_pool_handle()
                e.g_b = torch.cuda.CUDAGraph()
                with torch.cuda.stream(cap):
                    e.g_b.capture_begin(pool=e.pool_b)
                    try:
                        gin = list(torch.autograd.grad(diff, [s_in[i] for i in e.grad_in], e.s_gout, allow_unused=True))
                        # a parameter's gradient in the parameter's own memory layout (what AccumulateGrad guarantees and the
                        # fused optimizers require: channels-last convolution kernels), converted inside the graph
                        for j, i in enumerate(e.grad_in):
                            if gin[j] is not None and isinstance(s_in[i], torch.nn.Parameter) and gin[j].stride() != s_in[i].stride():
                                gin[j] = torch.empty_like(s_in[i], memory_format=torch.preserve_format).copy_(gin[j])
                    finally:
                        e.g_b.capture_end()
                if trace:
                    print(f"[graphed]   {self.name}: backward captured", flush=True)
                e.gin = list(gin)
                for g in e.gin:
                    if g is not None:
                        mark_static(g)
            with torch.no_grad():
                e.outs = [o.detach() for o in outs]
            for o in e.outs:
                mark_static(o)
            if not (grad and any(e.diff_out)):
                e.outs_graph = None              # nothing will ever back-propagate through it: drop the autograd graph
        finally:
            prof.enabled = was_prof
        cur.wait_stream(cap)
        self.captures += 1
        return e

    def entries(self) -> int:
        return len(self._cache)
