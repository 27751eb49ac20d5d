"""A process-wide counter of parameter / buffer (re-)registrations.

The host-side caches of folded / packed weights (detector.py, ops/pointnet2/fused.py) are keyed on every tensor's (identity,
storage, version).  Collecting those tensors means walking module trees — ~1 ms for the engine, and two hundred small walks per
step for the set-abstraction MLPs — which the 4-frame training step cannot afford on the host.  torch's global registration hooks
fire on every `register_parameter` / `register_buffer` / `register_module` (what an attribute assignment of a Parameter, load_state_dict(assign=True)
and parametrizations go through): a cached tensor list stays valid until EPOCH moves.  In-place updates, `.data` swaps and `module.to()` keep the
Parameter objects and are seen by the (identity, storage, version) signatures.  Code that writes `module._parameters[name]` / `_buffers[name]`
directly (torch.__future__.set_overwrite_module_params_on_conversion(True)) bypasses the hooks: call `invalidate()` afterwards.

One in-place update does NOT move `_version`: the FUSED optimizers (torch.optim.Adam(fused=True) -> torch._fused_adam_; probed on
torch 2.10: Adam / Adam(foreach=True) bump it, Adam(fused=True) leaves it at 0).  An engine that trains its RCNN with a fused Adam
(train_joint.rcnn_step) and then runs inference would keep its packed copies of the OLD weights.  A global optimizer-step post hook
therefore counts, per parameter, the optimizer steps that had a gradient for it (`OPT_GEN`); `tensor_sig` folds that count into the
signature."""
import weakref

import torch

EPOCH = [0]


def _bump(*_args):
    EPOCH[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_bump)
torch.nn.modules.module.register_module_buffer_registration_hook(_bump)
# re-parenting an already built submodule (`engine.rpn = other_rpn`, add_module) registers no parameter: its own hook
torch.nn.modules.module.register_module_module_registration_hook(_bump)

_orig_delattr = torch.nn.Module.__delattr__


def _delattr(self, name):            # `del module.child` / `del module.weight`: no registration hook fires
    _bump()
    return _orig_delattr(self, name)


torch.nn.Module.__delattr__ = _delattr


OPT_GEN = {}          # id(parameter) -> number of optimizer steps that updated it (ids of dead parameters only ever over-invalidate)


def _after_optimizer_step(optimizer, _args, _kwargs):
    gen = OPT_GEN
    for group in optimizer.param_groups:
        for p in group["params"]:
            if p.grad is not None:              # (a parameter without a gradient is not touched by the step)
                k = id(p)
                gen[k] = gen.get(k, 0) + 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

_register_step_hook(_after_optimizer_step)


def tensor_sig(t: torch.Tensor):
    """what a packed / folded copy of `t` is keyed on: identity, storage, in-place version, optimizer steps taken on it"""
    return (id(t), t.data_ptr(), t._version, OPT_GEN.get(id(t), 0))


def invalidate() -> None:
    """forget every cached tensor list (after parameters / buffers were replaced behind torch's registration API)"""
    _bump()


_tensors = weakref.WeakKeyDictionary()       # module -> (epoch, [parameters + buffers])


def module_tensors(module: torch.nn.Module):
    """list(module.parameters()) + list(module.buffers()), re-collected only after a registration anywhere in the process"""
    hit = _tensors.get(module)
    if hit is not None and hit[0] == EPOCH[0]:
        return hit[1]
    ts = list(module.parameters()) + list(module.buffers())
    _tensors[module] = (EPOCH[0], ts)
    return ts
