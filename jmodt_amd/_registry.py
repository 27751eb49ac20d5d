"""A process-wide counter of parameter / buffer (re-)registrations.

The host-side caches of folded / packed weights (detector.py, ops/pointnet2/fused.py) are keyed on every tensor's (identity,
storage, version).  Collecting those tensors means walking module trees — ~1 ms for the engine, and two hundred small walks per
step for the set-abstraction MLPs — which the 4-frame training step cannot afford on the host.  torch's global registration hooks
fire on every `register_parameter` / `register_buffer` / `register_module` (what an attribute assignment of a Parameter, load_state_dict(assign=True)
and parametrizations go through): a cached tensor list stays valid until EPOCH moves.  In-place updates, `.data` swaps and `module.to()` keep the
Parameter objects and are seen by the (identity, storage, version) signatures.  Code that writes `module._parameters[name]` / `_buffers[name]`
directly (torch.__future__.set_overwrite_module_params_on_conversion(True)) bypasses the hooks: call `invalidate()` afterwards."""
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
