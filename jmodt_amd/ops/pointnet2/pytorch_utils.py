"""Layer builders with the reference's module / parameter naming.

Mirror of jmodt/ops/pointnet2/pytorch_utils.py (SharedMLP :6-33, _ConvBase :36-102,
BatchNorm1d/2d :105-125, Conv1d/Conv2d :128-200, FC :203-236).  What has to match is the
state_dict layout — `layer{i}.conv.weight`, `layer{i}.bn.bn.{weight,bias,running_*}`, and for
the affinity heads `{0,2,3}.conv.{weight,bias}` (SURVEY.md §5 "Checkpoint / resume") — so that a
reference checkpoint loads into these modules unchanged; the construction code itself is this
package's own (one generic unit instead of the reference's per-rank subclasses).
"""
from typing import List, Optional, Sequence

import torch.nn as nn

_CONV = {1: nn.Conv1d, 2: nn.Conv2d}
_BN = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d}
_IN = {1: nn.InstanceNorm1d, 2: nn.InstanceNorm2d}


def _default_act():
    return nn.ReLU(inplace=True)


class _Norm(nn.Sequential):
    """wrapper that owns the norm under the child name `bn` (-> `<unit>.bn.bn.*` keys)"""

    def __init__(self, rank: int, channels: int, name: str = ""):
        super().__init__()
        norm = _BN[rank](channels)
        nn.init.constant_(norm.weight, 1.0)
        nn.init.constant_(norm.bias, 0.0)
        self.add_module(name + "bn", norm)


class BatchNorm1d(_Norm):
    def __init__(self, in_size: int, *, name: str = ""):
        super().__init__(1, in_size, name)


class BatchNorm2d(_Norm):
    def __init__(self, in_size: int, name: str = ""):
        super().__init__(2, in_size, name)


class _ConvUnit(nn.Sequential):
    """conv (+norm) (+activation) in post- or pre-activation order.  Children: conv, bn,
    activation, in — the only names a checkpoint can contain."""

    def __init__(self, rank, in_size, out_size, kernel_size, stride, padding, activation, bn, init, bias, preact, name,
                 instance_norm):
        super().__init__()
        conv = _CONV[rank](in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding,
                           bias=bias and not bn)
        init(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0)
        norm_width = in_size if preact else out_size
        tail: List = []
        if bn:
            tail.append((name + "bn", (BatchNorm1d if rank == 1 else BatchNorm2d)(norm_width)))
        if activation is not None:
            tail.append((name + "activation", activation))
        if instance_norm and not bn:
            tail.append((name + "in", _IN[rank](norm_width, affine=False, track_running_stats=False)))
        parts = tail + [(name + "conv", conv)] if preact else [(name + "conv", conv)] + tail
        for key, mod in parts:
            self.add_module(key, mod)


class Conv1d(_ConvUnit):
    def __init__(self, in_size: int, out_size: int, *, kernel_size: int = 1, stride: int = 1, padding: int = 0,
                 activation="default", bn: bool = False, init=nn.init.kaiming_normal_, bias: bool = True,
                 preact: bool = False, name: str = "", instance_norm: bool = False):
        act = _default_act() if isinstance(activation, str) else activation
        super().__init__(1, in_size, out_size, kernel_size, stride, padding, act, bn, init, bias, preact, name,
                         instance_norm)


class Conv2d(_ConvUnit):
    def __init__(self, in_size: int, out_size: int, *, kernel_size: Sequence[int] = (1, 1),
                 stride: Sequence[int] = (1, 1), padding: Sequence[int] = (0, 0), activation="default",
                 bn: bool = False, init=nn.init.kaiming_normal_, bias: bool = True, preact: bool = False,
                 name: str = "", instance_norm: bool = False):
        act = _default_act() if isinstance(activation, str) else activation
        super().__init__(2, in_size, out_size, kernel_size, stride, padding, act, bn, init, bias, preact, name,
                         instance_norm)


class SharedMLP(nn.Sequential):
    """stack of 1x1 Conv2d units named layer0, layer1, ... applied to (B, C, npoint, nsample)"""

    def __init__(self, args: List[int], *, bn: bool = False, activation="default", preact: bool = False,
                 first: bool = False, name: str = "", instance_norm: bool = False):
        super().__init__()
        act = _default_act() if isinstance(activation, str) else activation
        for i, (cin, cout) in enumerate(zip(args[:-1], args[1:])):
            bare = first and preact and i == 0  # a pre-activated first layer gets no norm / activation
            self.add_module(f"{name}layer{i}", Conv2d(cin, cout, bn=bn and not bare, activation=None if bare else act,
                                                      preact=preact, instance_norm=instance_norm))


class FC(nn.Sequential):
    def __init__(self, in_size: int, out_size: int, *, activation="default", bn: bool = False,
                 init: Optional[callable] = None, preact: bool = False, name: str = ""):
        super().__init__()
        act = _default_act() if isinstance(activation, str) else activation
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if fc.bias is not None:
            nn.init.constant_(fc.bias, 0)
        tail = []
        if bn:
            tail.append((name + "bn", BatchNorm1d(in_size if preact else out_size)))
        if act is not None:
            tail.append((name + "activation", act))
        parts = tail + [(name + "fc", fc)] if preact else [(name + "fc", fc)] + tail
        for key, mod in parts:
            self.add_module(key, mod)
