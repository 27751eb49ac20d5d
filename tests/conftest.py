import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _poison_empty():
    """JM_POISON_EMPTY=1 (a debugging tier, tools/fault_hunt.sh): every `torch.empty*` on the GPU comes back FILLED — NaN in floating
    point, 0x0fffffff in integers — so a kernel that consumes memory nobody wrote (a row count, a neighbour index, a partial sum) shows
    up deterministically (a device fault / NaN) instead of depending on what the caching allocator's recycled block happened to hold"""
    import torch
    real_empty, real_like, real_new = torch.empty, torch.empty_like, torch.Tensor.new_empty

    def fill(t):
        if isinstance(t, torch.Tensor) and t.is_cuda and t.numel():
            if t.is_floating_point():
                t.fill_(float("nan"))
            elif t.dtype in (torch.int32, torch.int64):
                t.fill_(0x0FFFFFFF)
            elif t.dtype == torch.uint8:
                t.fill_(0xA5)
        return t
    torch.empty = lambda *a, **k: fill(real_empty(*a, **k))
    torch.empty_like = lambda *a, **k: fill(real_like(*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: fill(real_new(self, *a, **k))


if os.environ.get("JM_POISON_EMPTY"):
    _poison_empty()


def has_reference() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, "jmodt"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
