"""Host logic of the training path that needs no GPU."""
import torch


def test_centre_canon_is_the_first_centre_of_every_class_of_copies():
    """train_rows._centre_canon: centres picked from copies of one point (cyclic RoI padding, roipool3d_kernel.cu:123-160) are mapped
    to the FIRST centre with the same canonical point; distinct points keep their own slot; composes level after level"""
    from jmodt_amd.train_rows import _centre_canon
    count = torch.tensor([3, 8, 1], dtype=torch.int32)
    n = 8
    canon = (torch.arange(n, dtype=torch.int32).view(1, n) % count.view(-1, 1)).int()             # canon_from_count: k % count
    pick = torch.tensor([[0, 4, 2, 3, 7, 6], [5, 1, 5, 0, 7, 2], [7, 0, 3, 3, 1, 2]], dtype=torch.int32)
    got = _centre_canon(canon, pick)
    assert got.dtype == torch.int32 and got.shape == pick.shape
    cp = torch.gather(canon, 1, pick.long())
    for r in range(pick.shape[0]):
        for i in range(pick.shape[1]):
            first = next(j for j in range(pick.shape[1]) if int(cp[r, j]) == int(cp[r, i]))
            assert int(got[r, i]) == first, (r, i)
    assert got[1].tolist() == [0, 1, 0, 3, 4, 5]                    # 8 distinct points: only the repeated pick 5 is a copy
    assert got[2].tolist() == [0] * 6                               # one distinct point: every centre is a copy of the first
    # next level: picks among the centres, classes through the representatives
    pick2 = torch.tensor([[1, 3, 5], [2, 0, 4], [5, 4, 3]], dtype=torch.int32)
    got2 = _centre_canon(got, pick2)
    rep = torch.gather(got, 1, pick2.long())
    for r in range(3):
        for i in range(3):
            assert int(got2[r, i]) == next(j for j in range(3) if int(rep[r, j]) == int(rep[r, i]))


def test_weight_gradient_split_and_workspace_queries_are_consistent():
    """host-only queries of the C-ABI (no device work): jm_rows_wgrad_splits / jm_rows_wgrad_workspace_bytes (include/jmodt_hip.h) —
    1 <= splits <= 256, at least 256 rows per split, at most ~1024 (tile, split) workgroups, workspace = splits (n k + n) floats or 0"""
    from jmodt_amd import _lib as L
    lib = L.load()
    for m in (0, 1, 255, 256, 511, 512, 5000, 65536, 524288, 2097152):
        for n, k in ((4, 4), (16, 16), (32, 64), (128, 128), (196, 128), (512, 256), (1024, 1536)):
            s = int(lib.jm_rows_wgrad_splits(m, n, k))
            tiles = -(-n // 128) * -(-k // 128)
            assert 1 <= s <= 256
            assert s == 1 or m // s >= 256, (m, n, k, s)
            assert s == 1 or (s - 1) * tiles < 1024, (m, n, k, s)
            want = 0 if s == 1 else s * (n * k + n) * 4
            assert int(lib.jm_rows_wgrad_workspace_bytes(m, n, k)) == want, (m, n, k, s)
