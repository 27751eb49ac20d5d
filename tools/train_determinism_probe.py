"""repeat the a16 training step (csrc/affinity_train.hip) on fixed inputs and report which of the outputs differ run to run
(bitwise).  Usage: python tools/train_determinism_probe.py [reps]"""
import sys

import torch

from jmodt_amd.ops.affinity import make_affinity_mlp
from jmodt_amd.ops.affinity_train import AffinityTrainState, _head_tensors, _train_steps


def main(reps=40):
    dev = torch.device("cuda:0")
    for frames, R, C, H, ntid in [(4, 64, 512, 512, 30), (6, 64, 512, 512, 9), (8, 128, 128, 96, 40), (8, 128, 512, 512, 30)]:
        g = torch.Generator().manual_seed(frames * R + C)
        feats = torch.relu(torch.randn(frames, R, C, generator=g)).to(dev)
        tids = torch.randint(0, ntid, (frames, R), generator=g).float().to(dev)
        torch.manual_seed(3)
        link, se = make_affinity_mlp(C, (H, H)).to(dev), make_affinity_mlp(C, (H, H)).to(dev)
        names = ["lp", "sp"] + [f"link.{n}" for n in ("w1", "b1", "w2", "b2", "w3", "b3")] + [f"se.{n}" for n in ("w1", "b1", "w2", "b2", "w3", "b3")]
        first, bad = None, {}
        for r in range(reps):
            st = AffinityTrainState(feats, tids)
            lp, sp, grads, out = _train_steps(st, st.counts, _head_tensors(link), _head_tensors(se), 1.0, 1.0, True)
            torch.cuda.synchronize()
            cur = [lp, sp] + grads + [out["link"], out["start"], out["end"]]
            cur = [t.clone() for t in cur]
            if first is None:
                first = cur
                continue
            for nm, a, b in zip(names + ["out.link", "out.start", "out.end"], first, cur):
                if not torch.equal(a, b):
                    d = (a - b).abs().max().item()
                    bad.setdefault(nm, []).append((r, d, d / max(a.abs().max().item(), 1e-30)))
        print((frames, R, C, H, ntid), "differences:", {k: (len(v), max(x[1] for x in v), max(x[2] for x in v)) for k, v in bad.items()} or "none")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
