"""400 steps of the default workload with every overlap on: results stay finite, allocated / reserved memory stay flat (the first
steps include MIOpen's kernel selection).  Usage: PYTHONPATH=. python tools/long_soak.py"""
import time, torch, bench
dev = torch.device("cuda:0")
st = bench.make_detect_state(8, 1236, dev)
t0 = time.time()
bad = 0
for i in range(400):
    cache, aff, inter = bench.detect_step(st)
    if i % 50 == 49:
        torch.cuda.synchronize()
        ok = all(bool(torch.isfinite(inter[k]).all()) for k in ("backbone_features", "rois", "rcnn_feat", "pred_boxes3d")) and bool(torch.isfinite(aff[0][0]).all())
        bad += 0 if ok else 1
        print(i + 1, "steps", f"{(time.time() - t0) / (i + 1) * 1e3:.2f} ms/step", "finite" if ok else "NON-FINITE", "alloc MB", torch.cuda.memory_allocated() >> 20, "reserved MB", torch.cuda.memory_reserved() >> 20, flush=True)
print("LONG_SOAK_OK" if not bad else "LONG_SOAK_FAILED")
