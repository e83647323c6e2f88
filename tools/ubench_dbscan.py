"""Dev: device DBSCAN at realistic foreground sizes (cars as 200-point blobs) and a worst-case chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from streammos_amd import ops
dev = "cuda:0"
rng = np.random.default_rng(0)
for n_obj in (10, 50, 100, 250):
    pts = np.concatenate([rng.normal(c, (0.8, 0.35, 0.3), (200, 3)) for c in rng.uniform(-45, 45, (n_obj, 3)) * (1, 1, 0.02)]).astype(np.float32)
    x = torch.from_numpy(pts).to(dev)
    ops.dbscan(x, 0.3, 5); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): lab = ops.dbscan(x, 0.3, 5)
    torch.cuda.synchronize()
    print("n=%6d  %.2f ms   clusters %d" % (len(pts), (time.perf_counter() - t) / 5 * 1e3, len(torch.unique(lab[lab >= 0]))), flush=True)
chain = np.stack((np.arange(20000) * 0.05, np.zeros(20000), np.zeros(20000)), 1).astype(np.float32)   # one 1 km wall
x = torch.from_numpy(chain).to(dev)
t = time.perf_counter(); lab = ops.dbscan(x, 0.3, 5); torch.cuda.synchronize()
print("chain n=20000  %.2f ms  clusters %d" % ((time.perf_counter() - t) * 1e3, len(torch.unique(lab[lab >= 0]))))
