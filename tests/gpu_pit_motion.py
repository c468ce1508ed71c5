"""Dev script (GPU): how far the spheres of the settled 4 M pit (config 4) move per step around step 506, and how many box pairs a kept list would carry at a given margin."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import engine as E, scenes as S
side = int(sys.argv[1]) if len(sys.argv) > 1 else 160
start = int(sys.argv[2]) if len(sys.argv) > 2 else 506
scene = S.ball_pit(side, side, side, seed=4)
nb = len(scene["body_transforms"]) - 1
w = E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP, max_contacts=5 * nb, max_pairs=7 * nb)
w.step(start); w.synchronize()
p0 = np.array(w.get_bodies()["transforms"]["position"][1:, :3], dtype=np.float64)
v0 = np.linalg.norm(np.array(w.get_bodies()["momentum"]["velocity"][1:, :3], dtype=np.float64), axis=1)
print("speed percentiles p50 p90 p99 max:", np.percentile(v0, [50, 90, 99, 100]), " (x 1/60 s per step)")
cum = np.zeros(len(p0))
for k in (1, 2, 4, 8, 16, 32):
    w.step(k - (0 if k == 1 else k // 2)); w.synchronize()
    p = np.array(w.get_bodies()["transforms"]["position"][1:, :3], dtype=np.float64)
    d = np.abs(p - p0).max(axis=1)
    qs = np.percentile(d, [50, 90, 99, 99.9, 100])
    print(f"after {k:2d} steps: max-axis displacement p50 {qs[0]:.5f} p90 {qs[1]:.5f} p99 {qs[2]:.5f} p99.9 {qs[3]:.5f} max {qs[4]:.4f};  beyond 1/64: {(d > 1/64).sum()}  1/32: {(d > 1/32).sum()}  1/16: {(d > 1/16).sum()}  1/8: {(d > 1/8).sum()}")
c = w.counts(); print("pairs", c["pairs"], "raw", c["raw_pairs"], "contacts", c["contacts"])
try:
    from scipy.spatial import cKDTree
    sub = p[(np.abs(p[:, 0]) < 20) & (np.abs(p[:, 2]) < 20)]
    tr = cKDTree(sub)
    for m in (0.0, 1 / 64, 1 / 32, 1 / 16, 1 / 8, 1 / 4):
        n = tr.count_neighbors(tr, 1.0 + 2 * m, p=np.inf)
        print(f"margin {m:.4f}: box pairs per sphere {(n - len(sub)) / 2 / len(sub):.3f}  (sample of {len(sub)})")
    n = tr.count_neighbors(tr, 1.0, p=2.0); print(f"touching: {(n - len(sub)) / 2 / len(sub):.3f}")
except Exception as e:
    print("scipy:", e)
w.close()
