"""Dev script (GPU): how a million boxes come to rest -- per 10 steps: still steps confirmed, replays, and what the failed ones found changed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S, engine as E
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 124
scene = S.grid_tiles(tiles, side=90, seed=2, lattice_cols=11 if tiles == 124 else None)
nb = len(scene["body_transforms"]) - 1
w = E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP, max_contacts=6 * nb)
prev = w.counts()
import torch
for s0 in range(0, 320, 10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w.step(10)
    w.synchronize(); dt = time.perf_counter() - t0
    c = w.counts()
    d = {k: c[k] - prev[k] for k in ("still_steps", "still_replays", "still_diff_key", "still_diff_count", "still_diff_feature", "still_diff_escape", "broadphase_rebuilds", "broadphase_inserts", "sort_reuses")}
    print(f"steps [{s0:3d},{s0+10:3d}) {1e2*dt:7.3f} ms/step contacts {c['contacts']:8d} active {c['active_bodies']:8d} " + " ".join(f"{k.replace('still_','').replace('broadphase_','bp_')}={v}" for k, v in d.items()), flush=True)
    prev = c
