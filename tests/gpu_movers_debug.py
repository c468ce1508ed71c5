"""Dev script (GPU): does a still step in movers form start at the narrowphase (xform ahead in movers form)?  One flier over a landed world, 12-step calls."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S, engine as E
big = len(sys.argv) > 1
scene = S.grid_tiles(124, side=90, seed=2) if big else S.grid_tiles(4, side=60, seed=2, lattice_cols=2)
nb = len(scene["body_transforms"])
w = E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP)
w.step(70)
m = w.get_bodies()["momentum"].copy(); m["velocity"][777] = (0.5, 30.0, 0.3); w.set_bodies(momentum=m)
prev = w.counts()
for k in range(8):
    w.step(12); w.synchronize(); c = w.counts()
    print(k, {x: c[x] - prev[x] for x in ("still_steps", "still_replays", "ahead_steps", "pair_steps", "broadphase_inserts", "broadphase_rebuilds", "still_diff_escape", "still_diff_key", "still_diff_count")})
    prev = c
w.enable_timing(True); w.step(24); w.synchronize()
kt = w.kernel_times()
print(", ".join(f"{k} {ms / 24 * 1e3:.1f} us x{n / 24:.2f}" for k, (ms, n) in sorted(kt.items(), key=lambda x: -x[1][0])[:10]))
w.close()
