"""Dev script (GPU): how the 16M-box world of config 5 goes through its landed window on ONE GPU -- still steps, replays, re-insertions per 20 steps, with and
without the bench's restore between the blocks.   python tests/gpu_still_trace16.py [tiles]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S, engine as E
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 1984
import math
cols = max(2, int(round(math.sqrt(tiles) - 0.5)))
scene = S.grid_tiles(tiles, side=90, seed=2, lattice_cols=cols)
nb = len(scene["body_transforms"]) - 1
w = E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP, max_contacts=6 * nb)
w.step(70); w.synchronize()
snap = w.snapshot()
keys = ("still_steps", "still_replays", "broadphase_inserts", "broadphase_rebuilds", "still_diff_key", "still_diff_count", "still_diff_feature", "still_diff_escape")
prev = w.counts()
for rnd in range(12):
    if rnd >= 6:
        w.restore(snap); w.step(1)
        prev = w.counts()
    w.synchronize(); t0 = time.perf_counter(); w.step(20); w.synchronize(); dt = time.perf_counter() - t0
    c = w.counts()
    print(("restore + " if rnd >= 6 else "") + "20 steps: %.3f ms/step  " % (50.0 * dt) + "  ".join(f"{k} +{c[k] - prev[k]}" for k in keys), flush=True)
    prev = c
