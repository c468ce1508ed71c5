"""What the LENGTH of an nh_step call costs (round 6: early counters).  The 1,004,400-box drop scene of bench.py, landed and awake (steps 70 ..): 200 steps (the world dozes off around step 300) taken as calls of
1, 2, 5, 20 and 100 sub-steps, steps/s by the wall clock (one synchronize at the end of the 200).  Columns: default | option no_early_counts (every call's last verdict waits for
its solver to END, as before).  The sample's own loop (example/main.cpp:274-328) is the one-sub-step row.
python tests/gpu_call_rates.py [tiles]        (-> profiles/rNN_call_rates.log)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S   # noqa: E402
from nudge_amd import engine as E   # noqa: E402

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 124
scene = S.grid_tiles(tiles, side=90, seed=2, lattice_cols=11 if tiles == 124 else None)
nb = len(scene["body_transforms"])
flags = E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP
print(f"{nb - 1:,} boxes in {tiles} tiles, landed and awake; 200 steps in calls of K sub-steps, best of 4; steps/s (wall clock)")
print("   K | default | no_early_counts")
w = E.World(scene, flags=flags, max_contacts=6 * nb)          # (ONE world, the option switched between the columns: two worlds differ by 2 % by where their arrays lie)
w.step(70); w.synchronize()
snap = w.snapshot()
for K in (1, 2, 5, 20, 100):
    row = []
    for off in (0, 1):
        w.set_option("no_early_counts", off)
        best = 0.0
        for rep in range(4):
            w.restore(snap); w.step(2); w.synchronize()
            t0 = time.perf_counter()
            for _ in range(200 // K):
                w.step(K)
            w.synchronize()
            best = max(best, 200 / (time.perf_counter() - t0))
        row.append(best)
    print(f"{K:4d} | {row[0]:8.1f} | {row[1]:8.1f}", flush=True)
c = w.counts()
assert c["error"] == 0, c
print("still steps / replays:", (c["still_steps"], c["still_replays"]))
