"""What a world that is PARTLY asleep costs (nh_internal.h, "SLEEPERS form").  The 1,004,400-box drop scene of bench.py falls asleep (step ~320); then the caller wakes the
boxes of `awake` of its 124 tiles (idle counters rewritten, nh_bodies_changed) -- they rest awake for ~255 steps while the others sleep.  Timed: 200 of those steps.
Columns: default (still steps in sleepers form) | NH_NO_LOCAL_STILL=1 (round 4: every such step in full) | NH_NO_STILL=1.
python tests/gpu_sleepers_rates.py [tiles]        (-> profiles/rNN_sleepers_rates.log)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S   # noqa: E402
from nudge_amd import engine as E   # noqa: E402

table = "--table" in sys.argv          # (per-kernel times of 20 extra steps of the default column, each kernel alone on the GPU)
if table: sys.argv.remove("--table")
only = None          # --only 0.5: that fraction, default library only (for a rocprofv3 run)
if "--only" in sys.argv:
    k = sys.argv.index("--only"); only = float(sys.argv[k + 1]); del sys.argv[k:k + 2]
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 124
scene = S.grid_tiles(tiles, side=90, seed=2, lattice_cols=11 if tiles == 124 else None)
nb = len(scene["body_transforms"])
tile_of = scene["tile_of_body"] if "tile_of_body" in scene else None
flags = E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP
print(f"{nb - 1:,} boxes in {tiles} tiles; the world asleep, then the boxes of K tiles woken; steps/s over 200 steps")
print("tiles awake | default (sleepers form) | NH_NO_LOCAL_STILL=1 | NH_NO_STILL=1")
order = np.argsort(scene["body_transforms"]["position"][1:, 0], kind="stable") + 1
for frac in ((only,) if only else (0.5, 0.1, 0.9)):
    row, detail = [], []
    for env in ((None,) if only else (None, "NH_NO_LOCAL_STILL", "NH_NO_STILL")):
        if env: os.environ[env] = "1"
        w = E.World(scene, flags=flags, max_contacts=6 * nb)
        if env: os.environ.pop(env, None)
        w.step(400); w.synchronize()
        c = w.counts()
        assert c["active_bodies"] == 0, c
        idle = w.get_bodies()["idle"].copy()
        idle[order[: int(frac * (nb - 1))]] = 0           # (the left part of the world, by x)
        w.set_bodies(idle=idle)
        w.step(5); w.synchronize()
        c0 = w.counts()
        t0 = time.perf_counter(); w.step(200); w.synchronize(); t = time.perf_counter() - t0
        c1 = w.counts()
        assert c1["error"] == 0 and 0 < c1["active_bodies"] < nb - 1, c1
        row.append(200 / t)
        if table and env is None:
            w.enable_timing(True); w.step(20); w.synchronize()
            kt = w.kernel_times()
            print(f"  [{frac:.0%} awake, per step] " + ", ".join(f"{k} {ms / 20 * 1e3:.1f} us x{n / 20:.1f}" for k, (ms, n) in sorted(kt.items(), key=lambda x: -x[1][0])[:12]), flush=True)
            w.enable_timing(False)
        detail.append(f"{c1['active_bodies']:,} active, {c1['sleeping_pairs']:,} sleeping pairs, {c1['still_steps'] - c0['still_steps']} still steps, {c1['still_replays'] - c0['still_replays']} replayed")
        w.close()
    print(f"{frac:4.0%} | " + " | ".join(f"{r:8.1f} ({d})" for r, d in zip(row, detail)), flush=True)
