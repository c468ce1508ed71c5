"""What a few moving bodies cost a large world at rest, with and without re-insertion into the kept pair list (nh_collide.hip, "7. re-insertion").
The 1,004,400-box drop scene of bench.py, landed.  "hop": every 12 steps K boxes are kicked (untimed) and hop across their neighbours -- they leave their
inflated boxes for a few steps, then rest again.  "fly": K boxes are thrown up once and are in the air for the whole window -- each of them leaves its box on
EVERY step.  Timed: 96 steps (between the kicks).  Three broadphase modes: default (leavers re-inserted), NH_NO_INCREMENTAL=1 (a leaver
forces a rebuild -- after eight in a row the library searches the grid every step), NH_NO_KEPT_PAIRS=1 (search every step).
python tests/gpu_movers_rates.py [tiles]        (-> profiles/rNN_movers_reinsertion.log)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S   # noqa: E402
from nudge_amd import engine as E   # noqa: E402

quick = "--quick" in sys.argv          # (round 5) three columns: local speculation (default), global speculation (NH_NO_LOCAL_STILL=1, round 4), none (NH_NO_STILL=1)
argv = [x for x in sys.argv[1:] if not x.startswith("--")]
tiles = int(argv[0]) if argv else 124
scene = S.grid_tiles(tiles, side=90, seed=2)
nb = len(scene["body_transforms"])
flags = E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP
print(f"{nb - 1:,} boxes in {tiles} tiles, landed and awake; steps/s over 96 steps")
_w = E.World(scene, flags=flags); _w.step(200); _w.synchronize(); _w.close()          # (clocks and allocator warm before the first measured world)
for mode in ("hop", "fly"):
    print(f"--- {mode}: " + ("K boxes kicked every 12 steps" if mode == "hop" else "K boxes thrown up once, in the air throughout"))
    columns = (None, "NH_NO_LOCAL_STILL", "NH_NO_STILL") if quick else (None, "NH_NO_LOCAL_STILL", "NH_NO_STILL", "NH_NO_INCREMENTAL", "NH_NO_KEPT_PAIRS")
    print("movers | " + " | ".join("default (still steps re-insert their own leavers)" if e is None else e + "=1" for e in columns))
    for K in ((0, 1, 16, 256, 2048) if mode == "hop" else (1, 16, 256, 2048)):
        row, detail = [], []
        for env in columns:
            if env: os.environ[env] = "1"
            w = E.World(scene, flags=flags)
            if env: os.environ.pop(env, None)
            w.step(70)
            rng = np.random.default_rng(5)
            movers = rng.choice(np.arange(1, nb), size=K, replace=False) if K else np.zeros(0, np.int64)
            c0 = w.counts()
            t = 0.0
            for rnd in range(8):
                if K and (mode == "hop" or rnd == 0):
                    m = w.get_bodies()["momentum"].copy()
                    kick = rng.uniform(-3.0, 3.0, size=(K, 3)).astype(np.float32); kick[:, 1] = 2.5 if mode == "hop" else 30.0
                    m["velocity"][movers] = kick
                    w.set_bodies(momentum=m)
                w.synchronize()
                t0 = time.perf_counter()
                w.step(12)
                w.synchronize()
                t += time.perf_counter() - t0
            c1 = w.counts()
            assert c1["error"] == 0
            row.append(8 * 12 / t)
            detail.append(f"{c1['broadphase_rebuilds'] - c0['broadphase_rebuilds']} rebuilds, {c1['broadphase_inserts'] - c0['broadphase_inserts']} re-insertions"
                          + (f", {c1['still_steps'] - c0['still_steps']} still steps, {c1['still_replays'] - c0['still_replays']} replayed" if env in (None, "NH_NO_LOCAL_STILL") else ""))
            w.close()
        print(f"{K:6d} | " + " | ".join(f"{r:8.1f} ({d})" for r, d in zip(row, detail)), flush=True)
