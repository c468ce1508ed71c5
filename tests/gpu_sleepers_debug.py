"""Debug aid (round 6, sleepers ahead): a small drop scene asleep, a third woken, calls of 30 sub-steps; the counters that say why a pair step was refused."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S, engine as E
scene = S.grid_tiles(4, side=40, seed=5, lattice_cols=2)
nb = len(scene["body_transforms"])
w = E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP, max_contacts=6 * nb)
w.step(400); w.synchronize()
print("asleep:", {k: w.counts()[k] for k in ("active_bodies", "sleeping_pairs", "contacts")})
order = np.argsort(scene["body_transforms"]["position"][1:, 0], kind="stable") + 1
idle = w.get_bodies()["idle"].copy(); idle[order[: (nb - 1) // 3]] = 0; w.set_bodies(idle=idle)
keys = ("active_bodies", "sleeping_pairs", "contacts", "still_steps", "still_replays", "pair_steps", "ahead_steps", "pair_diag_roles", "pair_diag_record", "pair_diag_scale", "pair_diag_owned", "still_diff_key", "still_diff_count", "still_diff_feature", "still_diff_escape", "error")
for k in range(12):
    w.step(10); w.synchronize()
    c = w.counts()
    print(k, {x: c.get(x) for x in keys}, flush=True)
