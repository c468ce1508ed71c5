import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S, engine as E
scene = S.grid_tiles(2, side=40, sphere_fraction=0.5, seed=53, lattice_cols=2)
w = E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP)
w.step(400)
for k in range(6):
    w.step(3)
    c = w.counts()
    print({k2: c[k2] for k2 in ("active_bodies", "contacts", "general_contacts", "sleeping_pairs", "cache", "culled", "pairs", "error", "asleep_steps", "still_steps")})
