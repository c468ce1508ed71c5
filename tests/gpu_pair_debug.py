"""Dev script (GPU): why are pair-ahead steps refused?  Prints the pair_diag counters of a small landed world call by call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import engine as E, scenes as S
scene = S.grid_tiles(2, side=36, seed=63, lattice_cols=2)
w = E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP)
w.step(90)
keys = ("still_steps", "still_replays", "ahead_steps", "pair_steps", "pair_diag_roles", "pair_diag_record", "pair_diag_scale", "pair_diag_owned", "still_diff_key", "still_diff_count", "still_diff_feature", "still_diff_escape", "records" if False else "pairs", "contacts")
prev = w.counts()
for n in (20, 20, 20, 5, 7, 20):
    w.step(n)
    c = w.counts()
    print(n, {k: c[k] - prev[k] if k not in ("pairs", "contacts") else c[k] for k in keys})
    prev = c
w.close()
