"""How stable is the contact key set from step to step in the landed window?  (dev script, GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S, engine as E
for name, scene in (("boxes 4 tiles", S.grid_tiles(4, side=90, seed=2, lattice_cols=2)), ("mixed 4 tiles", S.grid_tiles(4, side=90, sphere_fraction=0.5, seed=7, lattice_cols=2))):
    w = E.World(scene, flags=E.NH_FLAG_SYNC_COUNTS)
    w.step(70)
    prev, same_pairs, same_tags, n = None, 0, 0, 0
    for s in range(60):
        w.step(1)
        c = w.get_contacts()
        tags = c["tags"]
        pairs = np.unique(tags)
        if prev is not None:
            n += 1
            same_tags += int(len(tags) == len(prev[0]) and np.array_equal(tags, prev[0]))
            same_pairs += int(len(pairs) == len(prev[1]) and np.array_equal(pairs, prev[1]))
        prev = (tags, pairs)
    print(name, "steps", n, "identical contact tag arrays", same_tags, "identical pair key sets", same_pairs, "contacts", len(tags), "pairs", len(pairs))
    w.close()
