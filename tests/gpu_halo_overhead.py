"""Diagnostic (not a test): per-step cost of the halo exchange code path (pack, hand-over, unpack) on one GPU."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nudge_amd import scenes as S, engine as E, partition as PT
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 124
g = S.grid_tiles(tiles, side=90, seed=2, lattice_cols=22)
x = g["body_transforms"]["position"][1:, 0]
cuts = [float(np.median(x)) + 0.41]
parts, singles = [], []
for r in range(2):
    local, info = PT.split_scene(g, cuts, r)
    w = E.World(local, flags=0, capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=6 * info["capacity"]["bodies"])
    parts.append(PT.Partition(w, info, epoch=16))
    singles.append(E.World(local, flags=0, max_contacts=6 * info["capacity"]["bodies"]))
cl = PT.LocalCluster(parts)
cl.step(64); [s.step(64) for s in singles]
torch.cuda.synchronize()
for name, fn in (("partitioned (halo every step, refresh every 16)", lambda: cl.step(96)), ("two plain worlds", lambda: [s.step(96) for s in singles])):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {1e3 * dt / 96:.3f} ms per step of both halves")
print("ghosts", [p.stats["ghosts"] for p in parts], "bodies", [p.n_owned for p in parts])
