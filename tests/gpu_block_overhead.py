"""Dev script (GPU): where do the ~0.4 ms per timed block of bench.py go?  Same world, blocks of K steps from a device checkpoint, t(K) = a + b K."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nudge_amd import scenes as S, engine as E
scene = S.grid_tiles(124, side=90, seed=2, lattice_cols=11)
nb = len(scene["body_transforms"]) - 1
w = E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP, max_contacts=6 * nb)
w.step(70); w.synchronize()
snap = w.snapshot()
for mode in ("restore+1", "no restore"):
    for K in (1, 2, 5, 10, 20, 50, 100, 200):
        ts = []
        for rep in range(12):
            if mode == "restore+1":
                w.restore(snap); w.step(1)
            c0 = w.counts(); torch.cuda.synchronize()
            t0 = time.perf_counter(); w.step(K); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort(); t = sum(ts[2:-2]) / len(ts[2:-2])
        c = w.counts()
        print(f"{mode:12s} K {K:4d}: {1e3 * t:8.3f} ms per block = {1e3 * t / K:7.4f} ms per step   (still steps so far {c['still_steps']}, replays {c['still_replays']})", flush=True)
