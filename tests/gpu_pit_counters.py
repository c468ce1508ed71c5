"""Dev script (GPU): what the broadphase / sort of the settled 4 M-sphere pit (config 4) do step by step: rebuilds, re-insertions, sort re-uses, pairs, contacts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import engine as E, scenes as S
side = int(sys.argv[1]) if len(sys.argv) > 1 else 160
scene = S.ball_pit(side, side, side, seed=4)
nb = len(scene["body_transforms"]) - 1
w = E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP, max_contacts=5 * nb, max_pairs=7 * nb)
t0 = time.time(); w.step(500); w.synchronize(); print("500 steps", round(time.time() - t0, 1), "s")
prev = w.counts()
for k in range(8):
    t0 = time.time(); w.step(2); w.synchronize(); dt = time.time() - t0
    c = w.counts()
    print(k, {x: c[x] - prev[x] for x in ("broadphase_rebuilds", "broadphase_inserts", "sort_reuses")}, "pairs", c["pairs"], "raw", c["raw_pairs"], "contacts", c["contacts"], "general", c["general_contacts"], "active", c["active_bodies"], "ms/step", round(1e3 * dt / 2, 2))
    prev = c
w.close()
