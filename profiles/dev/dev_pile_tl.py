import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from nudge_amd import scenes as S, engine as E
scene = S.pile(n_boxes=1024, n_spheres=512, seed=9, iterations=20)
w = E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP)
w.step(600); w.synchronize()
t0 = time.perf_counter(); w.step(200); w.synchronize(); t = time.perf_counter() - t0
c = w.counts()
print(f"sample pile: {200 / t:.1f} steps/s, contacts {c['contacts']}, general {c['general_contacts']}, levels {c['levels']}, active {c['active_bodies']}, sleeping pairs {c['sleeping_pairs']}")
