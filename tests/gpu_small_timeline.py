"""Dev script (GPU box, under rocprofv3 --kernel-trace): the sample application's pile (1024 boxes + 512 spheres, 20 iterations) stepped on the resident C ABI; the
timeline of one steady step is printed by tools/gpu_timeline_cmd.sh from the trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S, engine as E
w = E.World(S.pile(n_boxes=1024, n_spheres=512, seed=1, iterations=20), flags=0)
w.step(900); w.synchronize()
t0 = time.perf_counter(); w.step(200); w.synchronize(); dt = time.perf_counter() - t0
print("wall: %.3f ms/step (%.0f steps/s)" % (1e3 * dt / 200, 200 / dt))
c = w.counts(); print({k: c[k] for k in ("contacts", "general_contacts", "levels", "pairs", "active_bodies", "still_steps")})
