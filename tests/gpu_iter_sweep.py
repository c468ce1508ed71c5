"""Diagnostic (not a test): cost of the one-body solver kernel versus the number of sweeps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S, engine as E
scene = S.grid_tiles(124, side=90, seed=2, lattice_cols=11)
n = len(scene["body_transforms"]) - 1
w = E.World(scene, flags=0, max_contacts=6 * n)
w.step(70)
for it in (0, 1, 2, 4, 8, 16):
    w.enable_timing(True)
    w.step(5, iterations=it)
    kt = w.kernel_times(reset=True)
    w.enable_timing(False)
    print(it, " ".join("%s %.3f" % (k, v[0] / 5) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])[:3]))
    w.step(3)
