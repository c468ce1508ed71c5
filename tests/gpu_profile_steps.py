"""Per-step kernel timing on the 1M-box scene (diagnostic, not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S, engine as E
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 124
scene = S.grid_tiles(tiles, side=90, seed=2, lattice_cols=11)
n = len(scene["body_transforms"]) - 1
w = E.World(scene, flags=0, max_contacts=6 * n)
w.enable_timing(True)
for s in range(int(sys.argv[2]) if len(sys.argv) > 2 else 70):
    w.step(1)
    kt = w.kernel_times(reset=True)
    c = w.counts()
    top = sorted(kt.items(), key=lambda kv: -kv[1][0])[:6]
    tot = sum(v[0] for v in kt.values())
    if s < 6 or s % 8 == 0:
        print(s, "K", c["contacts"], "P", c["pairs"], "large", c["large_colliders"], "total %.2f ms |" % tot, " ".join("%s %.3f" % (k, v[0]) for k, v in top))
