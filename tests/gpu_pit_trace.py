"""Diagnostic (not a test): wall time and counts of the ball pit every few steps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S, engine as E
side = int(sys.argv[1]) if len(sys.argv) > 1 else 160
total = int(sys.argv[2]) if len(sys.argv) > 2 else 160
t0 = time.time()
scene = S.ball_pit(side, side, side, seed=4)
n = len(scene["body_transforms"]) - 1
print("scene", n, "%.1fs" % (time.time() - t0), flush=True)
w = E.World(scene, flags=0, max_contacts=10 * n)
print("world %.1fs" % (time.time() - t0), flush=True)
for s in range(0, total, 10):
    t1 = time.time()
    w.step(10)
    w.synchronize()
    c = w.counts()
    print(s + 10, "%.1f ms/step" % ((time.time() - t1) * 100), "K", c["contacts"], "P", c["pairs"], "general", c["general_contacts"], "levels", c["levels"], "err", c["error"], flush=True)
