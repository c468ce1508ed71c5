"""Step rate of the 1M-box drop scene by phase (dev script, GPU box): falling, landed and awake, falling asleep, asleep.
   python tests/gpu_asleep_phase.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S, engine as E
scene = S.grid_tiles(124, side=90, seed=2, lattice_cols=11)
n = len(scene["body_transforms"]) - 1
w = E.World(scene, flags=E.NH_FLAG_FUSED_STEP, max_contacts=6 * n)
done = 0
for upto in (40, 60, 160, 300, 400, 600, 900, 1000, 1100):
    w.synchronize(); t0 = time.perf_counter()
    w.step(upto - done); w.synchronize()
    dt = time.perf_counter() - t0
    c = w.counts()
    print("steps [%4d, %4d): %7.1f steps/s   contacts %8d  active %8d  sleeping pairs %8d  broadphase rebuilds so far %d" %
          (done, upto, (upto - done) / dt, c["contacts"], c["active_bodies"], c["sleeping_pairs"], c["broadphase_rebuilds"]), flush=True)
    done = upto
# per-kernel table of the asleep phase (HIP events around every launch: slower than the rates above, the shares are what matters)
if hasattr(w, "enable_timing"):
    w.enable_timing(True)
    acc = {}
    for _ in range(20):
        w.step(1); w.synchronize()
        for name, (ms, launches) in w.kernel_times().items():
            a = acc.setdefault(name, [0.0, 0]); a[0] += ms; a[1] += launches
    tot = sum(a[0] for a in acc.values())
    print("asleep step, per kernel (ms/step, launches/step): total %.4f ms" % (tot / 20))
    for name, a in sorted(acc.items(), key=lambda kv: -kv[1][0])[:30]:
        print("  %-24s %8.4f  %5.1f" % (name, a[0] / 20, a[1] / 20))
