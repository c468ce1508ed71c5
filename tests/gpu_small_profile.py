"""Where a small world's step goes (dev script, GPU box): per-kernel GPU time and launch count, and host wall time per API call, for the sample
application's pile on the device-resident C ABI.   python tests/gpu_small_profile.py [exact]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nudge_amd import scenes as S, engine as E
flags = E.NH_FLAG_EXACT_ORDER if len(sys.argv) > 1 and sys.argv[1] == "exact" else 0
w = E.World(S.pile(n_boxes=1024, n_spheres=512, seed=1, iterations=20), flags=flags)
w.step(900); w.synchronize()
t0 = time.perf_counter(); w.step(200); w.synchronize(); dt = time.perf_counter() - t0
print("wall: %.3f ms/step (%.0f steps/s)" % (1e3 * dt / 200, 200 / dt))
# host wall time per call, each call followed by a synchronize (so: the call's own GPU work + its host round trips)
names = ["collide", "gravity", "read_cache", "setup", "apply", "update", "write_cache", "advance"]
acc = {n: 0.0 for n in names}
for _ in range(100):
    for n in names:
        t = time.perf_counter(); getattr(w, n)(); w.synchronize(); acc[n] += time.perf_counter() - t
print("per call (synchronised after each): " + "  ".join("%s %.3f" % (n, 1e3 * acc[n] / 100) for n in names), " sum %.3f ms" % (1e3 * sum(acc.values()) / 100))
w.enable_timing(True)
w.step(50); w.synchronize()
kt = w.kernel_times(reset=True)
tot = sum(v[0] for v in kt.values()); nl = sum(v[1] for v in kt.values())
print("GPU kernel time %.3f ms/step in %.1f launches/step" % (tot / 50, nl / 50))
for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %-24s %.4f ms/step  %.1f launches/step" % (k, v[0] / 50, v[1] / 50))
c = w.counts()
print("general contacts", c["general_contacts"], "levels", c["levels"], "contacts", c["contacts"])
