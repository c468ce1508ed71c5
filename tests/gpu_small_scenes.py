"""Steps/s of small worlds on the device-resident C ABI (dev script, run on the GPU box): the general (multi-body island) path with and
without the one-workgroup LDS-resident solver (NH_NO_RESIDENT=1 selects one launch per level and sweep).
  python tests/gpu_small_scenes.py"""
import os
import subprocess
import sys
import time

CASES = [("pile256 (config 1: 256 boxes, I = 4)", "pile", dict(n_boxes=256, seed=1, iterations=4), 600, 300),
         ("stacks 64 x 3", "stacks", dict(), 100, 300),
         ("pit 6^3 spheres", "ball_pit", dict(nx=6, ny=6, nz=6, seed=4), 200, 300),
         ("pit 9x9x9 spheres", "ball_pit", dict(nx=9, ny=9, nz=9, seed=4), 250, 300),
         ("pit 12^3 spheres", "ball_pit", dict(nx=12, ny=12, nz=12, seed=4), 300, 300),
         ("sample pile: 1024 boxes + 512 spheres, I = 20", "pile", dict(n_boxes=1024, n_spheres=512, seed=1, iterations=20), 900, 200)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nudge_amd import scenes as S, engine as E
    for name, fn, kw, warm, steps in CASES:
        for flags, label in ((E.NH_FLAG_EXACT_ORDER, "exact order"), (0, "default order")):
            w = E.World(getattr(S, fn)(**kw), flags=flags)
            w.step(warm)
            w.synchronize()
            t0 = time.perf_counter()
            w.step(steps)
            w.synchronize()
            dt = time.perf_counter() - t0
            c = w.counts()
            print(f"  {name:40s} {label:14s} {steps / dt:9.1f} steps/s   general contacts {c['general_contacts']:5d}  levels {c['levels']:4d}", flush=True)
            w.close()
else:
    for env, label in (({}, "LDS-resident solver (default)"), ({"NH_NO_RESIDENT": "1"}, "one launch per level (NH_NO_RESIDENT=1)")):
        print(label, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=True)
