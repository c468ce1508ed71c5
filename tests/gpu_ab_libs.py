"""Dev script (GPU): A/B of library builds on the landed 1M-box world (still steps): ms/step over 100-step nh_step calls + per-kernel table.
usage: python tests/gpu_ab_libs.py lib1.so lib2.so ...   (each is loaded in a fresh process)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import time, torch
    from nudge_amd import scenes as S, engine as E
    tiles = int(os.environ.get("AB_TILES", "124"))
    scene = S.grid_tiles(tiles, side=90, seed=2, lattice_cols=11 if tiles == 124 else None)
    nb = len(scene["body_transforms"]) - 1
    w = E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP, max_contacts=6 * nb)
    w.step(70); w.synchronize()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); w.step(60); w.synchronize(); best = min(best, (time.perf_counter() - t0) / 60)
    w.enable_timing(True); w.step(4); w.synchronize(); kt = w.kernel_times(); w.enable_timing(False)
    c = w.counts()
    print(json.dumps(dict(lib=os.environ.get("NUDGE_HIP_LIBRARY", "default"), ms_per_step=1e3 * best, still=c["still_steps"], replays=c["still_replays"],
                          kernels={k: round(v[0] / 4, 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])[:5]})))
    sys.exit(0)
for lib in sys.argv[1:] or ["default"]:
    env = dict(os.environ)
    if lib != "default":
        env["NUDGE_HIP_LIBRARY"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
    r = subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-500:]])[-1], flush=True)
