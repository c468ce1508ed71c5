"""A/B timing of two builds of libnudge_hip.so on ONE box (run-to-run differences between boxes are ~3 %, larger than most kernel
tweaks): alternates the builds, `rounds` times each, and prints steps/s + the per-kernel table of the chosen kernels.
  python tests/gpu_ab.py nudge_amd/_ab/libA.so nudge_amd/_ab/libB.so ... [rounds] [kernel,kernel...] [bench.py arguments]"""
import json
import os
import subprocess
import sys

libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
rounds = int(rest[0]) if rest else 3
kernels = rest[1].split(",") if len(rest) > 1 else ["solve_one_body", "narrowphase", "radix_pass"]
extra = rest[2:]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, NUDGE_HIP_LIBRARY=os.path.abspath(l))
        p = subprocess.run([sys.executable, "bench.py", "--no-cpu", "--profile-kernels"] + extra, env=env, capture_output=True, text=True, timeout=600)
        line = [x for x in p.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", p.stderr[-500:])
            continue
        v = json.loads(line[-1])["value"]
        ks = {}
        for x in p.stderr.splitlines():
            f = x.split()
            if len(f) >= 3 and f[0] in kernels:
                ks[f[0]] = float(f[1])
        res[l].append(v)
        print(f"{os.path.basename(l)} round {r}: {v:8.1f} steps/s  " + "  ".join(f"{k} {ks.get(k, 0) * 1e3:6.1f} us" for k in kernels), flush=True)
for l in libs:
    if res[l]:
        print(f"{os.path.basename(l)}: mean {sum(res[l]) / len(res[l]):8.1f}  best {max(res[l]):8.1f}")
