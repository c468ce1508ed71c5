"""Helpers for the call-site tests: scene files for tests/callsite/replay.cpp and its output."""
import os
import subprocess

import numpy as np

from nudge_amd import scenes as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPLAY_REF = os.path.join(ROOT, "oracle", "_ref", "replay_ref")                      # replay.cpp + the reference's nudge.cpp (checker)
REPLAY_HIP = os.path.join(ROOT, "nudge_amd", "compat", "_build", "replay_hip")       # replay.cpp + the C++ drop-in + libnudge_hip.so
RESIDENT_REPLAY = os.path.join(ROOT, "nudge_amd", "resident_replay")                          # examples/resident_replay.cpp: C++ host on the device-resident C ABI
PARTITION_RCCL = os.path.join(ROOT, "nudge_amd", "partition_rccl")                            # examples/partition_rccl.cpp: C++ host of the multi-GPU path (RCCL)
REF_TESTS_HIP = os.path.join(ROOT, "nudge_amd", "compat", "_build", "ref_tests_hip")  # the reference's own tests/main.cpp on the GPU backend


def write_scene(path, scene, max_contacts=None):
    """Reference (16-bit) layouts, in the order replay.cpp reads them."""
    assert S.narrow_ok(scene)
    nb, nbox, nsph = len(scene["body_transforms"]), len(scene["box_tags"]), len(scene["sphere_tags"])
    p = scene["params"]
    if max_contacts is None:
        max_contacts = max(4096, 16 * nb)
    with open(path, "wb") as f:
        f.write(np.array([nb, nbox, nsph, p["iterations"], max_contacts], dtype=np.uint32).tobytes())
        f.write(np.array([p["time_step"], p["gravity"], p["damping_rate"]], dtype=np.float32).tobytes())
        for k in ("body_transforms", "body_properties", "body_momentum", "idle_counters"):
            f.write(np.ascontiguousarray(scene[k]).tobytes())
        f.write(scene["box_tags"].astype(np.uint16).tobytes())
        f.write(np.ascontiguousarray(scene["box_data"]).tobytes())
        f.write(np.ascontiguousarray(scene["box_transforms"]).tobytes())
        f.write(scene["sphere_tags"].astype(np.uint16).tobytes())
        f.write(np.ascontiguousarray(scene["sphere_data"]).tobytes())
        f.write(np.ascontiguousarray(scene["sphere_transforms"]).tobytes())
    return nb


def read_result(path, nb):
    raw = open(path, "rb").read()
    o = 0
    t = np.frombuffer(raw, dtype=S.TRANSFORM, count=nb, offset=o); o += 32 * nb
    m = np.frombuffer(raw, dtype=S.MOMENTUM, count=nb, offset=o); o += 32 * nb
    idle = np.frombuffer(raw, dtype=np.uint8, count=nb, offset=o); o += nb
    sum_contacts = int(np.frombuffer(raw, dtype=np.uint64, count=1, offset=o)[0]); o += 8
    last_contacts, last_active = [int(x) for x in np.frombuffer(raw, dtype=np.uint32, count=2, offset=o)]; o += 8
    checksum = int(np.frombuffer(raw, dtype=np.uint64, count=1, offset=o)[0]) if len(raw) >= o + 8 else None
    return dict(transforms=t, momentum=m, idle=idle, sum_contacts=sum_contacts, last_contacts=last_contacts, last_active=last_active, checksum=checksum)


def run_replay(binary, scene_path, out_path, steps, timeout=600, extra=()):
    r = subprocess.run([binary, scene_path, out_path, str(steps)] + list(extra), capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"{binary} failed ({r.returncode}):\n{r.stdout}\n{r.stderr}")
    return r.stdout
