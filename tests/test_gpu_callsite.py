"""The drop-in boundary in the reference's own language: C++ call sites of `namespace nudge` (nudge.h:134-146),
linked against the GPU backend (nudge_amd/compat/nudge_compat.cpp over the C ABI of include/nudge_hip.h).

  * tests/callsite/replay.cpp is one call site compiled twice -- against the reference's nudge.cpp (exact-math
    variant, the checker) and against the drop-in; both must leave bit-identical worlds;
  * the reference's OWN test program (tests/main.cpp: 6 known-answer / property tests of collide) runs unmodified
    on the drop-in.
The binaries are built where the reference's header exists (`make -C nudge_amd/compat`, `make -C oracle`) and travel
with the repository; the tests fail, not skip, when a GPU is present and they are missing.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import callsite_util as CU      # noqa: E402
import make_golden as G         # noqa: E402
import parity_util as P         # noqa: E402
from nudge_amd import scenes as S   # noqa: E402
from oracle import refworld     # noqa: E402


def test_replay_call_site_is_the_reference_step_loop(tmp_path):
    """CPU: replay.cpp linked against the reference reproduces the golden fixture (so it IS a faithful call site)."""
    if not os.path.exists(CU.REPLAY_REF):
        pytest.skip("oracle/_ref/replay_ref not built (needs /root/reference)")
    scene, steps = G.build("pile256")
    nb = CU.write_scene(str(tmp_path / "s.bin"), scene)
    CU.run_replay(CU.REPLAY_REF, str(tmp_path / "s.bin"), str(tmp_path / "o.bin"), steps)
    r = CU.read_result(str(tmp_path / "o.bin"), nb)
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pile256.npz"))
    assert P.bits_equal(r["transforms"], fx["transforms"])
    assert P.bits_equal(r["momentum"]["velocity"], fx["momentum"]["velocity"])
    assert P.bits_equal(r["idle"], fx["idle"])


def _build(name):
    if name == "example":
        # the scene of the reference's sample application (example/main.cpp:390-432): 1024 boxes + 512 spheres dropped in one
        # column, 20 solver iterations (example/main.cpp:276) -- long enough that a pile has formed
        return S.pile(n_boxes=1024, n_spheres=512, seed=9, iterations=20), 420
    scene, steps = G.build(name)
    return scene, min(steps, 150)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pile256", "pile_mixed", "mixed20", "stacks", "pit6", "compound", "example"])
def test_same_call_site_two_backends_bit_identical(name, tmp_path):
    assert os.path.exists(CU.REPLAY_HIP), "nudge_amd/compat/_build/replay_hip missing: make -C nudge_amd/compat"
    scene, steps = _build(name)
    nb = CU.write_scene(str(tmp_path / "s.bin"), scene)
    CU.run_replay(CU.REPLAY_HIP, str(tmp_path / "s.bin"), str(tmp_path / "hip.bin"), steps)
    got = CU.read_result(str(tmp_path / "hip.bin"), nb)
    if os.path.exists(CU.REPLAY_REF):
        CU.run_replay(CU.REPLAY_REF, str(tmp_path / "s.bin"), str(tmp_path / "ref.bin"), steps)
        ref = CU.read_result(str(tmp_path / "ref.bin"), nb)
    else:
        assert refworld.available("exact"), "neither oracle/_ref/replay_ref nor the compiled reference travelled with the repo"
        w = refworld.RefWorld(scene)
        w.step(steps)
        b = w.bodies()
        ref = dict(transforms=b["transforms"], momentum=b["momentum"], idle=b["idle"], sum_contacts=w.sum_contacts(), last_contacts=None, last_active=None)
    assert P.bits_equal(got["transforms"], ref["transforms"])
    assert P.bits_equal(got["momentum"], ref["momentum"])          # including unused0 = mass_inverse (nudge.cpp:4198)
    assert P.bits_equal(got["idle"], ref["idle"])
    assert got["sum_contacts"] == ref["sum_contacts"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["zero_iterations", "peek_after_setup", "custom_impulses"])
@pytest.mark.parametrize("name", ["pile256", "mixed20"])
def test_unusual_call_patterns_two_backends_bit_identical(name, mode, tmp_path):
    """Call sites the plain loop does not cover, all legal use of the reference API: no apply_impulses at all (the warm start of
    setup_contact_constraints, nudge.cpp:4563-4632, must still reach momentum -- and stay in the caller's host arrays), reading
    momentum between setup and the first apply_impulses, and user impulses between two apply_impulses calls (the custom-constraint
    hook, example/main.cpp:314-317).  Same source file, reference vs GPU backend, bit-identical worlds and checksums."""
    assert os.path.exists(CU.REPLAY_HIP), "nudge_amd/compat/_build/replay_hip missing: make -C nudge_amd/compat"
    assert os.path.exists(CU.REPLAY_REF), "oracle/_ref/replay_ref missing: make -C oracle"
    scene, steps = _build(name)
    steps = min(steps, 110)
    nb = CU.write_scene(str(tmp_path / "s.bin"), scene)
    CU.run_replay(CU.REPLAY_HIP, str(tmp_path / "s.bin"), str(tmp_path / "hip.bin"), steps, extra=[mode])
    CU.run_replay(CU.REPLAY_REF, str(tmp_path / "s.bin"), str(tmp_path / "ref.bin"), steps, extra=[mode])
    got, ref = CU.read_result(str(tmp_path / "hip.bin"), nb), CU.read_result(str(tmp_path / "ref.bin"), nb)
    assert ref["sum_contacts"] > 0
    assert P.bits_equal(got["transforms"], ref["transforms"]), mode
    assert P.bits_equal(got["momentum"], ref["momentum"]), mode
    assert P.bits_equal(got["idle"], ref["idle"]), mode
    assert got["sum_contacts"] == ref["sum_contacts"] and got["checksum"] == ref["checksum"], mode


@pytest.mark.gpu
@pytest.mark.parametrize("name,steps", [("pile256", 300), ("grid30", 220), ("mixed20", 150)])
def test_custom_contacts_appended_after_collide_two_backends_bit_identical(name, steps, tmp_path):
    """"Custom contacts can be added here" (example/main.cpp:287): the call site appends contacts of its own behind the list collide() returned --
    a floor at y = 3 under the first dynamic bodies and a sideways push, with tags no collider pair produces -- and bumps contacts.count.  The
    reference sorts them in with everybody else (nudge.cpp:4027-4044); the drop-in uploads them, nh_append_contacts merges them into the
    device's tag order and counts the per-body bookkeeping again.  Same source file, two backends, bit-identical worlds."""
    assert os.path.exists(CU.REPLAY_HIP), "nudge_amd/compat/_build/replay_hip missing: make -C nudge_amd/compat"
    assert os.path.exists(CU.REPLAY_REF), "oracle/_ref/replay_ref missing: make -C oracle"
    scene, _ = _build(name)
    nb = CU.write_scene(str(tmp_path / "s.bin"), scene)
    CU.run_replay(CU.REPLAY_HIP, str(tmp_path / "s.bin"), str(tmp_path / "hip.bin"), steps, extra=["custom_contacts"])
    CU.run_replay(CU.REPLAY_REF, str(tmp_path / "s.bin"), str(tmp_path / "ref.bin"), steps, extra=["custom_contacts"])
    CU.run_replay(CU.REPLAY_REF, str(tmp_path / "s.bin"), str(tmp_path / "plain.bin"), steps)
    got, ref, plain = (CU.read_result(str(tmp_path / f), nb) for f in ("hip.bin", "ref.bin", "plain.bin"))
    assert not P.bits_equal(ref["transforms"], plain["transforms"]), "the custom contacts were meant to change the outcome"
    assert P.bits_equal(got["transforms"], ref["transforms"])
    assert P.bits_equal(got["momentum"], ref["momentum"])
    assert P.bits_equal(got["idle"], ref["idle"])
    assert got["sum_contacts"] == ref["sum_contacts"] and got["last_contacts"] == ref["last_contacts"]


@pytest.mark.gpu
def test_two_worlds_on_two_threads_are_independent(tmp_path):
    """The reference keeps no global mutable state: two threads may step two disjoint worlds at once (SURVEY 8(b) threading).  The drop-in keeps
    one session per world (found by the caller's arrays), each with its own context and stream: the call site steps two copies of a scene on
    two threads, both must equal the single-world reference run."""
    scene, steps = _build("pile256")
    nb = CU.write_scene(str(tmp_path / "s.bin"), scene)
    CU.run_replay(CU.REPLAY_HIP, str(tmp_path / "s.bin"), str(tmp_path / "hip.bin"), steps, extra=["two_worlds"])       # (exit code 3 if the two differ)
    CU.run_replay(CU.REPLAY_REF, str(tmp_path / "s.bin"), str(tmp_path / "ref.bin"), steps)
    got, ref = CU.read_result(str(tmp_path / "hip.bin"), nb), CU.read_result(str(tmp_path / "ref.bin"), nb)
    assert P.bits_equal(got["transforms"], ref["transforms"]) and P.bits_equal(got["momentum"], ref["momentum"]) and P.bits_equal(got["idle"], ref["idle"])


@pytest.mark.gpu
def test_reference_own_test_program_passes_on_the_gpu_backend():
    assert os.path.exists(CU.REF_TESTS_HIP), "nudge_amd/compat/_build/ref_tests_hip missing: make -C nudge_amd/compat"
    r = subprocess.run([CU.REF_TESTS_HIP], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "All tests passed." in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pile256", "mixed20", "pit6"])
def test_cpp_host_on_the_resident_abi_matches_reference(name, tmp_path):
    """examples/resident_replay.cpp: a C++ host that keeps every array in HBM and calls the C ABI directly (the performance path),
    against the reference stepping the same scene file."""
    assert os.path.exists(CU.RESIDENT_REPLAY), "nudge_amd/resident_replay missing: make -C nudge_amd/csrc"
    scene, steps = G.build(name)
    steps = min(steps, 150)
    nb = CU.write_scene(str(tmp_path / "s.bin"), scene)
    CU.run_replay(CU.RESIDENT_REPLAY, str(tmp_path / "s.bin"), str(tmp_path / "gpu.bin"), steps, extra=["exact"])
    got = CU.read_result(str(tmp_path / "gpu.bin"), nb)
    assert refworld.available("exact")
    w = refworld.RefWorld(scene)
    w.step(steps)
    b = w.bodies()
    assert P.bits_equal(got["transforms"], b["transforms"])
    assert P.bits_equal(got["momentum"], b["momentum"])
    assert P.bits_equal(got["idle"], b["idle"])
    assert got["sum_contacts"] == w.sum_contacts()


@pytest.mark.gpu
def test_cpp_host_with_one_nh_step_call_matches_reference(tmp_path):
    """The same C++ host making ONE call: nh_step(ctx, &args, steps) under NH_FLAG_FUSED_STEP -- the library drives the sample's loop itself (still steps with late
    verdicts while the boxes rest, the steps of the world asleep taken as done).  A drop tile through fall, rest and sleep: the reference's bits."""
    assert os.path.exists(CU.RESIDENT_REPLAY), "nudge_amd/resident_replay missing: make -C nudge_amd/csrc"
    scene, _ = G.build("grid30")
    steps = 420
    nb = CU.write_scene(str(tmp_path / "s.bin"), scene)
    CU.run_replay(CU.RESIDENT_REPLAY, str(tmp_path / "s.bin"), str(tmp_path / "gpu.bin"), steps, extra=["step"])
    got = CU.read_result(str(tmp_path / "gpu.bin"), nb)
    assert refworld.available("exact")
    w = refworld.RefWorld(scene)
    w.step(steps)
    b = w.bodies()
    assert P.bits_equal(got["transforms"], b["transforms"])
    assert P.bits_equal(got["momentum"], b["momentum"])
    assert P.bits_equal(got["idle"], b["idle"])
    assert (b["idle"][1:] == 0xff).all(), "the tile was meant to be asleep at the end"


@pytest.mark.gpu
def test_cpp_rccl_host_runs_a_slab_on_one_gpu():
    """examples/partition_rccl.cpp (C++ host of the multi-GPU path: x-slabs, ghost halo exchange with ncclSend / ncclRecv on the records of
    nh_halo_pack / nh_halo_unpack) with WORLD_SIZE = 1: the whole world is one slab, nothing is exchanged -- what a one-GPU box can check is that the
    program builds against RCCL, generates and steps its world, and that every box has landed on four contacts.  (Its communication path
    needs one GPU per rank: RCCL refuses two ranks on one device.)"""
    assert os.path.exists(CU.PARTITION_RCCL), "nudge_amd/partition_rccl missing: make -C nudge_amd/csrc"
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([CU.PARTITION_RCCL, "3", "2", "24", "90"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "3456 owned + 0 ghost bodies" in r.stdout and "(4.000 per owned+ghost body)" in r.stdout, r.stdout


@pytest.mark.gpu
def test_cpp_rccl_host_loopback_executes_rccl_between_the_halo_kernels():
    """RCCL on the one GPU there is: WORLD_SIZE = 1 and NUDGE_RCCL_LOOPBACK = 1 make examples/partition_rccl.cpp create a one-rank communicator and send the
    per-step halo records of 2304 bodies to ITSELF -- nh_halo_pack -> ncclGroupStart; ncclSend(self); ncclRecv(self); ncclGroupEnd -> nh_halo_update, all on
    the library's stream, every step.  The records update the bodies they were packed from, so the world must come out exactly as without the exchange
    (same checksum, four contacts per box): the communication path of the multi-GPU host has then really executed, kernels, RCCL and stream ordering."""
    assert os.path.exists(CU.PARTITION_RCCL), "nudge_amd/partition_rccl missing: make -C nudge_amd/csrc"
    base = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    outs = []
    for loop in ("0", "1"):
        r = subprocess.run([CU.PARTITION_RCCL, "2", "2", "24", "90"], capture_output=True, text=True, timeout=600, env=dict(base, NUDGE_RCCL_LOOPBACK=loop))
        assert r.returncode == 0, r.stdout + r.stderr
        assert "2304 owned + 0 ghost bodies" in r.stdout and "(4.000 per owned+ghost body)" in r.stdout, r.stdout
        outs.append(r.stdout)
    assert "loopback: 2304 halo records per step through ncclSend/ncclRecv(self), 90 steps" in outs[1], outs[1]
    checksum = [o.split("checksum")[1].split()[0] for o in outs]
    assert checksum[0] == checksum[1], checksum
    # ... and the same through nh_partition_step (round 6): the host hands communicator and entry points to the library, which runs all 90 sub-steps in ONE call and
    # enqueues the exchange between them itself -- same world, and the chain of still steps (one launch per step) ran through it
    r = subprocess.run([CU.PARTITION_RCCL, "--library-steps", "2", "2", "24", "90"], capture_output=True, text=True, timeout=600, env=dict(base, NUDGE_RCCL_LOOPBACK="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "2304 owned + 0 ghost bodies" in r.stdout and "(4.000 per owned+ghost body)" in r.stdout and "(inside nh_partition_step)" in r.stdout, r.stdout
    assert r.stdout.split("checksum")[1].split()[0] == checksum[0], (r.stdout, checksum)
    still = [int(o.split("still steps")[1].split(",")[0]) for o in (outs[1], r.stdout)]
    assert still[1] >= still[0] - 2, still


@pytest.mark.gpu
@pytest.mark.parametrize("sphere_fraction", [0.0, 0.5, "compound"])
def test_cpp_partition_host_matches_the_python_host_bit_for_bit(tmp_path, sphere_fraction):
    """examples/partition_rccl.cpp with NUDGE_LOCAL_RANKS=3 (three ranks of the C++ host in one process on one GPU, device-to-device copies as the transport) and
    nudge_amd/partition.py's LocalCluster are two hosts over the SAME C-ABI entry points (nh_partition_*): on one scene cut through its tiles, same epoch, same
    capacities, they must leave the same owned bodies bit for bit -- through refreshes, per-step halos, landing and rest -- which also pins the C++ host's refresh /
    migration / re-balancing logic, none of which a one-rank run executes."""
    import numpy as np
    import partition_util as PU
    import parity_util as P
    from nudge_amd import engine as E, partition as PT, scenes as S
    assert os.path.exists(CU.PARTITION_RCCL), "nudge_amd/partition_rccl missing: make -C nudge_amd/csrc"
    if sphere_fraction == "compound":
        scene = S.compound(n_bodies=300, seed=6)             # two boxes and a sphere per body: the records carry all three, the collider arrays are rebuilt with a body's own adjacent
        scene["body_momentum"]["velocity"][1::4, 0] = 5.0
    else:
        scene = S.grid_tiles(6, side=24, sphere_fraction=sphere_fraction, seed=33, lattice_cols=3)
    x = scene["body_transforms"]["position"][1:, 0]
    cuts = [float(np.quantile(x, 0.30)) + 0.41, float(np.quantile(x, 0.72)) + 0.41]          # uneven on purpose: the re-balancing moves them
    steps = 150
    scene_file, out_file = str(tmp_path / "scene.bin"), str(tmp_path / "owned.bin")
    PU.write_scene_file(scene, scene_file)
    env = dict(os.environ, NUDGE_LOCAL_RANKS="3", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([CU.PARTITION_RCCL, "--scene", scene_file, "--cuts", ",".join(repr(c) for c in cuts), "--out", out_file, "--min-extra", "2048", str(steps)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    got = PU.read_owned_file(out_file)
    parts = []
    for rk in range(3):
        local, info = PT.split_scene(scene, cuts, rk, min_extra=2048)
        w = E.World(local, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP, capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=8 * info["capacity"]["bodies"])
        parts.append(PT.Partition(w, info, epoch=16))
        assert parts[-1].hip is not None
    cl = PT.LocalCluster(parts)
    cl.step(steps)
    ref = cl.gather()
    assert np.array_equal(got["ids"], ref["ids"]) and np.array_equal(got["ids"], np.arange(1, len(scene["body_transforms"])))
    assert P.bits_equal(got["transforms"]["position"], ref["transforms"]["position"]) and P.bits_equal(got["transforms"]["rotation"], ref["transforms"]["rotation"])
    assert P.bits_equal(got["momentum"]["velocity"], ref["momentum"]["velocity"]) and P.bits_equal(got["momentum"]["angular_velocity"], ref["momentum"]["angular_velocity"])
    assert np.array_equal(got["idle"], ref["idle"])
    moves = sum(p.stats["cut_moves"] for p in parts)
    assert moves > 0 and f"cut moves" in r.stdout, (moves, r.stdout)
    if sphere_fraction == "compound":
        assert sum(p.stats["migrated_out"] for p in parts) > 0
    for q in parts:
        q.close(); q.e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["planks", "heap"])
def test_cpp_partition_host_with_single_ownership_matches_the_python_host_bit_for_bit(tmp_path, which):
    """examples/partition_rccl.cpp --single-owner: the C++ host drives the contact-ownership rule of SURVEY 8(e) (nudge.cpp:2074-2087, 2131-2132) and its per-iteration
    exchange entirely through the C ABI -- nh_set_first_ghost_body, nh_partition_mark_ghosts / pack_deltas / unpack_deltas / pack_momentum / unpack_momentum, one
    nh_apply_impulses(.., 1) per iteration and rank parity.  Two ranks in one process on one GPU against nudge_amd/partition.py's LocalCluster over the same entry points
    (which tests/test_gpu_partition.py pins to the reference engine bit for bit): planks lying across the cut, a heap forming across it."""
    import numpy as np
    import partition_util as PU
    import parity_util as P
    from nudge_amd import engine as E, partition as PT, scenes as S
    assert os.path.exists(CU.PARTITION_RCCL), "nudge_amd/partition_rccl missing: make -C nudge_amd/csrc"
    if which == "planks":
        scene, cut, steps = S.bridge(3), 0.0, 90
    else:
        scene = S.pile(n_boxes=120, n_spheres=0, seed=3)
        scene["body_transforms"]["position"][1:, 1] *= 0.08
        scene["body_transforms"]["position"][1:, 1] += 1.0
        cut, steps = 0.3, 80
    scene_file, out_file = str(tmp_path / "scene.bin"), str(tmp_path / "owned.bin")
    PU.write_scene_file(scene, scene_file)
    env = dict(os.environ, NUDGE_LOCAL_RANKS="2", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([CU.PARTITION_RCCL, "--scene", scene_file, "--cuts", repr(cut), "--out", out_file, "--min-extra", "512", "--epoch", "8", "--single-owner", str(steps)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    got = PU.read_owned_file(out_file)
    parts = []
    for rk in range(2):
        local, info = PT.split_scene(scene, [cut], rk, min_extra=512)
        w = E.World(local, flags=E.NH_FLAG_EXACT_ORDER, capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=16 * info["capacity"]["bodies"])
        parts.append(PT.Partition(w, info, epoch=8, per_iteration=True, single_owner=True))
        assert parts[-1].hip is not None
    cl = PT.LocalCluster(parts)
    cl.step(steps)
    ref = cl.gather()
    assert np.array_equal(got["ids"], ref["ids"]) and np.array_equal(got["ids"], np.arange(1, len(scene["body_transforms"])))
    assert P.bits_equal(got["transforms"]["position"], ref["transforms"]["position"]) and P.bits_equal(got["transforms"]["rotation"], ref["transforms"]["rotation"])
    assert P.bits_equal(got["momentum"]["velocity"], ref["momentum"]["velocity"]) and P.bits_equal(got["momentum"]["angular_velocity"], ref["momentum"]["angular_velocity"])
    assert sum(p.stats["ghosts"] for p in parts) > 0
    for q in parts:
        assert q.e.counts()["error"] == 0
        q.close(); q.e.close()
