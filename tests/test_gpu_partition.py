"""Partitioned world on the GPU: several partitions of one scene as separate device-resident worlds on cuda:0
(LocalCluster: the same Partition code the multi-process run uses, with tensor hand-over instead of RCCL), against the
single device-resident world."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_util as P                      # noqa: E402
from nudge_amd import engine as E           # noqa: E402
from nudge_amd import partition as PT       # noqa: E402
from nudge_amd import scenes as S           # noqa: E402

pytestmark = pytest.mark.gpu


def _cluster(scene, ranks, epoch=8, flags=0, **kw):
    x = scene["body_transforms"]["position"][1:, 0]
    cuts = [float(np.quantile(x, (r + 1) / ranks)) + 0.41 for r in range(ranks - 1)]
    parts = []
    for r in range(ranks):
        local, info = PT.split_scene(scene, cuts, r, min_extra=2048)
        w = E.World(local, flags=flags, capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=8 * info["capacity"]["bodies"])
        parts.append(PT.Partition(w, info, epoch=epoch, **kw))
    return PT.LocalCluster(parts), parts


def test_partitions_in_the_bench_mode_fall_asleep_like_the_single_world():
    """bench.py --gpus N steps every slab with NH_FLAG_SINGLE_APPLY | NH_FLAG_FUSED_STEP and the per-step halo as an UPDATE of the same bodies
    (nh_halo_update: the host's sleep prediction survives the exchange; only a refresh, every `epoch` steps, invalidates it).  Through the fall,
    the landing, the rest and the sleep of a drop scene cut through its tiles: the same bodies asleep as in the single world, state within the
    partitioned bar, no NH_ERR_STALE_HINT, and the island kernels really were skipped while nobody could be asleep."""
    scene = S.grid_tiles(4, side=30, seed=23, lattice_cols=2)
    flags = E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP
    cl, parts = _cluster(scene, 2, epoch=16, flags=flags)
    single = E.World(scene, flags=flags)
    parts[0].e.enable_timing(True)
    cl.step(120); single.step(120)
    names = parts[0].e.kernel_times(reset=True)
    parts[0].e.enable_timing(False)
    # nobody can be asleep in the first 120 steps: predicted, not computed -- except in the step after a refresh (new bodies in the ghost slots)
    assert names.get("active_flags", (0.0, 0))[1] <= 120 // 16 + 1 and names.get("uf_flatten", (0.0, 0))[1] <= 120 // 16 + 1, sorted(names)
    assert "gravity_damping" not in names and "advance" not in names, sorted(names)       # fused into the solver
    cl.step(480); single.step(480)
    got, ref = cl.gather(), single.get_bodies()
    for q in parts:
        assert q.e.counts()["error"] == 0
    asleep_ref = ref["idle"][1:] == 0xff
    assert asleep_ref.mean() > 0.9, "the scene was meant to have gone to sleep"
    assert float(((got["idle"] == 0xff) != asleep_ref).mean()) < 0.02       # (islands at the edge of falling asleep may differ by a step or two)
    assert P.rel_err(got["transforms"]["position"], ref["transforms"]["position"][1:]).max() <= 1e-4
    assert np.abs(got["momentum"]["velocity"] - ref["momentum"]["velocity"][1:]).max() <= 2e-4
    for q in parts:
        q.e.close()
    single.close()


@pytest.mark.parametrize("sphere_fraction", [0.0, 0.5])
def test_three_partitions_match_single_world(sphere_fraction):
    scene = S.grid_tiles(6, side=40, sphere_fraction=sphere_fraction, seed=21, lattice_cols=3)
    cl, parts = _cluster(scene, 3)
    single = E.World(scene, flags=0)
    # airborne: nothing depends on solver slots yet -> bit-identical
    cl.step(20); single.step(20)
    got, ref = cl.gather(), single.get_bodies()
    assert np.array_equal(got["ids"], np.arange(1, len(scene["body_transforms"])))
    assert P.bits_equal(got["transforms"]["position"], ref["transforms"]["position"][1:])
    assert P.bits_equal(got["momentum"]["velocity"], ref["momentum"]["velocity"][1:])
    assert sum(p.stats["ghosts"] for p in parts) > 0
    # landed and settled: SURVEY 8(d) bar (the slot-order effect explained in tests/test_partition_cpu.py)
    cl.step(280); single.step(280)
    got, ref = cl.gather(), single.get_bodies()
    p, pr = got["transforms"]["position"], ref["transforms"]["position"][1:]
    assert np.isfinite(p).all()
    assert P.rel_err(p, pr).max() <= 1e-4
    assert np.abs(got["momentum"]["velocity"] - ref["momentum"]["velocity"][1:]).max() <= 2e-4
    for q in parts:
        assert q.e.counts()["error"] == 0
        q.e.close()
    single.close()


def test_migration_on_the_gpu():
    scene = S.grid_tiles(1, side=40, seed=5)            # one tile: the median cut passes through it
    scene["body_momentum"]["velocity"][1:, 0] = 12.0      # 3 units (one pitch) in 30 steps, all still airborne
    cl, parts = _cluster(scene, 2, epoch=4, speed_floor=14.0)
    single = E.World(scene, flags=0)
    cl.step(30); single.step(30)
    got, ref = cl.gather(), single.get_bodies()
    assert parts[0].stats["migrated_out"] > 0 and parts[1].stats["migrated_in"] == parts[0].stats["migrated_out"]
    assert np.array_equal(got["ids"], np.arange(1, len(scene["body_transforms"])))
    assert P.bits_equal(got["transforms"]["position"], ref["transforms"]["position"][1:])
    assert P.bits_equal(got["transforms"]["rotation"], ref["transforms"]["rotation"][1:])
    for q in parts:
        q.e.close()
    single.close()


def test_per_iteration_ghost_exchange_on_the_gpu():
    """Partition(per_iteration=True) on the HIP engine: one nh_apply_impulses(…, 1) per sweep with the ghosts' momentum overwritten by their
    owners' in between (torch copies on the library's stream).  A heap forming across the cut: after every step each ghost equals its owner bit
    for bit, while with the per-step exchange it is off by up to 1e-1 in the same window; nh_apply_impulses(…, 1) x I equals one call of I sweeps."""
    scene = S.pile(n_boxes=120, n_spheres=0, seed=3)
    scene["body_transforms"]["position"][1:, 1] *= 0.08
    scene["body_transforms"]["position"][1:, 1] += 1.0

    def deviation(parts):
        worst, n = 0.0, 0
        for r, p in enumerate(parts):
            at = p.n_owned + 1
            for d in (-1, +1):
                m = p.ghost_in[d]
                if m:
                    owner = parts[r + d]
                    idx = owner.ghost_out[-d]
                    for mine, theirs in ((p.bt[at:at + m, :12], owner.bt[idx][:, :12]), (p.bt[at:at + m, 16:], owner.bt[idx][:, 16:]),
                                         (p.bm[at:at + m, :12], owner.bm[idx][:, :12]), (p.bm[at:at + m, 16:28], owner.bm[idx][:, 16:28])):
                        a = np.frombuffer(mine.contiguous().cpu().numpy().tobytes(), np.float32).astype(np.float64)
                        b = np.frombuffer(theirs.contiguous().cpu().numpy().tobytes(), np.float32).astype(np.float64)
                        worst = max(worst, float(np.abs(a - b).max()))
                    n += m
                at += m
        return worst, n

    for per_iteration in (False, True):
        parts = []
        for r in range(2):
            local, info = PT.split_scene(scene, [0.3], r, min_extra=2048)
            w = E.World(local, flags=0, capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=16 * info["capacity"]["bodies"])
            parts.append(PT.Partition(w, info, epoch=8, per_iteration=per_iteration, force_torch=True))      # (this test reads the torch-side ghost lists of both modes)
        cl = PT.LocalCluster(parts)
        cl.step(12)
        worst = 0.0
        for _ in range(60):
            cl.step(1)
            for q in parts:
                q.e.synchronize()
            dev, n = deviation(parts)
            assert n > 20
            worst = max(worst, dev)
        got = cl.gather()
        assert np.isfinite(got["transforms"]["position"]).all()
        if per_iteration:
            assert worst == 0.0, worst
        else:
            assert worst > 1e-4
        for q in parts:
            assert q.e.counts()["error"] == 0
            q.e.close()


def test_bench_under_torchrun_with_one_rank_executes_rccl_point_to_point():
    """`bench.py --gpus N` meets RCCL for the first time under the driver's 8-GPU run unless something executes that path earlier.  A one-GPU box can run
    everything but the second rank: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` with NUDGE_BENCH_LOOPBACK=1 creates the
    `nccl` (= RCCL) process group with its device id, sends the world through the partitioned code path (split_scene, Partition, DistCluster: one slab),
    and every step moves 4096 per-step halo records pack -> batch_isend_irecv(isend to self, irecv from self) on DEVICE tensors -> nh_halo_update; the
    barriers and the MAX / SUM all-reduces of the timing protocol run on the GPU as well.  The line must come out and carry contacts."""
    import json
    import socket
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, NUDGE_BENCH_LOOPBACK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "2", "--tiles", "8", "--side", "45", "--no-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["contacts_per_step"] > 0.9 * 4 * 8 * 45 * 45, out
    # (inside nh_partition_step the sub-step right after a refresh has no per-step exchange: 31 of 32 sub-steps carry the 4096 records)
    assert out["halo"]["backend"] == "nccl" and 0.9 * 4096 <= out["halo"]["loopback_records_per_step"] <= 4096, out["halo"]
    assert "nh_partition_step" in out["halo"]["per_step_transport"], out["halo"]
    # ... and the library-driven transport: an RCCL communicator of its own created through ctypes on the RCCL library torch has loaded, ncclGroupStart / ncclSend /
    # ncclRecv / ncclGroupEnd called by libnudge_hip.so through the addresses it was handed, checked with a pattern sent to this same rank
    assert out["halo"]["per_step_transport"].startswith("library-driven RCCL"), out["halo"]


def _two_clusters(scene, ranks, epoch, flags, **kw):
    """The same scene cut the same way twice: partitions driven by the C ABI (nh_partition_*: kernels) and by the torch implementation of the same operations."""
    x = scene["body_transforms"]["position"][1:, 0]
    cuts = [float(np.quantile(x, (r + 1) / ranks)) + 0.41 for r in range(ranks - 1)]
    out = []
    for force_torch in (False, True):
        parts = []
        for r in range(ranks):
            local, info = PT.split_scene(scene, cuts, r, min_extra=2048)
            w = E.World(local, flags=flags, capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=8 * info["capacity"]["bodies"])
            parts.append(PT.Partition(w, info, epoch=epoch, force_torch=force_torch, **kw))
        assert all((p.hip is not None) == (not force_torch) for p in parts)
        out.append((PT.LocalCluster(parts), parts))
    return out


@pytest.mark.parametrize("sphere_fraction", [0.0, 0.5])
def test_partition_kernels_behind_the_c_abi_match_the_torch_implementation(sphere_fraction):
    """nh_partition_* (include/nudge_hip.h "multi-GPU": migration, ghost listing / installation, collider rebuild, per-step records as HIP kernels with device-side
    counts) against the torch implementation of the same operations in nudge_amd/partition.py, on a drop scene cut through its tiles in the bench's mode: the same
    ghost counts at every refresh, and the owned bodies bit for bit over fall, landing, rest and sleep."""
    scene = S.grid_tiles(6, side=30, sphere_fraction=sphere_fraction, seed=27, lattice_cols=3)
    flags = E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP
    (ca, pa), (cb, pb) = _two_clusters(scene, 3, 16, flags)
    done = 0
    for upto in (1, 17, 60, 150, 340):
        ca.step(upto - done); cb.step(upto - done); done = upto
        ga, gb = ca.gather(), cb.gather()
        assert np.array_equal(ga["ids"], gb["ids"]) and np.array_equal(ga["ids"], np.arange(1, len(scene["body_transforms"])))
        assert P.bits_equal(ga["transforms"]["position"], gb["transforms"]["position"]) and P.bits_equal(ga["transforms"]["rotation"], gb["transforms"]["rotation"]), upto
        assert P.bits_equal(ga["momentum"]["velocity"], gb["momentum"]["velocity"]) and np.array_equal(ga["idle"], gb["idle"]), upto
        for x, y in zip(pa, pb):
            # (who is listed as a ghost may differ for a body exactly at the margin: reach and top speed are rounded differently by torch.norm and by the kernels)
            assert x.n_owned == y.n_owned and abs(x.stats["ghosts"] - y.stats["ghosts"]) <= 4, (upto, x.stats, y.stats)
    # (round 6: a boundary at which nothing would change is a per-step exchange -- both clusters must skip the same ones)
    assert getattr(ca, "quiet_refreshes", 0) == getattr(cb, "quiet_refreshes", 0) >= 3, (getattr(ca, "quiet_refreshes", 0), getattr(cb, "quiet_refreshes", 0))
    assert sum(p.stats["ghosts"] for p in pa) > 0 and pa[0].stats["refreshes"] + getattr(ca, "quiet_refreshes", 0) >= 20
    for q in pa + pb:
        assert q.e.counts()["error"] == 0
        q.close() if hasattr(q, "close") else None
        q.e.close()


def test_migration_and_cut_rebalancing_through_the_c_abi():
    """Bodies flying sideways across the cuts change owner through nh_partition_pack_migrants / unpack_migrants; a lopsided split is re-balanced through
    nh_partition_choose_cut / set_cut (a device-side sort picks the new cut).  Same owners, same cuts, same bits as the torch implementation."""
    scene = S.grid_tiles(2, side=32, seed=29, lattice_cols=2)
    scene["body_momentum"]["velocity"][1:, 0] = 10.0          # 2.5 units (most of a pitch) in 30 steps, all airborne
    x = scene["body_transforms"]["position"][1:, 0]
    cuts = [float(np.quantile(x, 0.2)) + 0.41, float(np.quantile(x, 0.8)) + 0.41]       # 20 % | 60 % | 20 %: the middle rank hands bodies to both sides
    clusters = []
    for force_torch in (False, True):
        parts = []
        for r in range(3):
            local, info = PT.split_scene(scene, cuts, r, min_extra=4096, slack=1.0)
            w = E.World(local, flags=0, capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=8 * info["capacity"]["bodies"])
            parts.append(PT.Partition(w, info, epoch=4, speed_floor=12.0, force_torch=force_torch))
        clusters.append((PT.LocalCluster(parts), parts))
    (ca, pa), (cb, pb) = clusters
    ca.step(30); cb.step(30)
    ga, gb = ca.gather(), cb.gather()
    assert np.array_equal(ga["ids"], np.arange(1, len(scene["body_transforms"]))) and np.array_equal(ga["ids"], gb["ids"])
    assert P.bits_equal(ga["transforms"]["position"], gb["transforms"]["position"]) and P.bits_equal(ga["momentum"]["velocity"], gb["momentum"]["velocity"])
    for xp, yp in zip(pa, pb):
        assert xp.n_owned == yp.n_owned and xp.lo == yp.lo and xp.hi == yp.hi, ((xp.n_owned, xp.lo, xp.hi), (yp.n_owned, yp.lo, yp.hi))
    assert sum(p.stats["migrated_out"] for p in pa) > 0 and sum(p.stats["cut_moves"] for p in pa) > 0
    assert sum(p.stats["migrated_out"] for p in pa) == sum(p.stats["migrated_out"] for p in pb)
    single = E.World(scene, flags=0)
    single.step(30)
    ref = single.get_bodies()
    assert P.bits_equal(ga["transforms"]["position"], ref["transforms"]["position"][1:])          # airborne: the partitioned world IS the single one
    for q in pa + pb:
        q.close()
        q.e.close()
    single.close()


def test_compound_bodies_across_a_cut_on_the_gpu():
    """Compound bodies (two boxes and a sphere each, local transforms: nudge.cpp:3023-3025, 3058-3060) in a world cut in two: ghost and migration records carry all
    three colliders, through the C ABI (nh_partition_*) and through the torch implementation alike -- same owners, same bits -- and while nothing touches, the
    partitioned world IS the single world; landed (contacts across the cut are solved on both ranks) it stays close to it."""
    scene = S.compound(n_bodies=120, seed=6)      # (the CPU test's scene: tests/test_partition_cpu.py)
    scene["body_momentum"]["velocity"][1::3, 0] = 6.0
    x = scene["body_transforms"]["position"][1:, 0]
    cuts = [float(np.quantile(x, 0.5)) + 0.13]
    clusters = []
    for force_torch in (False, True):
        parts = []
        for r in range(2):
            local, info = PT.split_scene(scene, cuts, r, min_extra=1024)
            # (bodies bump into each other in the air: the reference's own solver order, like the CPU test's engine -- the default order colours a world's contacts,
            #  and a slab is a different world than the union)
            w = E.World(local, flags=E.NH_FLAG_EXACT_ORDER, capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=16 * info["capacity"]["bodies"], max_pairs=16 * info["capacity"]["bodies"])
            parts.append(PT.Partition(w, info, epoch=4, speed_floor=8.0, force_torch=force_torch))
        clusters.append((PT.LocalCluster(parts), parts))
    (ca, pa), (cb, pb) = clusters
    single = E.World(scene, flags=E.NH_FLAG_EXACT_ORDER)
    ca.step(20); cb.step(20); single.step(20)
    ga, gb, ref = ca.gather(), cb.gather(), single.get_bodies()
    assert np.array_equal(ga["ids"], np.arange(1, len(scene["body_transforms"]))) and np.array_equal(ga["ids"], gb["ids"])
    assert P.bits_equal(ga["transforms"]["position"], gb["transforms"]["position"]) and P.bits_equal(ga["transforms"]["rotation"], gb["transforms"]["rotation"])
    assert P.bits_equal(ga["transforms"]["position"], ref["transforms"]["position"][1:]) and P.bits_equal(ga["transforms"]["rotation"], ref["transforms"]["rotation"][1:])
    assert pa[0].stats["migrated_out"] > 0 and sum(p.stats["ghosts"] for p in pa) > 0
    for xp, yp in zip(pa, pb):
        assert xp.n_owned == yp.n_owned
    ca.step(130); cb.step(130); single.step(130)
    ga, gb, ref = ca.gather(), cb.gather(), single.get_bodies()
    assert P.bits_equal(ga["transforms"]["position"], gb["transforms"]["position"]) and P.bits_equal(ga["momentum"]["velocity"], gb["momentum"]["velocity"])
    assert np.isfinite(ga["transforms"]["position"]).all()
    assert P.rel_err(ga["transforms"]["position"], ref["transforms"]["position"][1:]).max() <= 5e-2
    for q in pa + pb:
        assert q.e.counts()["error"] == 0
        q.close(); q.e.close()
    single.close()


@pytest.mark.parametrize("force_torch", [False, True])
def test_single_owner_of_cross_cut_contacts_on_the_gpu_is_the_reference_engine_bit_for_bit(force_torch):
    """SURVEY 8(e) "determinism rule" on the HIP engine (nh_set_first_ghost_body: nh_collide makes contacts only for the collider pairs a rank owns) under
    Partition(per_iteration=True, single_owner=True): ghost deltas back to the owners, owners' momentum out to the ghosts, even ranks sweeping before odd ones.
    The same two-rank cluster stepped with the compiled reference as every rank's engine (its contact list filtered by the same rule between collide and
    read_cached_impulses) leaves the same bits -- planks lying across the cut, and a heap forming across it; the contact lists of the two ranks are disjoint and
    together are the single world's."""
    import partition_util as PU
    from oracle import refworld
    assert refworld.available("exact")

    def low_pile():
        scene = S.pile(n_boxes=120, n_spheres=0, seed=3)
        scene["body_transforms"]["position"][1:, 1] *= 0.08
        scene["body_transforms"]["position"][1:, 1] += 1.0
        return scene

    for scene, cut, checkpoints in ((S.bridge(3), 0.0, (4, 30, 90)), (low_pile(), 0.3, (20, 50, 80))):
        gpu, cpu = [], []
        for r in range(2):
            local, info = PT.split_scene(scene, [cut], r, min_extra=512)
            w = E.World(local, flags=E.NH_FLAG_EXACT_ORDER, capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=16 * info["capacity"]["bodies"])
            # (force_torch=False, round 5: mark / deltas / momentum through nh_partition_mark_ghosts, pack_deltas, unpack_deltas, pack_momentum, unpack_momentum and
            # the refresh through nh_partition_*; True: the torch implementation of all of it -- both against the reference-engine cluster)
            gpu.append(PT.Partition(w, info, epoch=8, per_iteration=True, single_owner=True, force_torch=force_torch))
            assert (gpu[-1].hip is None) == force_torch
            cpu.append(PT.Partition(PU.OracleEngine(local, info["capacity"]), info, epoch=8, per_iteration=True, single_owner=True))
        cg, cc = PT.LocalCluster(gpu), PT.LocalCluster(cpu)
        done = 0
        for k in checkpoints:
            cg.step(k - done); cc.step(k - done); done = k
            a, b = cg.gather(), cc.gather()
            assert np.array_equal(a["ids"], b["ids"])
            assert P.bits_equal(a["transforms"]["position"], b["transforms"]["position"]) and P.bits_equal(a["transforms"]["rotation"], b["transforms"]["rotation"]), f"{scene['name']} step {k}"
            assert P.bits_equal(a["momentum"]["velocity"], b["momentum"]["velocity"]) and P.bits_equal(a["momentum"]["angular_velocity"], b["momentum"]["angular_velocity"]), f"{scene['name']} step {k}"
            lists = [q.e.get_contacts() for q in gpu]
            kept_gpu = sum(c["count"] for c in lists)
            kept_cpu = sum(q.e.ref.contacts()["count"] for q in cpu)
            assert kept_gpu == kept_cpu > 0
            if k == 4:
                # disjoint and complete: the two lists together are the single world's, while the worlds still agree (tests/test_partition_cpu.py does the same on the reference)
                single = E.World(scene, flags=E.NH_FLAG_EXACT_ORDER)
                single.step(4)
                c1 = single.get_contacts()
                key = lambda c: np.sort(np.stack([c["tags"], c["features"].astype(np.uint64)], axis=1).view(np.dtype([("t", np.uint64), ("f", np.uint64)])).ravel(), order=("t", "f"))   # noqa: E731
                mine = np.sort(np.concatenate([key(c) for c in lists]), order=("t", "f"))
                assert len(mine) == c1["count"] > 20 and np.array_equal(mine, key(c1))
                single.close()
        for q in gpu:
            assert q.e.counts()["error"] == 0
            q.e.close()


def test_deltas_against_a_mark_of_other_ghosts_are_refused():
    """Advisor (round 5, low).  nh_partition_pack_deltas packs `momentum now - momentum at the last nh_partition_mark_ghosts` per ghost.  A refresh replaces the ghost
    set; a host that forgets to mark again used to be caught only when the new set outgrew the reference buffer -- otherwise the deltas were taken against OTHER
    bodies' momenta and added on the owner rank without an error.  Now every nh_partition_unpack_ghosts (and every migration) starts a generation the mark has to match."""
    scene = S.bridge(3)
    gpu = []
    for r in range(2):
        local, info = PT.split_scene(scene, [0.0], r, min_extra=512)
        w = E.World(local, flags=E.NH_FLAG_EXACT_ORDER, capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=16 * info["capacity"]["bodies"])
        gpu.append(PT.Partition(w, info, epoch=8, per_iteration=True, single_owner=True))
    cg = PT.LocalCluster(gpu)
    cg.step(3)                                   # (refresh at step 0, marks and deltas every iteration since: all in order)
    q = gpu[0]
    assert q.hip is not None and sum(q.ghost_in.values()) > 0
    q.delta_pack()                               # a mark is on file for these ghosts: fine
    # a refresh of the ghost set behind the host's back: the same records installed again -- same count, so the reference buffer is large enough
    outs = [p.refresh_pack_ghosts() for p in gpu]
    routed = cg._route(outs)
    for p, inc in zip(gpu, routed):
        p.refresh_unpack_ghosts(inc)
    with pytest.raises(E.NudgeError, match="nh_partition_pack_deltas"):
        q.delta_pack()
    q.delta_mark()
    q.delta_pack()                               # marked again: accepted
    for p in gpu:
        p.e.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_sub_steps_in_one_library_call_with_ghosts_arriving_through_the_transport(overlap):
    """nh_partition_step (round 6): K sub-steps of a slab in ONE library call, the per-step halo enqueued by the library between them.  The chain of still steps (xform
    ahead, pair ahead) then runs through a partitioned world -- for the OWNED bodies; what a rank computes for a ghost is not what its owner computes, so ghosts' lanes
    leave the next step alone, k_xform_ghosts transforms their colliders when the halo has arrived and k_pair_begin evaluates their pairs.
    One GPU cannot hold two RCCL ranks, so the neighbour is a RECORDING: a two-partition cluster stepped one sub-step at a time (round 5's way) records what partition 1
    sends to partition 0 every step; a second, identical cluster then steps its partition 0 ALONE through nh_partition_step with a transport whose `recv` plays the
    recording back (hipMemcpyAsync on the library's stream) -- bit for bit the owned bodies of the first cluster's partition 0, and the chain did run."""
    import ctypes as C
    BENCH = E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP
    scene = S.grid_tiles(2, side=24, seed=91, lattice_cols=2)
    c1, p1 = _cluster(scene, 2, epoch=64, flags=BENCH)
    c2, p2 = _cluster(scene, 2, epoch=64, flags=BENCH)
    for p in p1 + p2:
        p.quiet_refresh = False          # (this test's timeline counts on the refresh at step 64 -- its full step ends the back-off of the landing; quiet refreshes: the tests above)
    c1.step(100); c2.step(100)          # landed and at rest; refreshes at steps 0 and 64, the next one at 128
    recorded = []
    orig = p1[1].step_pack

    def recording_pack():
        out = orig()
        recorded.append(out[-1].clone())
        return out
    p1[1].step_pack = recording_pack
    K = 24
    c1.step(K)
    assert len(recorded) == K and p1[0].ghost_in[+1] > 0 and recorded[0].shape[0] == p1[0].ghost_in[+1]
    # partition 0 of the second cluster, alone, through the library's own loop
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    played = [0]
    GROUP = C.CFUNCTYPE(C.c_int)
    XFER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
    group = GROUP(lambda: 0)
    send = XFER(lambda buf, count, dtype, peer, comm, stream: 0)

    def recv_fn(buf, count, dtype, peer, comm, stream):
        played[0] += 1
        msg = recorded[played[0] - 1]
        assert count == msg.numel()
        return hip.hipMemcpyAsync(buf, msg.data_ptr(), count, 3, stream)          # hipMemcpyDeviceToDevice, on the library's stream
    recv = XFER(recv_fn)
    q = p2[0]
    # (overlap: option "halo_overlap" -- the solver of a pair step in two launches, the bodies the neighbour waits for on a second stream and the next sub-step's halo behind
    # them, beside the interior's solver: the same recording, one message per sub-step, the same bits)
    if overlap:
        q.e.set_option("halo_overlap", 1)
    L = q.e.L
    L.nh_partition_set_transport.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int]
    E._check(L, L.nh_partition_set_transport(q.hip, C.c_void_p(1), C.cast(group, C.c_void_p), C.cast(group, C.c_void_p), C.cast(send, C.c_void_p), C.cast(recv, C.c_void_p), -1, 1), "set_transport")
    c0 = q.e.counts()
    q.library_steps(K, exchange_first=True)
    cb = q.e.counts()
    assert played[0] == K
    a, b = p1[0].owned_state(), q.owned_state()
    assert np.array_equal(a[0], b[0])
    assert P.bits_equal(a[1], b[1]), "transforms of the owned bodies differ"
    assert P.bits_equal(a[2], b[2]), "momentum of the owned bodies differs"
    assert np.array_equal(a[3], b[3])
    print(f"\n[nh_partition_step] {K} sub-steps in one call with {q.ghost_in[+1]} ghosts: still {cb['still_steps'] - c0['still_steps']}, ahead {cb['ahead_steps'] - c0['ahead_steps']}, pair {cb['pair_steps'] - c0['pair_steps']}, replays {cb['still_replays'] - c0['still_replays']}")
    assert cb["error"] == 0 and cb["ahead_steps"] - c0["ahead_steps"] >= K - 4 and cb["pair_steps"] - c0["pair_steps"] >= K - 6, (c0, cb)
    L.nh_partition_set_transport(q.hip, None, None, None, None, None, -1, -1)
    for p in p1 + p2:
        p.e.close()
