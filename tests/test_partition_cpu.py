"""Partitioned world (x-slabs + ghost halo exchange + migration, nudge_amd/partition.py) on the CPU:
the production partition code with the compiled reference as the stepping engine (tests/partition_util.py),
in-process and as two gloo ranks, against the single reference world on the same scene."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_util as P                      # noqa: E402
from nudge_amd import partition as PT       # noqa: E402
from nudge_amd import scenes as S           # noqa: E402
from oracle import refworld                 # noqa: E402

pytestmark = pytest.mark.skipif(not refworld.available("exact"), reason="compiled reference (oracle/_ref) not built")


def _drop_scene(side=24, sphere_fraction=0.0):
    return S.grid_tiles(1, side=side, sphere_fraction=sphere_fraction, seed=11)


def _cuts_through(scene, ranks):
    x = scene["body_transforms"]["position"][1:, 0]
    return [float(np.quantile(x, (r + 1) / ranks)) + 0.37 for r in range(ranks - 1)]      # through the tile, not along a gap


def _check_against_single_world(scene, got, steps, tol=1e-4, vtol=2e-4):
    """Bar of SURVEY 8(d): positions within 1e-4 relative.  The union is NOT bit-identical to the single world even though
    no contact crosses a cut: the reference's batch slot of a contact -- hence the order of a body's own four contacts
    inside one sweep -- depends on how many contacts precede it in the world (nudge.cpp:4206-4339), and that changes
    when the world is split.  Bodies untouched by that effect stay bit-identical (most of them)."""
    import partition_util as PU
    ref = PU.single_world_reference(scene, steps)
    ids = got["ids"]
    assert np.array_equal(ids, np.arange(1, len(scene["body_transforms"]))), "every dynamic body is owned by exactly one rank"
    p, pr = got["transforms"]["position"], ref["transforms"]["position"][ids]
    v, vr = got["momentum"]["velocity"], ref["momentum"]["velocity"][ids]
    assert np.isfinite(p).all()
    assert P.rel_err(p, pr).max() <= tol
    dv = np.linalg.norm(v.astype(np.float64) - vr.astype(np.float64), axis=1)
    assert (dv / np.maximum(np.linalg.norm(vr.astype(np.float64), axis=1), 1.0)).max() <= vtol
    same = (p.view(np.uint32) == pr.view(np.uint32)).all(axis=1).mean()
    assert same > 0.4, same
    return same


@pytest.mark.parametrize("ranks,sphere_fraction", [(2, 0.0), (3, 0.5)])
def test_partitions_in_one_process_match_single_world(ranks, sphere_fraction):
    import partition_util as PU
    scene = _drop_scene(sphere_fraction=sphere_fraction)
    cuts = _cuts_through(scene, ranks)
    parts = []
    for r in range(ranks):
        local, info = PT.split_scene(scene, cuts, r, min_extra=512)
        parts.append(PT.Partition(PU.OracleEngine(local, info["capacity"]), info, epoch=8))
    cl = PT.LocalCluster(parts)
    steps = 300                                           # boxes land around step 40 and have settled by 300
    cl.step(steps)
    assert sum(p.stats["ghosts"] for p in parts) > 0, "the cuts pass through the tile: there must be ghosts"
    _check_against_single_world(scene, cl.gather(), steps)


def test_bodies_migrate_across_the_cut():
    """A layer flying sideways: bodies cross the cut, change owner (first ghost, then owned), and while nothing has landed
    yet the union is BIT-identical to the single world; after the sliding impact the slot-order effect (see above,
    amplified by friction) is all that separates them."""
    import partition_util as PU
    scene = _drop_scene(side=16)
    scene["body_momentum"]["velocity"][1:, 0] = 8.0       # 8 units/s to +x: 2.6 units in 40 steps, pitch is 3
    cuts = _cuts_through(scene, 2)
    parts = []
    for r in range(2):
        local, info = PT.split_scene(scene, cuts, r, min_extra=512)
        parts.append(PT.Partition(PU.OracleEngine(local, info["capacity"]), info, epoch=4, speed_floor=10.0))
    cl = PT.LocalCluster(parts)
    cl.step(40)
    assert parts[0].stats["migrated_out"] > 0 and parts[1].stats["migrated_in"] == parts[0].stats["migrated_out"]
    got, ref = cl.gather(), PU.single_world_reference(scene, 40)
    assert np.array_equal(got["ids"], np.arange(1, len(scene["body_transforms"])))
    assert P.bits_equal(got["transforms"]["position"], ref["transforms"]["position"][1:])
    assert P.bits_equal(got["transforms"]["rotation"], ref["transforms"]["rotation"][1:])
    assert np.abs(got["momentum"]["velocity"] - ref["momentum"]["velocity"][1:]).max() <= 1e-6
    cl.step(50)
    _check_against_single_world(scene, cl.gather(), 90, tol=2e-3, vtol=5e-2)


def _gloo_worker(rank, world, port, steps, out_path):
    import torch.distributed as dist
    import partition_util as PU
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = _drop_scene()
    cuts = _cuts_through(scene, world)
    local, info = PT.split_scene(scene, cuts, rank, min_extra=512)
    part = PT.Partition(PU.OracleEngine(local, info["capacity"]), info, epoch=8)
    cl = PT.DistCluster(part)
    cl.step(steps)
    got = cl.gather(dst=0)
    if rank == 0:
        np.savez(out_path, ids=got["ids"], transforms=got["transforms"], momentum=got["momentum"], ghosts=np.int64(part.stats["ghosts"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks_match_single_world(tmp_path, world):
    """world = 3: the middle rank talks to two neighbours (the shape every rank but the first and last has on 8 GPUs)."""
    import torch.multiprocessing as mp
    steps = 200
    out = str(tmp_path / "gathered.npz")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_gloo_worker, args=(world, port, steps, out), nprocs=world, join=True)
    z = np.load(out)
    assert int(z["ghosts"]) > 0
    _check_against_single_world(_drop_scene(), dict(ids=z["ids"], transforms=z["transforms"], momentum=z["momentum"]), steps)


def test_static_colliders_are_filtered_to_the_slab_and_its_neighbours():
    """bench.py --gpus 8 cuts ONE 1984-tile world into 8 slabs; a rank keeps only the ground slabs that reach its own slab or a
    neighbour's (`static_margin`), not all 1984 of them.  The partitioned world must still match the single world."""
    import partition_util as PU
    scene = S.grid_tiles(8, side=8, seed=17, lattice_cols=8)          # eight tiles in a row along x
    ranks = 4
    cuts = _cuts_through(scene, ranks)
    parts, kept = [], []
    for r in range(ranks):
        local, info = PT.split_scene(scene, cuts, r, min_extra=256, static_margin=4.0)
        kept.append(info["n_static_box"])
        parts.append(PT.Partition(PU.OracleEngine(local, info["capacity"]), info, epoch=8))
    assert kept[0] < 8 and kept[-1] < 8 and max(kept) <= 7 and min(kept) >= 3, kept      # own slab (2 tiles) + what the neighbours stand on
    cl = PT.LocalCluster(parts)
    cl.step(120)
    _check_against_single_world(scene, cl.gather(), 120)
