"""Config 5 (BASELINE.json configs[4]: 16 M boxes, spatial decomposition into 8 x-slabs with ghost-body halo exchange) AT SIZE on ONE GPU
(pytest -m gpu): the 1984-tile world of bench.py --gpus 8 cut exactly as bench.py cuts it -- equal-count x-slabs whose cuts pass THROUGH tiles --
stepped by 8 `Partition`s in a `LocalCluster` on cuda:0 (the same Partition code the 8-process run uses; tensor hand-over instead of RCCL), against
  (1) the same 16,070,400-box world stepped as ONE world on the same GPU (the denominator of the north star's ">= 6 x single-GPU throughput"), and
  (2) the compiled reference stepping tiles on their own: EVERY tile a cut passes through (ghosts and owners on both sides of it) plus interior ones,
      to config 2's bar (tests/test_gpu_atsize.py).
No contact crosses a cut in a drop scene (boxes only touch the ground), so while the boxes are airborne the partitioned world is BIT-identical to the
single one; once landed the two differ by what union-versus-tile differs by -- the batch slot of a contact depends on how many contacts precede it in
ITS world's list (nudge.cpp:4206-4339) -- which the reference shows against itself (tests/test_cpu_oracle.py::test_reference_union_vs_tile_self_deviation)."""
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_util as P                            # noqa: E402
from nudge_amd import engine as E                  # noqa: E402
from nudge_amd import partition as PT              # noqa: E402
from nudge_amd import scenes as S                  # noqa: E402
from oracle import refworld                        # noqa: E402

pytestmark = pytest.mark.gpu

BENCH_FLAGS = E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP
RANKS, TILES_PER_RANK, SIDE = 8, 248, 90
POS_TOL, POS_TOL_TILE_MAX, VEL_TOL = 1e-4, 2e-4, 3e-4          # config 2's bars (tests/test_gpu_atsize.py)


def _bench_world():
    """The world and the cuts of `bench.py --gpus 8` (bench.py main(): halo branch)."""
    total = TILES_PER_RANK * RANKS
    cols = max(2, int(round(math.sqrt(total) - 0.5)))
    if cols % RANKS == 0:
        cols += 1
    g = S.grid_tiles(total, side=SIDE, seed=2, lattice_cols=cols)
    gx = g["body_transforms"]["position"][1:, 0]
    cuts = [float(np.quantile(gx, (r + 1) / RANKS)) + 0.41 for r in range(RANKS - 1)]
    return g, cuts


def test_c5_sixteen_million_boxes_in_eight_partitions_vs_single_world_and_reference():
    assert refworld.available("exact"), "oracle/_ref/libnudge_ref_exact.so did not travel to this box"
    g, cuts = _bench_world()
    n = len(g["body_transforms"]) - 1
    assert n == 16070400
    # ---- which tiles does a cut pass through?  (slab t is static box collider t: centre +- half extent along x) ----
    n_tiles = len(g["tile_of_static"])
    ox = g["box_transforms"]["position"][:n_tiles, 0].astype(np.float64)
    hx = g["box_data"]["size"][:n_tiles, 0].astype(np.float64)
    bx = g["body_transforms"]["position"][:, 0]
    cut_tiles = set()
    for c in cuts:
        for t in np.flatnonzero((ox - hx < c) & (ox + hx > c)):
            sel = g["tile_of_body"] == t
            if (bx[sel] < c).any() and (bx[sel] >= c).any():          # bodies of the tile on BOTH sides
                cut_tiles.add(int(t))
    assert len(cut_tiles) >= RANKS - 1, "the cuts were meant to pass through tiles"
    rng = np.random.default_rng(5)
    interior = [int(t) for t in rng.choice(n_tiles, size=24, replace=False) if int(t) not in cut_tiles]
    sample = sorted(cut_tiles) + interior
    assert len(sample) >= 32

    # ---- 8 partitions on one device, bench.py's flags and epoch ----
    parts = []
    for r in range(RANKS):
        local, info = PT.split_scene(g, cuts, r, static_margin=64.0)
        w = E.World(local, flags=BENCH_FLAGS, capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=6 * info["capacity"]["bodies"])
        parts.append(PT.Partition(w, info, epoch=16))
        del local
    cl = PT.LocalCluster(parts)
    single = E.World(g, flags=BENCH_FLAGS, max_contacts=6 * n)
    assert sum(p.n_owned for p in parts) == n

    # airborne (two refreshes: steps 0 and 16): bit-identical to the single world
    cl.step(32); single.step(32)
    got, ref = cl.gather(), single.get_bodies()
    assert np.array_equal(got["ids"], np.arange(1, n + 1))
    assert P.bits_equal(got["transforms"]["position"], ref["transforms"]["position"][1:])
    assert P.bits_equal(got["transforms"]["rotation"], ref["transforms"]["rotation"][1:])
    assert P.bits_equal(got["momentum"]["velocity"], ref["momentum"]["velocity"][1:])
    ghosts = sum(p.stats["ghosts"] for p in parts)
    assert ghosts > 0 and all(p.stats["refreshes"] >= 2 for p in parts)

    # landed, at rest and still awake (boxes land around steps 40-60 and fall asleep 255 quiet steps later): every box on its four contacts
    cl.step(288 - 32); single.step(288 - 32)
    contacts = 0
    for q in parts:
        c = q.e.counts()
        assert c["error"] == 0, c
        contacts += c["contacts"]
    cs = single.counts()
    assert cs["error"] == 0
    assert cs["contacts"] >= 0.95 * 4 * n and contacts >= cs["contacts"]            # (contacts of ghosts are counted on both sides)
    # ... and asleep: 352 steps, 22 refreshes; islands, sleeping pairs and cache culling have run on every partition
    steps = 352
    cl.step(steps - 288); single.step(steps - 288)
    got, ref = cl.gather(), single.get_bodies()
    for q in parts:
        assert q.e.counts()["error"] == 0
    assert single.counts()["error"] == 0
    assert sum(p.n_owned for p in parts) == n
    asleep_one = ref["idle"][1:] == 0xff
    assert asleep_one.mean() > 0.9, "the world was meant to have gone to sleep"
    assert float(((got["idle"] == 0xff) != asleep_one).mean()) < 0.02
    p_part, p_one = got["transforms"]["position"], ref["transforms"]["position"][1:]
    assert np.isfinite(p_part).all() and np.isfinite(p_one).all()
    e_single = P.rel_err(p_part, p_one)
    dv_single = float(np.abs(got["momentum"]["velocity"] - ref["momentum"]["velocity"][1:]).max())
    print(f"\n[c5 at size] 8 partitions vs the single 16,070,400-box world after {steps} steps: position error relative to |p| max {float(e_single.max()):.3e} (bar {POS_TOL:.0e}), "
          f"max |dv| {dv_single:.3e} (bar {VEL_TOL:.0e}); {ghosts} ghost bodies, {sum(p.stats['migrated_out'] for p in parts)} migrations, {sum(p.stats['cut_moves'] for p in parts)} cut moves, "
          f"contacts {contacts} (single world {cs['contacts']})")
    assert float(e_single.max()) <= POS_TOL
    assert dv_single <= VEL_TOL
    for q in parts:
        q.e.close()
    single.close()

    # ---- the tiled reference on every cut tile + interior ones ----
    worlds, maps = [], []
    for t in sample:
        sub, body_map = S.extract_tile(g, t)
        worlds.append(refworld.RefWorld(sub, max_contacts=6 * len(sub["body_transforms"])))
        maps.append(body_map)
    refworld.step_tiles(worlds, steps, threads=os.cpu_count() or 1, ftz=False)
    errs, worst_p, worst_v = [], 0.0, 0.0
    worst_p1 = 0.0
    for t, wr, body_map in zip(sample, worlds, maps):
        rb = wr.bodies()
        ids = body_map[1:] - 1                                     # rows of the gathered (id-sorted, body 0 dropped) arrays
        origin = g["box_transforms"]["position"][t].astype(np.float64)
        errs.append(P.rel_err(p_part[ids], rb["transforms"]["position"][1:], origin=origin))
        worst_p = max(worst_p, float(P.rel_err(p_part[ids], rb["transforms"]["position"][1:]).max()))
        worst_p1 = max(worst_p1, float(P.rel_err(p_one[ids], rb["transforms"]["position"][1:]).max()))
        worst_v = max(worst_v, float(np.abs(got["momentum"]["velocity"][ids] - rb["momentum"]["velocity"][1:]).max()))
        # (a body may fall asleep a step early or late: islands at the edge)
        assert float(((got["idle"][ids] == 0xff) != (rb["idle"][1:] == 0xff)).mean()) < 0.02, f"tile {t}: different bodies asleep"
    errs = np.concatenate(errs)
    q9999, tile_max = float(np.quantile(errs, 0.9999)), float(errs.max())
    print(f"[c5 at size] partitions vs the tiled reference on {len(sample)} tiles ({len(cut_tiles)} cut by a slab boundary): relative to |p| max {worst_p:.3e} "
          f"(single world: {worst_p1:.3e}; bar {POS_TOL:.0e}); relative to the tile origin 99.99 % {q9999:.3e} (bar {POS_TOL:.0e}), max {tile_max:.3e} (bar {POS_TOL_TILE_MAX:.0e}); "
          f"max |dv| {worst_v:.3e} (bar {VEL_TOL:.0e})")
    assert worst_p <= POS_TOL and worst_p1 <= POS_TOL
    assert q9999 <= POS_TOL and tile_max <= POS_TOL_TILE_MAX, (q9999, tile_max)
    assert worst_v <= VEL_TOL
