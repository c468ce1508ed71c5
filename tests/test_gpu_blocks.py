"""The spatially blocked, LDS-resident coloured solver (nudge_amd/csrc/nh_blocks.h) -- pytest -m gpu.

Large general contact sets (config 4's pit) are swept block by block in LDS instead of colour by colour from HBM.  The block order is one more
Gauss-Seidel order (like the colour order it replaces, a different one than the reference's batches, nudge.cpp:4646-4854), so against the
reference the bar is the statistical one of SURVEY 8(d) c4; what CAN be exact is checked exactly:
  * one block holding the whole world walks the colours in the very order of the per-colour launches: bit-identical to them;
  * the result does not depend on how workgroups are scheduled or in which order contacts arrive in their bins: two runs agree bit for bit,
    and so do one apply call of I iterations and I calls of one;
  * NH_BLK_CHECK=1 verifies on the device, every step, that every local index resolves to the body the contact names, that no two blocks of
    a launch touch the same body and that no block list holds two contacts of one colour on one body.
By default every block colours its own contacts (k_blk_prepare_local: no world-wide colouring, no adjacency lists of the general bodies) whenever no contact is
residual; NH_BLK_GLOBAL_COLOURS=1 keeps the world-wide colouring.  Both are checked by the same device-side structure check.
The knobs (read once in nh_create): NH_BLK_MIN (general contacts from which the blocked form is used; 200,000 by default), NH_BLK_TARGET (bodies
per average block: small values make many blocks out of a small world), NH_NO_BLOCKS=1 (per-colour launches), NH_NO_RESIDENT=1 (no one-workgroup
solvers)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from nudge_amd import scenes as S                  # noqa: E402
from nudge_amd import engine as E                  # noqa: E402
import parity_util as P                            # noqa: E402
from oracle import refworld                        # noqa: E402
from test_gpu_atsize import _reference_pit, _gpu_pit_stats     # noqa: E402

pytestmark = pytest.mark.gpu


def _world(monkeypatch, scene, env, flags=E.NH_FLAG_SINGLE_APPLY, **kw):
    keys = ("NH_BLK_MIN", "NH_BLK_TARGET", "NH_BLK_CHECK", "NH_NO_BLOCKS", "NH_NO_RESIDENT", "NH_BLK_GLOBAL_COLOURS", "NH_BLK_ROWS_GLOBAL", "NH_NO_BLK_CHAIN")
    for k in keys:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    w = E.World(scene, flags=flags, **kw)
    for k in keys:
        monkeypatch.delenv(k, raising=False)
    return w


def _same(a, b):
    return P.bits_equal(a["transforms"], b["transforms"]) and P.bits_equal(a["momentum"], b["momentum"]) and np.array_equal(a["idle"], b["idle"])


def test_one_block_is_bit_identical_to_the_per_colour_launches(monkeypatch):
    """A 12^3 pit in ONE block, with the world-wide colouring (NH_BLK_GLOBAL_COLOURS=1; by default a block colours its own contacts): the workgroup walks
    colours 1..L exactly like the L launches of k_level_exec do, with the momentum in LDS instead of HBM.  Same arithmetic, same order => the same bits, step
    after step, cache included."""
    scene = S.ball_pit(12, 12, 12, seed=21)
    nb = 12 ** 3
    a = _world(monkeypatch, scene, dict(NH_NO_RESIDENT=1, NH_BLK_MIN=1, NH_BLK_TARGET=1000000, NH_BLK_CHECK=1, NH_BLK_GLOBAL_COLOURS=1), max_contacts=8 * nb, max_pairs=10 * nb)
    b = _world(monkeypatch, scene, dict(NH_NO_RESIDENT=1, NH_NO_BLOCKS=1), max_contacts=8 * nb, max_pairs=10 * nb)
    for upto in (60, 150, 300):
        a.step(upto - a.steps_done); b.step(upto - b.steps_done)
        assert a.counts()["error"] == 0 and b.counts()["error"] == 0
        assert _same(a.get_bodies(), b.get_bodies()), upto
    ka, kb = a.get_cache(), b.get_cache()
    assert ka["count"] == kb["count"] and np.array_equal(ka["tags"], kb["tags"]) and P.bits_equal(ka["data"]["impulse"], kb["data"]["impulse"])
    assert a.counts()["general_contacts"] > 3000
    a.close(); b.close()


@pytest.mark.parametrize("target", [40, 300])
def test_many_blocks_structure_determinism_and_call_pattern(monkeypatch, target):
    """A 20^3 pit cut into ~200 (or ~30) blocks with ghosts on every side.  The device-side structure check runs every step; two runs agree bit
    for bit (bin arrival order and workgroup scheduling do not show); one apply call of 16 iterations equals 16 calls of one (the momentum
    copy is written back and re-read around every call)."""
    scene = S.ball_pit(20, 20, 20, seed=9)
    nb = 8000
    env = dict(NH_BLK_MIN=1, NH_BLK_TARGET=target, NH_BLK_CHECK=1)
    a = _world(monkeypatch, scene, env, max_contacts=8 * nb, max_pairs=10 * nb)
    b = _world(monkeypatch, scene, env, max_contacts=8 * nb, max_pairs=10 * nb)
    c = _world(monkeypatch, scene, env, flags=0, max_contacts=8 * nb, max_pairs=10 * nb)
    a.step(200); b.step(200)
    for _ in range(200):
        c.collide(); c.gravity(); c.read_cache(); c.setup()
        for _ in range(16):
            c.apply(1)
        c.update(); c.write_cache(); c.advance()
    ca = a.counts()
    assert ca["error"] == 0 and b.counts()["error"] == 0 and c.counts()["error"] == 0
    assert ca["general_contacts"] > 15000
    x, y, z = a.get_bodies(), b.get_bodies(), c.get_bodies()
    assert _same(x, y), "two runs of the blocked solver differ: a data race"
    assert _same(x, z), "16 x apply(1) differs from apply(16)"
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("target", [40, 300])
def test_chained_sweeps_are_the_bits_of_the_eight_launches(monkeypatch, target):
    """The default sweep (option no_blk_chain switches it off): the eight block colours of a pass as ONE launch -- a workgroup draws its (colour, block) task from a ticket counter and waits only for the
    neighbour blocks of earlier colours, which publish their momentum with system-scope stores (nh_blocks.h: k_blk_sweep<.., CHAIN>).  Same blocks, same colour
    order inside and between the blocks => the bits of the eight launches, step after step, with the structure check on; 16 x apply(1) included."""
    scene = S.ball_pit(20, 20, 20, seed=9)
    nb = 8000
    env = dict(NH_BLK_MIN=1, NH_BLK_TARGET=target, NH_BLK_CHECK=1)
    a = _world(monkeypatch, scene, env, max_contacts=8 * nb, max_pairs=10 * nb)
    b = _world(monkeypatch, scene, dict(env, NH_NO_BLK_CHAIN=1), max_contacts=8 * nb, max_pairs=10 * nb)
    c = _world(monkeypatch, scene, env, flags=0, max_contacts=8 * nb, max_pairs=10 * nb)
    for upto in (60, 200, 330):
        a.step(upto - a.steps_done); b.step(upto - b.steps_done)
        assert a.counts()["error"] == 0 and b.counts()["error"] == 0
        assert _same(a.get_bodies(), b.get_bodies()), f"chained sweeps differ from the eight launches at step {upto}"
        ca, cb = a.get_cache(), b.get_cache()
        assert ca["count"] == cb["count"] and P.bits_equal(ca["data"]["impulse"], cb["data"]["impulse"])
    for _ in range(330):
        c.collide(); c.gravity(); c.read_cache(); c.setup()
        for _ in range(16):
            c.apply(1)
        c.update(); c.write_cache(); c.advance()
    assert c.counts()["error"] == 0
    assert _same(a.get_bodies(), c.get_bodies()), "chained: 16 x apply(1) differs from apply(16)"
    assert a.counts()["general_contacts"] > 15000
    a.close(); b.close(); c.close()


def test_blocked_pit_against_the_reference_and_the_colour_launches(monkeypatch):
    """The 20^3 pit, checkpoint by checkpoint: the blocked sweep, the per-colour launches and the reference are three Gauss-Seidel orders on a
    chaotic island -- the invariants of all three agree within the bands of SURVEY 8(d) c4 (ii)."""
    assert refworld.available("exact"), "oracle/_ref did not travel to this box"
    checkpoints = (150, 300, 450)
    ref = _reference_pit(20, 20, 20, 9, checkpoints)
    scene = S.ball_pit(20, 20, 20, seed=9)
    nb = 8000
    blk = _world(monkeypatch, scene, dict(NH_BLK_MIN=1, NH_BLK_TARGET=120, NH_BLK_CHECK=1), max_contacts=8 * nb, max_pairs=10 * nb)
    glo = _world(monkeypatch, scene, dict(NH_BLK_MIN=1, NH_BLK_TARGET=120, NH_BLK_GLOBAL_COLOURS=1), max_contacts=8 * nb, max_pairs=10 * nb)
    col = _world(monkeypatch, scene, dict(NH_NO_BLOCKS=1), max_contacts=8 * nb, max_pairs=10 * nb)
    for s in checkpoints:
        blk.step(s - blk.steps_done); col.step(s - col.steps_done); glo.step(s - glo.steps_done)
        assert blk.counts()["error"] == 0 and glo.counts()["error"] == 0
        q = _gpu_pit_stats(glo, scene)
        assert q["inside"] == 1.0 and abs(q["top"] - ref[s]["top"]) <= 0.005 * ref[s]["top"] + 0.1 and q["max_pen"] <= 1.5 * ref[s]["max_pen"] + 0.01      # (blocks with the world-wide colours)
        g, h, r = _gpu_pit_stats(blk, scene), _gpu_pit_stats(col, scene), ref[s]
        print(f"\n[pit 20^3 step {s}] blocked / colours / reference: contacts {g['contacts']} / {h['contacts']} / {int(round(r['contacts_per_sphere'] * nb))}; "
              f"max penetration {g['max_pen']:.4f} / {h['max_pen']:.4f} / {r['max_pen']:.4f}; KE/sphere {g['ke']:.3e} / {h['ke']:.3e} / {r['ke']:.3e}; "
              f"top {g['top']:.3f} / {h['top']:.3f} / {r['top']:.3f}; interior degree {g['mean_degree']:.3f} / {h['mean_degree']:.3f} / {r['mean_degree']:.3f}")
        assert g["inside"] == 1.0
        assert abs(g["top"] - r["top"]) <= 0.005 * r["top"] + 0.1, (g["top"], r["top"])
        assert g["max_pen"] <= 1.5 * r["max_pen"] + 0.01, (g["max_pen"], r["max_pen"])
        assert r["ke"] / 1.5 - 1e-3 <= g["ke"] <= 1.5 * r["ke"] + 1e-3, (g["ke"], r["ke"])
        assert abs(g["contacts"] - r["contacts_per_sphere"] * nb) <= 0.03 * r["contacts_per_sphere"] * nb + 50
        if s >= 300:
            assert abs(g["mean_degree"] - r["mean_degree"]) <= 0.05 * r["mean_degree"], (g["mean_degree"], r["mean_degree"])
    blk.close(); col.close(); glo.close()


def test_blocked_heap_of_boxes_and_spheres_full_rows_and_residual_contacts(monkeypatch):
    """A heap of 3000 boxes and 1500 spheres of very different sizes in tiny blocks: full 40-float rows beside radial ones (the kernel with both
    paths), contacts between bodies too far apart for any block to own (the residual launches on the block-ordered momentum copy), blocks that
    overflow on the way (smaller blocks, retried).  Structure check on; deterministic; settles like the per-colour launches do."""
    scene = S.pile(3000, 1500, seed=33, iterations=8)
    nb = 4500
    env = dict(NH_BLK_MIN=1, NH_BLK_TARGET=24, NH_BLK_CHECK=1)
    a = _world(monkeypatch, scene, env, max_contacts=16 * nb, max_pairs=20 * nb)
    b = _world(monkeypatch, scene, env, max_contacts=16 * nb, max_pairs=20 * nb)
    c = _world(monkeypatch, scene, dict(NH_NO_BLOCKS=1), max_contacts=16 * nb, max_pairs=20 * nb)
    a.step(600); b.step(600); c.step(600)
    ca, cc = a.counts(), c.counts()
    assert ca["error"] == 0 and cc["error"] == 0
    x, y, z = a.get_bodies(), b.get_bodies(), c.get_bodies()
    assert _same(x, y), "two runs of the blocked solver differ: a data race"
    px, pz = x["transforms"]["position"][1:], z["transforms"]["position"][1:]
    assert np.isfinite(px).all() and np.isfinite(x["momentum"]["velocity"]).all()
    landed_x, landed_z = float((px[:, 1] < 60.0).mean()), float((pz[:, 1] < 60.0).mean())
    kx = float((x["momentum"]["velocity"][1:].astype(np.float64) ** 2).sum(axis=1).mean())
    kz = float((z["momentum"]["velocity"][1:].astype(np.float64) ** 2).sum(axis=1).mean())
    print(f"\n[heap 3000+1500, 600 steps] blocked / colours: contacts {ca['contacts']} / {cc['contacts']}, general {ca['general_contacts']} / {cc['general_contacts']}, "
          f"landed {landed_x:.3f} / {landed_z:.3f}, mean |v|^2 {kx:.3e} / {kz:.3e}, lowest centre {px[:, 1].min():.3f} / {pz[:, 1].min():.3f}")
    assert ca["general_contacts"] > 1000
    assert px[:, 1].min() > -10.0 - 0.5 and pz[:, 1].min() > -10.5          # nothing fell through the ground (top of the slab: y = -10)
    assert abs(landed_x - landed_z) < 0.05
    assert abs(ca["contacts"] - cc["contacts"]) < 0.1 * cc["contacts"] + 100
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("scene_name", ["pit", "boxes"])
def test_rows_built_block_by_block_are_the_bits_of_the_per_contact_kernel(monkeypatch, scene_name):
    """k_blk_rows (one workgroup per block, the block's bodies staged in LDS) against k_rows_general (one thread per contact, gathers from the caller's
    arrays; NH_BLK_ROWS_GLOBAL=1): same functions on the same operands, so the worlds must agree bit for bit -- spheres (radial rows) and boxes (full rows)."""
    if scene_name == "pit":
        scene, nb, cap = S.ball_pit(20, 20, 20, seed=9), 8000, 8
        env = dict(NH_BLK_MIN=1, NH_BLK_TARGET=120)
    else:
        scene, nb, cap = S.pile(2500, 0, seed=12, iterations=8), 2500, 24
        env = dict(NH_BLK_MIN=1, NH_BLK_TARGET=300, NH_NO_RESIDENT=1)
    a = _world(monkeypatch, scene, env, max_contacts=cap * nb, max_pairs=cap * nb)
    b = _world(monkeypatch, scene, dict(env, NH_BLK_ROWS_GLOBAL=1), max_contacts=cap * nb, max_pairs=cap * nb)
    for upto in (120, 300):
        a.step(upto - a.steps_done); b.step(upto - b.steps_done)
        assert a.counts()["error"] == 0 and b.counts()["error"] == 0
        assert _same(a.get_bodies(), b.get_bodies()), (scene_name, upto)
    ka, kb = a.get_cache(), b.get_cache()
    assert ka["count"] == kb["count"] and np.array_equal(ka["tags"], kb["tags"]) and P.bits_equal(ka["data"]["impulse"], kb["data"]["impulse"])
    assert a.counts()["general_contacts"] > 500
    a.close(); b.close()


def test_blocks_colour_their_own_boxes_full_rows_without_residual(monkeypatch):
    """A pile of 2500 equal boxes in blocks several boxes wide: no contact is residual, so every block colours its own contacts (k_blk_prepare_local) -- and
    the rows are the full 40-float ones (boxes), the kernel with both paths.  Structure check on every step; two runs agree bit for bit; the pile lands and
    stays above the ground like the one solved by the per-colour launches."""
    scene = S.pile(2500, 0, seed=12, iterations=8)
    nb = 2500
    env = dict(NH_BLK_MIN=1, NH_BLK_TARGET=300, NH_BLK_CHECK=1, NH_NO_RESIDENT=1)
    a = _world(monkeypatch, scene, env, max_contacts=24 * nb, max_pairs=24 * nb)
    b = _world(monkeypatch, scene, env, max_contacts=24 * nb, max_pairs=24 * nb)
    c = _world(monkeypatch, scene, dict(NH_NO_BLOCKS=1, NH_NO_RESIDENT=1), max_contacts=24 * nb, max_pairs=24 * nb)
    a.step(500); b.step(500); c.step(500)
    ca, cc = a.counts(), c.counts()
    assert ca["error"] == 0 and cc["error"] == 0
    x, y, z = a.get_bodies(), b.get_bodies(), c.get_bodies()
    assert _same(x, y), "two runs of the blocked solver differ: a data race"
    px, pz = x["transforms"]["position"][1:], z["transforms"]["position"][1:]
    assert np.isfinite(px).all() and np.isfinite(x["momentum"]["velocity"]).all()
    kx = float((x["momentum"]["velocity"][1:].astype(np.float64) ** 2).sum(axis=1).mean())
    kz = float((z["momentum"]["velocity"][1:].astype(np.float64) ** 2).sum(axis=1).mean())
    print(f"\n[2500 boxes, 500 steps] blocked (own colours) / per-colour launches: contacts {ca['contacts']} / {cc['contacts']}, general {ca['general_contacts']} / {cc['general_contacts']}, "
          f"mean |v|^2 {kx:.3e} / {kz:.3e}, lowest centre {px[:, 1].min():.3f} / {pz[:, 1].min():.3f}, mean height {px[:, 1].mean():.3f} / {pz[:, 1].mean():.3f}")
    assert ca["general_contacts"] > 1000
    assert px[:, 1].min() > -10.5 and pz[:, 1].min() > -10.5            # nothing fell through the ground (top of the slab: y = -10)
    assert abs(float(px[:, 1].mean()) - float(pz[:, 1].mean())) < 0.05 * abs(float(pz[:, 1].mean())) + 0.5
    assert abs(ca["contacts"] - cc["contacts"]) < 0.1 * cc["contacts"] + 200
    a.close(); b.close(); c.close()


def test_blocked_structure_check_on_a_million_spheres(monkeypatch):
    """The device-side structure check (NH_BLK_CHECK=1) at the scale the blocked solver is built for: a 100^3 pit -- a million spheres, ~1000 blocks of the
    default size, blocks that overflow while the pile collapses and are rebuilt smaller -- every step of the collapse and the first settling: every local
    index resolves to the body its contact names, no two blocks of a launch touch one body, no block list holds two contacts of one colour on one body."""
    scene = S.ball_pit(100, 100, 100, seed=6)
    nb = 100 ** 3
    w = _world(monkeypatch, scene, dict(NH_BLK_CHECK=1), max_contacts=5 * nb, max_pairs=7 * nb)
    blocked_steps = 0
    for upto in (100, 200, 300):
        w.step(upto - w.steps_done)
        c = w.counts()
        assert c["error"] == 0, c
        blocked_steps += int(c["general_contacts"] >= 200000)
    assert blocked_steps >= 2, "the blocked solver was meant to have run (general contacts above its threshold)"
    g = _gpu_pit_stats(w, scene)
    assert g["inside"] == 1.0 and g["max_pen"] <= 0.25, g
    w.close()
