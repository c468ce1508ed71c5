"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against
  (1) the compiled reference (oracle/_ref, exact-math variant) when it travelled with the repo, and
  (2) the committed golden fixtures (tests/golden/*.npz, generated from that reference) always.
Bar: BIT-EXACT body state / contacts for worlds the reference can run (<= 8192 colliders) in
NH_FLAG_EXACT_ORDER mode and, on conflict-free drop scenes, in the default closed-form order too;
<= 1e-4 relative for a union of tiles against the per-tile reference."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from nudge_amd import scenes as S                  # noqa: E402
from nudge_amd import engine as E                  # noqa: E402
import parity_util as P                            # noqa: E402
import make_golden as G                            # noqa: E402
from oracle import refworld                        # noqa: E402

pytestmark = pytest.mark.gpu
have_ref = refworld.available("exact")



def _needs_ref(fn):
    """The compiled reference travels to the GPU box with the repository (oracle/_ref, built by __graft_entry__.build()).  A box without it must
    FAIL these tests, not shrink the suite silently."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **kw):
        assert refworld.available("exact"), "oracle/_ref/libnudge_ref_exact.so did not travel to this box: the parity tests need the compiled reference"
        return fn(*a, **kw)
    return wrapper


needs_ref = _needs_ref
EXACT = E.NH_FLAG_SYNC_COUNTS | E.NH_FLAG_EXACT_ORDER
FAST = E.NH_FLAG_SYNC_COUNTS


def _fixture(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))


# ---- golden fixtures: always available ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(G.CASES))
def test_steps_match_golden_bit_exact(name):
    scene, steps = G.build(name)
    fx = _fixture(name)
    w = E.World(scene, flags=EXACT)
    w.step(steps)
    b = w.get_bodies()
    assert P.bits_equal(b["transforms"]["position"], fx["transforms"]["position"])
    assert P.bits_equal(b["transforms"]["rotation"], fx["transforms"]["rotation"])
    assert P.bits_equal(b["momentum"]["velocity"], fx["momentum"]["velocity"])
    assert P.bits_equal(b["momentum"]["angular_velocity"], fx["momentum"]["angular_velocity"])
    assert np.array_equal(b["idle"], fx["idle"])
    # contacts / active bodies / sleeping pairs of the next collide
    w.collide()
    c = w.get_contacts()
    tags = fx["contact_tags"]
    hi = tags >> np.uint64(32)
    key = (hi & np.uint64(0xFFFF)) | ((hi >> np.uint64(16)) << np.uint64(32))
    assert np.array_equal(c["tags"], key)
    assert np.array_equal(c["features"], (tags & np.uint64(0xFFFFFFFF)).astype(np.uint32))
    assert P.bits_equal(c["data"], fx["contact_data"])
    assert np.array_equal(c["bodies"], fx["contact_bodies"].astype(np.uint32))
    assert np.array_equal(w.get_active(), fx["active"].astype(np.uint32))
    assert np.array_equal(c["sleeping_pairs"], P.widen_sleeping(fx["sleeping_pairs"]))
    w.close()


@pytest.mark.parametrize("name", ["grid30", "grid30_awake", "mixed20"])
def test_default_order_is_exact_on_drop_scenes(name):
    """No lane conflicts on single-layer drops => the closed-form order IS the reference's order."""
    scene, steps = G.build(name)
    fx = _fixture(name)
    w = E.World(scene, flags=FAST)
    w.step(steps)
    b = w.get_bodies()
    assert P.bits_equal(b["transforms"], fx["transforms"])
    assert P.bits_equal(b["momentum"]["velocity"], fx["momentum"]["velocity"])
    w.close()


def test_run_to_run_determinism():
    scene, _ = G.build("pile_mixed")
    out = []
    for _ in range(2):
        w = E.World(scene, flags=FAST)
        w.step(150)
        out.append(w.get_bodies())
        w.close()
    assert P.bits_equal(out[0]["transforms"], out[1]["transforms"]) and P.bits_equal(out[0]["momentum"], out[1]["momentum"])


# ---- against the compiled reference, stage by stage -----------------------------------------------------------------
@needs_ref
@pytest.mark.parametrize("name,warm", [("pile256", 0), ("pile256", 300), ("pile_mixed", 300), ("grid30", 70), ("mixed20", 80), ("pit6", 120), ("stacks", 60), ("compound", 200)])
def test_collide_matches_reference(name, warm):
    scene, _ = G.build(name)
    ref = refworld.RefWorld(scene)
    ref.step(warm)
    b = ref.bodies()
    w = E.World(scene, flags=FAST)
    w.set_bodies(b["transforms"], b["momentum"], b["idle"])
    ref.collide()
    w.collide()
    od, ob, ok, of, _ = P.oracle_contacts_sorted(ref.contacts())
    c = w.get_contacts()
    assert c["count"] == len(ok)
    assert np.array_equal(c["tags"], ok) and np.array_equal(c["features"], of) and np.array_equal(c["bodies"], ob)
    assert P.bits_equal(c["data"], od)
    assert np.array_equal(w.get_active(), ref.active().astype(np.uint32))
    w.close()


@needs_ref
@pytest.mark.parametrize("name,warm,steps", [("pile256", 300, 40), ("pit6", 100, 40), ("stacks", 20, 60), ("pile_mixed", 380, 40), ("compound", 150, 60)])
def test_exact_order_steps_match_reference_on_piles(name, warm, steps):
    """Chaotic scenes: short horizon, but bit for bit, warm start and cache included."""
    scene, _ = G.build(name)
    ref0 = refworld.RefWorld(scene)
    ref0.step(warm)
    b = ref0.bodies()
    ref = refworld.RefWorld(scene)
    ref.set_bodies(b["transforms"], b["momentum"], b["idle"])
    w = E.World(scene, flags=EXACT)
    w.set_bodies(b["transforms"], b["momentum"], b["idle"])
    saw_general = 0
    for s in range(steps):
        ref.step(1)
        w.step(1)
        saw_general = max(saw_general, w.counts()["general_contacts"])
        rb, gb = ref.bodies(), w.get_bodies()
        assert P.bits_equal(rb["transforms"], gb["transforms"]), f"step {s}"
        assert P.bits_equal(rb["momentum"]["velocity"], gb["momentum"]["velocity"]), f"step {s}"
        assert P.bits_equal(rb["momentum"]["angular_velocity"], gb["momentum"]["angular_velocity"]), f"step {s}"
    rc, gc = ref.cache(), w.get_cache()
    assert rc["count"] == gc["count"]
    assert P.bits_equal(rc["data"]["impulse"], gc["data"]["impulse"])
    assert saw_general > 0, "scene was meant to exercise the level-scheduled path"
    w.close()


@needs_ref
def test_union_of_tiles_vs_tiled_reference():
    """the GPU steps the union world, the reference steps each tile on its own."""
    scene = S.grid_tiles(4, side=24, seed=11)
    steps = 300
    w = E.World(scene, flags=FAST)
    w.step(steps)
    gb = w.get_bodies()
    worst = 0.0
    for t in range(4):
        sub, body_map = S.extract_tile(scene, t)
        ref = refworld.RefWorld(sub)
        ref.step(steps)
        rb = ref.bodies()
        origin = scene["box_transforms"]["position"][t].astype(np.float64)
        err = P.rel_err(gb["transforms"]["position"][body_map[1:]], rb["transforms"]["position"][1:], origin=origin)
        worst = max(worst, float(err.max()))
        dv = np.abs(gb["momentum"]["velocity"][body_map[1:]] - rb["momentum"]["velocity"][1:]).max()
        assert dv <= 1e-4
    assert worst <= 1e-4, worst
    w.close()


# ---- error conventions of the boundary --------------------------------------------------------------------------------
def test_contact_capacity_overflow_is_reported():
    scene, _ = G.build("grid30_awake")
    fx = _fixture("grid30_awake")
    w = E.World(scene, flags=FAST, max_contacts=256, arena_bytes=256 << 20)
    w.set_bodies(fx["transforms"], fx["momentum"], fx["idle"])
    with pytest.raises(E.NudgeError, match="capacity"):
        w.collide()
    w.close()


def test_arena_exhaustion_is_reported():
    scene, _ = G.build("pile256")
    w = E.World(scene, flags=FAST, arena_bytes=4096)
    with pytest.raises(E.NudgeError, match="arena"):
        w.collide()
    w.close()


# (BASELINE.json sizes -- configs 2, 3, 4 at 1 M / 2 M / 4 M bodies: tests/test_gpu_atsize.py)


# ---- config 4 (ball pit): chaotic, so short-horizon exactness + invariants ---------------------------------
@needs_ref
def test_ball_pit_8000_short_horizon_exact():
    """(i) an 8000-sphere pit, exact solver order: bit-identical trajectory over the first steps with contacts."""
    scene = S.ball_pit(20, 20, 20, seed=9)
    w = E.World(scene, flags=E.NH_FLAG_SYNC_COUNTS | E.NH_FLAG_EXACT_ORDER, max_contacts=24 * 8000)
    ref = refworld.RefWorld(scene)
    w.step(60); ref.step(60)                     # free fall and first impacts with the floor
    assert w.counts()["contacts"] > 0
    for _ in range(10):
        w.step(1); ref.step(1)
        a, b = w.get_bodies(), ref.bodies()
        assert P.bits_equal(a["transforms"], b["transforms"])
        assert P.bits_equal(a["momentum"]["velocity"], b["momentum"]["velocity"])
    w.close()


@needs_ref
def test_ball_pit_invariants_default_order():
    """(ii) the default (graph-coloured) sweep is a different Gauss-Seidel order than the reference's, so the settled pit is compared
    through invariants against the reference's own run of the same instance: nothing NaN, nobody fell through the container,
    penetration, contacts per sphere and kinetic energy in the reference's band."""
    scene = S.ball_pit(20, 20, 20, seed=9)
    steps = 400
    w = E.World(scene, flags=0, max_contacts=24 * 8000)
    ref = refworld.RefWorld(scene)
    w.step(steps); ref.step(steps)
    c = w.counts()
    assert c["error"] == 0
    a, b = w.get_bodies(), ref.bodies()
    pa, pb = a["transforms"]["position"][1:], b["transforms"]["position"][1:]
    assert np.isfinite(pa).all() and np.isfinite(a["momentum"]["velocity"]).all()
    assert pa[:, 1].min() >= pb[:, 1].min() - 0.05                 # the floor holds
    assert abs(pa[:, 1].mean() - pb[:, 1].mean()) <= 0.02 * max(1.0, abs(pb[:, 1].mean()) + 1.0)   # same packing height
    m = 1.0 / scene["body_properties"]["mass_inverse"][1:]
    ke = lambda v: float((0.5 * m * (v[1:].astype(np.float64) ** 2).sum(axis=1)).sum())
    ke_a, ke_b = ke(a["momentum"]["velocity"]), ke(b["momentum"]["velocity"])
    assert ke_b / 3.0 - 1e-4 * len(m) <= ke_a <= 3.0 * ke_b + 1e-4 * len(m), (ke_a, ke_b)       # the same residual jitter, within a factor 3
    gc = w.get_contacts()
    ref.collide()
    rc = ref.contacts()
    assert abs(gc["count"] - rc["count"]) <= 0.02 * rc["count"] + 50, (gc["count"], rc["count"])
    assert gc["data"]["penetration"].max() <= 1.5 * rc["data"]["penetration"].max() + 0.005
    # contacts per sphere: same distribution (bins 0..12), each bin within 5 points
    deg = lambda bodies: np.bincount(np.minimum(np.bincount(bodies.reshape(-1).astype(np.int64), minlength=len(m) + 1)[1:], 12), minlength=13) / len(m)
    assert np.abs(deg(gc["bodies"]) - deg(rc["bodies"])).max() <= 0.05
    w.close()


# ---- deferred execution: unusual call patterns must still look like the reference's call-by-call semantics ---------------------
@needs_ref
@pytest.mark.parametrize("name,warm", [("grid30", 60), ("pile256", 330)])
def test_setup_without_apply_and_zero_iterations(name, warm):
    """nh_setup_contact_constraints defers the warm start of one-body bodies into the first nh_apply_impulses.  A caller that
    never applies (update / write / advance straight after setup), or applies zero sweeps, must get what the reference gives:
    momentum after the warm start only, and the cache rewritten from the warm-start impulses."""
    scene, _ = G.build(name)
    for pattern in ("no_apply", "zero_iterations", "synchronize_after_setup"):
        w = E.World(scene, flags=E.NH_FLAG_SYNC_COUNTS | E.NH_FLAG_EXACT_ORDER)
        ref = refworld.RefWorld(scene)
        w.step(warm); ref.step(warm)
        for _ in range(3):
            w.collide(); w.gravity(); w.read_cache(); w.setup()
            ref.collide(); ref.gravity(); ref.read_cache(); ref.setup()
            if pattern == "zero_iterations":
                w.apply(0)
            elif pattern == "synchronize_after_setup":
                # the warm start must be visible to whoever synchronises and reads momentum right after setup
                assert P.bits_equal(w.get_bodies()["momentum"]["velocity"], ref.bodies()["momentum"]["velocity"])
                w.apply(2); ref.apply(2)
            w.update(); w.write_cache(); w.advance()
            ref.update(); ref.write_cache(); ref.advance()
            a, b = w.get_bodies(), ref.bodies()
            assert P.bits_equal(a["transforms"], b["transforms"]), pattern
            assert P.bits_equal(a["momentum"]["velocity"], b["momentum"]["velocity"]), pattern
            ca, cb = w.get_cache(), ref.cache()
            assert ca["count"] == cb["count"]
            assert P.bits_equal(ca["data"]["impulse"], cb["data"]["impulse"]), pattern
        w.close()


# ---- BodyConnections (nudge.h:108-111): user constraints join islands (nudge.cpp:3799-3807) ----------------------------------
@needs_ref
def test_body_connections_join_islands():
    """Connected bodies sleep and wake together: a chain of connections ties resting boxes to one that keeps spinning, so the
    active list, idle counters and sleeping pairs differ from the unconnected world -- and match the reference bit for bit."""
    scene, _ = G.build("grid30")
    n = len(scene["body_transforms"]) - 1
    chain = np.arange(1, 121, dtype=np.uint32)
    scene["connections"] = np.stack([chain[:-1], chain[1:]], axis=1)
    scene["body_momentum"]["angular_velocity"][1] = (0.0, 40.0, 0.0)          # body 1 spins on for a long time
    plain = dict(scene); plain["connections"] = np.zeros((0, 2), np.uint32)
    w = E.World(scene, flags=E.NH_FLAG_SYNC_COUNTS | E.NH_FLAG_EXACT_ORDER)
    w0 = E.World(plain, flags=E.NH_FLAG_SYNC_COUNTS | E.NH_FLAG_EXACT_ORDER)
    ref = refworld.RefWorld(scene)
    differed = False
    for steps in (200, 150, 100):
        w.step(steps); w0.step(steps); ref.step(steps)
        a, b = w.get_bodies(), ref.bodies()
        assert P.bits_equal(a["transforms"], b["transforms"])
        assert P.bits_equal(a["momentum"]["velocity"], b["momentum"]["velocity"])
        assert np.array_equal(a["idle"], b["idle"])
        assert np.array_equal(w.get_active(), ref.active().astype(np.uint32))
        differed |= len(w.get_active()) != len(w0.get_active())
    assert differed, "the connections were meant to keep part of the grid awake"
    w.close(); w0.close()


@needs_ref
@pytest.mark.parametrize("name,warm", [("grid30", 70), ("pile256", 330)])
def test_custom_impulses_between_apply_calls(name, warm):
    """SURVEY 8(f)-1 / example/main.cpp:314-317: user code changes body momentum BETWEEN two apply_impulses calls (that is how custom
    constraints interleave with the contact solver).  nh_apply_impulses(..., 1) per sweep with momentum edited in between must match the
    reference doing the same, on the one-body path (grid30) and on the level-scheduled path (pile256), bit for bit."""
    scene, _ = G.build(name)
    w = E.World(scene, flags=EXACT)
    ref = refworld.RefWorld(scene)
    w.step(warm); ref.step(warm)

    def nudge(m):
        m = m.copy()
        m["angular_velocity"][1:9, 1] += np.float32(0.01)
        m["velocity"][1:9, 0] -= np.float32(0.002)
        return m

    for step in range(4):
        w.collide(); w.gravity(); w.read_cache(); w.setup()
        ref.collide(); ref.gravity(); ref.read_cache(); ref.setup()
        for it in range(scene["params"]["iterations"]):
            w.apply(1); ref.apply(1)
            gm, rm = w.get_bodies()["momentum"], ref.bodies()["momentum"]
            assert P.bits_equal(gm, rm), (step, it)
            w.set_bodies(momentum=nudge(gm)); ref.set_bodies(momentum=nudge(rm))
        w.update(); w.write_cache(); w.advance()
        ref.update(); ref.write_cache(); ref.advance()
        a, b = w.get_bodies(), ref.bodies()
        assert P.bits_equal(a["transforms"], b["transforms"]) and P.bits_equal(a["momentum"], b["momentum"]), step
        ca, cb = w.get_cache(), ref.cache()
        assert ca["count"] == cb["count"] and P.bits_equal(ca["data"]["impulse"], cb["data"]["impulse"])
    w.close()


@needs_ref
def test_hub_body_with_a_thousand_contacts_exact_order():
    """A dynamic tray carrying 306 boxes sits in 1228 contacts.  In the reference's batch scheduler (nudge.cpp:4206-4339) every one of
    them conflicts with every other, no batch ever fills and the list of open batches grows with the contact count: the replay spills
    it from LDS to global memory (k_greedy_replay).  Exact order, bit for bit against the reference."""
    scene = S.tray()
    w = E.World(scene, flags=EXACT)
    ref = refworld.RefWorld(scene)
    w.step(8); ref.step(8)
    seen = 0
    for s in range(14):
        w.step(1); ref.step(1)
        c = w.counts()
        assert c["error"] == 0, c
        seen = max(seen, c["contacts"])
        a, b = w.get_bodies(), ref.bodies()
        assert P.bits_equal(a["transforms"], b["transforms"]), s
        assert P.bits_equal(a["momentum"]["velocity"], b["momentum"]["velocity"]) and P.bits_equal(a["momentum"]["angular_velocity"], b["momentum"]["angular_velocity"]), s
    assert seen > 1100, seen
    w.close()


def test_second_setup_for_one_collide_is_refused():
    """include/nudge_hip.h note 8: the solver's inputs are laid out by nh_collide; a second nh_setup_contact_constraints for the same
    nh_collide returns NH_ERR_STALE_SETUP instead of computing garbage."""
    scene, _ = G.build("grid30_awake")
    w = E.World(scene, flags=FAST)
    w.step(75)
    w.collide(); w.gravity(); w.read_cache(); w.setup()
    with pytest.raises(E.NudgeError, match="setup_contact_constraints"):
        w.setup()
    w.apply(); w.update(); w.write_cache(); w.advance()
    w.step(2)
    assert w.counts()["error"] == 0
    w.close()


def test_pair_capacity_is_independent_of_contact_capacity():
    """nh_set_pair_capacity: a dense scene holds more overlapping AABB pairs than contacts.  With the default (contact capacity / 2 + 1024)
    a small contact capacity overflows the pair buffer (reported, not silent); sized from the colliders it runs."""
    scene = S.ball_pit(12, 12, 12, seed=3)
    n = 12 ** 3
    # the settled pit holds 3884 contacts (reference) and at least as many pairs; default pair capacity here: 4608 / 2 + 1024 = 3328
    w = E.World(scene, flags=FAST, max_contacts=4608, arena_bytes=512 << 20)
    with pytest.raises(E.NudgeError, match="pair capacity"):
        w.step(400)
    w.close()
    w = E.World(scene, flags=FAST, max_contacts=4608, max_pairs=8 * n)
    w.step(400)
    c = w.counts()
    assert c["error"] == 0 and c["pairs"] > 3328 and 3700 <= c["contacts"] <= 4100, c
    w.close()


def test_idle_counters_written_behind_the_librarys_back_are_reported():
    """The host predicts when a body can first be asleep and launches no island kernels before that (nh_collide); idle counters written
    from outside without nh_bodies_changed break the prediction: NH_ERR_STALE_HINT, not a silently wrong active list."""
    scene, _ = G.build("grid30_awake")
    w = E.World(scene, flags=FAST)
    w.step(75)                                   # everybody landed and awake; the largest idle counter is far from 0xff
    assert w.counts()["active_bodies"] == len(scene["body_transforms"]) - 1
    w._keep["bi"][5] = 255                       # put body 5 to sleep without telling anybody
    with pytest.raises(E.NudgeError, match="idle counters"):
        w.step(1)
    w.step(3)                                    # the failed step has looked at the counters: the library is back in sync
    assert w.counts()["error"] == 0
    b = w.get_bodies()
    idle = b["idle"].copy(); idle[7] = 255
    w.set_bodies(idle=idle)                      # the engine wrapper calls nh_bodies_changed
    w.step(2)
    assert w.counts()["error"] == 0
    w.close()


def test_single_apply_flag_gives_identical_results_and_refuses_a_second_apply():
    scene, _ = G.build("grid30_awake")
    a, b = E.World(scene, flags=0), E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY)
    a.step(90); b.step(90)
    x, y = a.get_bodies(), b.get_bodies()
    assert P.bits_equal(x["transforms"], y["transforms"]) and P.bits_equal(x["momentum"], y["momentum"])
    ca, cb = a.get_cache(), b.get_cache()
    assert ca["count"] == cb["count"] > 0 and P.bits_equal(ca["data"]["impulse"], cb["data"]["impulse"])
    b.collide(); b.gravity(); b.read_cache(); b.setup(); b.apply(4)
    with pytest.raises(E.NudgeError, match="invalid"):
        b.apply(4)
    a.close(); b.close()


@pytest.mark.parametrize("flags", [EXACT, FAST])
@pytest.mark.parametrize("name,kw,steps,full", [("pile", dict(n_boxes=256, seed=1), 400, True), ("pile", dict(n_boxes=120, n_spheres=60, seed=5), 400, True),
                                                ("ball_pit", dict(nx=6, ny=6, nz=6, seed=4), 260, False), ("ball_pit", dict(nx=9, ny=9, nz=9, seed=4), 300, False),
                                                ("ball_pit", dict(nx=12, ny=12, nz=12, seed=4), 300, False), ("pile", dict(n_boxes=1024, n_spheres=512, seed=1), 500, True)])
def test_lds_resident_solver_is_bit_identical_to_the_per_level_launches(name, kw, steps, full, flags, monkeypatch):
    """Small general sets are solved by ONE workgroup with the rows in LDS (k_solve_resident), larger ones in small worlds (the 12^3 pit, the sample
    application's 1536-body pile) by one workgroup with every body's momentum in LDS (k_solve_resident_bodies); NH_NO_RESIDENT=1 selects the
    launch-per-level path the large scenes use.  Same level schedule, same arithmetic: every body and every cached impulse must agree bit for bit,
    in both orders (the 9^3 pit only fits the first kernel in default order, where all its rows are radial)."""
    scene = getattr(S, name)(**kw)
    a = E.World(scene, flags=flags)
    monkeypatch.setenv("NH_NO_RESIDENT", "1")
    b = E.World(scene, flags=flags)
    monkeypatch.delenv("NH_NO_RESIDENT")
    seen = 0
    for _ in range(steps // 20):
        a.step(20); b.step(20)
        x, y = a.get_bodies(), b.get_bodies()
        assert P.bits_equal(x["transforms"], y["transforms"]) and P.bits_equal(x["momentum"], y["momentum"])
        seen = max(seen, a.counts()["general_contacts"])
    ca, cb = a.get_cache(), b.get_cache()
    assert ca["count"] == cb["count"] and P.bits_equal(ca["data"]["impulse"], cb["data"]["impulse"]) and P.bits_equal(ca["tags"], cb["tags"])
    assert seen > 20, "the scene never produced a general set"
    a.close(); b.close()


@needs_ref
@pytest.mark.parametrize("name,steps", [("grid30", 460), ("grid30_awake", 90), ("mixed20", 120), ("pile256", 420), ("stacks", 80), ("compound", 220)])
@pytest.mark.parametrize("flags", [0, E.NH_FLAG_EXACT_ORDER])
def test_fused_step_flag_changes_nothing(name, steps, flags):
    """NH_FLAG_FUSED_STEP: gravity / damping and the advance of the one-body bodies happen inside their solver kernel.  Same arithmetic in the
    same order: transforms, momentum, idle counters and the cache must equal the unfused run bit for bit -- while bodies fall, land, rest and
    (grid30: sleeping allowed) fall asleep, and for worlds where most bodies are NOT of the fused class (pile, compound)."""
    scene, _ = G.build(name)
    a, b = E.World(scene, flags=flags), E.World(scene, flags=flags | E.NH_FLAG_FUSED_STEP)
    for _ in range(steps // 10):
        a.step(10); b.step(10)
        x, y = a.get_bodies(), b.get_bodies()
        assert P.bits_equal(x["transforms"], y["transforms"]) and P.bits_equal(x["momentum"], y["momentum"])
        assert np.array_equal(x["idle"], y["idle"])
    ca, cb = a.get_cache(), b.get_cache()
    assert ca["count"] == cb["count"] and P.bits_equal(ca["data"]["impulse"], cb["data"]["impulse"]) and P.bits_equal(ca["tags"], cb["tags"])
    a.close(); b.close()


@pytest.mark.parametrize("flags", [E.NH_FLAG_SINGLE_APPLY, E.NH_FLAG_FUSED_STEP])
@pytest.mark.parametrize("name,steps", [("grid30_awake", 60), ("mixed20", 120), ("pile256", 200)])
def test_counts_or_synchronize_between_setup_and_apply_do_not_break_the_step(name, steps, flags):
    """nh_read_counts / nh_synchronize complete a deferred setup (header note 7).  Under NH_FLAG_SINGLE_APPLY that used to consume the setup without
    keeping the solver states, and the apply that followed was refused; now the completion keeps them and the step gives the bits of a plain world.
    A large collider count also has to survive the steps that re-use the kept pair list."""
    scene, _ = G.build(name)
    a, b = E.World(scene, flags=0), E.World(scene, flags=flags)
    a.step(steps); b.step(steps)
    for variant in range(3):
        for w in (a, b):
            w.collide(); w.gravity(); w.read_cache(); w.setup()
            if w is b and variant == 0: assert w.counts()["error"] == 0
            if w is b and variant == 1: w.synchronize()
            if w is b and variant == 2: w.get_contacts()
            w.apply(None); w.update(); w.write_cache(); w.advance()
        x, y = a.get_bodies(), b.get_bodies()
        assert P.bits_equal(x["transforms"], y["transforms"]) and P.bits_equal(x["momentum"], y["momentum"]) and np.array_equal(x["idle"], y["idle"]), variant
    ca, cb = a.get_cache(), b.get_cache()
    assert ca["count"] == cb["count"] and P.bits_equal(ca["data"]["impulse"], cb["data"]["impulse"])
    b.step(3); a.step(3)
    c = b.counts()
    assert c["error"] == 0 and c["large_colliders"] == a.counts()["large_colliders"]
    if c["sort_reuses"] or c["broadphase_rebuilds"] < b.steps_done:
        assert c["large_colliders"] >= 1, "the ground slab is a large collider on every step, not only on rebuild steps"
    x, y = a.get_bodies(), b.get_bodies()
    assert P.bits_equal(x["transforms"], y["transforms"]) and P.bits_equal(x["momentum"], y["momentum"])
    a.close(); b.close()


def test_fused_step_survives_calls_that_break_the_promised_order():
    """Any entry point other than read / setup / apply between gravity and the solver settles the pending gravity with the ordinary kernel, a
    flush of the setup (update without apply) consumes it without advancing anybody, and nh_advance with another time step is refused."""
    scene, _ = G.build("grid30_awake")
    a, b = E.World(scene, flags=0), E.World(scene, flags=E.NH_FLAG_FUSED_STEP)
    a.step(60); b.step(60)
    for variant in range(4):
        for w in (a, b):
            w.collide(); w.gravity()
            if variant == 1: w.synchronize()                      # settles gravity (the caller may look at body state after a synchronize)
            if variant == 2: w.gravity()                          # a second gravity call: the first is settled, the second pending
            w.read_cache(); w.setup()
            if variant == 3: w.update()                           # setup flushed without apply: lookup + warm start, no advance
            else: w.apply(4); w.update()
            w.write_cache(); w.advance()
        x, y = a.get_bodies(), b.get_bodies()
        assert P.bits_equal(x["transforms"], y["transforms"]) and P.bits_equal(x["momentum"], y["momentum"]) and np.array_equal(x["idle"], y["idle"]), variant
    b.collide(); b.gravity(); b.read_cache(); b.setup(); b.apply(4); b.update(); b.write_cache()
    with pytest.raises(E.NudgeError, match="invalid"):
        b.advance(time_step=1.0 / 60.0)
    a.close(); b.close()


@pytest.mark.parametrize("case", ["pit 12^3 settling", "pile falling asleep and woken", "pit 6^3 (resident solver)"])
def test_colours_seeded_from_the_cache_are_final_without_a_check(case, monkeypatch):
    """Default order: a contact that existed last step starts with last step's colour (it rides in the cache record), and k_colour_seed_final takes
    those colours as FINAL -- two cached colours cannot clash, because the two contacts already shared their body when a proper colouring wrote them.
    NH_COLOUR_CHECK_SEEDS=1 sends the seeds through a settle round (which would un-colour the loser of a clash): the two runs must stay bit-identical
    while a pit forms and settles, and while a pile falls asleep and is woken by a box dropped on it (cached entries of sleeping sets come back)."""
    if case.startswith("pit 12"):
        scene, steps = S.ball_pit(nx=12, ny=12, nz=12, seed=4), 400
    elif case.startswith("pit 6"):
        scene, steps = S.ball_pit(nx=6, ny=6, nz=6, seed=4), 300
    else:
        scene, steps = S.pile(n_boxes=256, n_spheres=0, seed=1, iterations=4), 1600
    a = E.World(scene, flags=FAST)
    monkeypatch.setenv("NH_COLOUR_CHECK_SEEDS", "1")
    b = E.World(scene, flags=FAST)
    monkeypatch.delenv("NH_COLOUR_CHECK_SEEDS")
    most_general, slept = 0, False
    for k in range(steps // 50):
        a.step(50); b.step(50)
        x, y = a.get_bodies(), b.get_bodies()
        assert P.bits_equal(x["transforms"], y["transforms"]) and P.bits_equal(x["momentum"], y["momentum"]) and np.array_equal(x["idle"], y["idle"]), (case, k)
        most_general = max(most_general, a.counts()["general_contacts"])
        if case.startswith("pile") and (x["idle"][1:] == 0xff).any() and not slept:
            slept = True                                  # somebody is asleep: drop a box onto the heap (same edit in both worlds)
            for w in (a, b):
                bodies = w.get_bodies()
                t, m, idle = bodies["transforms"].copy(), bodies["momentum"].copy(), bodies["idle"].copy()
                top = 1 + int(np.argmax(t["position"][1:, 1]))
                t["position"][top] = (t["position"][1:, 0].mean(), t["position"][1:, 1].max() + 6.0, t["position"][1:, 2].mean())
                m["velocity"][top] = (0.0, -12.0, 0.0)
                idle[top] = 0
                w.set_bodies(transforms=t, momentum=m, idle=idle)
    ca, cb = a.get_cache(), b.get_cache()
    assert ca["count"] == cb["count"] and P.bits_equal(ca["data"]["impulse"], cb["data"]["impulse"])
    assert most_general > 50
    if case.startswith("pile"):
        assert slept, "the pile never fell asleep: nothing was woken"
    a.close(); b.close()


@pytest.mark.parametrize("name,steps", [("grid30", 460), ("mixed20", 200), ("pile256", 600), ("pit6", 250), ("compound", 330)])
def test_kept_pair_list_gives_the_pairs_of_a_full_search(name, steps, monkeypatch):
    """The broadphase keeps the pairs of AABBs inflated by 1/32 of a cell and, while every AABB stays inside its inflated box, only re-tests the
    kept pairs with this step's exact boxes.  NH_NO_KEPT_PAIRS=1 regroups and searches every step.  Same pairs either way: bodies, contacts and
    the cache must agree bit for bit while a scene falls, lands, rests and sleeps -- and the default run must actually have re-used its list."""
    scene, _ = G.build(name)
    a = E.World(scene, flags=FAST)
    monkeypatch.setenv("NH_NO_KEPT_PAIRS", "1")
    b = E.World(scene, flags=FAST)
    monkeypatch.delenv("NH_NO_KEPT_PAIRS")
    for k in range(steps // 25):
        a.step(25); b.step(25)
        x, y = a.get_bodies(), b.get_bodies()
        assert P.bits_equal(x["transforms"], y["transforms"]) and P.bits_equal(x["momentum"], y["momentum"]) and np.array_equal(x["idle"], y["idle"]), (name, k)
    a.collide(); b.collide()
    ca, cb = a.get_contacts(), b.get_contacts()
    assert ca["count"] == cb["count"] and np.array_equal(ca["tags"], cb["tags"]) and P.bits_equal(ca["data"], cb["data"])
    assert np.array_equal(ca["sleeping_pairs"], cb["sleeping_pairs"])
    ra, rb = a.counts()["broadphase_rebuilds"], b.counts()["broadphase_rebuilds"]
    assert rb == steps // 25 * 25 + 1, rb
    assert ra < rb, (ra, rb)
    a.close(); b.close()


def test_kept_pair_list_survives_teleports_and_collider_edits():
    """Bodies moved by the caller, a collider resized in place and a collider handed to another body: each leaves its inflated box (or stays
    inside and changes nothing the kept list depends on); the next nh_collide must produce the same contacts as a world that searches every step."""
    import copy
    scene, _ = G.build("grid30_awake")
    worlds = []
    for env in (None, "1"):
        if env: os.environ["NH_NO_KEPT_PAIRS"] = env
        worlds.append(E.World(copy.deepcopy(scene), flags=FAST))
        os.environ.pop("NH_NO_KEPT_PAIRS", None)
    for w in worlds:
        w.step(120)
    for edit in range(3):
        for w in worlds:
            b = w.get_bodies()
            t = b["transforms"].copy()
            if edit == 0:
                t["position"][5] = t["position"][6] + np.float32(0.3)              # onto its neighbour
                w.set_bodies(transforms=t)
            elif edit == 1:
                raw = w.records("xd", 16)                                            # box sizes in place: box 9 grows by half
                sz = np.frombuffer(raw[9].cpu().numpy().tobytes(), np.float32).copy(); sz[:3] *= 1.5
                raw[9] = w.torch.from_numpy(sz.view(np.uint8).copy()).to(raw.device)
            else:
                raw = w.records("xx", 32)                                            # collider 12 now belongs to body 40
                rec = np.frombuffer(raw[12].cpu().numpy().tobytes(), np.uint32).copy(); rec[3] = 40
                raw[12] = w.torch.from_numpy(rec.view(np.uint8).copy()).to(raw.device)
            w.step(3)
        x, y = worlds[0].get_bodies(), worlds[1].get_bodies()
        assert P.bits_equal(x["transforms"], y["transforms"]) and P.bits_equal(x["momentum"], y["momentum"]), edit
    for w in worlds:
        w.close()


@pytest.mark.parametrize("name,warm", [("grid30_awake", 100), ("mixed20", 140)])
def test_a_few_movers_are_reinserted_into_the_kept_pair_list(name, warm):
    """A landed world in which a handful of bodies keep moving (kicked by the caller every few steps: they slide, hop onto their neighbours, fly off).  Each
    of them leaves its inflated box again and again; the kept pair list must survive that -- the leavers get new boxes and new pairs (k_reinsert), the rest of
    the list is untouched -- and give, step for step, the bits of a world that rebuilds the list whenever somebody leaves (NH_NO_INCREMENTAL=1) and of one that
    searches the grid every step (NH_NO_KEPT_PAIRS=1)."""
    import copy
    scene, _ = G.build(name)
    worlds = []
    for env in (None, "NH_NO_INCREMENTAL", "NH_NO_KEPT_PAIRS"):
        if env: os.environ[env] = "1"
        worlds.append(E.World(copy.deepcopy(scene), flags=FAST))
        if env: os.environ.pop(env, None)
    for w in worlds:
        w.step(warm)
    nb = worlds[0].get_bodies()["momentum"].shape[0]
    rng = np.random.default_rng(3)
    movers = rng.choice(np.arange(1, nb), size=7, replace=False)
    base = [w.counts() for w in worlds]
    for rnd in range(24):
        kick = rng.uniform(-3.0, 3.0, size=(len(movers), 3)).astype(np.float32)
        kick[:, 1] = np.abs(kick[:, 1]) * np.float32(0.5 if rnd % 3 else 2.0)
        for w in worlds:
            m = w.get_bodies()["momentum"].copy()
            m["velocity"][movers] = kick
            w.set_bodies(momentum=m)
            w.step(6)
        x = worlds[0].get_bodies()
        for k, w in enumerate(worlds[1:]):
            y = w.get_bodies()
            assert P.bits_equal(x["transforms"], y["transforms"]) and P.bits_equal(x["momentum"], y["momentum"]) and np.array_equal(x["idle"], y["idle"]), (name, rnd, k)
    for w in worlds:
        w.collide()
    ca = worlds[0].get_contacts()
    for w in worlds[1:]:
        cb = w.get_contacts()
        assert ca["count"] == cb["count"] and np.array_equal(ca["tags"], cb["tags"]) and P.bits_equal(ca["data"], cb["data"]) and np.array_equal(ca["bodies"], cb["bodies"])
    c = [w.counts() for w in worlds]
    steps = 24 * 6 + 1
    inserts = c[0]["broadphase_inserts"] - base[0]["broadphase_inserts"]
    rebuilds = [c[k]["broadphase_rebuilds"] - base[k]["broadphase_rebuilds"] for k in range(3)]
    print(f"\n[{name}: 7 movers, {steps} steps] re-insertions {inserts}; rebuilds: incremental {rebuilds[0]}, rebuild-on-leave {rebuilds[1]}, search every step {rebuilds[2]}")
    assert inserts > 50 and c[1]["broadphase_inserts"] == 0 and c[2]["broadphase_inserts"] == 0
    assert rebuilds[0] * 4 < rebuilds[1], rebuilds          # (the movers force a rebuild every step or two without re-insertion)
    assert rebuilds[2] == steps
    for w in worlds:
        w.close()


@pytest.mark.parametrize("name,steps", [("grid30", 460), ("mixed20", 250), ("pile256", 600), ("stacks", 150)])
def test_sort_reuse_changes_nothing(name, steps, monkeypatch):
    """When k_kept_filter writes the pairs in place, a contact record sits where it sat last step; when every record also has last step's key the
    tag sort of the records is skipped and last step's order is used.  NH_NO_SORT_REUSE=1 sorts every step: contacts (in tag order), bodies and the
    cache must be the same bit for bit -- and the default run must actually have skipped sorts on the scenes that come to rest."""
    scene, _ = G.build(name)
    a = E.World(scene, flags=FAST)
    monkeypatch.setenv("NH_NO_SORT_REUSE", "1")
    b = E.World(scene, flags=FAST)
    monkeypatch.delenv("NH_NO_SORT_REUSE")
    for k in range(steps // 25):
        a.step(25); b.step(25)
        x, y = a.get_bodies(), b.get_bodies()
        assert P.bits_equal(x["transforms"], y["transforms"]) and P.bits_equal(x["momentum"], y["momentum"]) and np.array_equal(x["idle"], y["idle"]), (name, k)
        a.collide(); b.collide()
        ca, cb = a.get_contacts(), b.get_contacts()
        assert ca["count"] == cb["count"] and np.array_equal(ca["tags"], cb["tags"]) and np.array_equal(ca["features"], cb["features"]) and P.bits_equal(ca["data"], cb["data"])
        assert np.array_equal(ca["bodies"], cb["bodies"]) and np.array_equal(ca["sleeping_pairs"], cb["sleeping_pairs"])
        for w in (a, b):                                  # (finish the step the collide above began)
            w.gravity(); w.read_cache(); w.setup(); w.apply(); w.update(); w.write_cache(); w.advance()
    ra, rb = a.counts()["sort_reuses"], b.counts()["sort_reuses"]
    assert rb == 0, rb
    if name != "pile256":
        assert ra > steps // 4, (ra, steps)
    a.close(); b.close()


# ---- degenerate inputs ------------------------------------------------------------------------------------------------------------
def _tiny_scene(n_boxes, n_spheres, apart=True):
    scene = S.pile(n_boxes=max(n_boxes, 1), n_spheres=n_spheres, seed=3)
    if n_boxes == 0:
        # keep only the static ground box (collider 0 on body 0) and the spheres
        keep_bodies = np.concatenate([[0], np.arange(2, len(scene["body_transforms"]))]).astype(np.int64)
        for k in ("body_transforms", "body_properties", "body_momentum", "idle_counters"):
            scene[k] = scene[k][keep_bodies]
        scene["box_tags"] = scene["box_tags"][:1]; scene["box_data"] = scene["box_data"][:1]; scene["box_transforms"] = scene["box_transforms"][:1]
        scene["sphere_transforms"]["body"] -= 1
    if apart:
        scene["body_transforms"]["position"][1:, 0] = 50.0 * np.arange(1, len(scene["body_transforms"]))      # nobody touches anybody
        scene["body_transforms"]["position"][1:, 1] = 500.0
    return scene


@needs_ref
@pytest.mark.parametrize("case", ["static_only", "one_box_free_fall", "spheres_only_no_contacts", "two_bodies_landing"])
def test_degenerate_worlds(case):
    """Empty and near-empty worlds go through every kernel with zero-length inputs (no pairs, no contacts, no active islands with
    contacts) and must match the reference like any other."""
    if case == "static_only":
        scene = _tiny_scene(1, 0)
        for k in ("body_transforms", "body_properties", "body_momentum", "idle_counters"):
            scene[k] = scene[k][:1]                                  # body 0 only
        scene["box_tags"] = scene["box_tags"][:1]; scene["box_data"] = scene["box_data"][:1]; scene["box_transforms"] = scene["box_transforms"][:1]
    elif case == "one_box_free_fall":
        scene = _tiny_scene(1, 0)
    elif case == "spheres_only_no_contacts":
        scene = _tiny_scene(0, 3)
    else:
        scene = _tiny_scene(1, 1, apart=False)
        scene["body_transforms"]["position"][1:, 1] = (1.0, 4.0)     # about to land on the ground / on each other
        scene["body_transforms"]["position"][1:, 0] = 0.0
        scene["body_transforms"]["position"][1:, 2] = 0.0
    for flags in (E.NH_FLAG_SYNC_COUNTS | E.NH_FLAG_EXACT_ORDER, 0):
        w = E.World(scene, flags=flags)
        ref = refworld.RefWorld(scene)
        for _ in range(4):
            w.step(40); ref.step(40)
            a, b = w.get_bodies(), ref.bodies()
            assert w.counts()["error"] == 0
            assert P.bits_equal(a["transforms"], b["transforms"]), (case, flags)
            assert P.bits_equal(a["momentum"]["velocity"], b["momentum"]["velocity"]), (case, flags)
            assert np.array_equal(a["idle"], b["idle"])
        w.close()


# ---- short division sequences: exhaustive proof obligations ---------------------------------------------------------
def test_short_rsqrt_sequences_are_exact_wherever_their_guard_passes():
    """nh_math.h: nh_rsqrt_guarded uses rsq/rcp estimates + one fused correction each and falls back to 1.0f / sqrtf(x) when a
    one-instruction predicate on the result fails.  tests/fastmath/exhaustive runs ALL 2^32 inputs and counts the inputs where
    the short result differs from the exact one although the predicate holds: must be zero."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "fastmath", "exhaustive")
    assert os.path.exists(exe), "tests/fastmath/exhaustive missing: make -C nudge_amd/csrc"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "EXHAUSTIVE OK" in r.stdout, r.stdout + r.stderr


# ---- the seeded tag sort's uncommon paths ----------------------------------------------------------------------------
@needs_ref
@pytest.mark.parametrize("tile,target", [("1", None), ("37", None), ("700", None), (None, "2"), ("3", "2")])
def test_seeded_sort_oversized_buckets_and_abrupt_change(tile, target, monkeypatch):
    """nh_util.hip: a bucket that outgrows its workgroup's tile is sorted through global memory.  NH_BUCKET_TILE shrinks the tile so
    that (nearly) every bucket of a forming pile takes that path; the world must still match the reference bit for bit, including
    across an abrupt change of the contact set (bodies teleported apart, then dropped again)."""
    if tile:
        monkeypatch.setenv("NH_BUCKET_TILE", tile)
    if target:
        monkeypatch.setenv("NH_BUCKET_TARGET", target)       # 2 keys per bucket: thousands of buckets, splitters searched in global memory
    scene = S.pile(n_boxes=700, n_spheres=300, seed=11, iterations=4)
    scene["body_transforms"]["position"][1:, 1] *= 0.08        # a dense column: many pairs from the first steps on
    w = E.World(scene, flags=EXACT, max_contacts=60000)
    ref = refworld.RefWorld(scene, max_contacts=60000)
    for phase in range(2):
        w.step(60)
        ref.step(60)
        b, rb = w.get_bodies(), ref.bodies()
        assert P.bits_equal(b["transforms"], rb["transforms"]) and P.bits_equal(b["momentum"]["velocity"], rb["momentum"]["velocity"])
        w.collide(); ref.collide()
        c, rc = w.get_contacts(), ref.contacts()
        assert c["count"] == rc["count"] and c["count"] > (1000 if phase == 0 else 100)
        order = np.argsort(rc["tags"], kind="stable")
        assert P.bits_equal(c["data"], rc["data"][order])
        if phase == 0:
            # scatter the bodies: almost every pair disappears at once, new ones form as they land again
            t = b["transforms"].copy()
            t["position"][1:, 0] += np.arange(len(t) - 1, dtype=np.float32) % 37 * 3.0
            t["position"][1:, 2] += np.arange(len(t) - 1, dtype=np.float32) // 37 * 3.0
            w.set_bodies(transforms=t); ref.set_bodies(transforms=t)
    w.close()


@pytest.mark.parametrize("mult,offset", [(1, 0), (257, 3), (70001, 12345), (1048573, 4000000000 - 1048573 * 500)])
@pytest.mark.parametrize("sort", ["seeded", "radix", "classic"])
def test_wide_sparse_collider_tags_do_not_change_the_world(mult, offset, sort, monkeypatch):
    """Collider tags are 32-bit in this ABI and only their ORDER matters (contacts come out in tag order, the solver order follows):
    an order-preserving re-tagging up to the top of the 32-bit range (9..32 significant bits: 2 to 8 digit passes of the packed key,
    seeded sort, one-kernel and three-kernel radix passes alike) must leave bodies bit-identical and map the contact tags one to one."""
    if sort != "seeded":
        monkeypatch.setenv("NH_SORT_RADIX" if sort == "radix" else "NH_SORT_CLASSIC", "1")
    scene = S.pile(n_boxes=300, n_spheres=150, seed=21, iterations=4)
    scene["body_transforms"]["position"][1:, 1] *= 0.1
    wide = dict(scene)
    f = lambda t: (t.astype(np.uint64) * np.uint64(mult) + np.uint64(offset)).astype(np.uint32)
    wide["box_tags"], wide["sphere_tags"] = f(scene["box_tags"]), f(scene["sphere_tags"])
    assert int(wide["sphere_tags"].max()) < 2 ** 32 and np.all(np.diff(np.concatenate([wide["box_tags"], wide["sphere_tags"]]).astype(np.int64)) > 0)
    a, b = E.World(scene, flags=EXACT, max_contacts=40000), E.World(wide, flags=EXACT, max_contacts=40000)
    a.step(45); b.step(45)
    ba, bb = a.get_bodies(), b.get_bodies()
    assert P.bits_equal(ba["transforms"], bb["transforms"]) and P.bits_equal(ba["momentum"], bb["momentum"])
    a.collide(); b.collide()
    ca, cb = a.get_contacts(), b.get_contacts()
    assert ca["count"] == cb["count"] and ca["count"] > 500
    lo, hi = ca["tags"] & np.uint64(0xFFFFFFFF), ca["tags"] >> np.uint64(32)
    assert np.array_equal((lo * np.uint64(mult) + np.uint64(offset)) | ((hi * np.uint64(mult) + np.uint64(offset)) << np.uint64(32)), cb["tags"])
    assert np.array_equal(ca["features"], cb["features"]) and P.bits_equal(ca["data"], cb["data"])
    a.close(); b.close()


# ---- sleeping: coarse (broadphase-level) and fine (contact-level) islands ------------------------------------------------
def _ref_cache_wide(rc):
    """reference cache tags (feature | (a16 | b16 << 16) << 32) -> this ABI's (a | b << 32, feature)"""
    t = rc["tags"]
    hi = t >> np.uint64(32)
    return (hi & np.uint64(0xFFFF)) | ((hi >> np.uint64(16)) << np.uint64(32)), (t & np.uint64(0xFFFFFFFF)).astype(np.uint32)


@needs_ref
@pytest.mark.parametrize("name,steps,checks", [("settling pile", 1500, (1100, 1300, 1500)), ("mixed tile falling asleep", 560, (300, 420, 560))])
def test_sleeping_pairs_and_cache_match_reference_while_a_world_falls_asleep(name, steps, checks):
    """nudge.cpp:3500-3703 drops the pairs of sleeping sets BEFORE the narrowphase (sets over broadphase pairs: a pair whose boxes
    overlap without touching is a sleeping pair too, and it is reported larger-tag-first), 3788-4006 those of sleeping contact islands
    after it; 4064-4101 keeps the cached impulses whose tag matches a sleeping pair (which, for box-sphere pairs dropped at the coarse
    level, it does not: the formats differ).  Bodies, active list, sleeping pairs and the cache must all follow the reference."""
    if name == "settling pile":
        scene = S.pile(n_boxes=256, n_spheres=0, seed=1, iterations=4)
    else:
        scene = S.grid_tiles(1, side=12, sphere_fraction=0.5, seed=7)
    w = E.World(scene, flags=EXACT)
    ref = refworld.RefWorld(scene)
    done, seen_sleeping = 0, 0
    for upto in checks:
        w.step(upto - done); ref.step(upto - done); done = upto
        b, rb = w.get_bodies(), ref.bodies()
        assert P.bits_equal(b["transforms"], rb["transforms"]) and np.array_equal(b["idle"], rb["idle"])
        kc, rk = w.get_cache(), ref.cache()
        wt, wf = _ref_cache_wide(rk)
        assert kc["count"] == rk["count"] and np.array_equal(kc["tags"], wt) and np.array_equal(kc["features"], wf)
        assert P.bits_equal(kc["data"]["impulse"], rk["data"]["impulse"])
        w.collide(); ref.collide()
        c, rc = w.get_contacts(), ref.contacts()
        assert c["count"] == rc["count"]
        assert np.array_equal(c["sleeping_pairs"], P.widen_sleeping(rc["sleeping_pairs"]))
        assert np.array_equal(w.get_active(), ref.active().astype(np.uint32))
        seen_sleeping = max(seen_sleeping, len(rc["sleeping_pairs"]))
    assert seen_sleeping > 0
    w.close()


# ---- round-2 advisor findings --------------------------------------------------------------------------------------------------
def test_restore_sleepers_then_counts_then_step():
    """World.restore() (nh_bodies_changed) followed by counts() and then step(): nh_read_counts must not bring back the idle bound an
    EARLIER nh_collide measured -- the restored state holds sleepers that bound knows nothing about.  (It used to: the next nh_collide then
    skipped the island kernels and failed with NH_ERR_STALE_HINT.)"""
    scene, _ = G.build("grid30")
    w = E.World(scene, flags=FAST)
    w.step(40)
    awake = w.snapshot()                       # everybody awake, small idle counters
    w.step(440)
    assert int((w.get_bodies()["idle"][1:] == 0xff).sum()) > 0, "the tile was meant to be asleep by now"
    asleep = w.snapshot()
    after = None
    for order in ("counts_then_step", "step"):
        w.restore(awake)
        w.step(3)                              # an nh_collide that measures a SMALL idle maximum
        w.restore(asleep)                      # ... which says nothing about this state
        if order == "counts_then_step":
            w.counts(); w.get_contacts()
        w.step(5)
        assert w.counts()["error"] == 0
        b = w.get_bodies()
        if after is None:
            after = b
        else:
            assert P.bits_equal(after["transforms"], b["transforms"]) and P.bits_equal(after["momentum"], b["momentum"]) and np.array_equal(after["idle"], b["idle"])
    w.close()


@pytest.mark.parametrize("side", [12, 20])
def test_garbage_colour_seeds_in_the_cache_are_caught(side):
    """The colouring of the default order is warm-started from a word of caller-owned cache memory (CachedContactImpulse.unused).  A cache the
    library did not write -- here: every entry claims colour 1 -- must not turn into two contacts of one body solved concurrently: the
    seeds are validated per body, clashes are coloured again, and the step stays deterministic (a race would show as run-to-run differences
    and, with thousands of clashes, as garbage).  12^3: the one-workgroup colouring; 20^3: the grid kernels."""
    scene = S.ball_pit(side, side, side, seed=5)
    nb = side ** 3
    outs = []
    for _ in range(2):
        w = E.World(scene, flags=E.NH_FLAG_SINGLE_APPLY, max_contacts=8 * nb, max_pairs=10 * nb)
        w.step(350)                              # settled
        w.synchronize()
        kd = w._keep["kd"].view(w.torch.int32).view(-1, 4)
        kd[:, 3] = 1                              # "colour 1" everywhere, as an integer bit pattern
        w.torch.cuda.synchronize()
        w.step(40)
        c = w.counts()
        assert c["error"] == 0 and c["general_contacts"] > 1000
        b = w.get_bodies()
        assert np.isfinite(b["transforms"]["position"]).all() and np.isfinite(b["momentum"]["velocity"]).all()
        assert float(np.abs(b["momentum"]["velocity"][1:]).max()) < 5.0        # a settled pit stays settled
        outs.append(b)
        w.close()
    assert P.bits_equal(outs[0]["transforms"], outs[1]["transforms"]) and P.bits_equal(outs[0]["momentum"], outs[1]["momentum"])


@pytest.mark.parametrize("shared_tag", [False, True])
def test_append_contacts_through_the_resident_abi(shared_tag):
    """nh_append_contacts on the device-resident boundary, default solver order, with and without NH_FLAG_SYNC_COUNTS: custom contacts that push the
    first dynamic boxes of a drop tile towards y = 3 (the tile's ground is at y = -10).  The two flag sets agree bit for bit, the pushed boxes end up
    well above everybody else, and a list that would overflow the capacity is refused.  `shared_tag`: all appended contacts carry ONE tag (the reference only
    sorts by tag and accepts that): the per-body bookkeeping must still be counted per body pair -- every held box is pushed, not only the first of the run."""
    scene, _ = G.build("grid30")
    held = np.arange(1, 9)

    def run(flags):
        w = E.World(scene, flags=flags)
        for _ in range(260):
            w.collide()
            pos = w.get_bodies()["transforms"]["position"][held]
            low = pos[:, 1] <= 3.0
            data = np.zeros(int(low.sum()), dtype=S.CONTACT)
            data["position"] = pos[low] - np.array([0.0, 0.25, 0.0], dtype=np.float32)
            data["penetration"] = np.minimum(3.0 - pos[low, 1], 0.2)
            data["normal"] = (0.0, 1.0, 0.0)
            data["friction"] = 0.25
            bodies = np.stack([np.zeros(int(low.sum()), np.uint32), held[low].astype(np.uint32)], axis=1)
            tags = (np.uint64(0x70000) | ((held[low].astype(np.uint64) + np.uint64(0x71000)) << np.uint64(32)))
            feats = np.ones(int(low.sum()), np.uint32)
            if shared_tag:
                tags[:] = np.uint64(0x70000) | (np.uint64(0x71000) << np.uint64(32)); feats = held[low].astype(np.uint32)
            w.append_contacts(data, bodies, tags, feats)
            w.gravity(); w.read_cache(); w.setup(); w.apply(); w.update(); w.write_cache(); w.advance()
        assert w.counts()["error"] == 0
        return w, w.get_bodies()

    wa, a = run(FAST)
    wb, b = run(0)
    assert P.bits_equal(a["transforms"], b["transforms"]) and P.bits_equal(a["momentum"], b["momentum"])
    y = a["transforms"]["position"][:, 1]
    assert np.all(y[9:] < -8.0)                                          # everybody else rests on the ground (top of the slab: y = -10)
    assert np.all(y[held] > y[9:].max() + 1.0), y[held]                  # the custom contacts push their boxes up, 0.2 of penetration at a time
    wa.collide()
    k = wa.counts()["contacts"]
    n = wa.max_contacts - k + 1
    with pytest.raises(E.NudgeError, match="capacity"):
        wa.append_contacts(np.zeros(n, dtype=S.CONTACT), np.zeros((n, 2), np.uint32), np.zeros(n, np.uint64), np.zeros(n, np.uint32))
    wa.close(); wb.close()
