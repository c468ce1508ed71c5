"""Still steps (include/nudge_hip.h note 9; nudge_amd/csrc/nh_internal.h: nh_StillStep) on the GPU (pytest -m gpu).

A step that follows a step which left the world at rest on static geometry is launched speculatively with three kernels that check what they rely on; when a
check fails the step is run again in full.  Whatever happens, a library that speculates must leave the same bits as one that never does (NH_NO_STILL=1), and
as the compiled reference: these tests drive worlds through both, through the transitions in and out of the still regime and through every way of leaving the
sample's call order in the middle of a speculative step."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_util as P                      # noqa: E402
from nudge_amd import engine as E           # noqa: E402
from nudge_amd import scenes as S           # noqa: E402
from oracle import refworld                 # noqa: E402

pytestmark = pytest.mark.gpu

BENCH_FLAGS = E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP


def _world(scene, still, flags=BENCH_FLAGS, env=(), **kw):
    names = list(env) + ([] if still else ["NH_NO_STILL"])
    for k in names:
        os.environ[k] = "1"
    try:
        return E.World(scene, flags=flags, **kw)
    finally:
        for k in names:
            os.environ.pop(k, None)


def _same_world(a, b, what=""):
    ba, bb = a.get_bodies(), b.get_bodies()
    assert P.bits_equal(ba["transforms"], bb["transforms"]), f"{what}: transforms differ"
    assert P.bits_equal(ba["momentum"], bb["momentum"]), f"{what}: momentum differs"
    assert np.array_equal(ba["idle"], bb["idle"]), f"{what}: idle counters differ"
    ca, cb = a.get_cache(), b.get_cache()
    assert ca["count"] == cb["count"], f"{what}: cache count {ca['count']} vs {cb['count']}"
    assert np.array_equal(ca["tags"], cb["tags"]) and np.array_equal(ca["features"], cb["features"]), f"{what}: cache tags differ"
    assert P.bits_equal(ca["data"]["impulse"], cb["data"]["impulse"]), f"{what}: cached impulses differ"
    ka, kb = a.get_contacts(), b.get_contacts()
    assert ka["count"] == kb["count"], f"{what}: contact count"
    assert np.array_equal(ka["tags"], kb["tags"]) and np.array_equal(ka["features"], kb["features"]) and np.array_equal(ka["bodies"], kb["bodies"]), f"{what}: contact identities differ"
    assert P.bits_equal(ka["data"], kb["data"]), f"{what}: contact data differ"


@pytest.mark.parametrize("sphere_fraction", [0.0, 0.5])
def test_still_steps_leave_the_bits_of_full_steps_through_fall_rest_and_sleep(sphere_fraction):
    """A drop scene from free fall to sleep: the speculating library against the one that never speculates, checkpoint by checkpoint, bit for bit (bodies, idle
    counters, contacts, cache), and both against the compiled reference.  At rest nearly every step must have gone through as a still step."""
    scene = S.grid_tiles(2, side=40, sphere_fraction=sphere_fraction, seed=31, lattice_cols=2)
    a, b = _world(scene, True), _world(scene, False)
    done = 0
    for cp in (30, 70, 100, 200, 290, 330, 420):
        a.step(cp - done); b.step(cp - done); done = cp
        _same_world(a, b, f"step {cp}")
    ca, cb = a.counts(), b.counts()
    assert ca["error"] == 0 and cb["error"] == 0
    assert cb["still_steps"] == 0 and cb["still_replays"] == 0
    # landed by ~70, asleep from ~300: two hundred steps at rest, all but the few after a checkpoint's counts() still
    print(f"\n[still steps, spheres {sphere_fraction}] {ca['still_steps']} still steps, {ca['still_replays']} replays of 420")
    assert ca["still_steps"] >= 150, ca
    if refworld.available("exact") and S.narrow_ok(scene):
        r = refworld.RefWorld(scene, max_contacts=8 * len(scene["body_transforms"]))
        r.step(420)
        rb, gb = r.bodies(), a.get_bodies()
        assert P.bits_equal(gb["transforms"], rb["transforms"]) and P.bits_equal(gb["momentum"]["velocity"], rb["momentum"]["velocity"])
        assert np.array_equal(gb["idle"], rb["idle"])
    a.close(); b.close()


def test_a_kick_in_the_middle_of_rest_fails_the_checks_and_is_replayed():
    """Bodies at rest, every step a still step; then the caller rewrites a few velocities (no nh_bodies_changed: momentum is the caller's to write).  The bodies leave
    their inflated boxes / change their contacts a few steps later: a still step's checks fail on the device, it is replayed in full, the world goes on exactly
    like the one that never speculates -- and returns to still steps once it is at rest again."""
    scene = S.grid_tiles(1, side=32, seed=37)
    a, b = _world(scene, True), _world(scene, False)
    a.step(120); b.step(120)
    c0 = a.counts()
    assert c0["still_steps"] > 20, c0
    for w in (a, b):
        m = w.get_bodies()["momentum"]
        m["velocity"][5::97, 1] = 4.0                    # a hop
        m["velocity"][11::131, 0] = 1.5                  # a shove along the ground
        w.set_bodies(momentum=m)
    a.step(100); b.step(100)
    _same_world(a, b, "after the kick")
    a.step(150); b.step(150)                               # (the kicked boxes slide and rock for a while)
    _same_world(a, b, "at rest again")
    c1 = a.counts()
    assert c1["still_replays"] > c0["still_replays"], (c0, c1)
    assert c1["still_steps"] > c0["still_steps"] + 10, (c0, c1)        # ... and back to still steps afterwards
    a.close(); b.close()


def test_leaving_the_call_order_inside_a_speculative_step():
    """Every entry point outside collide / gravity / read / setup / apply turns an unconfirmed still step into a full one before it looks at anything: counts and
    contacts read between nh_collide and the solver are a full step's, and the step ends in the same bits."""
    scene = S.grid_tiles(1, side=24, seed=41)
    a, b = _world(scene, True), _world(scene, False)
    a.step(100); b.step(100)
    assert a.counts()["still_steps"] > 10
    a.step(2); b.step(2)                                   # (the step after counts() is a full one; the next is still again)

    def step_with(w, hook_at):
        w.collide()
        if hook_at == 0: seen = (w.counts(), w.get_contacts())
        w.gravity()
        if hook_at == 1: seen = (w.counts(), w.get_contacts())
        w.read_cache()
        if hook_at == 2: seen = (w.counts(), w.get_contacts())
        w.setup()
        if hook_at == 3: seen = (w.counts(), w.get_contacts())
        w.apply(); w.update(); w.write_cache(); w.advance(); w.step_done()
        return seen

    for hook_at in (0, 1, 2, 3, 0):
        sa, sb = step_with(a, hook_at), step_with(b, hook_at)
        assert sa[0]["contacts"] == sb[0]["contacts"] and sa[0]["pairs"] == sb[0]["pairs"] and sa[0]["active_bodies"] == sb[0]["active_bodies"]
        assert np.array_equal(sa[1]["tags"], sb[1]["tags"]) and P.bits_equal(sa[1]["data"], sb[1]["data"])
        _same_world(a, b, f"hook {hook_at}")
        a.step(3); b.step(3)                               # back into the still regime before the next hook
        _same_world(a, b, f"after hook {hook_at}")
    ca = a.counts()
    assert ca["still_replays"] >= 4, ca
    a.close(); b.close()


def test_snapshot_restore_and_set_cache_count_leave_the_still_regime():
    """bench.py restores a device snapshot (bodies + cache) between its timed blocks: the step after must look at everything again."""
    scene = S.grid_tiles(1, side=24, seed=43)
    a, b = _world(scene, True), _world(scene, False)
    a.step(90); b.step(90)
    sa, sb = a.snapshot(), b.snapshot()
    a.step(25); b.step(25)
    _same_world(a, b, "before restore")
    a.restore(sa); b.restore(sb)
    a.step(25); b.step(25)
    _same_world(a, b, "after restore")
    assert a.counts()["still_steps"] > 30
    a.close(); b.close()


def test_still_steps_in_a_world_with_a_pile_never_start():
    """Bodies that touch each other are not the fused solver's: no step of such a world may be launched as a still step."""
    scene = S.pile(n_boxes=200, n_spheres=50, seed=3)
    a, b = _world(scene, True), _world(scene, False)
    a.step(200); b.step(200)
    _same_world(a, b, "pile")
    c = a.counts()
    assert c["still_steps"] == 0 and c["still_replays"] == 0 and c["general_contacts"] > 0, c
    a.close(); b.close()


def test_nh_step_with_late_verdicts_matches_the_eight_calls():
    """nh_step makes the eight calls itself and looks at a still step's verdict one step LATE (no host / GPU wait inside the loop); a failed still step and the one
    launched behind it did nothing and are run again.  Against a host that makes the eight calls itself (verdict inside every step) and one that never speculates: the
    same bits through rest, a kick that fails checks in the middle of a long nh_step call, and the return to rest; step counts add up."""
    scene = S.grid_tiles(1, side=32, seed=47)
    a, b, c = _world(scene, True), _world(scene, True), _world(scene, False)
    b.eight_calls = True
    for w in (a, b, c):
        w.step(110)
    _same_world(a, b, "rest: nh_step vs eight calls"); _same_world(a, c, "rest: nh_step vs no speculation")
    for w in (a, b, c):
        m = w.get_bodies()["momentum"]
        m["velocity"][7::89, 1] = 3.0
        m["velocity"][3::101, 2] = -1.2
        w.set_bodies(momentum=m)
    c0 = a.counts()
    a.step(260); b.step(260); c.step(260)                      # ONE nh_step call of 260 steps on `a`: hops, landings, slides, rest again
    _same_world(a, b, "kick: nh_step vs eight calls"); _same_world(a, c, "kick: nh_step vs no speculation")
    ca, cb = a.counts(), b.counts()
    assert ca["still_replays"] > c0["still_replays"] and ca["still_steps"] > c0["still_steps"] + 50, (c0, ca)
    assert ca["error"] == 0 and cb["error"] == 0
    a.close(); b.close(); c.close()


# ---- LOCAL speculation (nudge_amd/csrc/nh_internal.h: nh_StillStep, "LOCAL speculation"): movers in a world at rest ---------------------------------------------------
def _throw(worlds, idx, vy, lateral=0.0, seed=0):
    rng = np.random.default_rng(seed)
    kick = rng.uniform(-lateral, lateral, size=(len(idx), 3)).astype(np.float32) if lateral else np.zeros((len(idx), 3), np.float32)
    kick[:, 1] = vy
    for w in worlds:
        m = w.get_bodies()["momentum"].copy()
        m["velocity"][idx] = kick
        w.set_bodies(momentum=m)


@pytest.mark.parametrize("sphere_fraction", [0.0, 0.5])
def test_bodies_in_the_air_do_not_cost_the_world_its_still_steps(sphere_fraction):
    """A landed world in which a few bodies are thrown high up: every one of them leaves its inflated box on every step.  Round 4 failed each such still step as a whole
    (and then stopped offering them); now the step re-boxes and re-inserts its leavers itself (movers form) and the pairs they leave behind are records without
    contacts.  The bits must be those of a library that never speculates AND of one that speculates the old way (NH_NO_LOCAL_STILL=1), checkpoint by checkpoint --
    bodies, idle counters, contacts, cache -- and nearly every step must have gone through as a still step."""
    scene = S.grid_tiles(2, side=32, sphere_fraction=sphere_fraction, seed=61, lattice_cols=2)
    nb = len(scene["body_transforms"])
    a, b, c = _world(scene, True), _world(scene, False), _world(scene, True, env=["NH_NO_LOCAL_STILL"])
    for w in (a, b, c):
        w.step(120)
    _same_world(a, b, "rest")
    idx = np.arange(5, nb, 137)
    _throw((a, b, c), idx, 30.0, lateral=1.0, seed=7)           # 6 s in the air: far longer than the test
    c0, k0 = a.counts(), c.counts()
    done = 0
    for cp in (1, 2, 3, 10, 40, 41, 120, 200):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_world(a, b, f"flying, step {cp}: local speculation vs none")
        _same_world(c, b, f"flying, step {cp}: global speculation vs none")
    ca, kc = a.counts(), c.counts()
    print(f"\n[movers, spheres {sphere_fraction}] {len(idx)} bodies in the air for {done} steps: still steps {ca['still_steps'] - c0['still_steps']} (replays {ca['still_replays'] - c0['still_replays']}, "
          f"re-insertions {ca['broadphase_inserts'] - c0['broadphase_inserts']}); without local speculation {kc['still_steps'] - k0['still_steps']} (replays {kc['still_replays'] - k0['still_replays']})")
    assert ca["error"] == 0
    assert ca["broadphase_inserts"] > c0["broadphase_inserts"] + done // 2, (c0, ca)          # (they really did leave their boxes)
    assert ca["still_steps"] - c0["still_steps"] >= done - 30, (c0, ca)                      # (a few full steps: the first leave, the checkpoints' views)
    assert kc["still_steps"] - k0["still_steps"] <= done // 2, (k0, kc)                      # (what it used to be)
    a.close(); b.close(); c.close()


def test_a_body_that_comes_back_to_the_ground_it_left_keeps_its_record():
    """A body thrown straight up leaves its inflated box (re-boxed: its kept pairs die) and comes down on the same ground a second later.  Its pair with that ground is
    re-stamped where it stands instead of being appended anew (k_reinsert, still mode: nh_old_pair_of), so the record -- tag-order place, slots, body_rec -- lives on, empty
    while the body is in the air: take-off, flight AND landing are still steps.  Bit-identical to a library that never speculates at every checkpoint."""
    scene = S.grid_tiles(1, side=32, seed=89)
    nb = len(scene["body_transforms"])
    a, b = _world(scene, True), _world(scene, False)
    a.step(120); b.step(120)
    idx = np.arange(9, nb, 101)
    _throw((a, b), idx, 5.0)                                  # 1.0 s = 122 steps in the air, straight up and down
    c0 = a.counts()
    done = 0
    for cp in (1, 20, 60, 110, 125, 128, 131, 135, 145, 170, 230):
        a.step(cp - done); b.step(cp - done); done = cp
        _same_world(a, b, f"up and down, step {cp}")
    ca = a.counts()
    print(f"\n[up and down] {len(idx)} bodies: still steps {ca['still_steps'] - c0['still_steps']} of {done}, replays {ca['still_replays'] - c0['still_replays']}, "
          f"re-insertions {ca['broadphase_inserts'] - c0['broadphase_inserts']}")
    # (the bodies that flew are back on the ground, four contacts each; the rest of the world has meanwhile fallen asleep -- still steps in sleepers form)
    assert ca["error"] == 0 and ca["contacts"] == b.counts()["contacts"] and ca["contacts"] >= 4 * len(idx) - 4
    assert ca["broadphase_inserts"] > c0["broadphase_inserts"]
    assert ca["still_steps"] - c0["still_steps"] >= done - 40, (c0, ca)
    a.close(); b.close()


def test_hops_landings_and_slides_through_local_speculation():
    """Short hops: bodies leave their boxes, are in the air for a few dozen steps above their neighbours' boxes (appended pairs that overlap but must not touch), land -- a
    full step: a contact of a pair without a record -- slide, and come to rest.  nh_step with late verdicts, the eight calls and a library that never speculates end in
    the same bits at every checkpoint."""
    scene = S.grid_tiles(1, side=40, seed=67)
    nb = len(scene["body_transforms"])
    a, b, c = _world(scene, True), _world(scene, True), _world(scene, False)
    b.eight_calls = True
    for w in (a, b, c):
        w.step(110)
    idx = np.arange(3, nb, 53)
    c0 = a.counts()
    for rnd in range(2):                                                  # (the world stays awake: 110 + 2 x 80 steps)
        _throw((a, b, c), idx, 2.5, lateral=3.0, seed=11 + rnd)          # (tests/gpu_movers_rates.py "hop")
        at = 0
        for more in (5, 25, 50):
            for w in (a, b, c):
                w.step(more)
            at += more
            _same_world(a, c, f"hop {rnd}, step {at}: nh_step vs no speculation")
            _same_world(b, c, f"hop {rnd}, step {at}: eight calls vs no speculation")
    ca = a.counts()
    print(f"\n[hops] still steps {ca['still_steps'] - c0['still_steps']} of {2 * 80}, replays {ca['still_replays'] - c0['still_replays']}, re-insertions {ca['broadphase_inserts'] - c0['broadphase_inserts']}")
    assert ca["error"] == 0 and ca["active_bodies"] == nb - 1
    # (31 bodies hopping across their neighbours: the steps in which one of them touches another body -- a contact the fused solver does not own -- are full steps)
    assert ca["still_steps"] - c0["still_steps"] >= 40, (c0, ca)
    a.close(); b.close(); c.close()


def test_counts_and_synchronize_every_step_export_nothing_and_views_are_the_bits_of_a_full_step():
    """The sample reads active_bodies.count every sub-step (example/main.cpp:293).  nh_read_counts / nh_synchronize are cheap observers: they keep the world in the
    still regime and export no view; the dense contact list and the cache a caller gets AFTERWARDS (nh_export_views) are the bits of a library that never speculates."""
    scene = S.grid_tiles(1, side=32, seed=71)
    a, b = _world(scene, True), _world(scene, False)
    a.step(100); b.step(100)
    c0 = a.counts()
    for _ in range(40):
        a.step(1); b.step(1)
        ca, cb = a.counts(), b.counts()
        a.synchronize()
        for k in ("contacts", "active_bodies", "cache", "sleeping_pairs", "error"):
            assert ca[k] == cb[k], (k, ca[k], cb[k])
    c1 = a.counts()
    assert c1["still_steps"] - c0["still_steps"] >= 38, (c0, c1)
    _same_world(a, b, "views after forty observed steps")
    a.step(5); b.step(5)
    assert a.counts()["still_steps"] - c1["still_steps"] >= 4          # (exporting the views did not end the regime either)
    _same_world(a, b, "and again")
    a.close(); b.close()


def _same_sleepy_world(a, b, what=""):
    """Bodies, cache, contacts (_same_world) + what a world with sleepers adds: sleeping pairs, active list, the counters a caller sees."""
    _same_world(a, b, what)
    ka, kb = a.get_contacts(), b.get_contacts()
    assert np.array_equal(ka["sleeping_pairs"], kb["sleeping_pairs"]), f"{what}: sleeping pairs differ ({len(ka['sleeping_pairs'])} vs {len(kb['sleeping_pairs'])})"
    assert np.array_equal(a.get_active(), b.get_active()), f"{what}: active bodies differ"
    ca, cb = a.counts(), b.counts()
    for k in ("contacts", "sleeping_pairs", "active_bodies", "cache", "error"):
        assert ca[k] == cb[k], f"{what}: counter {k}: {ca[k]} vs {cb[k]}"


def test_the_legacy_option_makes_synchronize_export_the_views():
    """Hosts written against rounds 1-4 read the contact list / cache / active list from the device after nh_synchronize.  Since round 5 the cheap observers export
    nothing (above); nh_set_option("sync_exports_views", 1) restores the old contract: the caller's arrays, read WITHOUT nh_export_views, are a full step's bits."""
    from nudge_amd import engine as EE
    scene = S.grid_tiles(1, side=24, seed=71)
    a, b = _world(scene, True, env=("NH_SYNC_EXPORTS_VIEWS",)), _world(scene, False)
    a.step(100); b.step(100)
    c0 = a.counts()
    a.step(12); b.step(12)
    a.synchronize()
    n = b.counts()["contacts"]
    kb = b.get_contacts()
    assert a.counts()["still_steps"] - c0["still_steps"] >= 10
    # (the raw device arrays of world a, no nh_export_views call: World._down copies what is there)
    assert P.bits_equal(a._down("cd", EE.S.CONTACT, n), kb["data"]) and np.array_equal(a._down("ct", np.uint64, n), kb["tags"]) and np.array_equal(a._down("cf", np.uint32, n), kb["features"])
    cb = b.get_cache()
    assert np.array_equal(a._down("kt", np.uint64, cb["count"]), cb["tags"]) and P.bits_equal(a._down("kd", EE.S.IMPULSE, cb["count"])["impulse"], cb["data"]["impulse"])
    a.close(); b.close()


@pytest.mark.parametrize("sphere_fraction", [0.0, 0.5])
def test_a_world_that_is_partly_asleep_takes_still_steps(sphere_fraction):
    """SLEEPERS form.  Two tiles fall asleep; the caller wakes one of them (idle counters rewritten, nh_bodies_changed).  For the next ~250 steps half the world is awake
    and at rest, half asleep: round 4 ran every one of those steps in full (still steps needed everybody awake, asleep steps everybody asleep).  Now they are still
    steps in which the sleepers' pairs are sleeping pairs, their cache entries stay aside and their bodies are left alone -- bit-identical to a library that runs
    every step in full (bodies, idle counters, contacts, sleeping pairs, active list, cache, counters), through the second dozing-off and into the asleep steps."""
    scene = S.grid_tiles(2, side=32, sphere_fraction=sphere_fraction, seed=73, lattice_cols=2)
    nb = len(scene["body_transforms"])
    a, b = _world(scene, True), _plain_world(scene)
    a.step(420); b.step(420)
    _same_sleepy_world(a, b, "asleep")
    assert a.counts()["active_bodies"] == 0
    tile0 = np.flatnonzero(scene["body_transforms"]["position"][:, 0] < np.median(scene["body_transforms"]["position"][1:, 0]))
    tile0 = tile0[tile0 > 0]
    for w in (a, b):
        idle = w.get_bodies()["idle"].copy()
        idle[tile0] = 0
        w.set_bodies(idle=idle)
    c0 = a.counts()
    done = 0
    for cp in (1, 2, 3, 10, 60, 61, 150, 240, 252, 254, 255, 256, 257, 258, 262, 270, 300, 330):
        a.step(cp - done); b.step(cp - done); done = cp
        _same_sleepy_world(a, b, f"half awake, step {cp}")
        if cp == 150:
            cm = a.counts()
            assert 0 < cm["active_bodies"] < nb - 1 and cm["sleeping_pairs"] > 0 and cm["contacts"] > 0, cm
            still_mid = cm["still_steps"] - c0["still_steps"]
    c1 = a.counts()
    print(f"\n[sleepers, spheres {sphere_fraction}] half the world asleep: {still_mid} of the first 150 steps were still steps; {c1['still_steps'] - c0['still_steps']} still + {c1['asleep_steps'] - c0['asleep_steps']} asleep of {done}, replays {c1['still_replays'] - c0['still_replays']}")
    assert still_mid >= 120, (c0, cm)
    assert c1["active_bodies"] == 0 and c1["asleep_steps"] > c0["asleep_steps"]          # (everybody asleep again: the asleep steps took over)
    assert b.counts()["still_steps"] == 0
    a.close(); b.close()


def test_what_the_sleepers_form_skips_changes_nothing():
    """Round 6.  Inside one nh_step call, from its second sub-step on, a still step in sleepers form takes an asleep collider's bounds from the AABB on file, counts a record
    that was a sleeping pair and still has both ends static or asleep without fetching anything else, and lets solver waves of sleeping bodies leave at once; the sleeping
    pairs are counted in 64 places.  Calls of every length (a one-step call never skips), a body woken by the caller between two calls, the second dozing-off: the same bits
    as the library with the option off and as the one that runs every step in full."""
    scene = S.grid_tiles(2, side=32, sphere_fraction=0.5, seed=75, lattice_cols=2)
    nb = len(scene["body_transforms"])
    a, b, c = _world(scene, True), _world(scene, True, env=["NH_NO_SLEEPER_SKIP"]), _plain_world(scene)
    b2 = _world(scene, True, env=["NH_NO_SLEEPER_AHEAD"])          # (round 6, sleepers ahead: one solver launch per sub-step while the sleeping set stands still -- and the three kernels)
    class _Both:          # (b and b2 step and are compared together)
        def __init__(self, *ws): self.ws = ws
        def step(self, n): [w.step(n) for w in self.ws]
        def get_bodies(self): return self.ws[0].get_bodies()
        def set_bodies(self, **kw): [w.set_bodies(**kw) for w in self.ws]
        def close(self): [w.close() for w in self.ws]
    _same0 = globals()["_same_sleepy_world"]
    def _same_sleepy_world(x, y, what=""):
        if isinstance(y, _Both):
            for k, w in enumerate(y.ws): _same0(x, w, what + (" [sleepers ahead off]" if k else ""))
        else: _same0(x, y, what)
    b = _Both(b, b2)
    for w in (a, b, c):
        w.step(420)
    _same_sleepy_world(a, c, "asleep"); _same_sleepy_world(a, b, "asleep (option off)")
    order = np.argsort(scene["body_transforms"]["position"][1:, 0], kind="stable") + 1
    for w in (a, b, c):
        idle = w.get_bodies()["idle"].copy()
        idle[order[: (nb - 1) // 3]] = 0
        w.set_bodies(idle=idle)
    c0 = a.counts()
    done = 0
    for cp in (1, 2, 3, 5, 9, 40, 41, 120):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_sleepy_world(a, c, f"a third awake, step {cp}"); _same_sleepy_world(a, b, f"a third awake, step {cp} (option off)")
    # the caller wakes a few sleepers between two calls (idle counters rewritten: the library is told)
    for w in (a, b, c):
        idle = w.get_bodies()["idle"].copy()
        idle[order[-40:]] = 0
        w.set_bodies(idle=idle)
    for cp in (121, 125, 180, 240, 262, 300, 380, 400):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_sleepy_world(a, c, f"more woken, step {cp}"); _same_sleepy_world(a, b, f"more woken, step {cp} (option off)")
    c1 = a.counts()
    print(f"\n[sleeper skip] {c1['still_steps'] - c0['still_steps']} still + {c1['asleep_steps'] - c0['asleep_steps']} asleep steps of {done}; replays {c1['still_replays'] - c0['still_replays']}; pair steps {c1['pair_steps'] - c0['pair_steps']} (sleepers ahead)")
    assert c1["error"] == 0 and c1["still_steps"] - c0["still_steps"] >= 150, (c0, c1)
    a.close(); b.close(); c.close()


def test_early_counters_change_nothing():
    """Round 6.  A full step's counters are final when its one-body solver starts; the launch's first thread leaves them in pinned memory and the host, polling, reads them
    while the solver runs (nh_internal.h: early counters) instead of copying them behind it.  Which of the two the host does must not show anywhere: a drop scene through
    fall, rest and sleep in full steps, a pile (general contacts: the round trips after the first stay copies) and the speculating default, each against the same
    library with option no_early_counts -- bodies, idle counters, contacts, cache, counters bit for bit, in calls of one and of many steps."""
    for scene, cps in ((S.grid_tiles(2, side=32, sphere_fraction=0.5, seed=83, lattice_cols=2), (1, 2, 30, 60, 61, 100, 200, 330, 420)),
                       (S.pile(n_boxes=300, n_spheres=60, seed=9, iterations=8), (1, 2, 3, 40, 41, 120))):
        a, b = _plain_world(scene), _plain_world(scene)
        b.set_option("no_early_counts", 1)
        c, d = _world(scene, True), _world(scene, True, env=["NH_NO_EARLY_COUNTS"])
        done = 0
        for cp in cps:
            for w in (a, b, c, d):
                w.step(cp - done)
            done = cp
            _same_sleepy_world(a, b, f"{scene['name']} full steps, step {cp}")
            _same_sleepy_world(c, d, f"{scene['name']} default, step {cp}")
            _same_sleepy_world(a, c, f"{scene['name']} default against full steps, step {cp}")
        assert a.counts()["error"] == 0
        for w in (a, b, c, d):
            w.close()


def test_sleepers_ahead_through_falling_asleep_in_turns():
    """Round 6 (nh_internal.h: sleepers ahead).  While the sleeping set stands still a step in sleepers form is one solver launch behind k_pair_begin; a body that falls
    asleep fails the next step (replayed in full) and the form backs off.  A drop scene asleep, then three groups woken with idle counters 0 / 100 / 200 -- they fall asleep
    again in turns, 55, 155 and 255 steps later, each time in the middle of a long call in which the lanes work ahead -- and a few more woken by the caller between two calls:
    bodies, idle counters, contacts, sleeping pairs, active list, cache and counters bit for bit against the option off and against the library that runs every step in full."""
    scene = S.grid_tiles(2, side=32, sphere_fraction=0.5, seed=91, lattice_cols=2)
    nb = len(scene["body_transforms"])
    a, b, c = _world(scene, True), _world(scene, True, env=["NH_NO_SLEEPER_AHEAD"]), _plain_world(scene)
    for w in (a, b, c):
        w.step(420)
    _same_sleepy_world(a, c, "asleep"); _same_sleepy_world(a, b, "asleep (option off)")
    order = np.argsort(scene["body_transforms"]["position"][1:, 2], kind="stable") + 1
    third = (nb - 1) // 9
    for w in (a, b, c):
        idle = w.get_bodies()["idle"].copy()
        idle[order[:third]] = 0; idle[order[third:2 * third]] = 100; idle[order[2 * third:3 * third]] = 200
        w.set_bodies(idle=idle)
    c0 = a.counts()
    done = 0
    for cp in (64, 128, 129, 192, 256):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_sleepy_world(a, c, f"groups falling asleep in turns, step {cp}"); _same_sleepy_world(a, b, f"groups falling asleep in turns, step {cp} (option off)")
    for w in (a, b, c):
        idle = w.get_bodies()["idle"].copy()
        idle[order[-25:]] = 0
        w.set_bodies(idle=idle)
    for cp in (320, 384, 400, 520, 640):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_sleepy_world(a, c, f"more woken, step {cp}"); _same_sleepy_world(a, b, f"more woken, step {cp} (option off)")
    c1, cb = a.counts(), b.counts()
    print(f"\n[sleepers ahead] {c1['still_steps'] - c0['still_steps']} still + {c1['asleep_steps'] - c0['asleep_steps']} asleep steps of {done}; replays {c1['still_replays'] - c0['still_replays']}; "
          f"pair steps {c1['pair_steps'] - c0['pair_steps']} (option off: {cb['pair_steps']})")
    assert c1["error"] == 0 and c1["pair_steps"] - c0["pair_steps"] >= 200 and c1["still_replays"] - c0["still_replays"] <= 24, (c0, c1)
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("seed", [3, 17])
def test_sleepers_ahead_against_full_steps_through_random_wake_ups(seed):
    """Round 6 (sleepers ahead), a seeded random walk: a drop scene of boxes and spheres asleep; eight rounds in which the caller wakes a random few with random idle counters
    (they fall asleep again at random times, alone or together, in the middle of calls in which the lanes work ahead), gives one of them a shove now and then (a mover among
    sleepers), and steps in calls of random length.  After every call: bodies, idle counters, contacts, sleeping pairs, active list, cache and counters bit for bit against
    the library that runs every step in full."""
    rng = np.random.default_rng(seed)
    scene = S.grid_tiles(2, side=24, sphere_fraction=0.4, seed=100 + seed, lattice_cols=2)
    nb = len(scene["body_transforms"])
    a, c = _world(scene, True), _plain_world(scene)
    for w in (a, c):
        w.step(420)
    _same_sleepy_world(a, c, "asleep")
    c0 = a.counts()
    steps = 0
    for rnd in range(8):
        k = int(rng.integers(1, nb // 3))
        who = rng.choice(np.arange(1, nb), size=k, replace=False)
        idles = rng.integers(0, 255, size=k).astype(np.uint8)
        if rnd % 3 == 1:
            idles[:] = int(rng.integers(150, 250))          # (a group that falls asleep together)
        elif rnd % 3 == 2:
            idles[:] = 0; idles[: max(1, k // 8)] = rng.integers(100, 255, size=max(1, k // 8)).astype(np.uint8)          # (most of them rest awake for the whole round; a few doze off on the way)
        shove = rnd % 4 == 2
        for w in (a, c):
            bd = w.get_bodies()
            idle = bd["idle"].copy(); idle[who] = idles
            if shove:
                m = bd["momentum"].copy(); m["velocity"][who[0]] = np.float32([0.6, 1.2, -0.4]); idle[who[0]] = 0
                w.set_bodies(momentum=m, idle=idle)
            else:
                w.set_bodies(idle=idle)
        for n in rng.integers(1, 70, size=4):
            for w in (a, c):
                w.step(int(n))
            steps += int(n)
            _same_sleepy_world(a, c, f"seed {seed}, round {rnd}, step {steps}")
    c1 = a.counts()
    print(f"\n[sleepers ahead, random walk {seed}] {c1['still_steps'] - c0['still_steps']} still + {c1['asleep_steps'] - c0['asleep_steps']} asleep steps of {steps}; replays {c1['still_replays'] - c0['still_replays']}; pair steps {c1['pair_steps'] - c0['pair_steps']}")
    assert c1["error"] == 0
    a.close(); c.close()


def test_a_sleeper_next_to_an_awake_body_is_not_the_still_steps_business():
    """A sleeping body whose AABB overlaps an awake dynamic body's belongs to a set of more than one body (nudge.cpp:3575-3650): whether it sleeps is not local.  Such a
    step must fail as a still step and be run in full -- the world ends in the bits of the library that never speculates."""
    scene = S.grid_tiles(1, side=24, seed=79)
    nb = len(scene["body_transforms"])
    a, b = _world(scene, True), _plain_world(scene)
    a.step(420); b.step(420)
    for w in (a, b):
        bd = w.get_bodies()
        idle, t = bd["idle"].copy(), bd["transforms"].copy()
        idle[5::7] = 0                                             # every seventh body awake ...
        j = 5 + 7 * 3
        t["position"][j] = t["position"][j + 1] + np.float32([0.4, 2.2, 0.0])          # ... and one of them put right above its sleeping neighbour: it falls onto it
        w.set_bodies(transforms=t, idle=idle)
    done = 0
    for cp in (1, 5, 30, 80, 200):
        a.step(cp - done); b.step(cp - done); done = cp
        _same_sleepy_world(a, b, f"sleeper with an awake neighbour, step {cp}")
    a.close(); b.close()


def test_a_drop_from_the_slot_cache_in_a_step_that_fails_is_undone():
    """Advisor (round 5, medium).  In sleepers form the still narrowphase files the pair of a sleeping body with the static world as a sleeping pair and, when the
    pair's word differs from the contact key on file (a sphere on a slab whose tag is the smaller one: nudge.cpp:4064-4101 drops those cache entries), zeroes the slot
    cache's count of that record -- BEFORE the step's verdict is known.  Here sphere B's idle counter saturates after a run of confirmed still steps (so the caller's
    cache is stale and will be rebuilt from the slots) while an awake sphere A rests beside it with overlapping AABBs and no contact: {A, B} is a set with an awake
    body, the still step must fail, and the full replay keeps B awake and warm-starts its contact from the cache.  The drop must not have happened."""
    # (a world large enough for the kept pair list to stay "in place" through A's re-insertions: at least 15 of 16 kept pairs must overlap, k_kept_filter)
    scene = S.grid_tiles(1, side=16, sphere_fraction=0.5, seed=83)
    a, b = _world(scene, True), _plain_world(scene)
    a.step(150); b.step(150)

    sph_body = scene["sphere_transforms"]["body"]
    rad = scene["sphere_data"]["radius"]
    pos0 = scene["body_transforms"]["position"]
    # B: the sphere nearest the (+x, -z) corner of the lattice; A: the sphere farthest from it, put diagonally OUTSIDE the lattice next to B
    corner = np.float32([pos0[1:, 0].max(), 0.0, pos0[1:, 2].min()])
    d2 = ((pos0[sph_body][:, [0, 2]] - corner[[0, 2]]) ** 2).sum(axis=1)
    kb, ka = int(np.argmin(d2)), int(np.argmax(d2))
    B, A = int(sph_body[kb]), int(sph_body[ka])
    gap = np.float32(0.8) * (rad[ka] + rad[kb])
    for w in (a, b):
        bd = w.get_bodies()
        t, m, idle = bd["transforms"].copy(), bd["momentum"].copy(), bd["idle"].copy()
        t["position"][A] = t["position"][B] + np.float32([gap, 0.0, -gap])
        t["position"][A][1] = np.float32(-10.0) + rad[ka]
        m["velocity"][A] = 0; m["angular_velocity"][A] = 0
        idle[:] = 0
        w.set_bodies(transforms=t, momentum=m, idle=idle)
    # (A settles on the slab for a few steps, leaving its inflated box on the way: still steps offered meanwhile fail and are held back, up to 63 steps at a time --
    # nh_internal.h, back-off -- so give the world time to come back to the still regime)
    a.step(180); b.step(180)
    _same_sleepy_world(a, b, "A moved next to B")
    cs = a.counts()
    a.step(20); b.step(20)
    ce = a.counts()
    assert ce["still_steps"] - cs["still_steps"] >= 18, {k: (cs[k], ce[k]) for k in cs if k.startswith("still") or k in ("broadphase_rebuilds", "broadphase_inserts", "general_contacts", "has_other_bodies")}          # (two dynamic bodies with overlapping boxes and no contact: still steps all the same)
    for w in (a, b):
        idle = w.get_bodies()["idle"].copy()
        idle[:] = 0
        idle[B] = 0xff - 12
        w.set_bodies(idle=idle)
    c0 = a.counts()
    done = 0
    for cp in (8, 11, 12, 13, 14, 16, 20, 40):
        a.step(cp - done); b.step(cp - done); done = cp
        _same_sleepy_world(a, b, f"B saturates beside an awake A, step {cp}")
    c1 = a.counts()
    print(f"\n[undo] still steps {c1['still_steps'] - c0['still_steps']}, replays {c1['still_replays'] - c0['still_replays']}, idle[B] = {a.get_bodies()['idle'][B]}")
    assert c1["still_steps"] - c0["still_steps"] >= 6 and c1["still_replays"] > c0["still_replays"]          # (confirmed still steps first, then the failing one)
    assert a.get_bodies()["idle"][B] == 0xff          # (B's counter is saturated; whether its set sleeps is the full steps' business -- and the same in both worlds)
    a.close(); b.close()


# ---- asleep steps (nudge_amd/csrc/nh_internal.h: nh_AsleepState): nh_step on a world in which every body is asleep ---------------------------------------------
def _plain_world(scene, **kw):
    """Neither still steps nor asleep steps: every step in full."""
    os.environ["NH_NO_STILL"] = "1"; os.environ["NH_NO_ASLEEP"] = "1"
    try:
        return E.World(scene, flags=BENCH_FLAGS, **kw)
    finally:
        os.environ.pop("NH_NO_STILL", None); os.environ.pop("NH_NO_ASLEEP", None)


def _same_asleep_world(a, b, what=""):
    _same_world(a, b, what)
    ka, kb = a.get_contacts(), b.get_contacts()
    assert np.array_equal(ka["sleeping_pairs"], kb["sleeping_pairs"]), f"{what}: sleeping pairs differ"
    assert np.array_equal(a.get_active(), b.get_active()), f"{what}: active bodies differ"
    ca, cb = a.counts(), b.counts()
    for k in ("contacts", "sleeping_pairs", "active_bodies", "cache", "pairs", "error"):
        assert ca[k] == cb[k], f"{what}: counter {k}: {ca[k]} vs {cb[k]}"


def test_asleep_steps_change_nothing_and_leave_what_full_steps_leave():
    """A drop scene falls asleep around step 300.  From then on a full step reads everything and changes nothing (no active body, no contact, every pair a sleeping
    pair, the cache kept aside and written back: nudge.cpp:3669-3703, 4064-4101); nh_step checks once per call that nothing the caller owns has changed and takes the
    steps as done.  Against a library that runs every step in full: bodies, idle counters, contacts, sleeping pairs, active list, cache and counters bit for bit --
    through single-step calls, long calls, counts() in between; and against the compiled reference."""
    scene = S.grid_tiles(2, side=40, sphere_fraction=0.5, seed=53, lattice_cols=2)
    a, b = _world(scene, True), _plain_world(scene)
    done = 0
    for cp in (280, 330, 331, 332, 340, 500, 501, 900):
        a.step(cp - done); b.step(cp - done); done = cp
        _same_asleep_world(a, b, f"step {cp}")
    ca = a.counts()
    print(f"\n[asleep steps] {ca['asleep_steps']} of {done} steps taken as done ({ca['still_steps']} still steps)")
    assert ca["active_bodies"] == 0 and ca["contacts"] == 0 and ca["sleeping_pairs"] > 0
    assert ca["asleep_steps"] >= 500, ca
    assert b.counts()["asleep_steps"] == 0
    if refworld.available("exact") and S.narrow_ok(scene):
        r = refworld.RefWorld(scene, max_contacts=8 * len(scene["body_transforms"]))
        r.step(done)
        rb, gb = r.bodies(), a.get_bodies()
        assert P.bits_equal(gb["transforms"], rb["transforms"]) and np.array_equal(gb["idle"], rb["idle"])
    a.close(); b.close()


def test_whatever_the_caller_changes_while_the_world_sleeps_is_seen():
    """Between two nh_step calls the caller may do anything to its arrays.  Told (nh_bodies_changed: idle counters rewritten) or untold (a sleeping body teleported, a
    static collider moved, a collider's tag rewritten): the next call runs its steps in full and ends where the library that never skips ends."""
    scene = S.grid_tiles(1, side=24, seed=59)
    a, b = _world(scene, True), _plain_world(scene)
    a.step(400); b.step(400)
    a.step(20); b.step(20)
    c0 = a.counts()
    assert c0["active_bodies"] == 0 and c0["asleep_steps"] > 0
    # (1) a sleeping body is put somewhere else, nobody is told: it stays asleep (nudge.cpp:3669-3672), its overlaps -- the sleeping pairs -- are others
    for w in (a, b):
        t = w.get_bodies()["transforms"]
        t["position"][7] += np.float32(1.25)
        w.set_bodies(transforms=t)
    a.step(5); b.step(5)
    _same_asleep_world(a, b, "teleported sleeper")
    a.step(40); b.step(40)
    _same_asleep_world(a, b, "teleported sleeper, later")
    c1 = a.counts()
    assert c1["asleep_steps"] - c0["asleep_steps"] < 45 and c1["asleep_steps"] > c0["asleep_steps"]          # (seen, and asleep steps again afterwards)
    # (2) the ground is moved up a little under everybody (static collider 0), untold
    for w in (a, b):
        bx = w.records("xx", 32)
        raw = np.frombuffer(bx[:1].cpu().numpy().tobytes(), dtype=S.TRANSFORM).copy()
        raw["position"][0, 1] += np.float32(0.01)
        bx[:1] = w.torch.from_numpy(raw.view(np.uint8).reshape(1, 32).copy()).to(bx.device)
    a.step(3); b.step(3)
    _same_asleep_world(a, b, "moved ground")
    # (3) somebody is woken up and thrown: the world goes through contact, rest and sleep again
    for w in (a, b):
        bd = w.get_bodies()
        idle, m = bd["idle"].copy(), bd["momentum"].copy()
        idle[5::41] = 0
        m["velocity"][5::41, 1] = 3.0
        w.set_bodies(momentum=m, idle=idle)
    for cp in (30, 200, 420, 470):
        a.step(cp if cp == 30 else cp - prev); b.step(cp if cp == 30 else cp - prev); prev = cp
        _same_asleep_world(a, b, f"woken, step {cp}")
    c2 = a.counts()
    assert c2["active_bodies"] == 0 and c2["asleep_steps"] > c1["asleep_steps"]
    a.close(); b.close()


# ---- state streaming (include/nudge_hip.h: nh_stream_state): the GL-free viewer hook ------------------------------------------------------------------------------
def test_state_stream_delivers_the_transforms_of_the_steps_it_names():
    """Every 5th sub-step the body transforms travel into a pinned host ring without the world waiting for them.  Whatever frame `nh_stream_latest` hands out must be
    the transforms a second world has after exactly that many steps -- through full steps (fall, landing), still steps and the asleep steps (which take no frames:
    nothing moves) -- and the world itself must end in the bits of one that streams nothing."""
    scene = S.grid_tiles(1, side=24, seed=83)
    a, b, ref = _world(scene, True), _world(scene, True), _world(scene, True)
    a.stream_state(every=5, slots=4)
    done, seen = 0, 0
    for more in (7, 30, 33, 60, 41, 200, 120):
        a.step(more); b.step(more); done += more
        a.synchronize()
        frame, stats = a.stream_latest()
        assert frame is not None, stats
        step, xf = frame
        assert step % 5 == 0 and step <= done, (step, done, stats)
        if step > seen:
            ref.step(step - seen); seen = step
            assert P.bits_equal(xf, ref.get_bodies()["transforms"]), f"frame of step {step} (world at {done})"
    _same_world(a, b, "a world that streams its state against one that does not")
    frame, stats = a.stream_latest()
    print(f"\n[state stream] {stats['frames']} frames, {stats['dropped']} dropped over {done} steps")
    assert stats["frames"] >= 40
    a.stream_state(every=0)
    a.step(10); b.step(10)
    _same_world(a, b, "after disarming")
    a.close(); b.close(); ref.close()


@pytest.mark.parametrize("sphere_fraction", [0.0, 0.5])
def test_xform_ahead_leaves_the_bits_of_steps_that_transform_themselves(sphere_fraction):
    """Inside one nh_step call a plain still step's solver does the next step's k_xform<true> (nh_internal.h: xform ahead): that step starts at the narrowphase.  Against
    the same library with the option off, and against the one that never speculates, checkpoint by checkpoint: through rest (most steps ahead), through a hop and a
    shove (a collider outside its inflated box is seen by the solver lane that advanced it and fails the NEXT step), through falling asleep (the saturated idle
    counter likewise), and with calls of every length -- one step (never ahead), two, many."""
    scene = S.grid_tiles(2, side=36, sphere_fraction=sphere_fraction, seed=53, lattice_cols=2)
    a, b, c = _world(scene, True), _world(scene, True, env=["NH_NO_XFORM_AHEAD"]), _world(scene, False)
    done = 0
    for cp in (90, 91, 93, 120, 121, 160):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_world(a, b, f"step {cp} (ahead vs not)"); _same_world(a, c, f"step {cp} (ahead vs never speculating)")
    c0 = a.counts()
    assert c0["ahead_steps"] >= 40, c0
    assert b.counts()["ahead_steps"] == 0 and b.counts()["still_steps"] >= 40
    for w in (a, b, c):
        m = w.get_bodies()["momentum"]
        m["velocity"][7::89, 1] = 3.0
        m["velocity"][13::101, 2] = 1.0
        w.set_bodies(momentum=m)
    for cp in (200, 260, 330, 420, 520):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_world(a, b, f"step {cp} (ahead vs not)"); _same_world(a, c, f"step {cp} (ahead vs never speculating)")
    c1 = a.counts()
    print(f"\n[xform ahead, spheres {sphere_fraction}] {c1['ahead_steps']} of {c1['still_steps']} still steps started at the narrowphase; replays {c1['still_replays']} (without: {b.counts()['still_replays']})")
    # (after the kick the still steps are in movers form, then -- the idle counters of everybody else near saturation -- in sleepers form: few plain ones)
    assert c1["error"] == 0 and c1["ahead_steps"] > c0["ahead_steps"] and c1["still_replays"] > c0["still_replays"], (c0, c1)
    a.close(); b.close(); c.close()


def test_xform_ahead_is_refused_for_bodies_with_several_colliders():
    """A solver lane stands in for k_xform<true> for ONE collider.  Some boxes carry a second collider high above them (it touches nothing: the bodies stay in the class
    still steps own): the map check sees it on the device, the step that relied on the map fails itself, the host stops offering -- and every bit is what the library
    with the option off leaves."""
    scene = S.grid_tiles(1, side=20, seed=59)
    n = len(scene["body_transforms"]) - 1
    owners = np.arange(1, n + 1, dtype=np.uint32)[3::7]
    sph_t = np.zeros(len(owners), dtype=scene["sphere_transforms"].dtype)
    sph_t["rotation"][:, 3] = 1.0
    sph_t["position"][:, 1] = 2.5
    sph_t["body"] = owners
    sph_d = np.zeros(len(owners), dtype=scene["sphere_data"].dtype)
    sph_d["radius"] = 0.25
    first_tag = int(scene["box_tags"].max()) + 1
    scene.update(sphere_transforms=sph_t, sphere_data=sph_d, sphere_tags=np.arange(first_tag, first_tag + len(owners), dtype=np.uint32))
    a, b = _world(scene, True), _world(scene, True, env=["NH_NO_XFORM_AHEAD"])
    done = 0
    for cp in (100, 140, 141, 200):
        a.step(cp - done); b.step(cp - done); done = cp
        _same_world(a, b, f"step {cp}")
    ca, cb = a.counts(), b.counts()
    print(f"\n[xform ahead refused] ahead steps {ca['ahead_steps']}, still steps {ca['still_steps']} (option off: {cb['still_steps']}), replays {ca['still_replays']} / {cb['still_replays']}")
    assert ca["error"] == 0 and ca["still_steps"] >= 80
    assert ca["ahead_steps"] <= 4 and ca["still_replays"] <= cb["still_replays"] + 4, (ca, cb)
    a.close(); b.close()


def test_xform_ahead_in_a_world_of_spheres_on_one_slab():
    """One static box and nothing but spheres: the still narrowphase has no box-box launch (the sphere launch clears, reduces and checks for the step), every solver lane's
    collider is a sphere.  With and without the option, and against the library that never speculates."""
    f = np.float32
    n = 30 * 30
    st = S._identity_transforms(1)
    st["position"][0] = (0.0, -20.0, 0.0)
    ssz = np.array([[60.0, 10.0, 60.0]], dtype=np.float32)
    d = S._draws(0x9E3779B97F4A7C15 ^ 61, n)
    spt = S._identity_transforms(n)
    spt["position"][:, 0] = (np.arange(n) % 30).astype(np.float32) * f(3.0) - f(43.5) + d[:, 4] * f(0.2)
    spt["position"][:, 2] = (np.arange(n) // 30).astype(np.float32) * f(3.0) - f(43.5) + d[:, 5] * f(0.2)
    radius = (d[:, 0] * f(0.5) + f(0.5)).astype(np.float32)
    spt["position"][:, 1] = f(-10.0) + radius + d[:, 6] * f(0.5) + f(0.5)
    empty = S._identity_transforms(0)
    scene = S._assemble((st, ssz), (empty, np.zeros((0, 3), np.float32), np.zeros(0, S.PROPERTIES)), (spt, radius, S._sphere_properties(radius)),
                        dict(S.DEFAULT_PARAMS, iterations=8), name="spheres900")
    a, b, c = _world(scene, True), _world(scene, True, env=["NH_NO_XFORM_AHEAD"]), _world(scene, False)
    done = 0
    for cp in (80, 81, 120, 200, 320, 450):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_world(a, b, f"step {cp} (ahead vs not)"); _same_world(a, c, f"step {cp} (ahead vs never speculating)")
    ca = a.counts()
    print(f"\n[xform ahead, spheres only] {ca['ahead_steps']} of {ca['still_steps']} still steps started at the narrowphase; replays {ca['still_replays']}")
    assert ca["error"] == 0 and ca["ahead_steps"] >= 60, ca
    a.close(); b.close(); c.close()


# ---- pair ahead (nudge_amd/csrc/nh_internal.h: nh_StillStep) -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sphere_fraction", [0.0, 0.5])
def test_pair_ahead_leaves_the_bits_of_steps_that_run_their_narrowphase(sphere_fraction):
    """Inside one nh_step call the solver lane that has advanced a body and written its collider's new transform (xform ahead) also evaluates the body's one kept pair for
    the NEXT sub-step: contacts into the record's raw slots, count into its place in the tag order, a change onto the next step's list.  That sub-step is the solver alone
    behind a one-workgroup prologue.  Against the same library with the option off and against the one that never speculates, checkpoint by checkpoint: calls of every
    length, rest, a hop and a shove (the lanes see the leaver: the next step fails before it starts), falling asleep; the views a caller exports at the end of a call are
    the last step's (its contacts were written by the step before it)."""
    scene = S.grid_tiles(2, side=36, sphere_fraction=sphere_fraction, seed=63, lattice_cols=2)
    a, b, c = _world(scene, True), _world(scene, True, env=["NH_NO_PAIR_AHEAD"]), _world(scene, False)
    done = 0
    for cp in (90, 91, 93, 96, 100, 105, 111, 118, 126, 160, 161, 163, 200):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_world(a, b, f"step {cp} (pair ahead vs narrowphase launched)"); _same_world(a, c, f"step {cp} (pair ahead vs never speculating)")
    c0 = a.counts()
    assert c0["pair_steps"] >= 40, c0
    assert b.counts()["pair_steps"] == 0 and b.counts()["ahead_steps"] >= 40
    for w in (a, b, c):
        m = w.get_bodies()["momentum"]
        m["velocity"][7::89, 1] = 3.0
        m["velocity"][13::101, 2] = 1.0
        w.set_bodies(momentum=m)
    for cp in (240, 300, 370, 460, 560):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_world(a, b, f"step {cp} (pair ahead vs narrowphase launched)"); _same_world(a, c, f"step {cp} (pair ahead vs never speculating)")
    c1 = a.counts()
    print(f"\n[pair ahead, spheres {sphere_fraction}] {c1['pair_steps']} pair of {c1['ahead_steps']} ahead of {c1['still_steps']} still steps; replays {c1['still_replays']} (narrowphase launched: {b.counts()['still_replays']})")
    assert c1["error"] == 0 and c1["pair_steps"] > c0["pair_steps"], (c0, c1)
    a.close(); b.close(); c.close()


def test_pair_ahead_with_kept_pairs_that_are_nobodys():
    """Boxes that start high above the slab land later than their neighbours: while one hovers inside its inflated box's reach of the ground, its kept pair has no contact
    and is no body's contact-list record -- no solver lane evaluates it.  Such pairs are listed once per layout (k_pair_list) and evaluated by the step's prologue
    (k_pair_begin) the way the narrowphase did; the step in which one of them yields a contact fails and is run in full: the landing is seen, bit for bit."""
    scene = S.grid_tiles(1, side=28, seed=65)
    high = np.arange(1, len(scene["body_transforms"]))[5::37]
    scene["body_transforms"]["position"][high, 1] += 14.0
    for k in ("box_transforms",):
        pass
    a, b, c = _world(scene, True), _world(scene, True, env=["NH_NO_PAIR_AHEAD"]), _world(scene, False)
    done = 0
    for cp in (60, 90, 120, 150, 200, 260, 320):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_world(a, b, f"step {cp} (pair ahead vs narrowphase launched)"); _same_world(a, c, f"step {cp} (pair ahead vs never speculating)")
    ca, cb = a.counts(), b.counts()
    print(f"\n[pair ahead, late landers] pair {ca['pair_steps']} of {ca['ahead_steps']} ahead of {ca['still_steps']} still; replays {ca['still_replays']} / {cb['still_replays']}")
    assert ca["error"] == 0 and ca["still_replays"] <= cb["still_replays"] + 8, (ca, cb)
    a.close(); b.close(); c.close()


# ---- the one-kernel still step with a look-back (nudge_amd/csrc/nh_fused.hip; option "fused_still", off by default: bit-exact but measured slower) -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sphere_fraction", [0.0, 0.5])
def test_fused_still_steps_leave_the_bits_of_the_two_kernel_form(sphere_fraction):
    """Inside an nh_step call a plain still step that starts at the narrowphase and is not the call's last is ONE launch: the lane that solves a body evaluates the body's
    kept pair itself, the contacts never reach HBM, the dense contact indices travel from wave to wave by a look-back, and body state / slot cache / collider buffers are
    written out of place (nh_fused.hip).  Against the same library with the option off and against the one that never speculates, checkpoint by checkpoint: calls of
    every length (one and two sub-steps: never fused; three: never -- pairs; four and more: fused), through rest, a hop and a shove, and falling asleep."""
    scene = S.grid_tiles(2, side=36, sphere_fraction=sphere_fraction, seed=57, lattice_cols=2)
    a, b, c = _world(scene, True, env=["NH_FUSED_STILL"]), _world(scene, True), _world(scene, False)
    done = 0
    for cp in (90, 91, 93, 96, 100, 105, 111, 118, 126, 160, 161, 163, 200):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_world(a, b, f"step {cp} (fused vs two kernels)"); _same_world(a, c, f"step {cp} (fused vs never speculating)")
    c0 = a.counts()
    assert c0["fused_steps"] >= 40 and c0["fused_steps"] % 2 == 0, c0
    assert b.counts()["fused_steps"] == 0 and b.counts()["ahead_steps"] >= 40
    for w in (a, b, c):
        m = w.get_bodies()["momentum"]
        m["velocity"][7::89, 1] = 3.0
        m["velocity"][13::101, 2] = 1.0
        w.set_bodies(momentum=m)
    for cp in (240, 300, 370, 460, 560):
        for w in (a, b, c):
            w.step(cp - done)
        done = cp
        _same_world(a, b, f"step {cp} (fused vs two kernels)"); _same_world(a, c, f"step {cp} (fused vs never speculating)")
    c1 = a.counts()
    print(f"\n[fused, spheres {sphere_fraction}] {c1['fused_steps']} fused of {c1['ahead_steps']} ahead of {c1['still_steps']} still steps; replays {c1['still_replays']} (two kernels: {b.counts()['still_replays']})")
    assert c1["error"] == 0 and c1["fused_steps"] > c0["fused_steps"], (c0, c1)
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("fail_in", [1, 2, 5, 6])
def test_a_fused_step_that_fails_after_its_waves_have_written_is_run_again_from_the_state_it_was_given(fail_in):
    """The reason the fused kernel writes out of place: a wave may raise the failure flag when others have long advanced their bodies.  The test hook "fused_fail_in"
    makes the n-th fused launch do exactly that at the very end of one of its middle waves -- with the state it was given in the caller's arrays (odd n) and in the
    shadow arrays (even n).  The step and the one launched behind it are run again; every bit is the library's that never speculates."""
    scene = S.grid_tiles(1, side=40, sphere_fraction=0.5, seed=59)
    a, c = _world(scene, True, env=["NH_FUSED_STILL"]), _world(scene, False)
    a.step(100); c.step(100)
    _same_world(a, c, "at rest")
    c0 = a.counts()
    a.set_option("fused_fail_in", fail_in)
    a.step(14); c.step(14)
    _same_world(a, c, f"fused launch {fail_in} failed on purpose")
    c1 = a.counts()
    assert c1["still_replays"] > c0["still_replays"] and c1["fused_steps"] > c0["fused_steps"], (c0, c1)
    a.step(40); c.step(40)
    _same_world(a, c, "and on")
    c2 = a.counts()
    assert c2["error"] == 0 and c2["fused_steps"] > c1["fused_steps"]
    a.close(); c.close()


def test_two_dynamic_bodies_that_come_to_touch_inside_their_boxes_fail_a_fused_step_cleanly():
    """A sphere creeping towards its neighbour inside its inflated box: the kept pair of the two dynamic bodies is a record nobody's lane owns, and the step in which it
    first yields a contact must fail -- in whatever form it is launched -- and be run again in full.  Bits of the library that never speculates throughout."""
    scene = S.grid_tiles(1, side=16, sphere_fraction=0.5, seed=61)
    a, c = _world(scene, True, env=["NH_FUSED_STILL"]), _world(scene, False)
    a.step(150); c.step(150)
    sph_body = scene["sphere_transforms"]["body"]
    rad = scene["sphere_data"]["radius"]
    pos0 = scene["body_transforms"]["position"]
    corner = np.float32([pos0[1:, 0].max(), 0.0, pos0[1:, 2].min()])
    d2 = ((pos0[sph_body][:, [0, 2]] - corner[[0, 2]]) ** 2).sum(axis=1)
    kb, ka = int(np.argmin(d2)), int(np.argmax(d2))
    B, A = int(sph_body[kb]), int(sph_body[ka])
    for w in (a, c):
        bd = w.get_bodies()
        t, m, idle = bd["transforms"].copy(), bd["momentum"].copy(), bd["idle"].copy()
        t["position"][A] = t["position"][B] + np.float32([rad[ka] + rad[kb] + 0.02, 0.0, 0.0])          # 2 cm apart, along x
        t["position"][A][1] = np.float32(-10.0) + rad[ka]
        m["velocity"][A] = 0; m["angular_velocity"][A] = 0
        idle[:] = 0
        w.set_bodies(transforms=t, momentum=m, idle=idle)
    a.step(150); c.step(150)
    _same_world(a, c, "A beside B")
    c0 = a.counts()
    for w in (a, c):
        m = w.get_bodies()["momentum"]
        m["velocity"][A] = np.float32([-0.12, 0.0, 0.0])          # ... and rolling towards it at 1 mm per step
        m["angular_velocity"][A] = np.float32([0.0, 0.0, 0.12 / rad[ka]])
        w.set_bodies(momentum=m)
    done = 0
    for cp in (8, 16, 24, 32, 48, 64, 100):
        a.step(cp - done); c.step(cp - done); done = cp
        _same_world(a, c, f"A rolls into B, step {cp}")
    c1 = a.counts()
    print(f"\n[touching] fused {c1['fused_steps'] - c0['fused_steps']}, still {c1['still_steps'] - c0['still_steps']}, replays {c1['still_replays'] - c0['still_replays']}, general contacts now {c1['general_contacts']}")
    a.close(); c.close()
