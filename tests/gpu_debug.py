"""Ad-hoc GPU debugging run (not a test): python tests/gpu_debug.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nudge_amd import scenes as S
from nudge_amd import engine as E
from oracle.refworld import RefWorld
import parity_util as P


def compare_collide(scene, warm, flags=E.NH_FLAG_SYNC_COUNTS):
    ref = RefWorld(scene)
    ref.step(warm)
    b = ref.bodies()
    w = E.World(scene, flags=flags)
    w.set_bodies(b["transforms"], b["momentum"], b["idle"])
    ref.collide()
    w.collide()
    rc = ref.contacts()
    gc = w.get_contacts()
    od, ob, ok, of, _ = P.oracle_contacts_sorted(rc)
    cnt = w.counts()
    print(scene["name"], "warm", warm, "oracle K", rc["count"], "gpu", cnt)
    ok_tags = len(ok) == gc["count"] and np.array_equal(ok, gc["tags"]) and np.array_equal(of, gc["features"])
    print("   tags", ok_tags, "bodies", ok_tags and np.array_equal(ob, gc["bodies"]), "data bits", ok_tags and P.bits_equal(od, gc["data"]))
    if ok_tags and not P.bits_equal(od, gc["data"]):
        d = np.abs(od.view(np.float32).reshape(-1, 8) - gc["data"].view(np.float32).reshape(-1, 8))
        print("   max abs diff per field", d.max(axis=0))
    if not ok_tags:
        so = set(zip(ok.tolist(), of.tolist())); sg = set(zip(gc["tags"].tolist(), gc["features"].tolist()))
        print("   only oracle", len(so - sg), "only gpu", len(sg - so), "gpu sorted?", bool(np.all(np.diff(gc["tags"].astype(np.float64)) >= 0)))
    act = ref.active().astype(np.uint32)
    print("   active equal", np.array_equal(act, w.get_active()), len(act))
    return ref, w


def compare_steps(scene, warm, steps, flags):
    ref = RefWorld(scene)
    ref.step(warm)
    b = ref.bodies()
    w = E.World(scene, flags=flags)
    w.set_bodies(b["transforms"], b["momentum"], b["idle"])
    # NOTE: the oracle keeps its contact cache from the warm-up; start both from an empty cache instead
    ref2 = RefWorld(scene)
    ref2.set_bodies(b["transforms"], b["momentum"], b["idle"])
    for s in range(steps):
        ref2.step(1)
        w.step(1)
        rb, gb = ref2.bodies(), w.get_bodies()
        pe = P.bits_equal(rb["transforms"]["position"], gb["transforms"]["position"])
        re_ = P.bits_equal(rb["transforms"]["rotation"], gb["transforms"]["rotation"])
        ve = P.bits_equal(rb["momentum"]["velocity"], gb["momentum"]["velocity"])
        we = P.bits_equal(rb["momentum"]["angular_velocity"], gb["momentum"]["angular_velocity"])
        err = P.rel_err(gb["transforms"]["position"], rb["transforms"]["position"]).max()
        if s < 5 or s % 50 == 0 or s == steps - 1:
            print("   step", s, "pos", pe, "rot", re_, "vel", ve, "ang", we, "max rel", err, w.counts()["contacts"], ref2.contacts()["count"],
                  "general", w.counts()["general_contacts"], "levels", w.counts()["levels"])
    return ref2, w


if __name__ == "__main__":
    t0 = time.time()
    compare_collide(S.pile(256, 0), 300)
    compare_collide(S.pile(200, 100, seed=3), 300)
    compare_collide(S.grid_tiles(1, side=30), 70)
    compare_collide(S.grid_tiles(2, side=30, sphere_fraction=0.5), 0) if False else None
    compare_collide(S.ball_pit(8, 8, 8), 150)
    print("== steps, exact order, grid")
    compare_steps(S.grid_tiles(1, side=30), 40, 60, E.NH_FLAG_SYNC_COUNTS | E.NH_FLAG_EXACT_ORDER)
    print("== steps, closed-form order, grid")
    compare_steps(S.grid_tiles(1, side=30), 40, 60, E.NH_FLAG_SYNC_COUNTS)
    print("== steps, exact order, pile")
    compare_steps(S.pile(256, 0), 300, 30, E.NH_FLAG_SYNC_COUNTS | E.NH_FLAG_EXACT_ORDER)
    print("== steps, exact order, pit")
    compare_steps(S.ball_pit(8, 8, 8), 100, 30, E.NH_FLAG_SYNC_COUNTS | E.NH_FLAG_EXACT_ORDER)
    print("elapsed", time.time() - t0)
