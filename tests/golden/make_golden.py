"""Regenerates tests/golden/*.npz from the compiled reference (oracle/_ref, "exact" variant).

Run where /root/reference exists:   make -C oracle && python tests/golden/make_golden.py
Each fixture stores the reference's body state after N steps of a seeded scene (inputs are regenerated
from the seed by nudge_amd.scenes; a hash of the inputs is stored to detect generator drift), plus, for
the collide fixture, the reference's contact list.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from nudge_amd import scenes as S          # noqa: E402
from oracle.refworld import RefWorld       # noqa: E402

CASES = {
    # name: (builder, kwargs, steps)
    "pile256": (S.pile, dict(n_boxes=256, n_spheres=0, seed=1, iterations=4), 120),
    "pile_mixed": (S.pile, dict(n_boxes=160, n_spheres=96, seed=3, iterations=4), 400),
    "grid30": (S.grid_tiles, dict(n_tiles=1, side=30, seed=2), 450),
    "grid30_awake": (S.grid_tiles, dict(n_tiles=1, side=30, seed=2), 200),
    "mixed20": (S.grid_tiles, dict(n_tiles=1, side=20, sphere_fraction=0.5, seed=7), 200),
    "pit6": (S.ball_pit, dict(nx=6, ny=6, nz=6, seed=4), 150),
    "stacks": (S.stacks, dict(n_stacks=25, height=3, seed=5), 100),
    # compound bodies: two offset, rotated boxes + an offset sphere per body (local collider transforms, same-body pairs)
    "compound": (S.compound, dict(n_bodies=150, seed=6), 330),
}


def scene_hash(scene):
    h = hashlib.sha256()
    for k in ("body_transforms", "body_properties", "box_data", "box_transforms", "sphere_data", "sphere_transforms", "box_tags", "sphere_tags"):
        h.update(np.ascontiguousarray(scene[k]).tobytes())
    return h.hexdigest()


def build(name):
    fn, kw, steps = CASES[name]
    return fn(**kw), steps


def main():
    for name in (sys.argv[1:] or CASES):
        scene, steps = build(name)
        w = RefWorld(scene, variant="exact")
        w.step(steps)
        b = w.bodies()
        w.collide()
        c = w.contacts()
        order = np.argsort(c["tags"], kind="stable")
        np.savez_compressed(os.path.join(HERE, name + ".npz"),
                            input_hash=np.frombuffer(scene_hash(scene).encode(), dtype=np.uint8),
                            steps=np.int64(steps),
                            transforms=b["transforms"], momentum=b["momentum"], idle=b["idle"],
                            contact_tags=c["tags"][order], contact_data=c["data"][order], contact_bodies=c["bodies"][order],
                            active=w.active(), sleeping_pairs=c["sleeping_pairs"])
        print(name, steps, "contacts", c["count"], "active", len(w.active()), "sleeping pairs", len(c["sleeping_pairs"]))


if __name__ == "__main__":
    main()
