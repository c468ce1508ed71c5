"""Synthetic scene generators (host side, numpy) for the stepping engine.

The scenes are the configurations of BASELINE.json / SURVEY.md section 8(d): a pile dropping onto a
static ground box (config 1, the recipe of the reference's sample app, example/main.cpp:390-432),
single-layer grid-drop tiles (configs 2, 3, 5) and a sphere ball-pit (config 4).

All arrays use the *wide* record layouts of include/nudge_hip.h (32-bit body indices / collider
tags); `narrow_ok(scene)` tells whether a scene also fits the reference's 16-bit API (<= 8192
colliders, < 65535 bodies) so that the same arrays can be loaded into the compiled reference.

Randomness: a counter-based splitmix64 (identical on every host, no libc rand()):
    x_n = seed + (n+1)*0x9E3779B97F4A7C15 ; mix ; u01 = (z >> 40) * 2^-24
Every body draws a fixed block of 8 variates at counter 8*i .. 8*i+7 so generation vectorises.
"""
from __future__ import annotations

import numpy as np

TRANSFORM = np.dtype([("position", "<f4", 3), ("body", "<u4"), ("rotation", "<f4", 4)])  # nudge.h:34-38
PROPERTIES = np.dtype([("inertia_inverse", "<f4", 3), ("mass_inverse", "<f4")])             # nudge.h:40-43
MOMENTUM = np.dtype([("velocity", "<f4", 3), ("unused0", "<f4"),
                     ("angular_velocity", "<f4", 3), ("unused1", "<f4")])                 # nudge.h:45-50
BOX = np.dtype([("size", "<f4", 3), ("unused", "<f4")])                                      # nudge.h:56-59
SPHERE = np.dtype([("radius", "<f4")])                                                       # nudge.h:52-54
CONTACT = np.dtype([("position", "<f4", 3), ("penetration", "<f4"),
                    ("normal", "<f4", 3), ("friction", "<f4")])                            # nudge.h:61-66
IMPULSE = np.dtype([("impulse", "<f4", 3), ("unused", "<f4")])                               # nudge.h:113-116

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix_u01(seed: int, counters: np.ndarray) -> np.ndarray:
    """u01 float32 variates for the given counters (vectorised splitmix64)."""
    with np.errstate(over="ignore"):
        c = counters.astype(np.uint64)
        x = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (c + np.uint64(1)) * _GOLDEN
        z = (x ^ (x >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32)


def _draws(seed: int, n: int, k: int = 8) -> np.ndarray:
    idx = np.arange(n * k, dtype=np.uint64)
    return splitmix_u01(seed, idx).reshape(n, k)


def _identity_transforms(n: int) -> np.ndarray:
    t = np.zeros(n, dtype=TRANSFORM)
    t["rotation"][:, 3] = 1.0
    return t


def _box_properties(sx, sy, sz):
    """mass = 8*sx*sy*sz and inverse inertia exactly as the sample app's add_box (example/main.cpp:112-145)."""
    f = np.float32
    mass = (f(8.0) * sx) * sy * sz
    k = mass * (f(1.0) / f(3.0))
    kcx2 = k * sx * sx
    kcy2 = k * sy * sy
    kcz2 = k * sz * sz
    p = np.zeros(len(sx), dtype=PROPERTIES)
    p["mass_inverse"] = f(1.0) / mass
    p["inertia_inverse"][:, 0] = f(1.0) / (kcy2 + kcz2)
    p["inertia_inverse"][:, 1] = f(1.0) / (kcx2 + kcz2)
    p["inertia_inverse"][:, 2] = f(1.0) / (kcx2 + kcy2)
    return p


def _sphere_properties(r):
    """mass = 4.18879*r^3, inverse inertia 2.5/(m r^2) as add_sphere (example/main.cpp:147-174)."""
    f = np.float32
    mass = f(4.18879) * r * r * r
    k = f(2.5) / (mass * r * r)
    p = np.zeros(len(r), dtype=PROPERTIES)
    p["mass_inverse"] = f(1.0) / mass
    p["inertia_inverse"][:, 0] = k
    p["inertia_inverse"][:, 1] = k
    p["inertia_inverse"][:, 2] = k
    return p


def _assemble(static_boxes, dyn_boxes, dyn_spheres, params, tile_of_body=None, tile_of_static=None, name=""):
    """static_boxes: (transforms, sizes) on body 0; dyn_boxes: (body_transforms, sizes, props);
    dyn_spheres: (body_transforms, radii, props).  Bodies: 0 = static world, then boxes, then spheres
    interleaved if `order` is given.  Collider tags = global collider index (boxes first, spheres after)."""
    st, ssz = static_boxes
    bt, bsz, bp = dyn_boxes
    spt, sr, sp = dyn_spheres
    ns, nb, nsp = len(st), len(bt), len(spt)
    n_bodies = 1 + nb + nsp

    body_transforms = _identity_transforms(n_bodies)
    body_transforms[1:1 + nb] = bt
    body_transforms[1 + nb:] = spt
    props = np.zeros(n_bodies, dtype=PROPERTIES)
    props[1:1 + nb] = bp
    props[1 + nb:] = sp

    box_transforms = _identity_transforms(ns + nb)
    box_transforms[:ns] = st
    box_transforms["body"][:ns] = 0
    box_transforms["body"][ns:] = np.arange(1, 1 + nb, dtype=np.uint32)
    box_data = np.zeros(ns + nb, dtype=BOX)
    box_data["size"][:ns] = ssz
    box_data["size"][ns:] = bsz
    box_tags = np.arange(ns + nb, dtype=np.uint32)

    sphere_transforms = _identity_transforms(nsp)
    sphere_transforms["body"] = np.arange(1 + nb, 1 + nb + nsp, dtype=np.uint32)
    sphere_data = np.zeros(nsp, dtype=SPHERE)
    sphere_data["radius"] = sr
    sphere_tags = np.arange(ns + nb, ns + nb + nsp, dtype=np.uint32)

    scene = dict(
        name=name,
        body_transforms=body_transforms,
        body_properties=props,
        body_momentum=np.zeros(n_bodies, dtype=MOMENTUM),
        idle_counters=np.zeros(n_bodies, dtype=np.uint8),
        box_tags=box_tags, box_data=box_data, box_transforms=box_transforms,
        sphere_tags=sphere_tags, sphere_data=sphere_data, sphere_transforms=sphere_transforms,
        params=dict(params),
    )
    if tile_of_body is not None:
        scene["tile_of_body"] = tile_of_body
        scene["tile_of_static"] = tile_of_static
    return scene


DEFAULT_PARAMS = dict(time_step=1.0 / 120.0, gravity=9.82, damping_rate=0.25, iterations=8)


def pile(n_boxes=256, n_spheres=0, seed=1, iterations=4):
    """Config 1: the sample app's recipe (example/main.cpp:398-431) with our PRNG: ground half-extents
    (400,10,400) at y=-20 on body 0; boxes U[0.5,1.5)^3 / spheres r U[0.5,1.5), x,z U[-5,5), y U[0,300)."""
    f = np.float32
    d = _draws(0x9E3779B97F4A7C15 ^ seed, n_boxes + n_spheres)
    db, ds = d[:n_boxes], d[n_boxes:]
    st = _identity_transforms(1)
    st["position"][0] = (0.0, -20.0, 0.0)
    ssz = np.array([[400.0, 10.0, 400.0]], dtype=np.float32)

    sx, sy, sz = db[:, 0] + f(0.5), db[:, 1] + f(0.5), db[:, 2] + f(0.5)
    bt = _identity_transforms(n_boxes)
    bt["position"][:, 0] = db[:, 3] * f(10.0) - f(5.0)
    bt["position"][:, 1] = db[:, 4] * f(300.0)
    bt["position"][:, 2] = db[:, 5] * f(10.0) - f(5.0)
    bsz = np.stack([sx, sy, sz], axis=1)
    bp = _box_properties(sx, sy, sz)

    r = ds[:, 0] + f(0.5)
    spt = _identity_transforms(n_spheres)
    spt["position"][:, 0] = ds[:, 3] * f(10.0) - f(5.0)
    spt["position"][:, 1] = ds[:, 4] * f(300.0)
    spt["position"][:, 2] = ds[:, 5] * f(10.0) - f(5.0)
    sp = _sphere_properties(r)

    params = dict(DEFAULT_PARAMS, iterations=iterations)
    return _assemble((st, ssz), (bt, bsz, bp), (spt, r, sp), params, name=f"pile{n_boxes}+{n_spheres}")


def bridge(layers=2, iterations=8):
    """Bodies held from both sides of the plane x = 0: per layer two unit cubes at x = -1.25 / +1.25 and a plank (half extents 1.6, 0.5, 1) lying across them with its
    centre at x = 0.1, layer on layer, starting at rest on the ground.  Cut at x = 0 every plank has one support on each side: the test scene of contact ownership
    across a cut (nudge_amd/partition.py: single_owner)."""
    f = np.float32
    st = _identity_transforms(1)
    st["position"][0] = (0.0, -20.0, 0.0)
    ssz = np.array([[400.0, 10.0, 400.0]], dtype=f)
    pos, size, y = [], [], -9.0
    for _ in range(layers):
        pos += [(-1.25, y, 0.0), (1.25, y, 0.0), (0.1, y + 1.5, 0.0)]
        size += [(1.0, 1.0, 1.0), (1.0, 1.0, 1.0), (1.6, 0.5, 1.0)]
        y += 3.0
    n = len(pos)
    bt = _identity_transforms(n)
    bt["position"][:] = np.array(pos, dtype=f)
    bsz = np.array(size, dtype=f)
    bp = _box_properties(bsz[:, 0].copy(), bsz[:, 1].copy(), bsz[:, 2].copy())
    params = dict(DEFAULT_PARAMS, iterations=iterations)
    return _assemble((st, ssz), (bt, bsz, bp), (_identity_transforms(0), np.zeros(0, f), _sphere_properties(np.zeros(0, f))), params, name=f"bridge{layers}")


def grid_tiles(n_tiles=1, side=90, sphere_fraction=0.0, seed=2, iterations=8, lattice_cols=None):
    """Configs 2/3/5: `n_tiles` independent single-layer drop tiles.  Per tile one static ground slab
    (half-extents (side*1.5+2, 10, side*1.5+2), top at y=-10) and side*side bodies on a 3.0 pitch:
    boxes with half-extents U[0.5,1.0)^3, yaw U[0,0.5) rad about y, x/z jitter U[0,0.2), bottom face
    U[0.5,1.0) above the slab; with `sphere_fraction`=0.5 odd cells are spheres r U[0.5,1.0).
    Tiles sit on a lattice with a 4.0 gap between slabs.  Tile t's bodies are contiguous."""
    f = np.float32
    per = side * side
    n = n_tiles * per
    if lattice_cols is None:
        lattice_cols = int(np.ceil(np.sqrt(n_tiles)))
    half = f(side * 1.5 + 2.0)
    pitch_tile = f(2.0) * half + f(4.0)

    d = _draws(0x9E3779B97F4A7C15 ^ seed, n)
    tile = np.repeat(np.arange(n_tiles), per)
    cell = np.tile(np.arange(per), n_tiles)
    ix = (cell % side).astype(np.float32)
    iz = (cell // side).astype(np.float32)
    ox = (tile % lattice_cols).astype(np.float32) * pitch_tile
    oz = (tile // lattice_cols).astype(np.float32) * pitch_tile

    is_sphere = np.zeros(n, dtype=bool)
    if sphere_fraction > 0.0:
        is_sphere = ((cell % side + cell // side) % 2) == 1

    sx, sy, sz = d[:, 0] * f(0.5) + f(0.5), d[:, 1] * f(0.5) + f(0.5), d[:, 2] * f(0.5) + f(0.5)
    yaw = d[:, 3] * f(0.5)
    px = ox + (ix - f((side - 1) * 0.5)) * f(3.0) + d[:, 4] * f(0.2)
    pz = oz + (iz - f((side - 1) * 0.5)) * f(3.0) + d[:, 5] * f(0.2)
    drop = d[:, 6] * f(0.5) + f(0.5)
    radius = d[:, 0] * f(0.5) + f(0.5)

    pos = np.zeros((n, 3), dtype=np.float32)
    pos[:, 0] = px
    pos[:, 2] = pz
    pos[:, 1] = np.where(is_sphere, f(-10.0) + drop + radius, f(-10.0) + drop + sy)
    rot = np.zeros((n, 4), dtype=np.float32)
    h = (yaw * f(0.5)).astype(np.float32)
    # sin/cos in float64 then rounded: independent of numpy's float32 SIMD kernels
    rot[:, 1] = np.where(is_sphere, f(0.0), np.sin(h.astype(np.float64)).astype(np.float32))
    rot[:, 3] = np.where(is_sphere, f(1.0), np.cos(h.astype(np.float64)).astype(np.float32))
    # normalise like the reference's tests do (tests/main.cpp:106) so that |q| = 1 to fp32 rounding
    nrm = (f(1.0) / np.sqrt(rot[:, 1] * rot[:, 1] + rot[:, 3] * rot[:, 3])).astype(np.float32)
    rot[:, 1] *= nrm
    rot[:, 3] *= nrm

    bsel, ssel = ~is_sphere, is_sphere
    bt = _identity_transforms(int(bsel.sum()))
    bt["position"] = pos[bsel]
    bt["rotation"] = rot[bsel]
    bsz = np.stack([sx[bsel], sy[bsel], sz[bsel]], axis=1)
    bp = _box_properties(sx[bsel], sy[bsel], sz[bsel])
    spt = _identity_transforms(int(ssel.sum()))
    spt["position"] = pos[ssel]
    sr = radius[ssel]
    sp = _sphere_properties(sr)

    st = _identity_transforms(n_tiles)
    tt = np.arange(n_tiles)
    st["position"][:, 0] = (tt % lattice_cols).astype(np.float32) * pitch_tile
    st["position"][:, 1] = -20.0
    st["position"][:, 2] = (tt // lattice_cols).astype(np.float32) * pitch_tile
    ssz = np.tile(np.array([[half, 10.0, half]], dtype=np.float32), (n_tiles, 1))

    tile_of_body = np.concatenate([[-1], tile[bsel], tile[ssel]]).astype(np.int32)
    params = dict(DEFAULT_PARAMS, iterations=iterations)
    return _assemble((st, ssz), (bt, bsz, bp), (spt, sr, sp), params,
                     tile_of_body=tile_of_body, tile_of_static=tt.astype(np.int32),
                     name=f"grid{n_tiles}x{side}x{side}" + ("mixed" if sphere_fraction > 0 else ""))


def ball_pit(nx=20, ny=20, nz=20, seed=4, iterations=16):
    """Config 4: spheres r=0.5 on a jittered cubic lattice (pitch 1.1, jitter U[0,0.05)) dropped from rest
    into a container of 5 static boxes (floor + 4 walls, thickness 10) on body 0."""
    f = np.float32
    n = nx * ny * nz
    d = _draws(0x9E3779B97F4A7C15 ^ seed, n)
    idx = np.arange(n)
    ix, iy, iz = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    spt = _identity_transforms(n)
    spt["position"][:, 0] = (ix.astype(np.float32) - f((nx - 1) * 0.5)) * f(1.1) + d[:, 0] * f(0.05)
    spt["position"][:, 1] = f(0.6) + iy.astype(np.float32) * f(1.1) + d[:, 1] * f(0.05)
    spt["position"][:, 2] = (iz.astype(np.float32) - f((nz - 1) * 0.5)) * f(1.1) + d[:, 2] * f(0.05)
    r = np.full(n, 0.5, dtype=np.float32)
    sp = _sphere_properties(r)

    wx, wz = f(nx * 1.1 * 0.5 + 1.0), f(nz * 1.1 * 0.5 + 1.0)
    hy = f(ny * 1.1 + 10.0)
    st = _identity_transforms(5)
    ssz = np.zeros((5, 3), dtype=np.float32)
    st["position"][0] = (0.0, -10.0, 0.0); ssz[0] = (wx + 20.0, 10.0, wz + 20.0)       # floor
    st["position"][1] = (-(wx + 10.0), hy * 0.5, 0.0); ssz[1] = (10.0, hy, wz + 20.0)  # -x wall
    st["position"][2] = ((wx + 10.0), hy * 0.5, 0.0); ssz[2] = (10.0, hy, wz + 20.0)   # +x wall
    st["position"][3] = (0.0, hy * 0.5, -(wz + 10.0)); ssz[3] = (wx, hy, 10.0)         # -z wall
    st["position"][4] = (0.0, hy * 0.5, (wz + 10.0)); ssz[4] = (wx, hy, 10.0)          # +z wall

    empty_t = _identity_transforms(0)
    params = dict(DEFAULT_PARAMS, iterations=iterations)
    return _assemble((st, ssz), (empty_t, np.zeros((0, 3), np.float32), np.zeros(0, PROPERTIES)),
                     (spt, r, sp), params, name=f"pit{nx}x{ny}x{nz}")


def stacks(n_stacks=64, height=3, seed=5, iterations=8):
    """Small-island scene for solver tests: `n_stacks` separated stacks of `height` boxes each."""
    f = np.float32
    n = n_stacks * height
    d = _draws(0x9E3779B97F4A7C15 ^ seed, n)
    st = _identity_transforms(1)
    st["position"][0] = (0.0, -20.0, 0.0)
    ssz = np.array([[400.0, 10.0, 400.0]], dtype=np.float32)
    s = np.arange(n) // height
    lvl = np.arange(n) % height
    side = int(np.ceil(np.sqrt(n_stacks)))
    sx, sy, sz = d[:, 0] * f(0.3) + f(0.7), np.full(n, 0.5, np.float32), d[:, 2] * f(0.3) + f(0.7)
    bt = _identity_transforms(n)
    bt["position"][:, 0] = (s % side).astype(np.float32) * f(4.0) + d[:, 3] * f(0.1)
    bt["position"][:, 1] = f(-10.0) + f(0.6) + lvl.astype(np.float32) * f(1.2)
    bt["position"][:, 2] = (s // side).astype(np.float32) * f(4.0) + d[:, 4] * f(0.1)
    bsz = np.stack([sx, sy, sz], axis=1)
    bp = _box_properties(sx, sy, sz)
    empty_t = _identity_transforms(0)
    params = dict(DEFAULT_PARAMS, iterations=iterations)
    return _assemble((st, ssz), (bt, bsz, bp), (empty_t, np.zeros(0, np.float32), np.zeros(0, PROPERTIES)),
                     params, name=f"stacks{n_stacks}x{height}")


def compound(n_bodies=150, seed=6, iterations=8):
    """Compound bodies (nudge.h:34-38: a collider transform is LOCAL to its body, `body` names the owner; composed at
    nudge.cpp:3021-3079): every dynamic body carries two rotated, offset boxes and an offset sphere, dropped in two layers so
    that they also land on each other.  Pairs of colliders of one body never collide (nudge.cpp:3683)."""
    f = np.float32
    n = n_bodies
    d = _draws(0x9E3779B97F4A7C15 ^ seed, n)
    st = _identity_transforms(1)
    st["position"][0] = (0.0, -20.0, 0.0)
    ssz = np.array([[400.0, 10.0, 400.0]], dtype=np.float32)
    side = int(np.ceil(np.sqrt((n + 1) // 2)))
    cell = np.arange(n) // 2
    layer = np.arange(n) % 2
    yaw = d[:, 5] * f(3.0)
    bt = _identity_transforms(n)
    bt["position"][:, 0] = (cell % side).astype(np.float32) * f(2.6) + d[:, 3] * f(0.4)
    bt["position"][:, 1] = f(-10.0) + f(1.0) + d[:, 4] * f(0.5) + layer.astype(np.float32) * f(2.2)
    bt["position"][:, 2] = (cell // side).astype(np.float32) * f(2.6) + d[:, 6] * f(0.4)
    bt["rotation"][:, 1] = np.sin(yaw * f(0.5))
    bt["rotation"][:, 3] = np.cos(yaw * f(0.5))
    bp = _box_properties(np.full(n, 1.1, np.float32), np.full(n, 0.5, np.float32), np.full(n, 0.4, np.float32))
    empty_t = _identity_transforms(0)
    params = dict(DEFAULT_PARAMS, iterations=iterations)
    scene = _assemble((st, ssz), (bt, np.zeros((n, 3), np.float32), bp), (empty_t, np.zeros(0, np.float32), np.zeros(0, PROPERTIES)),
                      params, name=f"compound{n}")
    owner = np.arange(1, n + 1, dtype=np.uint32)
    # two boxes per body: A at (-0.6, 0, 0) turned 0.3 rad about z, B at (+0.6, 0.1, 0) turned 0.5 rad about y
    box_t = _identity_transforms(1 + 2 * n)
    box_t[0] = st[0]
    box_t["body"][1::2] = owner
    box_t["body"][2::2] = owner
    box_t["position"][1::2] = (-0.6, 0.0, 0.0)
    box_t["rotation"][1::2] = (0.0, 0.0, np.sin(f(0.15)), np.cos(f(0.15)))
    box_t["position"][2::2] = (0.6, 0.1, 0.0)
    box_t["rotation"][2::2] = (0.0, np.sin(f(0.25)), 0.0, np.cos(f(0.25)))
    box_d = np.zeros(1 + 2 * n, dtype=BOX)
    box_d["size"][0] = ssz[0]
    box_d["size"][1::2] = np.stack([d[:, 0] * f(0.2) + f(0.4), np.full(n, 0.3, np.float32), np.full(n, 0.3, np.float32)], axis=1)
    box_d["size"][2::2] = np.stack([np.full(n, 0.3, np.float32), d[:, 1] * f(0.2) + f(0.4), np.full(n, 0.3, np.float32)], axis=1)
    sph_t = _identity_transforms(n)
    sph_t["body"] = owner
    sph_t["position"][:] = (0.0, 0.6, 0.2)
    sph_d = np.zeros(n, dtype=SPHERE)
    sph_d["radius"] = d[:, 2] * f(0.1) + f(0.3)
    scene.update(box_tags=np.arange(1 + 2 * n, dtype=np.uint32), box_data=box_d, box_transforms=box_t,
                 sphere_tags=np.arange(1 + 2 * n, 1 + 3 * n, dtype=np.uint32), sphere_data=sph_d, sphere_transforms=sph_t)
    return scene


def tray(nx=17, nz=18, seed=8, iterations=8):
    """A dynamic HUB body: one wide tray box resting on the ground carries nx*nz small boxes, so a single dynamic body sits in
    4*nx*nz + 4 contacts (> 1000 for the default).  Every contact on the tray conflicts with every other one in the reference's
    batch scheduler (nudge.cpp:4206-4339): no batch ever fills, the open list grows with the contact count."""
    f = np.float32
    n = nx * nz
    d = _draws(0x9E3779B97F4A7C15 ^ seed, n + 1)
    st = _identity_transforms(1)
    st["position"][0] = (0.0, -20.0, 0.0)
    ssz = np.array([[400.0, 10.0, 400.0]], dtype=np.float32)
    hx, hz = f(nx * 1.1 + 1.0), f(nz * 1.1 + 1.0)
    bt = _identity_transforms(n + 1)
    bsz = np.zeros((n + 1, 3), dtype=np.float32)
    bt["position"][0] = (0.0, -10.0 + 0.5 + 0.01, 0.0)
    bsz[0] = (hx, 0.5, hz)
    i = np.arange(n)
    sx = d[1:, 0] * f(0.2) + f(0.4)
    bt["position"][1:, 0] = ((i % nx).astype(np.float32) - f((nx - 1) * 0.5)) * f(2.2) + d[1:, 1] * f(0.1)
    bt["position"][1:, 1] = f(-10.0 + 1.0 + 0.02) + sx
    bt["position"][1:, 2] = ((i // nx).astype(np.float32) - f((nz - 1) * 0.5)) * f(2.2) + d[1:, 2] * f(0.1)
    bsz[1:] = np.stack([sx, sx, sx], axis=1)
    bp = _box_properties(bsz[:, 0].copy(), bsz[:, 1].copy(), bsz[:, 2].copy())
    empty_t = _identity_transforms(0)
    params = dict(DEFAULT_PARAMS, iterations=iterations)
    return _assemble((st, ssz), (bt, bsz, bp), (empty_t, np.zeros(0, np.float32), np.zeros(0, PROPERTIES)), params, name=f"tray{nx}x{nz}")


def narrow_ok(scene) -> bool:
    """True if the scene fits the reference's 16-bit API limits (nudge.cpp:3010, nudge.h:69-70)."""
    c = len(scene["box_tags"]) + len(scene["sphere_tags"])
    return c <= 8192 and len(scene["body_transforms"]) < 65535


def extract_tile(scene, t):
    """Sub-scene holding only tile `t` (its slab on body 0 + its bodies) at the SAME world coordinates,
    for the tiled oracle.  Returns (sub_scene, body_index_map) where body_index_map[j] is the
    union-world body index of sub-scene body j."""
    tob = scene["tile_of_body"]
    body_sel = np.flatnonzero(tob == t)
    body_map = np.concatenate([[0], body_sel]).astype(np.int64)
    remap = np.full(len(tob), -1, dtype=np.int64)
    remap[body_map] = np.arange(len(body_map))

    bb = scene["box_transforms"]["body"]
    n_static = len(scene["tile_of_static"])
    box_sel = np.concatenate([np.flatnonzero(scene["tile_of_static"] == t),
                              n_static + np.flatnonzero(remap[bb[n_static:]] > 0)])
    sb = scene["sphere_transforms"]["body"]
    sph_sel = np.flatnonzero(remap[sb] > 0)

    sub = dict(name=f"{scene['name']}/tile{t}", params=dict(scene["params"]))
    for k in ("body_transforms", "body_properties", "body_momentum", "idle_counters"):
        sub[k] = scene[k][body_map].copy()
    sub["box_data"] = scene["box_data"][box_sel].copy()
    sub["box_transforms"] = scene["box_transforms"][box_sel].copy()
    sub["box_transforms"]["body"] = remap[sub["box_transforms"]["body"]].astype(np.uint32)
    sub["sphere_data"] = scene["sphere_data"][sph_sel].copy()
    sub["sphere_transforms"] = scene["sphere_transforms"][sph_sel].copy()
    sub["sphere_transforms"]["body"] = remap[sub["sphere_transforms"]["body"]].astype(np.uint32)
    nbx = len(box_sel)
    sub["box_tags"] = np.arange(nbx, dtype=np.uint32)
    sub["sphere_tags"] = np.arange(nbx, nbx + len(sph_sel), dtype=np.uint32)
    return sub, body_map
