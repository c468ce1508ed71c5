"""oracle/refworld.py -- TEST INFRASTRUCTURE: ctypes binding of the compiled reference (oracle/_ref).

`RefWorld(scene, variant="exact")` loads a scene (nudge_amd.scenes layout, narrowed to the reference's
16-bit indices) into the unmodified reference engine and replays the reference's own step loop
(example/main.cpp:274-328) stage by stage.  Used as the parity checker by tests/ and smoke(), and as
the `cpu_baseline` of bench.py.  Never used by the product path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


class _View(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "body_transforms", "body_properties", "body_momentum", "idle_counters",
        "box_tags", "box_data", "box_transforms", "sphere_tags", "sphere_data", "sphere_transforms",
        "contact_data", "contact_bodies", "contact_tags", "sleeping_pairs", "cache_tags", "cache_data",
        "active_indices", "connections")] + [(n, C.c_uint32) for n in (
        "body_count", "box_count", "sphere_count", "contact_count", "sleeping_count", "cache_count",
        "active_count", "connection_count", "max_bodies", "max_boxes", "max_spheres", "max_contacts")]


def available(variant="exact") -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", f"libnudge_ref_{variant}.so"))


def lib(variant="exact"):
    if variant in _LIBS:
        return _LIBS[variant]
    path = os.path.join(_HERE, "_ref", f"libnudge_ref_{variant}.so")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} missing: run `make -C oracle` where /root/reference exists")
    L = C.CDLL(path)
    L.ref_world_create.restype = C.c_void_p
    L.ref_world_create.argtypes = [C.c_uint32] * 4 + [C.c_uint64]
    L.ref_world_destroy.argtypes = [C.c_void_p]
    L.ref_world_set_counts.argtypes = [C.c_void_p] + [C.c_uint32] * 4
    L.ref_world_set_cache_count.argtypes = [C.c_void_p, C.c_uint32]
    if hasattr(L, "ref_world_set_contact_count"):
        L.ref_world_set_contact_count.argtypes = [C.c_void_p, C.c_uint32]
    L.ref_world_view.argtypes = [C.c_void_p, C.POINTER(_View)]
    L.ref_set_ftz.argtypes = [C.c_int]
    L.ref_variant.restype = C.c_char_p
    L.ref_simd_width.restype = C.c_int
    for name in ("ref_stage_collide", "ref_stage_read_cache", "ref_stage_setup", "ref_stage_update",
                 "ref_stage_write_cache"):
        getattr(L, name).argtypes = [C.c_void_p]
    L.ref_stage_gravity.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    L.ref_stage_apply.argtypes = [C.c_void_p, C.c_uint32]
    L.ref_stage_advance.argtypes = [C.c_void_p, C.c_float]
    L.ref_constraint_batches.argtypes = [C.c_void_p]
    L.ref_constraint_batches.restype = C.c_uint32
    L.ref_culled_count.argtypes = [C.c_void_p]
    L.ref_culled_count.restype = C.c_uint32
    for name in ("ref_constraint_to_contact", "ref_sorted_contacts", "ref_contact_impulses"):
        getattr(L, name).argtypes = [C.c_void_p]
        getattr(L, name).restype = C.c_void_p
    L.ref_world_step.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float]
    L.ref_world_sum_contacts.argtypes = [C.c_void_p]
    L.ref_world_sum_contacts.restype = C.c_uint64
    L.ref_tiles_step.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_int]
    L.ref_tiles_step.restype = C.c_double
    _LIBS[variant] = L
    return L


def _view_array(ptr, dtype, count):
    if count == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    nbytes = np.dtype(dtype).itemsize * count
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


class RefWorld:
    """One world inside the compiled reference.  All getters return COPIES unless stated."""

    def __init__(self, scene, variant="exact", max_contacts=None, arena_bytes=None, ftz=False):
        from nudge_amd import scenes as S
        if not S.narrow_ok(scene):
            raise ValueError("scene exceeds the reference's limits (<= 8192 colliders, < 65535 bodies)")
        self.S = S
        self.L = lib(variant)
        self.variant = variant
        self.L.ref_set_ftz(1 if ftz else 0)
        nb = len(scene["body_transforms"])
        nbox, nsph = len(scene["box_tags"]), len(scene["sphere_tags"])
        if max_contacts is None:
            max_contacts = max(4096, 16 * nb)
        if arena_bytes is None:
            # worst temp of collide = coarse_count^2 u32 (nudge.cpp:3276) + scheduler buckets (4222-4223)
            arena_bytes = (64 << 20) + max_contacts * 1400
        self.params = dict(scene["params"])
        self.h = self.L.ref_world_create(nb, max(nbox, 1), max(nsph, 1), max_contacts, arena_bytes)
        self.max_contacts = max_contacts
        v = self._view()
        _view_array(v.body_transforms, S.TRANSFORM, nb)[:] = scene["body_transforms"]
        _view_array(v.body_properties, S.PROPERTIES, nb)[:] = scene["body_properties"]
        _view_array(v.body_momentum, S.MOMENTUM, nb)[:] = scene["body_momentum"]
        _view_array(v.idle_counters, np.uint8, nb)[:] = scene["idle_counters"]
        if nbox:
            _view_array(v.box_tags, np.uint16, nbox)[:] = scene["box_tags"].astype(np.uint16)
            _view_array(v.box_data, S.BOX, nbox)[:] = scene["box_data"]
            _view_array(v.box_transforms, S.TRANSFORM, nbox)[:] = scene["box_transforms"]
        if nsph:
            _view_array(v.sphere_tags, np.uint16, nsph)[:] = scene["sphere_tags"].astype(np.uint16)
            _view_array(v.sphere_data, S.SPHERE, nsph)[:] = scene["sphere_data"]
            _view_array(v.sphere_transforms, S.TRANSFORM, nsph)[:] = scene["sphere_transforms"]
        ncon = 0
        if "connections" in scene and len(scene["connections"]):
            con = np.asarray(scene["connections"], dtype=np.uint16).reshape(-1, 2)
            ncon = len(con)
            _view_array(v.connections, np.uint16, 2 * ncon)[:] = con.reshape(-1)
        self.L.ref_world_set_counts(self.h, nb, nbox, nsph, ncon)
        self.n_bodies, self.n_boxes, self.n_spheres = nb, nbox, nsph

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.ref_world_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _view(self):
        v = _View()
        self.L.ref_world_view(self.h, C.byref(v))
        return v

    # ---- stepping -------------------------------------------------------------------------------
    def step(self, steps=1, iterations=None, time_step=None, gravity=None, damping_rate=None):
        p = self.params
        self.L.ref_world_step(self.h, steps, p["iterations"] if iterations is None else iterations,
                              p["time_step"] if time_step is None else time_step,
                              p["gravity"] if gravity is None else gravity,
                              p["damping_rate"] if damping_rate is None else damping_rate)

    def collide(self): self.L.ref_stage_collide(self.h)
    def gravity(self):
        p = self.params
        self.L.ref_stage_gravity(self.h, p["time_step"], p["gravity"], p["damping_rate"])
    def keep_owned_contacts(self, first_ghost):
        """Between collide() and read_cache(): drops the contacts a rank of a partitioned world does not own (include/nudge_hip.h: nh_set_first_ghost_body -- a contact
        with the static world belongs to its dynamic body, a contact of two dynamic bodies to the body whose collider has the larger tag; bodies >= first_ghost are ghosts).
        The list stays in the reference's order."""
        v = self._view()
        n = v.contact_count
        if n == 0:
            return 0
        ab = _view_array(v.contact_bodies, np.uint16, 2 * n).reshape(-1, 2)
        tags = _view_array(v.contact_tags, np.uint64, n)
        data = _view_array(v.contact_data, self.S.CONTACT, n)
        a, b = ab[:, 0].astype(np.int64), ab[:, 1].astype(np.int64)
        ta, tb = (tags >> np.uint64(32)) & np.uint64(0xffff), (tags >> np.uint64(48)) & np.uint64(0xffff)
        owner = np.where(a == 0, b, np.where(b == 0, a, np.where(ta > tb, a, b)))
        keep = np.flatnonzero(owner < first_ghost)
        m = len(keep)
        if m < n:
            ab[:m] = ab[keep].copy(); tags[:m] = tags[keep].copy(); data[:m] = data[keep].copy()
            self.L.ref_world_set_contact_count(self.h, m)
        return n - m

    def read_cache(self): self.L.ref_stage_read_cache(self.h)
    def setup(self): self.L.ref_stage_setup(self.h)
    def apply(self, iterations=None):
        self.L.ref_stage_apply(self.h, self.params["iterations"] if iterations is None else iterations)
    def update(self): self.L.ref_stage_update(self.h)
    def write_cache(self): self.L.ref_stage_write_cache(self.h)
    def advance(self): self.L.ref_stage_advance(self.h, self.params["time_step"])

    # ---- state ----------------------------------------------------------------------------------
    def set_bodies(self, transforms=None, momentum=None, idle=None):
        v = self._view()
        if transforms is not None:
            _view_array(v.body_transforms, self.S.TRANSFORM, self.n_bodies)[:] = transforms
        if momentum is not None:
            _view_array(v.body_momentum, self.S.MOMENTUM, self.n_bodies)[:] = momentum
        if idle is not None:
            _view_array(v.idle_counters, np.uint8, self.n_bodies)[:] = idle

    def bodies(self):
        v = self._view()
        return dict(
            transforms=_view_array(v.body_transforms, self.S.TRANSFORM, self.n_bodies).copy(),
            momentum=_view_array(v.body_momentum, self.S.MOMENTUM, self.n_bodies).copy(),
            idle=_view_array(v.idle_counters, np.uint8, self.n_bodies).copy())

    def contacts(self):
        v = self._view()
        n = v.contact_count
        ab = _view_array(v.contact_bodies, np.uint16, 2 * n).reshape(-1, 2).copy()
        return dict(
            count=n,
            data=_view_array(v.contact_data, self.S.CONTACT, n).copy(),
            bodies=ab,
            tags=_view_array(v.contact_tags, np.uint64, n).copy(),
            sleeping_pairs=_view_array(v.sleeping_pairs, np.uint32, v.sleeping_count).copy())

    def active(self):
        v = self._view()
        return _view_array(v.active_indices, np.uint16, v.active_count).copy()

    def cache(self):
        v = self._view()
        n = v.cache_count
        return dict(count=n, tags=_view_array(v.cache_tags, np.uint64, n).copy(),
                    data=_view_array(v.cache_data, self.S.IMPULSE, n).copy())

    def snapshot(self):
        """Everything that persists from one step to the next (bodies + contact cache), copied to the host side of the caller."""
        v = self._view()
        n = v.cache_count
        return dict(bodies=self.bodies(), cache_count=n, cache_tags=_view_array(v.cache_tags, np.uint64, n).copy(),
                    cache_data=_view_array(v.cache_data, self.S.IMPULSE, n).copy())

    def restore(self, snap):
        b = snap["bodies"]
        self.set_bodies(b["transforms"], b["momentum"], b["idle"])
        v = self._view()
        n = snap["cache_count"]
        if n:
            _view_array(v.cache_tags, np.uint64, n)[:] = snap["cache_tags"]
            _view_array(v.cache_data, self.S.IMPULSE, n)[:] = snap["cache_data"]
        self.L.ref_world_set_cache_count(self.h, n)

    def solver_order(self):
        """constraint_to_contact (batches x 8 contact indices) of the last setup (nudge.cpp:4203, 4354)."""
        nb = self.L.ref_constraint_batches(self.h)
        w = self.L.ref_simd_width()
        return _view_array(self.L.ref_constraint_to_contact(self.h), np.uint32, nb * w).reshape(nb, w).copy()

    def sorted_contacts(self):
        n = self._view().contact_count
        return _view_array(self.L.ref_sorted_contacts(self.h), np.uint32, n).copy()

    def contact_impulses(self):
        n = self._view().contact_count
        return _view_array(self.L.ref_contact_impulses(self.h), self.S.IMPULSE, n).copy()

    def sum_contacts(self):
        return int(self.L.ref_world_sum_contacts(self.h))


def step_tiles(worlds, steps, threads=1, ftz=True):
    """Steps independent tile worlds on `threads` host threads (tiled oracle / CPU baseline).  Returns seconds."""
    if not worlds:
        return 0.0
    L = worlds[0].L
    p = worlds[0].params
    arr = (C.c_void_p * len(worlds))(*[w.h for w in worlds])
    return float(L.ref_tiles_step(arr, len(worlds), steps, p["iterations"], p["time_step"], p["gravity"],
                                  p["damping_rate"], threads, 1 if ftz else 0))
