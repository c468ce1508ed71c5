"""Test-only engine for nudge_amd.partition on the CPU: the compiled reference (oracle/_ref) behind the four
members `Partition` needs (records, set_counts, step, params).  Arrays are kept in the WIDE layouts of
include/nudge_hip.h as CPU torch tensors and mirrored into a reference world (16-bit tags) around every step, so the
partition / halo logic that runs on the GPU in production is exercised here with gloo and no GPU."""
import numpy as np
import torch

from nudge_amd import scenes as S
from oracle import refworld
from oracle.refworld import _view_array


class OracleEngine:
    _REC = dict(bt=32, bp=16, bm=32, bi=1, xt=4, xd=16, xx=32, st=4, sd=4, sx=32)
    _KEY = dict(bt="body_transforms", bp="body_properties", bm="body_momentum", bi="idle_counters", xt="box_tags", xd="box_data",
                xx="box_transforms", st="sphere_tags", sd="sphere_data", sx="sphere_transforms")

    def __init__(self, scene, capacity):
        self.params = dict(scene["params"])
        self.cap = dict(capacity)
        self.t = {}
        for name, w in self._REC.items():
            src = scene[self._KEY[name]]
            if name in ("xt", "st"):
                src = src.astype(np.uint32)
            which = "bodies" if name[0] == "b" else ("boxes" if name[0] == "x" else "spheres")
            t = torch.zeros((self.cap[which], w), dtype=torch.uint8)
            raw = np.ascontiguousarray(src).view(np.uint8).reshape(len(src), -1) if len(src) else np.zeros((0, w), np.uint8)
            t[:len(src)] = torch.from_numpy(raw.copy())
            self.t[name] = t
        self.nb, self.nbox, self.nsph = len(scene["body_transforms"]), len(scene["box_tags"]), len(scene["sphere_tags"])
        pad = dict(scene)
        pad["body_transforms"] = np.zeros(self.cap["bodies"], S.TRANSFORM); pad["body_properties"] = np.zeros(self.cap["bodies"], S.PROPERTIES)
        pad["body_momentum"] = np.zeros(self.cap["bodies"], S.MOMENTUM); pad["idle_counters"] = np.zeros(self.cap["bodies"], np.uint8)
        pad["box_tags"] = np.zeros(self.cap["boxes"], np.uint32); pad["box_data"] = np.zeros(self.cap["boxes"], S.BOX); pad["box_transforms"] = np.zeros(self.cap["boxes"], S.TRANSFORM)
        pad["sphere_tags"] = np.zeros(self.cap["spheres"], np.uint32); pad["sphere_data"] = np.zeros(self.cap["spheres"], S.SPHERE); pad["sphere_transforms"] = np.zeros(self.cap["spheres"], S.TRANSFORM)
        self.ref = refworld.RefWorld(pad, max_contacts=max(4096, 16 * self.cap["bodies"]))

    def records(self, name, record_bytes):
        assert self._REC[name] == record_bytes
        return self.t[name]

    def set_counts(self, bodies, boxes, spheres):
        assert bodies <= self.cap["bodies"] and boxes <= self.cap["boxes"] and spheres <= self.cap["spheres"]
        self.nb, self.nbox, self.nsph = bodies, boxes, spheres

    def _np(self, name, dtype, n):
        return np.frombuffer(self.t[name][:n].numpy().tobytes(), dtype=dtype, count=n)

    # the step in phases (Partition(per_iteration=True)): the caller-side tensors are the truth between phases, so the momentum goes in
    # before every sweep and comes out after it
    def _upload(self):
        self._phase = True
        self.step(1, only_upload=True)

    def set_first_ghost(self, first_ghost):
        """Contact ownership (include/nudge_hip.h: nh_set_first_ghost_body), on the reference: the contacts this rank does not own are removed from the list between
        collide and read_cached_impulses (example/main.cpp:287 is where the reference lets its caller edit the list)."""
        self.first_ghost = int(first_ghost)

    def collide(self):
        self._upload(); self.ref.collide()
        if getattr(self, "first_ghost", 0):
            self.dropped_contacts = self.ref.keep_owned_contacts(self.first_ghost)
    def gravity(self):
        self.ref.gravity()
        self._download(momentum_only=True)          # (single ownership measures what the solver does to a ghost from here)
    def read_cache(self): self.ref.read_cache()

    def setup(self):
        self.ref.setup()
        self._download(momentum_only=True)

    def apply(self, iterations):
        v = self.ref._view()
        _view_array(v.body_momentum, S.MOMENTUM, self.nb)[:] = self._np("bm", S.MOMENTUM, self.nb)
        self.ref.apply(iterations)
        self._download(momentum_only=True)

    def update(self):
        v = self.ref._view()
        _view_array(v.body_momentum, S.MOMENTUM, self.nb)[:] = self._np("bm", S.MOMENTUM, self.nb)
        self.ref.update()

    def write_cache(self): self.ref.write_cache()

    def advance(self):
        self.ref.advance()
        self._download()

    def _download(self, momentum_only=False):
        nb = self.nb
        out = self.ref.bodies()
        self.t["bm"][:nb] = torch.from_numpy(np.ascontiguousarray(out["momentum"]).view(np.uint8).reshape(nb, 32).copy())
        if momentum_only:
            return
        gid = self._np("bt", S.TRANSFORM, nb)["body"].copy()
        bt = out["transforms"].copy()
        bt["body"] = gid                                             # the global id rides in the unused field
        self.t["bt"][:nb] = torch.from_numpy(np.ascontiguousarray(bt).view(np.uint8).reshape(nb, 32).copy())
        self.t["bi"][:nb, 0] = torch.from_numpy(out["idle"].copy())

    def step(self, steps=1, only_upload=False):
        r, v = self.ref, self.ref._view()
        nb, nx, ns = self.nb, self.nbox, self.nsph
        for _ in range(steps):
            _view_array(v.body_transforms, S.TRANSFORM, nb)[:] = self._np("bt", S.TRANSFORM, nb)
            _view_array(v.body_properties, S.PROPERTIES, nb)[:] = self._np("bp", S.PROPERTIES, nb)
            _view_array(v.body_momentum, S.MOMENTUM, nb)[:] = self._np("bm", S.MOMENTUM, nb)
            _view_array(v.idle_counters, np.uint8, nb)[:] = self._np("bi", np.uint8, nb)
            if nx:
                tags = self._np("xt", np.uint32, nx)
                assert tags.max() < 65536
                _view_array(v.box_tags, np.uint16, nx)[:] = tags.astype(np.uint16)
                _view_array(v.box_data, S.BOX, nx)[:] = self._np("xd", S.BOX, nx)
                _view_array(v.box_transforms, S.TRANSFORM, nx)[:] = self._np("xx", S.TRANSFORM, nx)
            if ns:
                tags = self._np("st", np.uint32, ns)
                assert tags.max() < 65536
                _view_array(v.sphere_tags, np.uint16, ns)[:] = tags.astype(np.uint16)
                _view_array(v.sphere_data, S.SPHERE, ns)[:] = self._np("sd", S.SPHERE, ns)
                _view_array(v.sphere_transforms, S.TRANSFORM, ns)[:] = self._np("sx", S.TRANSFORM, ns)
            r.L.ref_world_set_counts(r.h, nb, nx, ns, 0)
            r.n_bodies, r.n_boxes, r.n_spheres = nb, nx, ns
            if only_upload:
                return
            r.step(1)
            self._download()


def single_world_reference(scene, steps):
    w = refworld.RefWorld(scene)
    w.step(steps)
    return w.bodies()


# ---- the scene as a file for the C++ host (examples/partition_rccl.cpp: load_scene) and its result file -----------------------------------------------------
def write_scene_file(scene, path):
    """'NHSC' | u32 version = 1, nb, nbox, nsph, iterations | f64 time_step, gravity, damping_rate | body transforms, properties, momentum, idle counters,
    box tags (u32), data, transforms, sphere tags (u32), data, transforms -- the record layouts of include/nudge_hip.h."""
    p = scene["params"]
    with open(path, "wb") as f:
        f.write(b"NHSC")
        f.write(np.array([1, len(scene["body_transforms"]), len(scene["box_tags"]), len(scene["sphere_tags"]), int(p["iterations"])], dtype=np.uint32).tobytes())
        f.write(np.array([p["time_step"], p["gravity"], p["damping_rate"]], dtype=np.float64).tobytes())
        for k in ("body_transforms", "body_properties", "body_momentum", "idle_counters"):
            f.write(np.ascontiguousarray(scene[k]).tobytes())
        f.write(np.ascontiguousarray(scene["box_tags"], dtype=np.uint32).tobytes()); f.write(np.ascontiguousarray(scene["box_data"]).tobytes()); f.write(np.ascontiguousarray(scene["box_transforms"]).tobytes())
        f.write(np.ascontiguousarray(scene["sphere_tags"], dtype=np.uint32).tobytes()); f.write(np.ascontiguousarray(scene["sphere_data"]).tobytes()); f.write(np.ascontiguousarray(scene["sphere_transforms"]).tobytes())


def read_owned_file(path):
    """What partition_rccl --out writes: per owned body its transform (the global id in `body`), momentum and idle counter; returned sorted by id like gather()."""
    rec = np.dtype([("xf", S.TRANSFORM), ("mom", S.MOMENTUM), ("idle", np.uint8)])
    a = np.fromfile(path, dtype=rec)
    order = np.argsort(a["xf"]["body"], kind="stable")
    a = a[order]
    return dict(ids=a["xf"]["body"].astype(np.int64), transforms=a["xf"].copy(), momentum=a["mom"].copy(), idle=a["idle"].copy())
