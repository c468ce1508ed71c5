"""ctypes binding of libnudge_hip.so (include/nudge_hip.h) + a `World` convenience wrapper.

Host-side plumbing only: torch owns the device arrays (plain uint8 tensors reinterpreted as the POD records
of nudge_hip.h), the HIP stream is torch's current stream, and every entry point of the step loop is the
C-ABI function that replaces the reference call of the same name (reference example/main.cpp:274-328).

There is NO CPU fallback: importing works anywhere, but constructing a `World` raises unless the HIP
library is built and a GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import warnings

import numpy as np

from . import scenes as S

_HERE = os.path.dirname(os.path.abspath(__file__))
# NUDGE_HIP_LIBRARY: another build of the same library (A/B timing of kernel variants on one box)
_LIB_PATH = os.environ.get("NUDGE_HIP_LIBRARY") or os.path.join(_HERE, "libnudge_hip.so")
_LIB = None

NH_FLAG_SYNC_COUNTS = 1
NH_FLAG_EXACT_ORDER = 2
NH_FLAG_SINGLE_APPLY = 4
NH_FLAG_FUSED_STEP = 8
NH_VIEW_CONTACTS, NH_VIEW_CACHE, NH_VIEW_ACTIVE, NH_VIEW_ALL = 1, 2, 4, 7
# environment variables NH_<NAME> that World() forwards to nh_set_option (tests and dev scripts; include/nudge_hip.h lists what each does)
OPTION_NAMES = ("no_still", "no_kept_pairs", "no_incremental", "no_sort_reuse", "sort_classic", "sort_radix", "bucket_tile", "bucket_target", "colour_jp",
                "colour_check_seeds", "no_resident", "solver_waves", "fp_sub", "cooperative", "no_blocks", "blk_check", "blk_min", "blk_target", "blk_rows_global",
                "blk_global_colours", "blk_profile", "no_asleep", "no_blk_chain", "no_local_still", "no_xform_ahead", "measure_skip_chains", "sync_exports_views", "fused_still", "fused_fail_in", "no_pair_ahead", "no_sleeper_skip", "halo_overlap", "no_early_counts", "no_sleeper_ahead")

EXPORTS = [
    "nh_create", "nh_destroy", "nh_set_flags", "nh_synchronize", "nh_read_counts", "nh_export_views", "nh_set_cache_count",
    "nh_set_tag_bits", "nh_set_pair_capacity", "nh_set_option", "nh_bodies_changed", "nh_error_string", "nh_last_hip_error", "nh_collide", "nh_apply_gravity_damping",
    "nh_read_cached_impulses", "nh_write_cached_impulses", "nh_setup_contact_constraints", "nh_apply_impulses",
    "nh_update_cached_impulses", "nh_advance", "nh_contact_impulses_device", "nh_enable_timing", "nh_set_timing_filter", "nh_kernel_times",
    "nh_halo_pack", "nh_halo_unpack", "nh_halo_update", "nh_append_contacts", "nh_step", "nh_stream_state", "nh_stream_latest",
    "nh_partition_create", "nh_partition_destroy", "nh_partition_info", "nh_partition_pack_migrants", "nh_partition_unpack_migrants", "nh_partition_pack_ghosts",
    "nh_partition_top_speed", "nh_partition_set_peer_speeds", "nh_partition_refresh_is_quiet", "nh_partition_mark_ghosts", "nh_partition_pack_deltas", "nh_partition_unpack_deltas",
    "nh_partition_pack_momentum", "nh_partition_unpack_momentum", "nh_partition_exchange_iteration", "nh_partition_unpack_ghosts", "nh_partition_pack_step", "nh_partition_unpack_step", "nh_partition_choose_cut", "nh_partition_set_cut",
    "nh_partition_set_transport", "nh_partition_exchange_step", "nh_partition_step", "nh_partition_transport_check", "nh_partition_transport_result", "nh_set_first_ghost_body",
]
HALO_RECORD_BYTES = 64


class Arena(C.Structure):
    _fields_ = [("data", C.c_void_p), ("size", C.c_size_t)]


class ContactData(C.Structure):
    _fields_ = [("data", C.c_void_p), ("bodies", C.c_void_p), ("tags", C.c_void_p), ("features", C.c_void_p),
                ("capacity", C.c_uint32), ("count", C.c_uint32), ("sleeping_pairs", C.c_void_p), ("sleeping_count", C.c_uint32)]


class _Shapes(C.Structure):
    _fields_ = [("tags", C.c_void_p), ("data", C.c_void_p), ("transforms", C.c_void_p), ("count", C.c_uint32)]


class ColliderData(C.Structure):
    _fields_ = [("boxes", _Shapes), ("spheres", _Shapes)]


class BodyData(C.Structure):
    _fields_ = [("transforms", C.c_void_p), ("properties", C.c_void_p), ("momentum", C.c_void_p),
                ("idle_counters", C.c_void_p), ("count", C.c_uint32)]


class BodyConnections(C.Structure):
    _fields_ = [("data", C.c_void_p), ("count", C.c_uint32)]


class ContactCache(C.Structure):
    _fields_ = [("tags", C.c_void_p), ("features", C.c_void_p), ("data", C.c_void_p), ("capacity", C.c_uint32), ("count", C.c_uint32)]


class ActiveBodies(C.Structure):
    _fields_ = [("indices", C.c_void_p), ("capacity", C.c_uint32), ("count", C.c_uint32)]


class Counts(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("colliders", "pairs", "contacts", "sleeping_pairs", "active_bodies", "cache", "culled",
                                          "large_colliders", "general_contacts", "levels", "error", "has_other_bodies", "unleveled", "raw_pairs", "broadphase_rebuilds", "sort_reuses", "broadphase_inserts",
                                          "still_steps", "still_replays", "still_diff_key", "still_diff_count", "still_diff_feature", "still_diff_escape",
                                          "blk_blocks", "blk_bodies", "blk_ghosts", "asleep_steps", "ahead_steps", "fused_steps", "pair_steps", "pair_diag_roles", "pair_diag_record", "pair_diag_scale", "pair_diag_owned")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class StepArgs(C.Structure):
    _fields_ = [("active_bodies", C.POINTER(ActiveBodies)), ("contacts", C.POINTER(ContactData)), ("bodies", C.POINTER(BodyData)), ("colliders", C.POINTER(ColliderData)),
                ("body_connections", C.POINTER(BodyConnections)), ("contact_cache", C.POINTER(ContactCache)), ("arena", Arena),
                ("time_step", C.c_float), ("gravity", C.c_float * 3), ("damping_rate", C.c_float), ("iterations", C.c_uint32)]


class PartitionConfig(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("ranks", C.c_uint32), ("lo", C.c_double), ("hi", C.c_double), ("n_owned", C.c_uint32), ("n_static_box", C.c_uint32), ("n_static_sph", C.c_uint32),
                ("body_capacity", C.c_uint32), ("box_capacity", C.c_uint32), ("sphere_capacity", C.c_uint32), ("epoch", C.c_uint32),
                ("time_step", C.c_double), ("gravity", C.c_double), ("speed_floor", C.c_double), ("max_reach", C.c_double), ("cut_slack", C.c_double)]


class PartitionInfo(C.Structure):
    _fields_ = [("n_owned", C.c_uint32), ("n_bodies", C.c_uint32), ("n_boxes", C.c_uint32), ("n_spheres", C.c_uint32), ("ghost_out", C.c_uint32 * 2), ("ghost_in", C.c_uint32 * 2),
                ("lo", C.c_double), ("hi", C.c_double), ("migrated_out", C.c_uint64), ("migrated_in", C.c_uint64), ("refreshes", C.c_uint64), ("cut_moves", C.c_uint64), ("quiet_refreshes", C.c_uint64)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_double), ("launches", C.c_uint32), ("reserved", C.c_uint32)]


def build(force=False):
    """Compile libnudge_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if force:
        subprocess.check_call(cmd + ["clean"])
    subprocess.check_call(cmd)
    return _LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run `make -C nudge_amd/csrc` (there is no CPU fallback)")
        L = C.CDLL(_LIB_PATH)
        L.nh_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_uint32]
        L.nh_destroy.argtypes = [C.c_void_p]
        L.nh_destroy.restype = None
        L.nh_set_flags.argtypes = [C.c_void_p, C.c_uint32]
        L.nh_set_tag_bits.argtypes = [C.c_void_p, C.c_uint32]
        L.nh_set_pair_capacity.argtypes = [C.c_void_p, C.c_uint32]
        L.nh_bodies_changed.argtypes = [C.c_void_p]
        L.nh_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.nh_step.argtypes = [C.c_void_p, C.POINTER(StepArgs), C.c_uint32]
        L.nh_synchronize.argtypes = [C.c_void_p]
        L.nh_read_counts.argtypes = [C.c_void_p, C.POINTER(Counts)]
        L.nh_export_views.argtypes = [C.c_void_p, C.c_uint32]
        L.nh_stream_state.argtypes = [C.c_void_p, C.POINTER(BodyData), C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
        L.nh_stream_latest.argtypes = [C.c_void_p, C.POINTER(StreamInfo)]
        L.nh_set_cache_count.argtypes = [C.c_void_p, C.c_uint32]
        L.nh_error_string.argtypes = [C.c_int]
        L.nh_error_string.restype = C.c_char_p
        L.nh_last_hip_error.argtypes = [C.c_void_p]
        L.nh_collide.argtypes = [C.c_void_p, C.POINTER(ActiveBodies), C.POINTER(ContactData), C.POINTER(BodyData),
                                 C.POINTER(ColliderData), C.POINTER(BodyConnections), Arena]
        L.nh_apply_gravity_damping.argtypes = [C.c_void_p, C.POINTER(ActiveBodies), C.POINTER(BodyData), C.c_float,
                                               C.POINTER(C.c_float), C.c_float]
        L.nh_read_cached_impulses.argtypes = [C.c_void_p, C.POINTER(ContactCache), C.POINTER(ContactData), C.POINTER(Arena), C.POINTER(C.c_void_p)]
        L.nh_write_cached_impulses.argtypes = [C.c_void_p, C.POINTER(ContactCache), C.POINTER(ContactData), C.c_void_p]
        L.nh_setup_contact_constraints.argtypes = [C.c_void_p, C.POINTER(ActiveBodies), C.POINTER(ContactData), C.POINTER(BodyData),
                                                   C.c_void_p, C.POINTER(Arena), C.POINTER(C.c_void_p)]
        L.nh_apply_impulses.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BodyData), C.c_uint32]
        L.nh_update_cached_impulses.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.nh_advance.argtypes = [C.c_void_p, C.POINTER(ActiveBodies), C.POINTER(BodyData), C.c_float]
        L.nh_contact_impulses_device.argtypes = [C.c_void_p]
        L.nh_contact_impulses_device.restype = C.c_void_p
        L.nh_enable_timing.argtypes = [C.c_void_p, C.c_int]
        L.nh_set_timing_filter.argtypes = [C.c_void_p, C.c_char_p]
        L.nh_kernel_times.argtypes = [C.c_void_p, C.POINTER(KernelTime), C.c_int, C.c_int]
        L.nh_halo_pack.argtypes = [C.c_void_p, C.POINTER(BodyData), C.c_void_p, C.c_uint32, C.c_void_p]
        L.nh_halo_unpack.argtypes = [C.c_void_p, C.POINTER(BodyData), C.c_uint32, C.c_uint32, C.c_void_p]
        L.nh_halo_update.argtypes = [C.c_void_p, C.POINTER(BodyData), C.c_uint32, C.c_uint32, C.c_void_p]
        L.nh_append_contacts.argtypes = [C.c_void_p, C.POINTER(ContactData), C.POINTER(BodyData), C.c_uint32, C.c_void_p, Arena]
        # multi-GPU: one x-slab per context (include/nudge_hip.h, "multi-GPU")
        L.nh_partition_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(PartitionConfig), C.POINTER(BodyData), C.POINTER(ColliderData)]
        L.nh_partition_destroy.argtypes = [C.c_void_p]
        L.nh_partition_destroy.restype = None
        L.nh_partition_info.argtypes = [C.c_void_p, C.POINTER(PartitionInfo)]
        L.nh_partition_pack_migrants.argtypes = [C.c_void_p, C.POINTER(BodyData), C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32 * 2)]
        L.nh_partition_unpack_migrants.argtypes = [C.c_void_p, C.POINTER(BodyData), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.nh_partition_pack_ghosts.argtypes = [C.c_void_p, C.POINTER(BodyData), C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32 * 2)]
        L.nh_partition_unpack_ghosts.argtypes = [C.c_void_p, C.POINTER(BodyData), C.POINTER(ColliderData), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.nh_partition_top_speed.argtypes = [C.c_void_p, C.POINTER(BodyData), C.POINTER(C.c_double)]
        L.nh_partition_set_peer_speeds.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.nh_partition_refresh_is_quiet.argtypes = [C.c_void_p, C.POINTER(BodyData), C.POINTER(C.c_int)]
        L.nh_partition_pack_step.argtypes = [C.c_void_p, C.POINTER(BodyData), C.c_void_p, C.c_void_p]
        L.nh_partition_mark_ghosts.argtypes = [C.c_void_p, C.POINTER(BodyData)]
        L.nh_partition_exchange_iteration.argtypes = [C.c_void_p, C.POINTER(BodyData)]
        for _fn in (L.nh_partition_pack_deltas, L.nh_partition_unpack_deltas, L.nh_partition_pack_momentum, L.nh_partition_unpack_momentum):
            _fn.argtypes = [C.c_void_p, C.POINTER(BodyData), C.c_void_p, C.c_void_p]
        L.nh_partition_unpack_step.argtypes = [C.c_void_p, C.POINTER(BodyData), C.c_void_p, C.c_void_p]
        L.nh_partition_choose_cut.argtypes = [C.c_void_p, C.POINTER(BodyData), C.c_int, C.c_uint32, C.POINTER(C.c_double)]
        L.nh_partition_set_cut.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.nh_partition_set_transport.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.nh_partition_exchange_step.argtypes = [C.c_void_p, C.POINTER(BodyData)]
        L.nh_partition_step.argtypes = [C.c_void_p, C.POINTER(StepArgs), C.c_uint32, C.c_uint32, C.c_uint32]
        L.nh_partition_transport_check.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.nh_partition_transport_result.argtypes = [C.c_void_p]
        L.nh_set_first_ghost_body.argtypes = [C.c_void_p, C.c_uint32]
        _LIB = L
    return _LIB


class NudgeError(RuntimeError):
    pass


class StreamInfo(C.Structure):
    _fields_ = [("slot", C.c_uint32), ("valid", C.c_uint32), ("step", C.c_uint64), ("frames", C.c_uint64), ("dropped", C.c_uint64)]


def _check(L, rc, what):
    if rc != 0:
        raise NudgeError(f"{what}: {L.nh_error_string(rc).decode()} ({rc})")


class World:
    """A world resident in HBM, stepped through the C ABI.  `scene` uses the layouts of nudge_amd.scenes."""

    def __init__(self, scene, device=0, max_contacts=None, arena_bytes=None, flags=NH_FLAG_SYNC_COUNTS, tag_bits=None, capacity=None, max_pairs=None):
        """`capacity` = dict(bodies=, boxes=, spheres=): allocate room for more records than the scene holds (the
        partitioned world appends ghost / migrated bodies, nudge_amd/partition.py); counts then change via set_counts().
        `max_pairs`: capacity of the broadphase pair buffer (nh_set_pair_capacity); default max_contacts / 2 + 1024."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("nudge_amd.World needs a HIP device (no CPU fallback)")
        self.torch = torch
        self.L = lib()
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.params = dict(scene["params"])
        nb = len(scene["body_transforms"])
        nbox, nsph = len(scene["box_tags"]), len(scene["sphere_tags"])
        if max_contacts is None:
            max_contacts = max(4096, 8 * nb)
        self.nb, self.nbox, self.nsph, self.max_contacts = nb, nbox, nsph, max_contacts
        capacity = capacity or {}
        cap_b, cap_x, cap_s = max(nb, capacity.get("bodies", 0)), max(nbox, capacity.get("boxes", 0)), max(nsph, capacity.get("spheres", 0))
        self.capacity = dict(bodies=cap_b, boxes=cap_x, spheres=cap_s)
        self._keep = {}

        def up(name, arr, min_bytes=16, cap=0):
            a = np.ascontiguousarray(arr)
            raw = a.view(np.uint8).reshape(-1)
            t = torch.zeros(max(raw.size, min_bytes, cap * a.dtype.itemsize), dtype=torch.uint8, device=self.dev)
            if raw.size:
                t[:raw.size] = torch.from_numpy(raw.copy()).to(self.dev)
            self._keep[name] = t
            return t

        def alloc(name, nbytes):
            t = torch.zeros(max(int(nbytes), 16), dtype=torch.uint8, device=self.dev)
            self._keep[name] = t
            return t

        self.bodies = BodyData(up("bt", scene["body_transforms"], cap=cap_b).data_ptr(), up("bp", scene["body_properties"], cap=cap_b).data_ptr(),
                               up("bm", scene["body_momentum"], cap=cap_b).data_ptr(), up("bi", scene["idle_counters"], cap=cap_b).data_ptr(), nb)
        self.colliders = ColliderData(
            _Shapes(up("xt", scene["box_tags"].astype(np.uint32), cap=cap_x).data_ptr(), up("xd", scene["box_data"], cap=cap_x).data_ptr(),
                    up("xx", scene["box_transforms"], cap=cap_x).data_ptr(), nbox),
            _Shapes(up("st", scene["sphere_tags"].astype(np.uint32), cap=cap_s).data_ptr(), up("sd", scene["sphere_data"], cap=cap_s).data_ptr(),
                    up("sx", scene["sphere_transforms"], cap=cap_s).data_ptr(), nsph))
        con = np.asarray(scene.get("connections", np.zeros((0, 2), np.uint32)), dtype=np.uint32).reshape(-1, 2)
        self.connections = BodyConnections(up("cn", con).data_ptr(), len(con))
        K = max_contacts
        self.contacts = ContactData(alloc("cd", 32 * K).data_ptr(), alloc("cb", 8 * K).data_ptr(), alloc("ct", 8 * K).data_ptr(),
                                    alloc("cf", 4 * K).data_ptr(), K, 0, alloc("cs", 8 * K).data_ptr(), 0)
        self.cache = ContactCache(alloc("kt", 8 * K).data_ptr(), alloc("kf", 4 * K).data_ptr(), alloc("kd", 16 * K).data_ptr(), K, 0)
        self.active = ActiveBodies(alloc("ai", 4 * cap_b).data_ptr(), cap_b, 0)
        if arena_bytes is None:
            ncol = cap_x + cap_s
            nb = cap_b
            cells = 1
            while cells < 4 * ncol:
                cells <<= 1
            cells = min(max(cells, 1 << 16), 1 << 24)
            pairs = max_pairs if max_pairs else K // 2 + 1024
            arena_bytes = (64 << 20) + ncol * 200 + cells * 8 + K * 320 + pairs * 216 + nb * 64
        self.arena_t = alloc("arena", arena_bytes)
        self.arena = Arena(self.arena_t.data_ptr(), arena_bytes)

        self.ctx = C.c_void_p()
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        _check(self.L, self.L.nh_create(C.byref(self.ctx), device, C.c_void_p(stream), flags), "nh_create")
        self.flags = flags
        # A/B switches for tests and dev scripts: environment variables NH_<NAME>=<int> become nh_set_option(name, value) calls HERE -- the library itself never
        # reads the environment (include/nudge_hip.h: nh_set_option)
        for key, val in sorted(os.environ.items()):
            if not key.startswith("NH_"):
                continue
            name = key[3:].lower()
            if name not in OPTION_NAMES:           # (a stale or unrelated NH_* variable must not keep a world from being created)
                if name not in ("gather_32",):
                    warnings.warn(f"nudge_amd: environment variable {key} names no library option; ignored")
                continue
            try:
                num = int(val)
            except ValueError:
                num = 1
            rc = self.L.nh_set_option(self.ctx, name.encode(), num)
            if rc:                                 # (e.g. measure_skip_chains on a build without -DNH_MEASURE)
                warnings.warn(f"nudge_amd: nh_set_option({name}, {num}) refused: {self.L.nh_error_string(rc).decode()}")
        if tag_bits is None:
            mt = 1
            for k in ("box_tags", "sphere_tags"):
                if len(scene[k]):
                    mt = max(mt, int(scene[k].max()))
            tag_bits = max(8, int(mt).bit_length())
        self.L.nh_set_tag_bits(self.ctx, tag_bits)
        if max_pairs:
            _check(self.L, self.L.nh_set_pair_capacity(self.ctx, int(max_pairs)), "nh_set_pair_capacity")
        self.steps_done = 0
        self.sum_contacts = 0
        self._imp = C.c_void_p()
        self._con = C.c_void_p()

    def close(self):
        if getattr(self, "ctx", None):
            self.L.nh_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the reference's step loop (example/main.cpp:274-328), one C-ABI call per reference call ----
    def collide(self):
        _check(self.L, self.L.nh_collide(self.ctx, C.byref(self.active), C.byref(self.contacts), C.byref(self.bodies),
                                         C.byref(self.colliders), C.byref(self.connections), self.arena), "nh_collide")
        self._temp = Arena(self.arena.data, self.arena.size)

    def append_contacts(self, data, bodies, tags, features, count=None):
        """Custom contacts behind the list collide() returned (reference example/main.cpp:287): numpy arrays of S.CONTACT records, (n, 2) uint32
        body pairs, uint64 tags (a_tag | b_tag << 32) and uint32 features; `count` = contacts so far (read from the device when not given)."""
        torch = self.torch
        n = len(tags)
        if n == 0:
            return
        k = self.counts()["contacts"] if count is None else count
        if k + n > self.max_contacts:
            raise NudgeError("append_contacts: contact capacity")
        for name, arr, width in (("cd", np.ascontiguousarray(data), 32), ("cb", np.ascontiguousarray(bodies, dtype=np.uint32), 8),
                                 ("ct", np.ascontiguousarray(tags, dtype=np.uint64), 8), ("cf", np.ascontiguousarray(features, dtype=np.uint32), 4)):
            raw = arr.view(np.uint8).reshape(-1)
            assert raw.size == width * n, name
            self._keep[name][width * k:width * (k + n)] = torch.from_numpy(raw.copy()).to(self.dev)
        _check(self.L, self.L.nh_append_contacts(self.ctx, C.byref(self.contacts), C.byref(self.bodies), n, None, self.arena), "append_contacts")

    def gravity(self):
        p = self.params
        g = (C.c_float * 3)(0.0, p["gravity"], 0.0)
        _check(self.L, self.L.nh_apply_gravity_damping(self.ctx, C.byref(self.active), C.byref(self.bodies), p["time_step"], g, p["damping_rate"]), "gravity")

    def read_cache(self):
        _check(self.L, self.L.nh_read_cached_impulses(self.ctx, C.byref(self.cache), C.byref(self.contacts), C.byref(self._temp), C.byref(self._imp)), "read_cached_impulses")

    def setup(self):
        _check(self.L, self.L.nh_setup_contact_constraints(self.ctx, C.byref(self.active), C.byref(self.contacts), C.byref(self.bodies),
                                                           self._imp, C.byref(self._temp), C.byref(self._con)), "setup_contact_constraints")

    def apply(self, iterations=None):
        it = self.params["iterations"] if iterations is None else iterations
        _check(self.L, self.L.nh_apply_impulses(self.ctx, self._con, C.byref(self.bodies), it), "apply_impulses")

    def update(self):
        _check(self.L, self.L.nh_update_cached_impulses(self.ctx, self._con, self._imp), "update_cached_impulses")

    def write_cache(self):
        _check(self.L, self.L.nh_write_cached_impulses(self.ctx, C.byref(self.cache), C.byref(self.contacts), self._imp), "write_cached_impulses")

    def advance(self, time_step=None):
        dt = self.params["time_step"] if time_step is None else time_step
        _check(self.L, self.L.nh_advance(self.ctx, C.byref(self.active), C.byref(self.bodies), dt), "advance")

    def step(self, steps=1, iterations=None):
        """`steps` sub-steps of the sample's loop (example/main.cpp:274-328).  Worlds that do not ask for host-visible counts go through nh_step -- the eight calls made
        by the library itself, one crossing of the ABI for all the steps; the others (NH_FLAG_SYNC_COUNTS: `sum_contacts` is kept per step) make the eight calls here."""
        if steps > 0 and not (self.flags & NH_FLAG_SYNC_COUNTS) and not getattr(self, "eight_calls", False):
            p = self.params
            it = p["iterations"] if iterations is None else iterations
            # (rebuilt every call -- a dozen host-side stores -- so that a changed time step, gravity or damping is never ignored; the structs it points to are this
            # object's own: counts changed by set_counts are seen through the pointers)
            args = self._step_args = StepArgs(C.pointer(self.active), C.pointer(self.contacts), C.pointer(self.bodies), C.pointer(self.colliders), C.pointer(self.connections),
                                              C.pointer(self.cache), self.arena, p["time_step"], (C.c_float * 3)(0.0, p["gravity"], 0.0), p["damping_rate"], it)
            _check(self.L, self.L.nh_step(self.ctx, C.byref(args), steps), "nh_step")
            self.steps_done += steps
            return
        for _ in range(steps):
            self.collide()
            self.gravity()
            self.read_cache()
            self.setup()
            self.apply(iterations)
            self.update()
            self.write_cache()
            self.advance()
            self.step_done()

    def partition_step(self, partition, steps, exchange_first, loopback_records=0, iterations=None):
        """`steps` sub-steps of a partitioned world in ONE library call (nh_partition_step): the per-step halo exchange is enqueued by the library between two sub-steps
        over the transport of nh_partition_set_transport; the chain of still steps runs through the call."""
        p = self.params
        it = p["iterations"] if iterations is None else iterations
        args = self._step_args = StepArgs(C.pointer(self.active), C.pointer(self.contacts), C.pointer(self.bodies), C.pointer(self.colliders), C.pointer(self.connections),
                                          C.pointer(self.cache), self.arena, p["time_step"], (C.c_float * 3)(0.0, p["gravity"], 0.0), p["damping_rate"], it)
        _check(self.L, self.L.nh_partition_step(partition, C.byref(args), steps, 1 if exchange_first else 0, loopback_records), "nh_partition_step")
        self.steps_done += steps

    def step_done(self):
        """Book-keeping of one finished step; callers that drive the eight calls themselves (partition.py, per-iteration exchange) call it after advance()."""
        self.steps_done += 1
        if self.flags & NH_FLAG_SYNC_COUNTS:
            self.sum_contacts += self.contacts.count

    # ---- checkpoint / restore of the caller-owned persistent state (bodies + contact cache), device to device ----
    def snapshot(self):
        """Everything that persists from one step to the next on the caller's side (nudge.h: BodyData transforms / momentum / idle
        counters and the ContactCache), cloned on the device."""
        self.export_views(NH_VIEW_CACHE)
        c = self.counts()
        return dict(cache_count=c["cache"], arrays={k: self._keep[k].clone() for k in ("bt", "bm", "bi", "kt", "kf", "kd")})

    def restore(self, snap):
        self.synchronize()
        for k, t in snap["arrays"].items():
            self._keep[k].copy_(t)
        self.L.nh_bodies_changed(self.ctx)
        _check(self.L, self.L.nh_set_cache_count(self.ctx, snap["cache_count"]), "set_cache_count")

    # ---- variable membership (partitioned worlds) ----
    def set_counts(self, bodies, boxes, spheres):
        c = self.capacity
        if bodies > c["bodies"] or boxes > c["boxes"] or spheres > c["spheres"]:
            raise NudgeError(f"set_counts({bodies}, {boxes}, {spheres}) exceeds the capacity {c}")
        self.nb, self.nbox, self.nsph = bodies, boxes, spheres
        self.L.nh_bodies_changed(self.ctx)            # (the caller has rewritten body records, idle counters included)
        self.bodies.count = bodies
        self.colliders.boxes.count = boxes
        self.colliders.spheres.count = spheres

    def set_first_ghost(self, first_ghost):
        """Contact ownership of a partitioned world (include/nudge_hip.h: nh_set_first_ghost_body): bodies >= first_ghost are ghosts; 0 = off."""
        _check(self.L, self.L.nh_set_first_ghost_body(self.ctx, int(first_ghost)), "nh_set_first_ghost_body")

    def records(self, name, record_bytes):
        """Device tensor behind one caller-owned array, as [capacity, record_bytes] uint8 (no copy)."""
        t = self._keep[name]
        n = t.numel() // record_bytes
        return t[:n * record_bytes].view(n, record_bytes)

    # ---- halo records of the partitioned world (include/nudge_hip.h: nh_halo_pack / nh_halo_unpack) ----
    def halo_pack(self, indices_i32):
        """indices_i32: int32 device tensor of body slots -> uint8 device tensor [len, 64] of their per-step records."""
        n = int(indices_i32.numel())
        out = self.torch.empty((n, HALO_RECORD_BYTES), dtype=self.torch.uint8, device=self.dev)
        _check(self.L, self.L.nh_halo_pack(self.ctx, C.byref(self.bodies), C.c_void_p(indices_i32.data_ptr() if n else 0), n,
                                           C.c_void_p(out.data_ptr() if n else 0)), "halo_pack")
        return out

    def halo_unpack(self, first_slot, records, same_bodies=False):
        """same_bodies: the records update the bodies already in those slots (per-step halo: nh_halo_update keeps the sleep prediction)."""
        n = int(records.shape[0])
        if n:
            assert records.is_contiguous() and records.shape[1] == HALO_RECORD_BYTES
            fn = self.L.nh_halo_update if same_bodies else self.L.nh_halo_unpack
            _check(self.L, fn(self.ctx, C.byref(self.bodies), first_slot, n, C.c_void_p(records.data_ptr())), "halo_unpack")

    # ---- state ----
    def synchronize(self):
        _check(self.L, self.L.nh_synchronize(self.ctx), "synchronize")

    def export_views(self, what=NH_VIEW_ALL):
        """The dense contact list / the contact cache in the caller's arrays brought up to date after still steps (include/nudge_hip.h note 9)."""
        _check(self.L, self.L.nh_export_views(self.ctx, what), "export_views")

    def counts(self):
        c = Counts()
        _check(self.L, self.L.nh_read_counts(self.ctx, C.byref(c)), "read_counts")
        return c.as_dict()

    def _down(self, name, dtype, count):
        t = self._keep[name]
        nbytes = np.dtype(dtype).itemsize * count
        return np.frombuffer(t[:nbytes].cpu().numpy().tobytes(), dtype=dtype, count=count).copy()

    def get_bodies(self):
        self.synchronize()
        return dict(transforms=self._down("bt", S.TRANSFORM, self.nb), momentum=self._down("bm", S.MOMENTUM, self.nb),
                    idle=self._down("bi", np.uint8, self.nb))

    def set_option(self, name, value=1):
        """nh_set_option on the live world (tests, A/B): include/nudge_hip.h lists the names."""
        _check(self.L, self.L.nh_set_option(self.ctx, name.encode(), int(value)), f"nh_set_option({name})")

    def set_bodies(self, transforms=None, momentum=None, idle=None):
        torch = self.torch
        for name, arr in (("bt", transforms), ("bm", momentum), ("bi", idle)):
            if arr is not None:
                raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
                self._keep[name][:raw.size] = torch.from_numpy(raw.copy()).to(self.dev)
        if idle is not None:
            self.L.nh_bodies_changed(self.ctx)

    def get_contacts(self):
        self.export_views(NH_VIEW_CONTACTS)
        c = self.counts()
        n, ns = c["contacts"], c["sleeping_pairs"]
        return dict(count=n, data=self._down("cd", S.CONTACT, n), bodies=self._down("cb", np.uint32, 2 * n).reshape(-1, 2),
                    tags=self._down("ct", np.uint64, n), features=self._down("cf", np.uint32, n),
                    sleeping_pairs=self._down("cs", np.uint64, ns))

    def get_active(self):
        self.export_views(NH_VIEW_ACTIVE)
        c = self.counts()
        return self._down("ai", np.uint32, c["active_bodies"])

    def get_cache(self):
        self.export_views(NH_VIEW_CACHE)
        c = self.counts()
        n = c["cache"]
        return dict(count=n, tags=self._down("kt", np.uint64, n), features=self._down("kf", np.uint32, n), data=self._down("kd", S.IMPULSE, n))

    def get_contact_impulses(self):
        if not self._imp:
            raise NudgeError("get_contact_impulses: no nh_ContactImpulseData handle -- step() went through nh_step, which keeps the handles inside the library; "
                             "set world.eight_calls = True (or call the eight entry points) to look at the per-contact impulses of a step")
        c = self.counts()
        n = c["contacts"]
        ptr = self.L.nh_contact_impulses_device(self._imp)
        off = ptr - self.arena_t.data_ptr()
        raw = self.arena_t[off:off + 16 * n].cpu().numpy().tobytes()
        return np.frombuffer(raw, dtype=S.IMPULSE, count=n).copy()

    # ---- state streaming (include/nudge_hip.h: nh_stream_state): body transforms into a pinned host ring every `every` steps, without stopping the world ----
    def stream_state(self, every, slots=4, count=None):
        torch = self.torch
        n = self.nb if count is None else count
        if every:
            self._stream_ring = torch.zeros((slots, n * 32), dtype=torch.uint8).pin_memory()
            self._stream_count = n
            _check(self.L, self.L.nh_stream_state(self.ctx, C.byref(self.bodies), n, C.c_void_p(self._stream_ring.data_ptr()), slots, every), "nh_stream_state")
        else:
            _check(self.L, self.L.nh_stream_state(self.ctx, None, 0, None, 0, 0), "nh_stream_state")

    def stream_latest(self):
        """(step, transforms) of the newest frame that has landed in the ring, or None; plus the stream's counters."""
        info = StreamInfo()
        _check(self.L, self.L.nh_stream_latest(self.ctx, C.byref(info)), "nh_stream_latest")
        stats = dict(frames=int(info.frames), dropped=int(info.dropped))
        if not info.valid:
            return None, stats
        raw = self._stream_ring[info.slot].numpy().tobytes()
        return (int(info.step), np.frombuffer(raw, dtype=S.TRANSFORM, count=self._stream_count).copy()), stats

    # ---- measurement ----
    def enable_timing(self, on=True, only=None):
        self.L.nh_set_timing_filter(self.ctx, only.encode() if only else None)
        self.L.nh_enable_timing(self.ctx, 1 if on else 0)

    def kernel_times(self, reset=True):
        buf = (KernelTime * 128)()
        n = self.L.nh_kernel_times(self.ctx, buf, 128, 1 if reset else 0)
        return {buf[i].name.decode(): (buf[i].ms, int(buf[i].launches)) for i in range(n)}
