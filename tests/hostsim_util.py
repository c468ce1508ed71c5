"""ctypes access to tests/hostsim/libhostsim.so (CPU build of the kernels' per-item device functions)."""
import ctypes as C
import os
import subprocess

import numpy as np

from nudge_amd import scenes as S

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_DIR, "libhostsim.so")
        src = os.path.join(_DIR, "hostsim.cpp")
        hdrs = [os.path.join(_DIR, "..", "..", "nudge_amd", "csrc", h) for h in ("nh_math.h", "nh_narrowphase.h", "nh_solver.h")]
        newest = max(os.path.getmtime(p) for p in [src] + hdrs)
        if not os.path.exists(so) or os.path.getmtime(so) < newest:
            subprocess.check_call(["g++", "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-fPIC", "-shared",
                                   "-std=c++14", src, "-o", so])
        _LIB = C.CDLL(so)
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def collide(body_transforms, scene, cap=1 << 20):
    L = lib()
    nbox, nsph = len(scene["box_tags"]), len(scene["sphere_tags"])
    bt = np.ascontiguousarray(body_transforms)
    oc = np.zeros(cap, dtype=S.CONTACT)
    ob = np.zeros((cap, 2), dtype=np.uint32)
    ok = np.zeros(cap, dtype=np.uint64)
    of = np.zeros(cap, dtype=np.uint32)
    L.hs_collide.restype = C.c_int
    n = L.hs_collide(_p(bt), C.c_uint32(nbox), _p(scene["box_transforms"]), _p(scene["box_data"]), _p(scene["box_tags"]),
                     C.c_uint32(nsph), _p(scene["sphere_transforms"]), _p(scene["sphere_data"]), _p(scene["sphere_tags"]),
                     C.c_uint32(cap), _p(oc), _p(ob), _p(ok), _p(of))
    assert n <= cap
    return dict(count=n, data=oc[:n], bodies=ob[:n], keys=ok[:n], features=of[:n])


def narrow_tags(keys, features):
    """(a_tag | b_tag<<32, feature) -> the reference's u64 tag (nudge.h:76)."""
    a = keys & np.uint64(0xFFFF)
    b = (keys >> np.uint64(32)) & np.uint64(0xFFFF)
    return ((a | (b << np.uint64(16))) << np.uint64(32)) | features.astype(np.uint64)


def solve(body_transforms, props, momentum, contacts, bodies, cached, order, iterations, one_body=False, radial=False):
    """`one_body`: contacts against body 0 use the one-body forms of nh_solver.h (what the fused GPU kernel runs);
    `radial`: contacts that qualify use the radial 12-float rows (the colour-scheduled sweeps of the default order)."""
    L = lib()
    mom = np.ascontiguousarray(momentum).copy()
    n = len(contacts)
    out = np.zeros(n, dtype=S.IMPULSE)
    order = np.ascontiguousarray(order, dtype=np.uint32)
    bodies = np.ascontiguousarray(bodies, dtype=np.uint32)
    fn = L.hs_solve_radial if radial else (L.hs_solve_one_body if one_body else L.hs_solve)
    fn.restype = C.c_int
    taken = fn(C.c_uint32(len(body_transforms)), _p(np.ascontiguousarray(body_transforms)), _p(np.ascontiguousarray(props)), _p(mom),
               C.c_uint32(n), _p(np.ascontiguousarray(contacts)), _p(bodies), _p(np.ascontiguousarray(cached)),
               C.c_uint32(len(order)), _p(order), C.c_uint32(iterations), _p(out))
    if one_body or radial:
        return mom, out, taken
    return mom, out


def advance(active, body_transforms, momentum, idle, dt):
    L = lib()
    bt = np.ascontiguousarray(body_transforms).copy()
    idl = np.ascontiguousarray(idle).copy()
    act = np.ascontiguousarray(active, dtype=np.uint32)
    L.hs_advance(C.c_uint32(len(act)), _p(act), _p(bt), _p(np.ascontiguousarray(momentum)), _p(idl), C.c_float(dt))
    return bt, idl


def gravity(active, momentum, dt, g, rate):
    L = lib()
    mom = np.ascontiguousarray(momentum).copy()
    act = np.ascontiguousarray(active, dtype=np.uint32)
    L.hs_gravity(C.c_uint32(len(act)), _p(act), _p(mom), C.c_float(dt), C.c_float(g), C.c_float(rate))
    return mom
