"""Helpers shared by the parity tests: oracle <-> wide-format conversions and comparisons."""
import numpy as np


def oracle_contacts_sorted(c):
    """Oracle contacts (narrow tags) in tag order -> (data, bodies[u32], pair key a|b<<32, feature)."""
    order = np.argsort(c["tags"], kind="stable")
    tags = c["tags"][order]
    hi = (tags >> np.uint64(32)).astype(np.uint64)
    a = hi & np.uint64(0xFFFF)
    b = (hi >> np.uint64(16)) & np.uint64(0xFFFF)
    key = a | (b << np.uint64(32))
    feat = (tags & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return c["data"][order], c["bodies"][order].astype(np.uint32), key, feat, order


def widen_sleeping(sp):
    sp = sp.astype(np.uint64)
    return (sp & np.uint64(0xFFFF)) | ((sp >> np.uint64(16)) << np.uint64(32))


def bits_equal(x, y):
    x = np.ascontiguousarray(x)
    y = np.ascontiguousarray(y)
    return x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8))


def rel_err(p, pref, origin=None):
    """Per-body ||p - pref|| / max(||pref - origin||, 1)  (SURVEY 8d)."""
    d = np.linalg.norm(p.astype(np.float64) - pref.astype(np.float64), axis=1)
    base = pref.astype(np.float64) if origin is None else pref.astype(np.float64) - origin
    return d / np.maximum(np.linalg.norm(base, axis=1), 1.0)
