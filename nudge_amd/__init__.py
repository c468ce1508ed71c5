"""nudge_amd -- MI355X-native rigid-body stepping engine (collide -> setup -> apply_impulses -> advance).

The product is the C-ABI library built from nudge_amd/csrc (include/nudge_hip.h); this Python package
is host-side plumbing for tests and benchmarks: scene generation (scenes.py) and a ctypes binding
(engine.py).  There is no CPU fallback: engine.py raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"
