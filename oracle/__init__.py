"""oracle/ -- TEST INFRASTRUCTURE ONLY (the parity checker), never imported by the product.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this package.

Contents
  _ref/            the UNMODIFIED reference engine compiled from /root/reference by oracle/Makefile
                   (git-ignored build output; shipped to the GPU box like any built .so)
  ref_shim.cpp     extern "C" harness around the reference's own 7 entry points (nudge.h:134-146)
  exact_intrinsics.h   re-routes the 4 hardware-approximate rsqrt/rcp intrinsics to IEEE sequences
  refworld.py      ctypes binding: load a scene, step it stage by stage, read state back
Parity pinning: oracle/_ref/ref_tests_{exact,shipped} are the reference's own tests/main.cpp
(6 known-answer / property tests), which pass against both variants (`make -C oracle check`).
"""
