#!/usr/bin/env python3
"""bench.py -- steps/s and contacts solved/s of the stepping hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the driver's way) the ranks are the launcher's; started
plainly, `python bench.py --gpus N` re-executes itself under that launcher with N ranks (self_launch).  It refuses when the node shows fewer than N devices or when a
launcher's WORLD_SIZE disagrees with --gpus: no line ever reports fewer GPUs than were asked for.  `--dry-launch` proves the launch alone (gloo, no GPU).

A "step" is one full pass of the hot path (collide -> gravity -> read cache -> setup -> apply x I -> update ->
write cache -> advance, reference example/main.cpp:274-328) over one world resident in HBM.

WHAT IS TIMED.  The metric is quoted on the LANDED, AWAKE scene (while the boxes fall there are no
contacts, once they sleep there are none either, nudge.cpp:4896-4898).  So whatever --warmup says, the bench first
steps (untimed) until the scene has landed -- contacts >= 95 % of the scene's resting count and every body awake --
and fails loudly if that never happens; --warmup steps follow (untimed); then K steps are timed between
barrier + synchronize.  At N = 1 that K-step block is repeated from a device-side checkpoint of the landed state
until about three seconds have been timed (the awake window is only ~250 steps long): restore, ONE untimed step (a
restore rewrites body state and cache from outside, after which the library looks at everything again for a step --
an artefact of the checkpoint, not of the workload), then exactly the same K timed steps in every block;
`ms_per_step` is the mean over all of them.  `landed_window` times 200 consecutive steps of the same window in one
piece.  A line whose timed window saw no contacts is refused.

Workload at N = 1: BASELINE.json configs[1] -- 1,004,400 boxes in 124 single-layer drop tiles, 8 solver
iterations, fp32.  For N > 1 (config 5's shape) the world is ONE lattice of 248 x N tiles (N = 8: the 1984 tiles /
16 M boxes of configs[4]) cut into N x-slabs of equal body count -- cuts pass through tiles -- and every rank
steps its slab with a ghost-body halo exchange per step over RCCL point-to-point (nudge_amd/partition.py); there is no
collective on the data path.  `value` = bodies stepped per second by all ranks / 1,004,400: steps/s of a config-2 sized
world, the same unit at every N.  NUDGE_BENCH_HALO=0 falls back to independent worlds per rank (no exchange at all).

The JSON line also carries
  roofline     -- the dominant kernel's algorithmic bytes / its mean duration (HIP events on the launch
                  stream, recorded during the timed region) against the 8 TB/s HBM peak;
  step_traffic -- sum of the algorithmic bytes of every kernel of a step / ms_per_step against the same peak (and
                  the same with the PMC-measured bytes when profiles/ holds them);
  cpu_baseline -- the compiled reference (oracle/_ref) timed on this box's host cores on a bounded sample of
                  the same workload, all cores and one core (rank 0, N = 1 only).
"""
import argparse
import glob
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EPOCH = 32                     # steps between two refreshes of the partitioned world (migration, ghost lists): nudge_amd/partition.py
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)
C2_BODIES = 1004400

# Algorithmic bytes per unit for each kernel (DESIGN.md "Kernels"; itemisation of SURVEY.md 8(d)).
# unit: B bodies, C colliders, P pairs, K contacts, R contact records (= pairs), G general contacts, T grid table entries
KERNEL_BYTES = {
    "collide_begin": ("B", 12 + 1),                        # degree / cursor / info words cleared, idle counter read
    "xform_aabb": ("C", 32 + 32 + 16 + 4 + 32 + 32 + 4),   # local+body transform, shape, tag -> world transform, AABB, tag
    "cell_keys": ("C", 32 + 8),                            # AABB -> cell key, rank in the cell
    "cell_scan": ("T", 6),                                 # per-cell counts -> starts: read twice, written once, over two launches
    "cell_scatter": ("C", 8 + 32 + 4 + 32),                # key/rank + AABB -> key + 32-byte AABB record in cell order
    "find_pairs": ("C", 32 + 4),                           # AABB in cell order + key (+ 8 B per emitted pair)
    "large_pairs": ("P", 32 + 8),                          # candidates of the large colliders: AABB read, pair written
    "kept_filter": ("P", 8 + 2 * 32 + 8),                  # per kept pair (~ per pair at rest): the pair, two exact AABBs (gathered), the pair written
    "narrowphase": ("P", 8 + 2 * (32 + 16 + 4) + 4 * 36 + 24),  # pair, 2 x (transform, shape, tag) -> ~4 contacts x 36 B + record
    "narrowphase_bb": ("P", 8 + 2 * (32 + 16 + 4) + 4 * 36 + 24),
    "narrowphase_sph": ("P", 8 + 2 * (32 + 4 + 4) + 36 + 24),
    "bucket_count": ("R", 8 + 8),                          # key -> bucket, rank
    "bucket_scatter": ("R", 8 + 12 + 12),
    "bucket_sort": ("R", 12 + 12),
    "sorted_counts": ("R", 4 + 16 + 8),
    "gather_contacts": ("K", 36 + 52),                     # raw contact + feature -> contact, pair, tag, feature
    "cache_lookup": ("K", 12 + 28 + 16),                   # tag+feature, cache probe (tag, feature, impulse) -> impulse
    "gravity_damping": ("B", 4 + 32 + 32),
    "adjacency_simple": ("B", 4 + 4 + 4 + 1 + 8),          # degree, pair info, first contact -> class, contact list record
    # fused one-body solver (lookup + rows + warm start + all sweeps + export), priced by what the instantiation bench.py runs MOVES
    # (k_solve_one_body<4, true, CONTIG, 1> under NH_FLAG_SINGLE_APPLY | NH_FLAG_FUSED_STEP; checked against the ISA and the PMC write side):
    #   per contact: contact 32 r, tag 8 r, feature 4 r, cache probe (tag 8 + feature 4 + impulse 16) r, impulse 16 w               = 72 r + 16 w
    #   per body (4 contacts): class 1, contact-list record 8, hint 4 r + 4 w, transform 32 r + 32 w, properties 16, momentum 32 r + 32 w,
    #                          idle 1 r + 1 w                                                                      = 94 r + 69 w -> 23.5 r + 17.25 w
    #   NOT moved in this mode: the body pair (the CONTIG instantiation never loads it), the solver state (16 B w, only kept for a further
    #   apply call), CSR offsets / adjacency (the one-record contact list replaces them)
    "solve_one_body": ("K", 72 + 16 + 23.5 + 17.25),       # = 128.75; main() adds the 16-byte state when the world keeps states, drops the fused advance's 8.5 when it does not fuse
    # still steps (include/nudge_hip.h note 9): transforms + AABBs + the idle scan; the narrowphase straight from the kept pair list into the pairs' own raw slots;
    # the fused solver from the slots and the slot cache
    "xform_still": ("C", 32 + 32 + 16 + 4 + 32 + 32 + 4 + 32 + 1),      # as xform_aabb, + the inflated box it checks (32 r), the idle counter (1 r)
    "narrowphase_still": ("P", 8 + 2 * 32 + 2 * (32 + 16 + 4) + 8 + 4 + 4 + 4 * 36 + 4 + 4),   # kept pair, 2 AABBs, 2 x (transform, shape, tag), key, tag-order position, body record -> 4 contacts x 36 B, count x 2
    "narrowphase_sph_still": ("P", 8 + 2 * 32 + 2 * (32 + 4 + 4) + 8 + 4 + 4 + 36 + 4 + 4),
    #   per contact: contact 32 r, cached impulse 16 r, solved impulse 16 w                                                                = 48 r + 16 w
    #   per body (4 contacts): record 4, tag-order position 4, count 4, first contact 4, 4 feature words 16, cached feature words 16 r + 16 w, cached count 4 r + 4 w,
    #                          transform 32 r + 32 w, properties 16, momentum 32 r + 32 w, idle 1 r + 1 w                                   = 133 r + 85 w -> 33.25 r + 21.25 w
    "solve_still": ("K", 48 + 16 + 33.25 + 21.25),         # = 118.5
    # the ONE-KERNEL still step (nh_fused.hip), per dynamic body with four contacts -- what the lane MOVES (the static partner's transform / box / shape are shared by the
    # ~8100 bodies of a tile: cache hits, not priced):
    #   position table 16 r; own collider: box 32 + world transform 32 + tag 4 + shape 16 r; record: key 8, count 4, start 4, cached count 4 r + 4 w, cached feature words
    #   16 r + 16 w, cached impulses 64 r, solved impulses 64 w; body: momentum 32 r + 32 w, properties 16 r, transform 32 r + 32 w, idle 1 r + 1 w; next step's transforms
    #   (xform ahead): local transform 32 + size 16 + inflated box 32 r, world transform 32 + box 32 w                                                       = 574 B
    #   a sphere body (one contact): 16 + 16 instead of 64 + 64 impulse bytes, 4 + 4 instead of 16 + 16 feature bytes                                        = 454 B
    # NO contact ever reaches HBM: the two-kernel form writes 4 x 36 B per pair and reads them back (narrowphase_still 352 + solve_still 4 x 118.5 + 148 = 974 B per box body)
    "still_fused": ("B", 574.0), "fused_map": ("R", 4 + 8 + 16 + 16), "fused_copy_static": ("T", 0), "fused_copy_bodies": ("B", 2 * 65),
    "pair_begin": ("T", 0), "pair_owned": ("B", 4),
    "ahead_map": ("C", 16 + 4), "ahead_check": ("C", 16 + 4),      # once per nh_step call (xform ahead): AABB min + body of every collider -> body's collider / several colliders? + the static world's bounds
    "asleep_check": ("C", 32 + 32 + 16 + 4 + 32 + 32 + 4 + 1),   # nh_step on a world asleep: local + body transform, shape, tag, the remembered AABB and tag, the idle counter
    "slot_counts": ("R", 4 + 4 + 4), "slots_to_cache": ("K", 16 + 4 + 28), "cache_to_slots": ("K", 4 + 16 + 4 + 20),
    "gravity_rest": ("B", 4 + 1),                           # active list + class of every body; momentum only of bodies outside the fused solver
    "advance_rest": ("B", 4 + 1),
    "solve_one_body8": ("K", 112 + 36),
    "apply_static": ("K", 160 + 32 + 36),
    "rows_general": ("G", 4 + 32 + 8 + 2 * (32 + 16) + 160 + 8),
    "warm_level": ("G", 160 + 8 + 16 + 16 + 2 * 64),
    "apply_level": ("G", 160 + 8 + 16 + 16 + 2 * 64),      # one sweep of one colour class: rows, pair, state rw, two momentum records rw
    "solve_resident": ("GI", 2 * 64),                      # one workgroup, rows in LDS: per contact-sweep two momentum records rw (plus the rows once)
    "grid_setup": ("T", 0),                                # one wavefront: bounds -> cell size and table dimensions
    "adjacency_from_simple": ("B", 8 + 4 + 4 * 4),         # contact list record -> CSR offsets + entries (only when a general path needs the CSR)
    "solve_colours": ("GI", 2 * 64),                       # persistent coloured sweeps: per contact-iteration two momentum records rw (rows stay on chip)
    "update_impulses": ("G", 160 + 16 + 16),
    "write_cache": ("K", 12 + 16 + 28),
    "advance": ("B", 4 + 32 + 32 + 1 + 32 + 1),
    # spatially blocked solver (nh_blocks.h): per contact and sweep 60 B streamed in block order; and per sweep every momentum record a block holds once in, once
    # out (2 x 32 B): the general bodies as their block's own + the ghost copies the blocks borrow (nh_Counts.blk_bodies / blk_ghosts) -- main() turns that into
    # bytes per contact for the scene at hand
    "blk_sweep": ("GI", 32 + 4 + 24), "blk_warm": ("G", 32 + 4 + 24),   # nh_blocks.h (k_blk_sweep): first 8 row floats, local index pair, 3 state floats in + out = 60 B
    "blk_bounds": ("B", 1 + 12), "blk_body": ("B", 1 + 12 + 8), "blk_contact": ("G", 4 + 8 + 2 * 8 + 8), "blk_scatter": ("G", 4 + 4 + 4 + 4),
    "blk_prepare": ("G", 2 * (4 + 8 + 2 * 8) + 2 * 4 + 16 + 4 + 3 * 4),      # k_blk_prepare_local: two passes over (order, pair, two block ranks), the index pair parked and fetched, the seed, order again; order, index pair and colour out
    "reinsert": ("T", 0), "blk_gather": ("B", 8 + 64), "blk_scatter_back": ("B", 8 + 64), "blk_residual_pairs": ("G", 0),
    "colour_seed": ("G", 4 + 16 + 8), "colour_validate": ("G", 2 * 8), "blk_reset": ("T", 0), "blk_grid": ("T", 0), "zero_u32": ("T", 0),
}


def grid_table_entries(C):
    """Entries of the broadphase grid table for C colliders (nh_collide: a power of two >= 4 C, 2^16 .. 2^24)."""
    want = 1
    while want < 4 * C:
        want <<= 1
    want = max(want, 1 << 16)
    return min(want, 1 << 24)

# timer name (nh_* launch sites) -> kernel symbol prefix in the rocprofv3 PMC export (profiles/export_pmc.py)
PMC_KERNEL = {"still_fused": "k_still_fused", "solve_one_body": "k_solve_one_body<4, true, true, 1, false>", "solve_still": "k_solve_one_body<4, true, true, 1, true, true, true>", "narrowphase_still": "k_narrowphase<false, true>",
              "xform_still": "k_xform<true>", "apply_static": "k_apply_static<4, true>", "narrowphase": "k_narrowphase<false, false>",
              "gather_contacts": "k_gather_contacts", "find_pairs": "k_find_pairs", "radix_pass": "os_pass<unsigned long, unsigned int, true>",
              "apply_level": "k_level_exec<false>", "write_cache": "k_write_cache", "xform_aabb": "k_xform", "blk_sweep": "k_blk_sweep<false, true>"}


AHEAD_BYTES_PER_BODY = 4 + 32 + 16 + 32 + 32 + 32       # xform ahead (main): what a still solver lane moves for the next step's k_xform<true>
# pair ahead (main): what the lane moves MORE to evaluate its body's own kept pair for the next step -- the pair 8 r, the key on file 8 r, two tags 8 r, the record's count 4 w,
# per contact 32 + 4 w (the static partner's transform / box / shape: shared by the ~8100 bodies of a tile, cache hits) -- priced per body with its contacts for the scene at hand
PAIR_BYTES_PER_BODY = 8 + 8 + 8 + 4
PAIR_BYTES_PER_CONTACT = 36
PMC_SUFFIX = ""       # "_c4" for the pit: its own export (profiles/collect.sh)


def _pmc_file():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_traffic{PMC_SUFFIX}.json")))
    return files[-1] if files else None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC export (FETCH_SIZE x 2 + WRITE_SIZE, see
    profiles/export_pmc.py and MI355X_MICROARCH.md): measured in a separate rocprofv3 --pmc run of this same command."""
    f, sym = _pmc_file(), PMC_KERNEL.get(kernel)
    if not f or not sym:
        return None, None
    try:
        ks = json.load(open(f))["kernels"]
        k = ks.get(sym) or next((v for name, v in ks.items() if name.startswith(sym.rstrip(">"))), None)      # (template arguments may have grown)
    except Exception:
        return None, None
    # corrected = 2 x FETCH + WRITE (an upper bound: gathers are not half-counted), raw = FETCH + WRITE (a lower bound); both are reported
    return ((k["hbm_bytes_corrected"], k.get("hbm_bytes_raw")), os.path.basename(f)) if k else (None, None)


# ---- VALU issue ceiling (round 5; VERDICT r4 "a compute ceiling for the dominant kernel") --------------------------------------------------------------------------
# Measured on an MI355X with tools/valu_rate.hip (profiles/r05_valu_rate.log), wave64 fp32 instructions per microsecond and SIMD:
#   independent instructions, two or more waves per SIMD: 810 (the SIMD's issue limit);
#   ONE dependent chain per wave: 247 with one wave per SIMD, 470 with two, 655 with three, 768 with four.
# The fused one-body solver is a dependent chain (Gauss-Seidel inside a body, in the reference's operation order) at two waves per SIMD.
VALU_PEAK_PER_US_PER_SIMD = 810.0
VALU_DEPENDENT_PER_US_PER_WAVE = 247.0
SIMDS = 256 * 4
# ... and by instruction (same log; 8 independent chains, two waves per SIMD): what ONE wave-instruction costs the SIMD's VALU pipe, in microseconds
VALU_COST_US = {"mul_add_sub_fmac": 1 / 950.0,          # v_mul / v_add / v_sub / v_fmac _f32 (VOP2): 927-984 per us
                "fma_vop3": 1 / 826.0,                  # v_fma_f32 and other three-operand encodings
                "mov_minmax_cmp_select": 1 / 560.0,     # v_mov_b32 594, v_max_f32 551, v_bfi_b32 535, v_cndmask_b32 with an SGPR mask 545, v_cmp -> vcc / sgpr 580
                "rcp_rsq": 1 / 287.0,                   # v_rcp_f32 284, v_rsq_f32 290
                "select_through_vcc": 1 / 148.0 - 1 / 583.0}      # v_cmp + v_cndmask_b32_e32 reading VCC: 148 pairs per us (the v_cmp is counted above)
# one sweep of the still solver over a body's four contacts (ISA of k_solve_one_body<4,true,true,1,true>, the loop the landed boxes take: 388 VALU instructions)
SWEEP_MIX = {"mul_add_sub_fmac": 260, "fma_vop3": 40, "mov_minmax_cmp_select": 64, "rcp_rsq": 12, "select_through_vcc": 12}


def valu_ceiling(kernel, launch_ms, waves_per_simd=2):
    """VALU instructions per wave of `kernel` from the newest committed SQ export (SQ_INSTS_VALU / SQ_WAVES, profiles/export_sq.py) against what the SIMDs can issue."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_sq_wave_states.json")))
    sym = PMC_KERNEL.get(kernel)
    if not files or not sym or not launch_ms:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"].get(sym)
        per_wave, waves = k["valu_instructions_per_wave"], k["counters_per_launch"]["SQ_WAVES"]
    except Exception:
        return None
    rate = per_wave * waves / SIMDS / (launch_ms * 1e3)          # wave-instructions per microsecond and SIMD
    # what the instructions cost the VALU pipe, by the measured price of each kind (the sweep's mix stands for the whole kernel: 63 % of its instructions)
    sweep_us = sum(SWEEP_MIX[k] * VALU_COST_US[k] for k in SWEEP_MIX)
    per_instr_us = sweep_us / sum(SWEEP_MIX.values())
    wave_life_us = launch_ms * 1e3 / (waves / (SIMDS * waves_per_simd))
    parked = k.get("parked_at_waitcnt_share_of_wave_cycles") or 0.0
    busy_kernel = waves_per_simd * per_wave * per_instr_us / wave_life_us
    busy_compute_phase = busy_kernel / max(1e-9, 1.0 - parked)
    return {"valu_pipe_busy_by_instruction_prices": busy_kernel, "valu_pipe_busy_while_the_waves_compute": busy_compute_phase,
            "price_model": {"sweep_mix": SWEEP_MIX, "us_of_pipe_per_wave_instruction": VALU_COST_US, "pipe_us_per_sweep_and_wave": sweep_us,
                            "note": "busy = waves per SIMD x instructions per wave x mean price / wave lifetime; 'while the waves compute' divides by the share of the wave's "
                                    "cycles NOT parked at s_waitcnt (SQ) -- the memory phases at the head and tail of a wave, during which its SIMD partner is mostly parked too"},"valu_instructions_per_wave": per_wave, "waves_per_launch": waves, "achieved_wave_instr_per_us_per_simd": rate,
            "peak_wave_instr_per_us_per_simd": VALU_PEAK_PER_US_PER_SIMD, "valu_frac": rate / VALU_PEAK_PER_US_PER_SIMD,
            "dependent_chain_rate_at_this_occupancy": VALU_DEPENDENT_PER_US_PER_WAVE * waves_per_simd,
            "frac_of_dependent_chain_rate": rate / (VALU_DEPENDENT_PER_US_PER_WAVE * waves_per_simd), "waves_per_simd": waves_per_simd,
            "sq_share_of_wave_cycles": {key: k.get(key + "_share_of_wave_cycles") for key in ("active", "valu", "parked_at_waitcnt", "issue_stall")},
            "source": [os.path.basename(files[-1]), "profiles/r05_valu_rate.log (tools/valu_rate.hip)"],
            "note": "valu_frac: of what the SIMD issues when fed independent FMAs -- but this kernel's instructions are not FMAs: priced one by one (mov / min / max / "
                    "compare / select at 1.7 x a multiply, rcp / rsq at 3.3 x, a select through VCC at 6 x) two waves keep the VALU pipe about two thirds busy over the whole "
                    "kernel and ~1.0 in the cycles in which they are not parked at s_waitcnt (valu_pipe_busy_*).  Two experiments say what the rest is NOT: a third wave per "
                    "SIMD (168 VGPRs, 12,288 B LDS; resident by PMC) made every wave 1.55 x slower -- 170 us against 153 (profiles/r05_solver_three_waves_ab.log, "
                    "tools/experiments/solver_three_waves.py); and the same two waves with EVERY load prefetched under the previous group's sweeps (persistent waves, LDS-direct "
                    "loads: tools/experiments/solver_pipelined.py) take 177 us against 171 -- not latency either.  What is left is two throughputs that overlap imperfectly: "
                    "the VALU pipe at this instruction mix and 2.7 - 4.0 TB/s (PMC) moved in 16- and 32-byte pieces (DESIGN.md 7.3)"}


def pmc_step_total():
    """Sum over ALL kernels of the PMC export: HBM bytes of one landed step (each kernel's per-launch mean x its launches per step
    when the export carries them, else x 1)."""
    f = _pmc_file()
    if not f:
        return None, None
    try:
        ks = json.load(open(f))["kernels"]
        return sum(v["hbm_bytes_corrected"] * v.get("launches_per_step", 1) for v in ks.values()), os.path.basename(f)
    except Exception:
        return None, None


def _host_cpu_facts():
    """What the host really offers this process: affinity mask, cgroup CPU quota, model name (so that `cores` means something)."""
    facts = {"os_cpu_count": os.cpu_count()}
    try:
        facts["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        facts["affinity"] = None
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                facts["cgroup_cpu_max"] = " ".join(txt)
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
                facts["cgroup_cpu_max"] = f"{int(q)} {int(per)}"
                if q > 0:
                    quota = q / per
            break
        except Exception:
            continue
    facts["cgroup_quota_cores"] = quota
    try:
        model, sockets, cores_per, threads_per = None, set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                sockets.add(line.split(":", 1)[1].strip())
            elif line.startswith("cpu cores") and cores_per is None:
                cores_per = int(line.split(":", 1)[1])
            elif line.startswith("siblings") and threads_per is None:
                threads_per = int(line.split(":", 1)[1])
        facts["cpu_model"] = model
        if cores_per and sockets:
            facts["physical_cores"] = cores_per * len(sockets)
            facts["smt"] = (threads_per // cores_per) if threads_per and cores_per else None
    except Exception:
        pass
    return facts


def cpu_baseline(scene, land_steps, tiles_total, budget_s=20.0):
    """Compiled reference on the host cores, same phase of the simulation (landed, awake).  The reference is single-threaded and re-entrant for
    disjoint worlds, so the host baseline is "one tile per thread"; how many threads actually help is MEASURED: a thread-scaling
    curve (1, 2, 4, ... threads, every point the same landed tiles restored from a snapshot and stepped for the same phase), and the best point is
    the quoted value with ITS thread count as `cores`.  Also: one core stepping the tiles one after another (what the unmodified reference does)."""
    from nudge_amd import scenes as S
    from oracle import refworld
    if not refworld.available("exact"):
        return None
    variant = "shipped" if refworld.available("shipped") else "exact"
    facts = _host_cpu_facts()
    usable = facts.get("affinity") or os.cpu_count() or 1
    if facts.get("cgroup_quota_cores"):
        usable = max(1, min(usable, int(math.ceil(facts["cgroup_quota_cores"]))))
    # (one point BEYOND the quota, when the affinity mask allows it: the curve must go flat there -- that is what tells a CPU quota from a slow reference)
    over = min(facts.get("affinity") or usable, 2 * usable) if facts.get("cgroup_quota_cores") else usable
    ntiles = max(1, min(over, tiles_total))
    worlds = []
    for t in range(ntiles):
        sub, _ = S.extract_tile(scene, t)
        worlds.append(refworld.RefWorld(sub, variant=variant, ftz=True))
    refworld.step_tiles(worlds, land_steps + 10, threads=ntiles, ftz=True)          # untimed: let the boxes land
    snaps = [w.snapshot() for w in worlds]
    per_tile = len(scene['body_transforms']) // max(1, tiles_total)
    points, t = [], 1
    while t < min(usable, ntiles):
        points.append(t)
        t *= 2
    points.append(min(usable, ntiles))
    points.append(ntiles)
    if facts.get("physical_cores") and 1 < facts["physical_cores"] < ntiles and facts["physical_cores"] not in points:
        points.append(facts["physical_cores"])                                      # one thread per physical core (no SMT sharing)
    points = sorted(set(points))
    per_point = budget_s / len(points)
    curve = []
    for t in points:
        sub = worlds[:t]
        for w, sn in zip(sub, snaps):
            w.restore(sn)
        probe = refworld.step_tiles(sub, 4, threads=t, ftz=True)
        steps = int(max(8, min(120, 0.8 * per_point / max(probe / 4.0, 1e-6))))    # <= 124 more steps: inside the awake window
        wall = refworld.step_tiles(sub, steps, threads=t, ftz=True)
        curve.append({"threads": t, "tile_steps_per_s": t * steps / wall, "steps": steps, "per_thread": steps / wall})
    best = max(curve, key=lambda c: c["tile_steps_per_s"])
    one = curve[0]
    return {"value": best["tile_steps_per_s"] / tiles_total, "unit": "steps/s", "cores": best["threads"], "kind": "reference",
            "sample": f"best point of a thread-scaling curve: {best['threads']} tiles ({per_tile} bodies each) on {best['threads']} host threads x {best['steps']} steps after "
                      f"{land_steps + 10} untimed steps (landed, awake), one tile per thread, unmodified reference ({variant}) -O3 -mavx2 -mfma, FTZ/DAZ on; "
                      f"whole scene = {tiles_total} tiles: value = tile steps/s / {tiles_total}",
            "host": facts,
            "thread_scaling": [{"threads": c["threads"], "tile_steps_per_s": round(c["tile_steps_per_s"], 2), "per_thread": round(c["per_thread"], 2),
                                "efficiency_vs_one_thread": round(c["per_thread"] / one["per_thread"], 3)} for c in curve],
            "tile_steps_per_s_per_core": best["tile_steps_per_s"] / best["threads"],
            "one_core": {"value": one["tile_steps_per_s"] / tiles_total, "unit": "steps/s", "cores": 1,
                         "sample": f"1 tile x {one['steps']} steps on one thread (tile steps/s {one['tile_steps_per_s']:.1f}); whole scene = {tiles_total} tiles one after another"}}


def single_gpu_rate(tiles, side, device_index, steps, min_time=1.5):
    """Landed-window rate of ONE independent world of `tiles` drop tiles on this GPU, in bench.py's N = 1 mode and unit (config-2-sized worlds stepped
    per second): the same-load per-GPU reference the N > 1 line is compared with (rank 0 measures it after the timed region, the other ranks wait)."""
    import torch
    from nudge_amd import scenes as S
    from nudge_amd import engine as E
    scene = S.grid_tiles(tiles, side=side, seed=2, lattice_cols=11 if tiles == 124 else None)
    nb = len(scene["body_transforms"]) - 1
    w = E.World(scene, device=device_index, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP, max_contacts=max(4096, 6 * nb))
    done, landed = 0, False
    while done < 400 and not landed:
        w.step(4); done += 4
        c = w.counts()
        landed = c["contacts"] >= 0.95 * 4.0 * nb and c["active_bodies"] >= nb
    if not landed:
        w.close()
        return None
    w.step(5); w.synchronize()
    snap = w.snapshot()
    dt, blocks = 0.0, 0
    while dt < min_time and blocks < 5000:
        if blocks:
            w.restore(snap)
        w.counts(); torch.cuda.synchronize()
        t0 = time.perf_counter(); w.step(steps); torch.cuda.synchronize(); dt += time.perf_counter() - t0
        blocks += 1
    w.close()
    sec = dt / (blocks * steps)
    return {"value": nb / C2_BODIES / sec, "unit": "config-2-sized worlds stepped per second", "bodies": nb, "tiles": tiles, "ms_per_step": 1e3 * sec,
            "timed_steps": blocks * steps, "landed_after_steps": done}


def committed_single_gpu_16M():
    """The 16 M-box world of config 5 stepped as ONE world on ONE GPU (python bench.py --gpus 1 --tiles 1984), as committed under profiles/: the
    denominator of the north star's ">= 6 x single-GPU throughput at 8 GPUs"."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_c5_single_gpu_16M.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "bodies": d["config"]["bodies_per_gpu"], "source": os.path.basename(files[-1])}
    except Exception:
        return None


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n, dry):
    """Re-executes this command line under `python -m torch.distributed.run` with n ranks on this node (one per GPU) and never returns.  Refuses when the node
    shows fewer than n devices -- except for --dry-launch, which touches no GPU -- so `--gpus 8` can never come back as a one-GPU number."""
    if not dry and os.environ.get("NUDGE_BENCH_ONE_DEVICE", "0") != "1":
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            raise SystemExit(f"bench.py --gpus {n}: {have} HIP device(s) visible on this node -- refusing to run (nothing is measured on fewer GPUs than asked for)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execvpe(sys.executable, cmd, env)


def dry_launch(rank, world):
    """--dry-launch: the ranks of this invocation find each other (gloo, CPU) and rank 0 prints what it saw.  The launch path of `--gpus N` without a GPU."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.zeros(world, dtype=torch.int64)
        t[rank] = 1 + int(os.environ.get("LOCAL_RANK", "0"))
        dist.all_reduce(t)
        seen, locals_ = int((t > 0).sum()), [int(v) - 1 for v in t.tolist()]
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen, locals_ = 1, [0]
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks_seen": seen, "local_ranks": locals_,
                          "launched_by": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "environment"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5, help="extra untimed steps AFTER the scene has landed")
    ap.add_argument("--tiles", type=int, default=None, help="tiles per GPU (default 124 = config 2 at one GPU, 248 = config 5's share otherwise)")
    ap.add_argument("--side", type=int, default=None)
    ap.add_argument("--scene", default="boxes", choices=["boxes", "mixed", "pit"])
    ap.add_argument("--min-time", type=float, default=3.0, help="seconds to time at N = 1 (the K-step block is repeated from a checkpoint)")
    ap.add_argument("--no-full-window", action="store_true", help="skip the steps [0, 1000) figure of SURVEY 8(d) (drop scenes, N = 1)")
    ap.add_argument("--land-cap", type=int, default=None, help="give up if the scene has not landed after this many steps")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--profile-kernels", action="store_true", help="also print the per-kernel time table (stderr)")
    ap.add_argument("--dry-launch", action="store_true", help="only prove the launch: every rank joins a gloo group, rank 0 prints how many ranks it saw (no GPU needed)")
    args = ap.parse_args()

    # ---- launch: `python bench.py --gpus N` with no torch.distributed.run around it starts its own N ranks (missing 1: it used to step ONE GPU and
    # print n_gpus 1).  A launcher that did set the environment must agree with --gpus; nothing here ever measures fewer GPUs than it was asked for.
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and not launched:
        self_launch(args.gpus, args.dry_launch)          # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to measure {world} rank(s) under the name of {args.gpus}")
    if args.dry_launch:
        return dry_launch(rank, world)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    if os.environ.get("NUDGE_BENCH_ONE_DEVICE", "0") != "1" and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) visible: refusing (one rank per GPU; NUDGE_BENCH_ONE_DEVICE=1 is the one-GPU rehearsal)")
    # NUDGE_BENCH_BACKEND=gloo + NUDGE_BENCH_ONE_DEVICE=1: rehearsal of the N > 1 path on a box with ONE GPU (all ranks share
    # cuda:0, halo messages staged through the host).  The driver's runs use neither: one rank per GPU over RCCL.
    backend = os.environ.get("NUDGE_BENCH_BACKEND", "nccl")
    device_index = 0 if os.environ.get("NUDGE_BENCH_ONE_DEVICE", "0") == "1" else local_rank
    torch.cuda.set_device(device_index)
    # NUDGE_BENCH_LOOPBACK=1 (only meaningful at N = 1 under torch.distributed.run): the process group is created although there is one rank, the
    # world goes through the partitioned code path (one slab, no neighbours) and every step sends 4096 halo records to this same rank over the
    # backend's point-to-point ops -- RCCL executed on the one GPU a test box has (tests/test_gpu_partition.py)
    loopback = world == 1 and os.environ.get("NUDGE_BENCH_LOOPBACK") == "1"
    use_dist = world > 1 or loopback
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from nudge_amd import scenes as S
    from nudge_amd import engine as E

    if args.tiles is None:
        args.tiles = 124 if world == 1 else 248
    if args.side is None:
        args.side = 160 if args.scene == "pit" else 90
    halo = use_dist and os.environ.get("NUDGE_BENCH_HALO", "1") != "0" and args.scene in ("boxes", "mixed")
    sphere_fraction = 0.5 if args.scene == "mixed" else 0.0
    cluster = None
    if halo:
        # ONE world of world x tiles (N = 8: 1984 tiles = 16 M boxes, a 44-column lattice like SURVEY 8(d) c5); equal-count x-slabs.  With
        # 44 / N columns per rank some cuts pass through tiles and some along the gaps between them
        from nudge_amd import partition as PT
        total = args.tiles * world
        cols = max(2, int(round(math.sqrt(total) - 0.5)))
        if cols % world == 0:
            cols += 1                                       # never ALL cuts in gaps: at least the odd ones cross tiles
        g = S.grid_tiles(total, side=args.side, sphere_fraction=sphere_fraction, seed=2, lattice_cols=cols)
        gx = g["body_transforms"]["position"][1:, 0]
        cuts = [float(np.quantile(gx, (r + 1) / world)) + 0.41 for r in range(world - 1)]
        scene, info = PT.split_scene(g, cuts, rank, static_margin=64.0)
        del g, gx
        sharding = (f"{world} x-slabs of one {total}-tile world ({total * args.side * args.side:,} bodies), ghost halo exchange every step "
                    f"({'RCCL' if backend == 'nccl' else backend} p2p), refresh + migration every {EPOCH} steps")
    elif args.scene in ("boxes", "mixed"):
        # every rank owns its own lattice of independent tiles (different seed per rank)
        scene = S.grid_tiles(args.tiles, side=args.side, sphere_fraction=sphere_fraction, seed=(2 if args.scene == "boxes" else 3) + 1000 * rank,
                             lattice_cols=11 if args.tiles == 124 else None)
        sharding = "independent tiles per rank, no data-path collective" if world > 1 else "single GPU"
    else:
        scene = S.ball_pit(args.side, args.side, args.side, seed=4 + rank)
        sharding = "independent replicas" if world > 1 else "single GPU"
    n_units = args.tiles * args.side * args.side
    workload = {"boxes": (f"c2: {n_units:,} boxes in {args.tiles} single-layer drop tiles ({args.side}x{args.side}), 8 solver iterations, dt 1/120" if args.tiles == 124 else
                          f"c5 on ONE GPU: {n_units:,} boxes in {args.tiles} drop tiles stepped as one world (no partition), 8 solver iterations, dt 1/120") if world == 1 else
                         f"c5: {n_units * world:,} boxes in {args.tiles * world} drop tiles cut into {world} x-slabs, 8 solver iterations",
                "mixed": f"c3: {n_units:,} mixed boxes+spheres in {args.tiles} tiles",
                "pit": f"c4: {args.side ** 3:,}-sphere ball pit, 16 iterations"}[args.scene]
    I = scene["params"]["iterations"]
    nb = len(scene["body_transforms"]) - 1
    n_box_dyn = int((scene["box_transforms"]["body"] > 0).sum())
    n_sph_dyn = int((scene["sphere_transforms"]["body"] > 0).sum())
    if halo:
        # (the partitioned step keeps the sample's call order -- the halo arrives before nh_collide -- so gravity and advance fuse into the solver here too)
        w = E.World(scene, device=device_index, flags=E.NH_FLAG_SINGLE_APPLY | (0 if os.environ.get("NUDGE_BENCH_NO_FUSED") else E.NH_FLAG_FUSED_STEP), capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=6 * info["capacity"]["bodies"])
        cluster = PT.DistCluster(PT.Partition(w, info, epoch=EPOCH), host_staging=(backend != "nccl"), loopback=int(os.environ.get("NUDGE_BENCH_LOOPBACK_RECORDS", "4096")) if loopback else 0)
        # the per-step halo driven by the library over its own RCCL communicator (pack, ncclSend / ncclRecv, unpack in one C call); checked with a pattern exchange
        # under a time-out and agreed on by all ranks -- any doubt keeps the torch.distributed point-to-point transport (NUDGE_BENCH_RCCL_DIRECT=0 forces that)
        if backend == "nccl" and os.environ.get("NUDGE_BENCH_RCCL_DIRECT", "1") != "0":
            cluster.use_rccl_direct(device_index, loopback=loopback)
        # (the sub-steps between two refreshes in ONE library call -- nh_partition_step -- whenever the transport is the library's; NUDGE_BENCH_MULTI_STEP=0: one call per sub-step, round 5's way)
        cluster.multi_step = os.environ.get("NUDGE_BENCH_MULTI_STEP", "1") != "0"
    else:
        w = E.World(scene, device=device_index, flags=E.NH_FLAG_SINGLE_APPLY | E.NH_FLAG_FUSED_STEP, max_contacts=max(4096, (6 if args.scene != "pit" else 5) * nb),
                    max_pairs=None if args.scene != "pit" else 7 * nb)

    def advance_world(n):
        if cluster is not None:
            cluster.step(n)
        else:
            w.step(n)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def all_ranks(flag):
        if not use_dist:
            return flag
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    snap0 = w.snapshot() if (cluster is None and world == 1 and args.scene != "pit" and not args.no_full_window) else None      # step 0, for the [0, 1000) figure

    # ---- untimed: step until the scene has LANDED and is still awake (the configuration the metric is quoted on) ----
    if args.scene == "pit":
        k_rest, chunk, cap = 2.6 * nb, 20, args.land_cap or 900             # the settled pit holds ~2.8 contacts per sphere
    else:
        k_rest, chunk, cap = 0.95 * (4.0 * n_box_dyn + 1.0 * n_sph_dyn), 4, args.land_cap or 400   # a resting box: 4 face contacts, a sphere: 1
    landed_at = None
    done = 0
    while done < cap:
        advance_world(chunk)
        done += chunk
        c = w.counts()
        if c["error"]:
            raise SystemExit(f"device error {c['error']} while landing")
        # (a partition counts its ghosts among the active bodies; its own share of the resting contacts is what matters)
        ok = c["contacts"] >= k_rest and (halo or args.scene == "pit" or c["active_bodies"] >= nb)
        if all_ranks(ok):
            landed_at = done
            break
    if landed_at is None:
        raise SystemExit(f"bench.py: the scene did not land within {cap} steps (contacts {c['contacts']} < {k_rest:.0f}): refusing to time free fall")
    advance_world(args.warmup)
    w.synchronize()
    first_timed = landed_at + args.warmup

    # ---- one fully instrumented pass over 3 landed steps: per-kernel table, dominant kernel (state restored afterwards) ----
    # N > 1 (round 5): every rank checkpoints ITS OWN bodies (owned + ghosts) and contact cache on its device; a block is restored from it on all ranks between two
    # barriers.  Membership (who owns what, who is whose ghost) is not part of the checkpoint -- in the landed drop scene it does not change -- and the partition's step
    # counter runs on across blocks, so the refresh every `EPOCH` steps falls INSIDE timed blocks at its true duty cycle instead of inside or outside one 20-step window by chance.
    snap = w.snapshot() if (cluster is None or args.scene == "boxes") else None
    owned0 = cluster.p.n_owned if cluster is not None else None

    def restore_block(b):
        """Back to the checkpoint.  A partition's ghosts are not part of it (which bodies a neighbour lists changes from refresh to refresh with the top speeds): the
        restored step counter makes the untimed head step of the block a REFRESH, which installs the ghosts of the restored owners; the counter then continues from a
        phase that walks through the epoch block by block, so that over the timed blocks a refresh falls inside a block at its true rate (K / EPOCH per block)."""
        w.restore(snap)
        if cluster is not None:
            cluster.p.steps = EPOCH * 4096

    w.enable_timing(True)
    advance_world(3)
    w.synchronize()
    ktimes = w.kernel_times(reset=True)
    w.enable_timing(False)
    dominant = max(ktimes.items(), key=lambda kv: kv[1][0])[0] if ktimes else None
    # (per-kernel timing of EVERY launch keeps the library out of its pipelined loop, where the steady step is ONE launch -- the one-kernel still step, nh_fused.hip: a probe
    # of a few untimed steps tells whether that is what the timed region will run; its events then go around that kernel)
    fused_probe = 0
    if cluster is None and snap is not None and dominant in ("solve_still", "narrowphase_still"):
        w.restore(snap); w.step(1)
        cp0 = w.counts(); w.step(12); cp1 = w.counts()
        fused_probe = cp1.get("fused_steps", 0) - cp0.get("fused_steps", 0)
        if fused_probe >= 6:
            dominant = "still_fused"
    if args.profile_kernels and rank == 0:
        tot = sum(v[0] for v in ktimes.values())
        for k, (ms, n) in sorted(ktimes.items(), key=lambda kv: -kv[1][0]):
            print(f"  {k:24s} {ms / 3:9.4f} ms/step  {n / 3:6.1f} launches/step  {100 * ms / tot:5.1f} %", file=sys.stderr)
    if snap is not None and cluster is None:
        w.restore(snap)
    elif snap is None:
        first_timed += 3

    # ---- timed region: blocks of exactly K steps between barrier + synchronize; only the dominant kernel carries events ----
    if dominant:
        w.enable_timing(True, only=dominant)
    dt, blocks, contact_sum, pair_sum, general_sum = 0.0, 0, 0.0, 0.0, 0.0
    fused_sum = 0          # timed steps that were ONE launch (nh_fused.hip)
    pairstep_sum = 0       # timed steps that started at the solver (pair ahead)
    ahead_sum = 0          # timed steps that started at the narrowphase: the step before them did their k_xform<true> inside its solver (include/nudge_hip.h note 9, xform ahead)
    block_ms = []
    timed_acc = [0.0, 0, 0.0, 0]          # inside the timed blocks only: seconds in refreshes, refreshes, seconds enqueueing per-step halos, per-step halos
    while True:
        if snap is not None and cluster is not None and not all_ranks(cluster.p.n_owned == owned0):
            snap = None          # (a body changed owner: the checkpoint no longer describes this rank's slots -- no more restores, the blocks timed so far stand)
            if blocks:
                break
        if snap is not None and (blocks or cluster is not None):
            restore_block(blocks)
        if snap is not None:
            # one UNTIMED step at the head of every block: restoring a checkpoint (an artefact of this measurement: it rewrites body state and cache from outside) makes
            # the library look at everything again for one step, which a world stepping through its landed window never does.  Every block -- the first one too --
            # therefore times the same K steps, the ones after that step.
            advance_world(1)
            if cluster is not None:
                cluster.p.steps = EPOCH * 4096 + 1 + (blocks * args.steps) % EPOCH
        c0 = w.counts()
        barrier()
        if cluster is not None:
            acc0 = (cluster.t_refresh, cluster.n_refresh, cluster.t_halo, cluster.n_halo)
        t0 = time.perf_counter()
        advance_world(args.steps)
        barrier()
        t1 = time.perf_counter()
        if cluster is not None:
            timed_acc = [a + (b - c) for a, b, c in zip(timed_acc, (cluster.t_refresh, cluster.n_refresh, cluster.t_halo, cluster.n_halo), acc0)]
        c1 = w.counts()
        if c1["error"]:
            raise SystemExit(f"device error {c1['error']}")
        dt += t1 - t0
        block_ms.append(1e3 * (t1 - t0) / args.steps)
        blocks += 1
        contact_sum += 0.5 * (c0["contacts"] + c1["contacts"])
        pair_sum += 0.5 * (c0["pairs"] + c1["pairs"])
        general_sum += 0.5 * (c0["general_contacts"] + c1["general_contacts"])
        ahead_sum += c1.get("ahead_steps", 0) - c0.get("ahead_steps", 0)
        fused_sum += c1.get("fused_steps", 0) - c0.get("fused_steps", 0)
        pairstep_sum += c1.get("pair_steps", 0) - c0.get("pair_steps", 0)
        if snap is None or all_ranks(dt >= args.min_time) or blocks >= 5000:          # (N > 1: every rank leaves after the same block)
            break
    dom = w.kernel_times(reset=True).get(dominant) if dominant else None
    w.enable_timing(False)
    if snap is not None:
        first_timed += 1
    # the landed, awake window in ONE piece (SURVEY 8(d)'s sub-window): 200 consecutive steps from the same checkpoint, no restore, no host synchronisation inside
    landed_window = None
    if snap is not None and args.scene != "pit" and world == 1:
        w.restore(snap); advance_world(1); barrier()
        t0 = time.perf_counter(); advance_world(200); barrier(); t_lw = time.perf_counter() - t0
        c_lw = w.counts()
        landed_window = {"steps": f"[{first_timed}, {first_timed + 200})", "steps_per_s": 200.0 / t_lw, "ms_per_step": 1e3 * t_lw / 200.0, "seconds": t_lw,
                         "all_awake_at_the_end": bool(c_lw["active_bodies"] >= nb), "contacts_at_the_end": c_lw["contacts"],
                         "note": "200 consecutive steps of the landed, awake world timed as one block (the K-step blocks above repeat the first K of them)"}
    timed_steps = blocks * args.steps
    K, P, G = contact_sum / blocks, pair_sum / blocks, general_sum / blocks

    per_rank = None
    if use_dist:
        mine = torch.tensor([dt] + [float(v) for v in timed_acc], dtype=torch.float64, device=red_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [[float(v) for v in e.tolist()] for e in every]
        tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        sums = torch.tensor([K, float(nb if not halo else cluster.p.n_owned)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        contacts_all, bodies_all = float(sums[0].item()), float(sums[1].item())
    else:
        contacts_all, bodies_all = K, float(nb)

    if rank == 0:
        if contacts_all <= 0:
            raise SystemExit("bench.py: no contacts in the timed window: not a measurement of the hot path")
        C = c1["colliders"]
        B = nb + 1
        sec_per_step = dt / timed_steps
        if args.scene == "boxes":
            value = bodies_all / C2_BODIES / sec_per_step       # = steps/s at N = 1
            unit = "steps/s" if (world == 1 and args.tiles == 124) else "steps/s x (bodies / 1,004,400): config-2-sized worlds stepped per second by all GPUs"
        else:
            value, unit = world / sec_per_step, "steps/s"
        out = {
            "metric": "simulation steps/s (contacts solved/s alongside), 1M-box drop scene",
            "value": value, "unit": unit,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * sec_per_step, "world_steps_per_s": 1.0 / sec_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "bodies_per_gpu": nb, "colliders_per_gpu": C, "iterations": I,
                       "sharding": sharding, "bodies_all_gpus": bodies_all,
                       "contacts_per_step": K, "pairs_per_step": P, "general_contacts_per_step": G,
                       "landed_after_steps": landed_at,
                       "timed_window": f"steps [{first_timed}, {first_timed + args.steps}) (landed and awake), timed {blocks}x from a device checkpoint (one untimed step after every restore)" if snap is not None
                                       else f"steps [{first_timed}, {first_timed + args.steps}) (landed and awake)",
                       "timed_blocks": blocks, "timed_steps": timed_steps, "timed_seconds": dt,
                       # the same K steps block after block: how steady the clock is over the timed seconds (ms per step of the first, fastest, slowest, last block)
                       "block_ms_per_step": {"first": block_ms[0], "min": min(block_ms), "max": max(block_ms), "last": block_ms[-1]}},
            "contacts_solved_per_sec": contacts_all * I / sec_per_step,
        }
        units = {"B": B, "C": C, "P": P, "K": K, "R": P, "G": G, "GI": G * I, "T": grid_table_entries(C)}
        if args.scene == "pit":
            # sphere contacts in default order use the 48-byte radial rows: 112 bytes less per contact than the 160-byte general row
            for name in ("rows_general", "warm_level", "apply_level", "update_impulses"):
                KERNEL_BYTES[name] = (KERNEL_BYTES[name][0], KERNEL_BYTES[name][1] - 112)
            # blocked sweeps: + the momentum records a sweep loads and stores -- every general body once as its block's own, and once more per block that borrows it as a
            # ghost (nh_Counts.blk_bodies + blk_ghosts of the last timed step: the block tables are rebuilt every step)
            blk_records = float(c1.get("blk_bodies", 0) + c1.get("blk_ghosts", 0)) or float(nb)
            out["config"]["blocked_solver"] = {"blocks": c1.get("blk_blocks"), "bodies_binned": c1.get("blk_bodies"), "ghost_copies": c1.get("blk_ghosts")}
            for name in ("blk_sweep", "blk_warm"):
                KERNEL_BYTES[name] = (KERNEL_BYTES[name][0], KERNEL_BYTES[name][1] + 64.0 * blk_records / max(G, 1.0))
        # the one-body solver as this world runs it (see KERNEL_BYTES): + 16 B of solver state per contact when a further apply call may follow,
        # - the fused advance's transform / idle traffic when gravity and advance are separate kernels
        fused = bool(w.flags & E.NH_FLAG_FUSED_STEP)
        single = fused or bool(w.flags & E.NH_FLAG_SINGLE_APPLY)
        KERNEL_BYTES["solve_one_body"] = ("K", KERNEL_BYTES["solve_one_body"][1] + (0.0 if single else 16.0) - (0.0 if fused else 8.5))
        # the still solver with XFORM AHEAD: in `ahead_frac` of its launches every lane also does the next step's k_xform<true> for its body's collider -- collider index
        # 4 r, local transform 32 r, size 16 r, inflated box 32 r, world transform 32 w, AABB 32 w = 148 B per body (the body's own transform is in registers already:
        # 37 B less than k_xform<true> moves per collider) -- priced per contact for the scene at hand
        ahead_frac = min(1.0, ahead_sum / float(timed_steps)) if timed_steps else 0.0
        if ahead_frac > 0.0 and K > 0:
            KERNEL_BYTES["solve_still"] = ("K", KERNEL_BYTES["solve_still"][1] + AHEAD_BYTES_PER_BODY * ahead_frac * nb / K)
        # PAIR AHEAD: in `pair_frac` of the timed steps the narrowphase was not launched at all -- the solver lanes of the step before had evaluated every body's own pair
        # (their launches carry the extra bytes: the launches that DID the evaluation are as many as the steps that profited, one step earlier)
        pair_frac = min(1.0, pairstep_sum / float(timed_steps)) if timed_steps else 0.0
        if pair_frac > 0.0 and K > 0:
            KERNEL_BYTES["solve_still"] = ("K", KERNEL_BYTES["solve_still"][1] + pair_frac * (PAIR_BYTES_PER_BODY * nb / K + PAIR_BYTES_PER_CONTACT))
        out["config"]["pair_ahead"] = {"timed_steps_started_at_the_solver": pairstep_sum, "share_of_timed_steps": pair_frac,
                                       "bytes_added_to_solve_still": {"per_body": PAIR_BYTES_PER_BODY, "per_contact": PAIR_BYTES_PER_CONTACT},
                                       "note": "a plain still step's solver lane also evaluates its body's own kept pair for the next sub-step (nh_step, note 9): that step is ONE solver launch behind a one-workgroup prologue"}
        fused_frac = min(1.0, fused_sum / float(timed_steps)) if timed_steps else 0.0
        # the one-kernel still step, priced for the scene at hand: per dynamic box body 574 B, per dynamic sphere body 454 B (KERNEL_BYTES), per body of the world
        if n_box_dyn + n_sph_dyn:
            KERNEL_BYTES["still_fused"] = ("B", (574.0 * n_box_dyn + 454.0 * n_sph_dyn) / float(nb + 1))
        out["config"]["fused_still"] = {"timed_steps_that_were_one_launch": fused_sum, "share_of_timed_steps": fused_frac,
                                        "note": "a plain still step that starts at the narrowphase and is not the last sub-step of its nh_step call is ONE launch (nh_fused.hip): the lane that solves "
                                                "a body evaluates the body's kept pair itself; fused steps come in pairs (body state is written out of place), the sub-steps a call begins and ends with are the two- / three-launch form"}
        out["config"]["xform_ahead"] = {"timed_steps_started_at_the_narrowphase": ahead_sum, "share_of_timed_steps": ahead_frac,
                                        "bytes_per_body_added_to_solve_still": AHEAD_BYTES_PER_BODY,
                                        "note": "a plain still step's solver does the next sub-step's world transforms / AABBs / box test / scene bounds (nh_step, note 9): that step has no k_xform<true>"}
        # whole step: sum of the per-kernel algorithmic bytes over the kernels the instrumented pass saw, against the peak
        # (a kernel launched once per colour class / level moves its share of the items per launch: priced per step, not per launch)
        # A kernel the instrumented pass (3 steps, one synchronisation at its end) saw less than once per step runs once per SYNCHRONISATION, not per step: the views
        # of the contact list and the cache a world in the still regime exports when somebody looks (include/nudge_hip.h note 9) -- the timed region looks once per block
        alg, unpriced = 0.0, []
        for name, (_ms, launches) in ktimes.items():
            kb = KERNEL_BYTES.get(name)
            if kb is None:
                unpriced.append(name)
                continue
            weight = 1.0 if launches >= 3 else 1.0 / args.steps
            # (the instrumented pass runs outside the pipelined loop: every still step there launches k_xform<true> and the narrowphase -- the timed steps did so only
            # when they did not start further down the chain)
            if name == "xform_still":
                weight *= max(0.0, 1.0 - ahead_frac)
            elif name in ("narrowphase_still", "narrowphase_sph_still"):
                weight *= max(0.0, 1.0 - pair_frac)
            alg += weight * kb[1] * units[kb[0]] * (I if name == "apply_level" else 1)      # ("GI" units already carry the iterations)
        if fused_frac > 0.0:
            # `alg` so far prices the launches of the instrumented pass, which never fuses: the timed steps were `fused_frac` one-launch steps and the rest as priced
            alg = fused_frac * KERNEL_BYTES["still_fused"][1] * units["B"] + (1.0 - fused_frac) * alg
        pmc_sum, pmc_src = pmc_step_total() if args.scene == "boxes" and args.tiles == 124 and world == 1 else (None, None)
        pmc_frac = (pmc_sum / sec_per_step / 1e9 / HBM_PEAK_GBS) if pmc_sum else None
        pmc_refused = None
        if pmc_frac is not None and pmc_frac > 1.0:
            # more bytes per second than the chip can move: the export's launches_per_step are wrong (round 5 counted steps by k_xform dispatches, which xform ahead
            # removed from most steps) -- an impossible figure is not printed
            pmc_refused = f"{pmc_src}: {pmc_sum:.4g} B per step would be {pmc_frac:.2f} x the HBM peak -- the export's step count is wrong; figure withheld"
            pmc_sum, pmc_frac = None, None
        out["step_traffic"] = {"bound": "hbm", "algorithmic_bytes_per_step": alg, "achieved": alg / sec_per_step / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": alg / sec_per_step / 1e9 / HBM_PEAK_GBS, "unpriced_kernels": sorted(unpriced),
                               "pmc_bytes_per_step": pmc_sum, "pmc_frac": pmc_frac, "pmc_source": pmc_src, "pmc_refused": pmc_refused}
        if dom and dominant in KERNEL_BYTES and units.get(KERNEL_BYTES[dominant][0]) is not None:
            unit_k, per = KERNEL_BYTES[dominant]
            launches_per_step = dom[1] / timed_steps
            ms = dom[0] / dom[1]
            n_items = units[unit_k]
            if dominant in ("apply_level", "warm_level"):
                n_items = n_items * (I if dominant == "apply_level" else 1) / max(1.0, launches_per_step)      # one colour class per launch
            elif dominant in ("blk_sweep", "blk_warm"):
                n_items = n_items / max(1.0, launches_per_step)                                                  # one block colour of one sweep per launch
            bytes_per_launch = per * n_items
            # (one committed PMC export per configuration of profiles/collect.sh: c2 -- no suffix --, c3, c4, and the 16 M-box world of config 5 on one GPU)
            global PMC_SUFFIX
            which = {("boxes", 124): "", ("mixed", 248): "_c3", ("boxes", 1984): "_c5"}.get((args.scene, args.tiles)) if args.scene != "pit" else ("_c4" if args.side == 160 else None)
            comparable = world == 1 and which is not None and not any(k.startswith("NH_") for k in os.environ)
            if comparable:
                PMC_SUFFIX = which
            tr, traffic_src = pmc_traffic(dominant) if comparable else (None, None)          # (the committed PMC exports describe the default paths of exactly these scenes)
            traffic, traffic_raw = tr if tr else (None, None)
            out["roofline"] = {"bound": "hbm", "kernel": dominant, "achieved": bytes_per_launch / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": bytes_per_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_raw": traffic_raw,
                               "traffic_note": "PMC bytes per launch: traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 half-counts wide reads: an upper bound where the kernel gathers), "
                                               "traffic_raw = FETCH_SIZE + WRITE_SIZE (lower bound)" if traffic else None,
                               "traffic_source": traffic_src,
                               "avg_launch_ms": ms, "launches_per_step": launches_per_step,
                               "algorithmic_bytes_per_launch": bytes_per_launch, "share_of_step": dom[0] / (1e3 * dt),
                               # the PMC figure `traffic` is an UPPER bound of what the kernel moved: algorithmic bytes above it mean the price list is wrong
                               "pricing_check": (None if not traffic else ("ok" if bytes_per_launch <= 1.02 * traffic else "ALGORITHMIC BYTES EXCEED THE PMC UPPER BOUND: frac overstated")),
                               "avg_launch_note": "avg_launch_ms: HIP events around this kernel only, during the timed region (neighbouring kernels overlap its head and tail); "
                                                  "kernel_ms_per_step: 3 separate steps with events around EVERY launch (each kernel alone on the GPU)"}
            # `bound` stays "hbm": achieved / peak above are the HBM roofline the metric is defined on.  What actually limits a kernel of dependent fp32 arithmetic is
            # said next to it: its VALU issue rate against the SIMD's limit and against the rate of one dependency chain per wave (valu_ceiling)
            vc = valu_ceiling(dominant, ms) if (dominant in ("solve_still", "solve_one_body", "still_fused") and comparable and which == "") else None          # (the SQ export describes c2's launches)
            if vc:
                out["roofline"]["valu"] = vc
                out["roofline"]["limited_by"] = "VALU throughput at this kernel's instruction mix (roofline.valu: the pipe is ~1.0 busy while the waves compute, two thirds of the kernel) next to the bytes it moves in 16- and 32-byte pieces; neither more waves nor prefetching helps (DESIGN.md 7.3)"

        elif dom:
            out["roofline"] = {"bound": "hbm", "kernel": dominant, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                               "avg_launch_ms": dom[0] / dom[1]}
        if landed_window:
            out["landed_window"] = landed_window
        try:
            cc = w.counts()
            out["still_steps"] = {"confirmed": cc["still_steps"], "replayed": cc["still_replays"], "note": "steps since the world was created that went through as still steps (include/nudge_hip.h note 9) / were launched as one and run again in full"}
        except Exception:
            pass
        if world == 1 and not args.no_cpu and args.scene in ("boxes", "mixed"):
            try:
                cb = cpu_baseline(scene, landed_at + args.warmup, args.tiles)
            except Exception as e:      # the baseline is a report, never a reason to lose the GPU number
                cb = {"error": repr(e)}
            if cb:
                out["cpu_baseline"] = cb
        if snap0 is not None:
            # SURVEY 8(d)'s whole-run figure: steps [0, 1000) from the initial state -- fall, landing, rest, falling asleep -- timed as one block with no host
            # synchronisation inside; then the same 1000 steps once more, untimed, with the counters read every 10 steps for sum K_s, sum P_s
            w.restore(snap0); barrier()
            cf0 = w.counts()
            t0 = time.perf_counter(); advance_world(1000); barrier(); t_full = time.perf_counter() - t0
            cf1 = w.counts()
            w.restore(snap0)
            sum_k, sum_p, asleep_at, prev = 0.0, 0.0, None, None
            for s0 in range(0, 1000, 10):
                advance_world(10)
                cc = w.counts()
                cur = (cc["contacts"], cc["pairs"])
                if prev is not None:
                    sum_k += 5.0 * (prev[0] + cur[0]); sum_p += 5.0 * (prev[1] + cur[1])
                else:
                    sum_k += 5.0 * cur[0]; sum_p += 5.0 * cur[1]
                prev = cur
                if asleep_at is None and s0 > landed_at and cc["active_bodies"] < nb:
                    asleep_at = s0 + 10
            out["full_window"] = {"steps": "[0, 1000)", "steps_per_s": 1000.0 / t_full, "seconds": t_full, "sum_contacts": sum_k, "sum_pairs": sum_p,
                                  "contacts_solved_per_sec": sum_k * I / t_full, "first_sleepers_by_step": asleep_at,
                                  # how the 1000 steps went through (include/nudge_hip.h: note 9, nh_step): as still steps of a world at rest; as steps of a world in which every
                                  # body is asleep and nothing has changed (a fixed point of the step: taken as done after one check); the rest in full
                                  "still_steps": cf1.get("still_steps", 0) - cf0.get("still_steps", 0), "asleep_steps": cf1.get("asleep_steps", 0) - cf0.get("asleep_steps", 0),
                                  "note": "whole run from the initial state (free fall, landing, rest, sleep); sums by the trapezoid rule over counters read every 10 steps of an untimed replay"}
        if ktimes:
            # per-kernel milliseconds per step from the three fully instrumented landed steps (events around EVERY launch: the sum exceeds
            # ms_per_step, small kernels are inflated; the timed region only instruments the dominant kernel)
            out["kernel_ms_per_step"] = {k: round(ms / 3.0, 5) for k, (ms, n) in sorted(ktimes.items(), key=lambda kv: -kv[1][0])[:16]}
            out["launches_per_step"] = round(sum(n for _, n in ktimes.values()) / 3.0, 1)
        if world > 1 and args.scene == "boxes" and os.environ.get("NUDGE_BENCH_NO_REFERENCE") != "1":
            # apples to apples for whoever computes scaling from these lines: (1) what ONE GPU does with the bodies one rank has here (independent tiles,
            # no halo), measured now on rank 0's GPU while the other ranks wait; (2) the whole world of this run on ONE GPU, when committed (config 5)
            try:
                ref = single_gpu_rate(args.tiles, args.side, device_index, args.steps)
            except Exception as e:
                ref = {"error": repr(e)}
            out["per_gpu_reference"] = ref
            if ref and ref.get("value"):
                out["per_gpu_reference"]["note"] = f"N = 1 rate at this run's bodies per GPU ({args.tiles} tiles); value / (n_gpus x this) = parallel efficiency at equal load"
                out["efficiency_vs_same_load_single_gpu"] = value / (world * ref["value"])
            one = committed_single_gpu_16M()
            if one and abs(one["bodies"] - bodies_all) <= 0.01 * bodies_all:
                out["single_gpu_same_world"] = one
                out["speedup_vs_single_gpu_same_world"] = value / one["value"]
        if per_rank:
            # what every rank measured for the same blocks (the line's time is the slowest rank's): wall time per step; the refreshes that fell into the timed blocks
            # and what one cost the host (its own device synchronisations included); host time to enqueue the per-step halo (pack, RCCL group, unpack -- no wait)
            out["per_rank"] = [{"rank": r, "ms_per_step": 1e3 * e[0] / timed_steps, "refreshes_in_timed_blocks": int(e[2]), "refresh_ms": (1e3 * e[1] / e[2]) if e[2] else None,
                                # (with nh_partition_step the halo is enqueued by the library inside the call that also steps: there is no host time of its own to report)
                                "halo_enqueue_us_per_step": ((1e6 * e[3] / e[4]) if e[4] else None) if not (cluster is not None and cluster.direct and cluster.multi_step) else None}
                               for r, e in enumerate(per_rank)]
        if cluster is not None:
            st = cluster.p.stats
            out["halo"] = {"backend": backend, "per_step_transport": (("library-driven RCCL inside nh_partition_step (all sub-steps between two refreshes in one call)" if cluster.multi_step else "library-driven RCCL (nh_partition_exchange_step), one call per sub-step") if cluster.direct else f"torch.distributed point-to-point ({cluster.direct_why or 'not asked'})"), "loopback_records_per_step": (cluster.loopback_records / max(1, getattr(cluster, "loopback_steps", 0))) if loopback else None,
                           "ghost_bodies_rank0": st["ghosts"], "refreshes": st["refreshes"], "quiet_refreshes": getattr(cluster, "quiet_refreshes", 0), "migrated_out_rank0": st["migrated_out"],
                           # (bytes and steps over the same window: every step this partition ever took -- `p.steps` is the epoch clock the timed blocks rewind)
                           "bytes_sent_per_step_rank0": st["step_bytes"] / max(1, st["steps_total"]),
                           "loopback_bytes_per_step_through_rccl": (64.0 * cluster.loopback_records / max(1, cluster.loopback_steps)) if (loopback and getattr(cluster, "loopback_steps", 0)) else None}
        print(json.dumps(out))
    w.close()
    if use_dist:
        dist.barrier()                 # (rank 0 may still be measuring the single-GPU reference)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
