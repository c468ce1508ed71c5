#!/usr/bin/env python3
"""bench.py -- steps/s and contacts solved/s of the stepping hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one full pass of the hot path (collide -> gravity -> read cache -> setup -> apply x I -> update ->
write cache -> advance, reference example/main.cpp:274-328) over one world resident in HBM.
Workload at N = 1: BASELINE.json configs[1] -- 1,004,400 boxes in 124 single-layer drop tiles, 8 solver
iterations, fp32.  For N > 1 (config 5's shape, weak scaling) the world is ONE lattice of 124 x N tiles cut into N
x-slabs of equal body count -- the cuts pass through tiles -- and every rank steps its slab with a ghost-body halo
exchange per step over RCCL point-to-point (nudge_amd/partition.py); there is no collective on the data path.
`value` = (1,004,400-box partitions stepped by all ranks) per second.  NUDGE_BENCH_HALO=0 falls back to N
independent 124-tile worlds (no exchange at all).

The JSON line also carries
  roofline     -- the dominant kernel's algorithmic bytes / its mean duration (HIP events on the launch
                  stream, recorded during the timed region) against the 8 TB/s HBM peak;
  cpu_baseline -- the compiled reference (oracle/_ref) timed on this box's host cores on a bounded sample of
                  the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)

# Algorithmic bytes per unit for each kernel (DESIGN.md "Kernels"; itemisation of SURVEY.md 8(d)).
# unit: B bodies, C colliders, P pairs, K contacts, KI contact-iterations, R contact records (= pairs with contacts)
KERNEL_BYTES = {
    "xform_aabb": ("C", 32 + 32 + 16 + 4 + 32 + 32 + 4),   # local+body transform, shape, tag -> world transform, AABB, tag
    "cell_keys": ("C", 32 + 8 + 8),                        # AABB -> cell key, value, morton key
    "cell_bounds": ("C", 8 + 32 + 32),                     # key/val + AABB -> AABB in cell order
    "find_pairs": ("C", 32 + 4),                           # AABB in cell order + key (+ 8 B per emitted pair)
    "narrowphase": ("P", 8 + 2 * (32 + 16 + 4) + 4 * 36 + 24),  # pair, 2 x (transform, shape, tag) -> ~4 contacts x 36 B + record
    "gather_contacts": ("K", 36 + 52),                     # raw contact + feature -> contact, pair, tag, feature
    "cache_lookup": ("K", 12 + 28 + 16),                   # tag+feature, cache probe (tag, feature, impulse) -> impulse
    "setup_static": ("K", 32 + 16 + 8 + 160 + 16 + 36),    # contact, cached impulse, pair -> row + state (+ body state/4)
    "apply_static": ("K", 160 + 32 + 36),                  # row + state rw (+ body state/4) for ALL iterations of one call
    # fused one-body solver (lookup + rows + warm start + all sweeps + export), per contact: contact 32, pair 8, tag+feature 12,
    # cache probe 12 + 16, state 16 w, impulse 16 w = 112; per body (4 contacts): class/offsets/adjacency/hint 33, transform 32,
    # properties 16, momentum 32 r + 32 w = 145 -> 36 per contact
    "solve_one_body": ("K", 112 + 36),
    "update_impulses": ("K", 160 + 16 + 16),
    "write_cache": ("K", 12 + 16 + 28),
    "radix_hist": ("N", 8),
    "radix_scatter": ("N", 24),
    "radix_pass": ("P", 24),                               # one 8-bit pass over the pair records: key 8 + index 4, read and written
    "advance": ("B", 32 + 32 + 1 + 32 + 1),
    "gravity_damping": ("B", 64 + 4),
}


# timer name (nh_* launch sites) -> kernel symbol prefix in the rocprofv3 PMC export (profiles/export_pmc.py)
PMC_KERNEL = {"solve_one_body": "k_solve_one_body<4, true, true>", "apply_static": "k_apply_static<4, true>", "narrowphase": "k_narrowphase",
              "gather_contacts": "k_gather_contacts", "find_pairs": "k_find_pairs", "radix_pass": "os_pass<unsigned long, unsigned int, true>",
              "apply_level": "k_level_exec<false>", "write_cache": "k_write_cache", "xform_aabb": "k_xform"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC export (FETCH_SIZE x 2 + WRITE_SIZE, see
    profiles/export_pmc.py and MI355X_MICROARCH.md): measured in a separate rocprofv3 --pmc run of this same command."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), key=os.path.getmtime)
    sym = PMC_KERNEL.get(kernel)
    if not files or not sym:
        return None, None
    try:
        k = json.load(open(files[-1]))["kernels"].get(sym)
    except Exception:
        return None, None
    return (k["hbm_bytes_corrected"], os.path.basename(files[-1])) if k else (None, None)


def step_algorithmic_bytes(B, C, P, K, I):
    """A = 146 B + 260 C + 96 P + 504 K + 128 K I   (bytes per step)."""
    return 146.0 * B + 260.0 * C + 96.0 * P + 504.0 * K + 128.0 * K * I


def cpu_baseline(scene, warmup, tiles_total, budget_s=20.0):
    """Compiled reference on the host cores: `cores` tiles in parallel, same phase of the simulation."""
    from nudge_amd import scenes as S
    from oracle import refworld
    if not refworld.available("exact"):
        return None
    cores = os.cpu_count() or 1
    ntiles = min(cores, tiles_total)
    worlds = []
    for t in range(ntiles):
        sub, _ = S.extract_tile(scene, t)
        worlds.append(refworld.RefWorld(sub, variant="shipped" if refworld.available("shipped") else "exact", ftz=True))
    refworld.step_tiles(worlds, warmup, threads=cores, ftz=True)          # untimed: let the boxes land
    probe = refworld.step_tiles(worlds, 10, threads=cores, ftz=True)
    steps = int(max(10, min(2000, budget_s / max(probe / 10.0, 1e-6))))
    wall = refworld.step_tiles(worlds, steps, threads=cores, ftz=True)
    tile_steps_per_s = ntiles * steps / wall
    value = tile_steps_per_s / tiles_total                               # steps/s of the whole 124-tile world
    per_tile = len(scene['body_transforms']) // max(1, tiles_total)
    return {"value": value, "unit": "steps/s", "cores": cores, "kind": "reference",
            "sample": f"{ntiles} of {tiles_total} tiles ({per_tile} bodies each) x {steps} steps after {warmup + 10} warm-up steps, "
                      f"one tile per host thread, unmodified reference -O3 -mavx2 -mfma, FTZ/DAZ on; scaled by tiles",
            "tile_steps_per_s_per_core": tile_steps_per_s / cores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--tiles", type=int, default=124, help="tiles per GPU (124 = config 2)")
    ap.add_argument("--side", type=int, default=90)
    ap.add_argument("--scene", default="boxes", choices=["boxes", "mixed", "pit"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--profile-kernels", action="store_true", help="also print the per-kernel time table (stderr)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    # NUDGE_BENCH_BACKEND=gloo + NUDGE_BENCH_ONE_DEVICE=1: rehearsal of the N > 1 path on a box with ONE GPU (all ranks share
    # cuda:0, halo messages staged through the host).  The driver's runs use neither: one rank per GPU over RCCL.
    backend = os.environ.get("NUDGE_BENCH_BACKEND", "nccl")
    device_index = 0 if os.environ.get("NUDGE_BENCH_ONE_DEVICE", "0") == "1" else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from nudge_amd import scenes as S
    from nudge_amd import engine as E

    halo = world > 1 and os.environ.get("NUDGE_BENCH_HALO", "1") != "0" and args.scene in ("boxes", "mixed")
    sphere_fraction = 0.5 if args.scene == "mixed" else 0.0
    cluster = None
    if halo:
        # ONE world of world x tiles; equal-count x-slabs whose cuts pass through tiles
        from nudge_amd import partition as PT
        # 4 x world + 1 columns of tiles (~30 rows at 124 tiles per GPU; the odd column count puts the equal-count cuts INSIDE tiles,
        # not in the gaps between them): a compact 2-D lattice like config 5's, so that coordinates stay
        # small enough for fp32 (a 1-D strip of 88 columns would put boxes at x = 24,000 where one ulp is 2 mm)
        g = S.grid_tiles(args.tiles * world, side=args.side, sphere_fraction=sphere_fraction, seed=2, lattice_cols=4 * world + 1)
        gx = g["body_transforms"]["position"][1:, 0]
        cuts = [float(np.quantile(gx, (r + 1) / world)) + 0.41 for r in range(world - 1)]
        scene, info = PT.split_scene(g, cuts, rank)
        del g, gx
        sharding = (f"{world} x-slabs of one {args.tiles * world}-tile world, ghost halo exchange every step "
                    f"({'RCCL' if backend == 'nccl' else backend} p2p), refresh + migration every 16 steps")
    elif args.scene in ("boxes", "mixed"):
        # every rank owns its own lattice of independent tiles (different seed per rank)
        scene = S.grid_tiles(args.tiles, side=args.side, sphere_fraction=sphere_fraction, seed=(2 if args.scene == "boxes" else 3) + 1000 * rank,
                             lattice_cols=11 if args.scene == "boxes" else 16)
        sharding = "independent tiles per rank, no data-path collective" if world > 1 else "single GPU"
    else:
        scene = S.ball_pit(args.side, args.side, args.side, seed=4 + rank)
        sharding = "independent replicas" if world > 1 else "single GPU"
    n_units = args.tiles * args.side * args.side
    workload = {"boxes": f"c2: {n_units:,} boxes in {args.tiles} single-layer drop tiles ({args.side}x{args.side}), 8 solver iterations, dt 1/120",
                "mixed": f"c3: {n_units:,} mixed boxes+spheres in {args.tiles} tiles",
                "pit": f"c4: {args.side ** 3:,}-sphere ball pit, 16 iterations"}[args.scene]
    I = scene["params"]["iterations"]
    nb = len(scene["body_transforms"]) - 1
    if halo:
        w = E.World(scene, device=device_index, flags=0, capacity=info["capacity"], tag_bits=info["tag_bits"], max_contacts=6 * info["capacity"]["bodies"])
        cluster = PT.DistCluster(PT.Partition(w, info, epoch=16), host_staging=(backend != "nccl"))
    else:
        w = E.World(scene, device=device_index, flags=0, max_contacts=max(4096, (6 if args.scene != "pit" else 10) * nb))

    def advance_world(n):
        if cluster is not None:
            cluster.step(n)
        else:
            w.step(n)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (untimed): boxes land; also one fully profiled step to find the dominant kernel ----
    advance_world(max(args.warmup - 3, 0))
    w.enable_timing(True)
    advance_world(min(3, args.warmup))
    w.synchronize()
    ktimes = w.kernel_times(reset=True)
    w.enable_timing(False)
    dominant = max(ktimes.items(), key=lambda kv: kv[1][0])[0] if ktimes else None
    if args.profile_kernels and rank == 0:
        tot = sum(v[0] for v in ktimes.values())
        for k, (ms, n) in sorted(ktimes.items(), key=lambda kv: -kv[1][0]):
            print(f"  {k:24s} {ms / 3:9.4f} ms/step  {n // 3:4d} launches/step  {100 * ms / tot:5.1f} %", file=sys.stderr)
    counts0 = w.counts()

    # ---- timed region: exactly K steps, only the dominant kernel carries events ----
    if dominant:
        w.enable_timing(True, only=dominant)
    barrier()
    t0 = time.perf_counter()
    advance_world(args.steps)
    barrier()
    t1 = time.perf_counter()
    dt = t1 - t0
    dom = w.kernel_times(reset=True).get(dominant) if dominant else None
    w.enable_timing(False)
    counts1 = w.counts()
    if counts1["error"]:
        raise SystemExit(f"device error {counts1['error']}")

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        ksum = torch.tensor([float(counts1["contacts"])], dtype=torch.float64, device=red_dev)
        dist.all_reduce(ksum, op=dist.ReduceOp.SUM)
        contacts_all = float(ksum.item())
    else:
        contacts_all = float(counts1["contacts"])

    if rank == 0:
        K = 0.5 * (counts0["contacts"] + counts1["contacts"])
        P = 0.5 * (counts0["pairs"] + counts1["pairs"])
        C = counts1["colliders"]
        B = nb + 1
        steps_per_s = world * args.steps / dt
        out = {
            "metric": "simulation steps/s (contacts solved/s alongside), 1M-box drop scene",
            "value": steps_per_s,
            "unit": "steps/s" if world == 1 else "partition-steps/s (one 1,004,400-box partition per GPU)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "bodies_per_gpu": nb, "colliders_per_gpu": C, "iterations": I,
                       "sharding": sharding,
                       "contacts_per_step": K, "pairs_per_step": P, "timed_window": f"steps [{args.warmup}, {args.warmup + args.steps})"},
            "contacts_solved_per_sec": contacts_all * I * args.steps / dt,
        }
        A = step_algorithmic_bytes(B, C, P, K, I)
        out["step_roofline"] = {"bound": "hbm", "algorithmic_bytes_per_step": A, "achieved": A / (dt / args.steps) / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": A / (dt / args.steps) / 1e9 / HBM_PEAK_GBS}
        if dom and dominant in KERNEL_BYTES:
            unit, per = KERNEL_BYTES[dominant]
            units = {"B": B, "C": C, "P": P, "K": K, "N": K}[unit]
            launches_per_step = dom[1] / args.steps
            ms = dom[0] / dom[1]
            bytes_per_launch = per * units
            traffic, traffic_src = pmc_traffic(dominant) if args.scene == "boxes" and args.tiles == 124 else (None, None)
            out["roofline"] = {"bound": "hbm", "kernel": dominant, "achieved": bytes_per_launch / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": bytes_per_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                               "avg_launch_ms": ms, "launches_per_step": launches_per_step,
                               "algorithmic_bytes_per_launch": bytes_per_launch, "share_of_step": dom[0] / (1e3 * dt)}
        elif dom:
            out["roofline"] = {"bound": "hbm", "kernel": dominant, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                               "avg_launch_ms": dom[0] / dom[1]}
        if world == 1 and not args.no_cpu and args.scene in ("boxes", "mixed"):
            try:
                cb = cpu_baseline(scene, args.warmup, args.tiles)
            except Exception as e:      # the baseline is a report, never a reason to lose the GPU number
                cb = {"error": repr(e)}
            if cb:
                out["cpu_baseline"] = cb
        if ktimes:
            # per-kernel milliseconds per step from the three fully instrumented warm-up steps (events around EVERY launch: the sum exceeds
            # ms_per_step, small kernels are inflated; the timed region only instruments the dominant kernel)
            out["kernel_ms_per_step"] = {k: round(ms / 3.0, 5) for k, (ms, n) in sorted(ktimes.items(), key=lambda kv: -kv[1][0])[:16]}
        if cluster is not None:
            st = cluster.p.stats
            out["halo"] = {"ghost_bodies_rank0": st["ghosts"], "refreshes": st["refreshes"], "migrated_out_rank0": st["migrated_out"],
                           "bytes_sent_per_step_rank0": st["step_bytes"] / max(1, cluster.p.steps)}
        print(json.dumps(out))
    w.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
