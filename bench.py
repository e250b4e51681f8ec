#!/usr/bin/env python3
"""Headline benchmark: Mparticle-steps/s of the SPH step on a synthetic dam-break.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one pass of the hot path (cell build + density + acceleration + integrate,
FULL neighbour mode) over the whole particle set, state resident in HBM.  Prints ONE JSON
line (rank 0).  See DESIGN.md §Measurement for the definitions used in `roofline` and
`cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
DENSITY_FORCE_BYTES = 64       # algorithmic bytes per particle of the density+force pass
                               # (density 20 B + force 44 B)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed steps first (the first ~5 steps after an upload run 5-10 %% "
                         "slower than the steady state)")
    ap.add_argument("--particles", type=int, default=4 * 1024 * 1024,
                    help="particles of the 1-GPU workload (default: BASELINE config C3, 4M)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = --particles per GPU in a box N times as long in z "
                         "(one unit box per slab); strong = --particles in total, unit box")
    ap.add_argument("--no-other-scaling", action="store_true",
                    help="N > 1: skip the shorter measurement of the other scaling mode")
    ap.add_argument("--cpu-sample", type=int, default=524288,
                    help="particles in the CPU-baseline sample (0 disables); with --cpu-steps "
                         "sized for ~10 s of single-thread work")
    ap.add_argument("--cpu-steps", type=int, default=12)
    return ap.parse_args()


def cpu_baseline(n_sample, steps, n_total):
    """Times the CPU restatement (oracle/, kind 'port', 1 thread) on a thinner slice of the
    same column: same number density, same h, same neighbour count as the GPU workload."""
    from oracle.oracle import Oracle, OracleParams
    import ctypes as C
    from smoothed_particle_hydrodynamics_amd import scenes
    frac = n_sample / float(n_total)
    # same density: shrink the column's z extent by the particle ratio
    p, pos, vel, mass = scenes.dam_break(n_sample, box=(1.0, 1.0, frac))
    op = OracleParams()
    C.memmove(C.byref(op), C.byref(p), C.sizeof(op))
    orc = Oracle()
    orc.step(op, pos, vel, mass, mode="full")  # untimed: page in, warm caches
    t0 = time.perf_counter()
    for _ in range(steps):
        out = orc.step(op, pos, vel, mass, mode="full")
    dt = time.perf_counter() - t0
    return {
        "value": n_sample * steps / dt / 1e6,
        "unit": "Mparticle-steps/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d-particle slice of the same dam-break column (same density, h, %.1f "
                  "neighbours/particle), %d FULL-mode steps, %.1f s; host has %d cores" % (
                      n_sample, float(out["ncount"].mean()), steps, dt, os.cpu_count()),
    }


def reference_scene(S, steps=8, n=32768):
    """The reference's own compiled src/sph.cpp (oracle/_ref, built where /root/reference exists
    and shipped as a .so) timed on this box on ITS default scene - srand(42) sphere, 32 768
    particles, shipped sampled neighbour search, one thread like the original - next to the HIP
    library in REF mode on the same initial state, with a live check that both end in the same
    bits.  None when the reference build is not present."""
    import ctypes as C
    from oracle import oracle as orc
    if not orc.reference_available():
        return None
    p = S.default_params()
    op = orc.OracleParams()
    C.memmove(C.byref(op), C.byref(p), C.sizeof(op))
    ref = orc.Reference()
    ref.configure(op, n)
    ref.init_sphere()
    s0 = ref.get_state()
    ref.step()                                   # untimed: first touch
    ref.set_state(s0["pos"], s0["vel"], s0["mass"])
    t0 = time.perf_counter()
    for _ in range(steps):
        ref.step()
    dt_cpu = (time.perf_counter() - t0) / steps
    s1 = ref.get_state()
    with S.SPH(n, p, mode=S.MODE_REF) as sph:
        sph.setParticles(s0["pos"], s0["vel"], s0["mass"])
        for _ in range(steps):
            sph.step()
        part = sph.getParticles()
        same = bool(np.array_equal(part.mPosition, s1["pos"]) and
                    np.array_equal(part.mVelocity, s1["vel"]) and
                    np.array_equal(part.mDensity, s1["rho"]) and
                    np.array_equal(part.mNeighborCount, s1["ncount"]))
        sph.setParticles(s0["pos"], s0["vel"], s0["mass"])
        sph.setTiming(S.TIMING_OFF)
        sph.run(5)
        sph.synchronize()
        k = 200
        t0 = time.perf_counter()
        sph.run(k)
        sph.synchronize()
        dt_gpu = (time.perf_counter() - t0) / k
    return {
        "workload": "the reference's default scene: srand(42) sphere, %d particles, shipped "
                    "sampled neighbour search (REF mode)" % n,
        "cpu": {"value": n / dt_cpu / 1e6, "unit": "Mparticle-steps/s", "ms_per_step": dt_cpu * 1e3,
                "cores": 1, "kind": "reference",
                "sample": "%d x SPH::step() of the compiled src/sph.cpp" % steps},
        "gpu": {"value": n / dt_gpu / 1e6, "unit": "Mparticle-steps/s", "ms_per_step": dt_gpu * 1e3},
        "speedup": dt_cpu / dt_gpu,
        "identical_after_steps": steps if same else 0,
    }


def measured_traffic(n):
    """HBM bytes per density+acceleration launch pair from the committed PMC passes
    (profiles/r1_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs,
    FETCH doubled as the gfx950 guide prescribes).  None when no profile matches the workload."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")))
        if prof.get("particles") == n:
            return prof["density_plus_acceleration_hbm_bytes"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def phase_split(ctx, S, step, synchronize, set_timing, phase_totals, steps=5):
    """Per-phase milliseconds over `steps` fully instrumented steps (outside the timed region)."""
    set_timing(S.TIMING_PHASES)
    for _ in range(steps):
        step()
    synchronize()
    t, k = phase_totals()
    return {"cell_build": t[0] / k, "density": t[2] / k, "acceleration": t[4] / k,
            "integrate": t[5] / k, "steps": k}


def run_single(args, S, scenes, torch, local_rank):
    """N = 1: one context holds the whole grid."""
    n = args.particles
    p, pos, vel, mass = scenes.dam_break(n)
    sph = S.SPH(n, p, mode=S.MODE_FULL, device=local_rank)
    sph.setParticles(pos, vel, mass)
    # Timed region: HIP events (on the context's stream) bracket only the density+acceleration
    # pair of every step - each event record costs ~10 us of stream time, so the full per-phase
    # split is taken over a few extra steps after the timed region.
    sph.setTiming(S.TIMING_SUMS)
    for _ in range(args.warmup):
        sph.step()
    sph.synchronize()
    torch.cuda.synchronize()
    sph.resetTimings()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sph.step()
    sph.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pair, covered = sph.phaseTotals()
    totals = phase_split(sph, S, lambda: sph.step(), sph.synchronize, sph.setTiming, sph.phaseTotals)
    totals["pair_ms"] = pair[2] / covered
    nb_mean = float(sph.getParticles().mNeighborCount.mean())
    ke, pe = sph.energy()
    assert np.isfinite(ke) and np.isfinite(pe)
    return p, dt, totals, covered, n, nb_mean, "1 GPU"


def run_slabs(args, S, scenes, torch, rank, world, local_rank, scaling, steps, warmup):
    """N > 1: one z-slab of the cell grid per GPU, neighbour exchange over RCCL (xGMI).

    weak: the column is `world` unit boxes long in z and holds world x --particles (every slab
    is the 1-GPU workload plus its halos); strong: the 1-GPU workload itself is cut in `world`
    slabs.  Every rank derives the same cuts from the z coordinates alone and generates only
    the particles it owns (counter-based PRNG: any subset of the scene on any rank)."""
    import torch.distributed as dist
    from smoothed_particle_hydrodynamics_amd import slab as SL
    n = args.particles * world if scaling == "weak" else args.particles
    box = (1.0, 1.0, float(world)) if scaling == "weak" else (1.0, 1.0, 1.0)
    p, hi = scenes.dam_break_params(n, box)
    z = scenes.box_fill_axis(n, (0.0, 0.0, 0.0), hi, 2)
    planes = SL.plane_of(p, z)
    cuts = SL.plan_cuts(p, z, world)
    hist = np.bincount(planes, minlength=p.full_cells_z)
    cap, msg = SL.slab_capacities(hist, cuts, rank, slack=1.5)
    mine = np.nonzero((planes >= cuts[rank]) & (planes < cuts[rank + 1]))[0]
    del z, planes
    slab = SL.HipSlab(p, cuts[rank], cuts[rank + 1], cap, msg, device=local_rank,
                      has_left=rank > 0, has_right=rank + 1 < world)
    slab.upload(mine.astype(np.uint32), scenes.box_fill_subset(mine, (0.0, 0.0, 0.0), hi),
                np.zeros(3 * mine.size, np.float32), np.ones(mine.size, np.float32),
                all_masses_equal=True)
    # SPH_SLAB_TRANSPORT=host stages the messages through host memory (rehearsal without P2P);
    # =native lets libsph_hip.so issue the RCCL calls itself (no Python in the step loop)
    mode = os.environ.get("SPH_SLAB_TRANSPORT", "torch")
    if mode == "native":
        stepper = SL.NativeSlabStepper(slab, rank, world)
    else:
        transport = (SL.HostStagedTransport if mode == "host" else SL.DistTransport)(rank, world)
        stepper = SL.DistSlabStepper(slab, transport)

    def fence():
        slab.synchronize()
        torch.cuda.synchronize()
        dist.barrier()

    slab.set_timing(S.TIMING_SUMS)
    for _ in range(warmup):
        stepper.step()
    fence()
    slab.reset_timings()
    t0 = time.perf_counter()
    for _ in range(steps):
        stepper.step()
    fence()
    dt_local = time.perf_counter() - t0
    t = torch.tensor([dt_local], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    st = slab.status()
    flags = torch.tensor([st["errors"], st["owned"]], dtype=torch.int64, device="cuda")
    err = flags[:1].clone()
    dist.all_reduce(err, op=dist.ReduceOp.MAX)
    own = flags[1:].clone()
    dist.all_reduce(own, op=dist.ReduceOp.SUM)
    if int(err.item()) != 0 or int(own.item()) != n:
        raise SystemExit("slab run inconsistent: error bits %d, owned %d of %d" %
                         (int(err.item()), int(own.item()), n))
    pair, covered = slab.phase_totals()
    totals = phase_split(slab, S, stepper.step, fence, slab.set_timing, slab.phase_totals)
    totals["pair_ms"] = pair[2] / covered
    d = slab.download()
    nb_mean = float(d["ncount"].mean())
    slab.close()
    return {"params": p, "dt": dt, "totals": totals, "covered": covered, "n": n,
            "n_rank": st["owned"], "neighbors_mean": nb_mean, "steps": steps, "box": box,
            "parallelism": "z-slab x%d, RCCL halo (%s)" % (
                world, {"native": "ncclSend/ncclRecv issued by libsph_hip.so",
                        "host": "host-staged rehearsal"}.get(mode, "torch.distributed P2P"))}


def main():
    args = parse_args()
    import torch
    import smoothed_particle_hydrodynamics_amd as S
    from smoothed_particle_hydrodynamics_amd import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SPH_BENCH_ONE_DEVICE") == "1":   # rehearsal: all ranks share device 0
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                         "--nproc-per-node %d" % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    # The library is built in-tree by __graft_entry__.build(); only a missing one is compiled
    # here, and only when no other rank could be loading it at the same time.
    if not os.path.exists(S.library_path()):
        if world > 1:
            raise SystemExit("libsph_hip.so is missing: run `python -c 'import __graft_entry__ as "
                             "g; g.build()'` before a multi-rank launch")
        S.build_library()
    n = args.particles
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if (os.environ.get("SPH_SLAB_TRANSPORT") == "host" or
                os.environ.get("SPH_BENCH_ONE_DEVICE") == "1"):   # rehearsals: ranks share a GPU
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        dist.barrier()
        r = run_slabs(args, S, scenes, torch, rank, world, local_rank, args.scaling, args.steps,
                      args.warmup)
        p, dt, totals, covered = r["params"], r["dt"], r["totals"], r["covered"]
        n, n_rank, nb_mean, par, box = r["n"], r["n_rank"], r["neighbors_mean"], r["parallelism"], r["box"]
        other = None
        if not args.no_other_scaling:
            # the other scaling mode, shorter, reported beside the headline (never as `value`)
            mode = "strong" if args.scaling == "weak" else "weak"
            o = run_slabs(args, S, scenes, torch, rank, world, local_rank, mode,
                          max(5, args.steps // 2), min(args.warmup, 2))
            other = {"scaling": mode, "particles": o["n"], "box": list(o["box"]),
                     "value": o["n"] * o["steps"] / o["dt"] / 1e6, "unit": "Mparticle-steps/s",
                     "steps": o["steps"], "ms_per_step": o["dt"] / o["steps"] * 1e3}
    else:
        p, dt, totals, covered, n_rank, nb_mean, par = run_single(args, S, scenes, torch,
                                                                  local_rank)
        box, other = (1.0, 1.0, 1.0), None

    if rank == 0:
        # density+acceleration pair: HIP events on the context's stream over the timed steps
        df_ms = totals["pair_ms"]
        achieved = DENSITY_FORCE_BYTES * n_rank / (df_ms * 1e-3) / 1e9
        line = {
            "metric": "Mparticle-steps/sec (whole node), dam-break",
            "value": n * args.steps / dt / 1e6,
            "unit": "Mparticle-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "dam-break %d particles%s in a %gx%gx%g box, fp32, FULL neighbour "
                            "mode, cell grid rebuilt every step" % (
                                n, {262144: " (BASELINE configs[1], C2)",
                                    4194304: " (BASELINE configs[2], C3)",
                                    16777216: " (BASELINE configs[3], C4)"}.get(n, "")
                                if world == 1 or args.scaling == "strong" else
                                " = %d x the 1-GPU workload%s, one unit box per slab along z" % (
                                    world, " (BASELINE configs[2], C3)"
                                    if args.particles == 4194304 else ""),
                                box[0], box[1], box[2]),
                "particles": n,
                "particles_per_gpu": n // world,
                "h": float(p.h),
                "grid": [p.full_cells_x, p.full_cells_y, p.full_cells_z],
                "neighbors_mean": nb_mean,
                "parallelism": par,
            },
            "phases_ms": {
                "cell_build": totals["cell_build"], "density": totals["density"],
                "acceleration": totals["acceleration"], "integrate": totals["integrate"],
                "note": "%d fully instrumented steps after the timed region" % totals["steps"],
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "density+acceleration pass (k_full_density_tiled + k_full_accel_lists)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(n) if world == 1 else None,
                "traffic_unit": "bytes per launch pair (profiles/r1_hbm_traffic.json)",
                "bytes_per_particle": DENSITY_FORCE_BYTES,
                "particles_per_launch": n_rank,
                "ms_per_launch_pair": df_ms,
                "note": "VALU-bound gather-sum: see DESIGN.md (Roofline); frac is against the "
                        "64 B/particle compulsory-traffic figure of SURVEY.md 8(d)",
            },
        }
        if other is not None:
            line["other_scaling"] = other
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, n), args.cpu_steps, n)
            ref_scene = reference_scene(S)
            if ref_scene is not None:
                line["reference_scene"] = ref_scene
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
