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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", type=int, default=4 * 1024 * 1024,
                    help="total particles (default: BASELINE config C3, 4M)")
    ap.add_argument("--cpu-sample", type=int, default=131072,
                    help="particles in the CPU-baseline sample (0 disables)")
    ap.add_argument("--cpu-steps", type=int, default=2)
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


def main():
    args = parse_args()
    import torch
    import smoothed_particle_hydrodynamics_amd as S
    from smoothed_particle_hydrodynamics_amd import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                         "--nproc-per-node %d" % (args.gpus, world, args.gpus))
    if world > 1:
        raise SystemExit("multi-GPU slab decomposition is not wired into bench.py yet")

    torch.cuda.set_device(local_rank)
    S.build_library()
    n = args.particles
    p, pos, vel, mass = scenes.dam_break(n)
    sph = S.SPH(n, p, mode=S.MODE_FULL, device=local_rank)
    sph.setParticles(pos, vel, mass)

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

    totals, covered = sph.phaseTotals()
    # phases: 0 voxelize(cell build) 1 findNeighbors 2 density 3 pressure 4 acceleration 5 integrate
    df_ms = (totals[2] + totals[4]) / covered
    achieved = DENSITY_FORCE_BYTES * n / (df_ms * 1e-3) / 1e9
    nb_mean = float(sph.getParticles().mNeighborCount.mean())
    ke, pe = sph.energy()

    line = {
        "metric": "Mparticle-steps/sec (whole node), dam-break",
        "value": n * args.steps / dt / 1e6,
        "unit": "Mparticle-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "dam-break %d particles (BASELINE configs[2], C3) in the unit box, fp32, "
                        "FULL neighbour mode, cell grid rebuilt every step" % n,
            "particles": n,
            "h": float(p.h),
            "grid": [p.full_cells_x, p.full_cells_y, p.full_cells_z],
            "neighbors_mean": nb_mean,
            "parallelism": "1 GPU" if world == 1 else "slab x%d" % world,
        },
        "phases_ms": {
            "cell_build": totals[0] / covered, "density": totals[2] / covered,
            "acceleration": totals[4] / covered, "integrate": totals[5] / covered,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "density+acceleration pass (k_full_density + k_full_accel)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "bytes_per_particle": DENSITY_FORCE_BYTES,
            "ms_per_launch_pair": df_ms,
        },
    }
    if args.cpu_sample > 0:
        line["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, n), args.cpu_steps, n)
    assert np.isfinite(ke) and np.isfinite(pe)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
