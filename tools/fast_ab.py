"""Exact vs tolerance-mode pair arithmetic on ONE box: step time, density / acceleration phase
times, and how far the FAST results are from the exact ones (which equal the CPU oracle bit for
bit, tests/test_gpu_full_mode.py) - on a moving dam-break column, so that the viscous sum is live.
    python tools/fast_ab.py [sizes...]      (on the GPU box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smoothed_particle_hydrodynamics_amd as S  # noqa: E402
from smoothed_particle_hydrodynamics_amd import scenes  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [262144, 4194304]
VARIANTS = [("exact", S.MODE_FULL), ("fast", S.MODE_FULL_FAST)]


def vec_rel(a, b):
    a = a.astype(np.float64).reshape(-1, 3)
    b = b.astype(np.float64).reshape(-1, 3)
    den = np.maximum(np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1))
    den[den == 0] = 1.0
    return np.linalg.norm(a - b, axis=1) / den


for n in sizes:
    p, pos, vel, mass = scenes.dam_break(n, speed=0.05)
    term = float(p.kernel1) * float(p.hscaled6)     # largest single-neighbour density term (m = 1)
    results = {}
    for name, mode in VARIANTS:
        sph = S.SPH(n, p, mode=mode)
        sph.setParticles(pos, vel, mass)
        sph.run(3)                       # moved state
        sph.step()
        part = sph.getParticles()
        results[name] = {k: getattr(part, k).copy() for k in ("mDensity", "mAcceleration", "mNeighborCount",
                                                             "mPosition", "mVelocity")}
        sph.setParticles(pos, vel, mass)
        sph.run(10)
        sph.synchronize()
        K = 60
        t0 = time.perf_counter()
        sph.run(K)
        sph.synchronize()
        wall = (time.perf_counter() - t0) / K * 1e6
        sph.setTiming(S.TIMING_PHASES)
        for _ in range(10):
            sph.step()
        sph.synchronize()
        t, k = sph.phaseTotals()
        ts = sph.tileStats()
        print("%9d %-11s step %7.1f us | build %5.0f density %5.0f accel %5.0f integrate %4.0f | caps %d/%d untiled %d/%d" % (
            n, name, wall, t[0] / k * 1e3, t[2] / k * 1e3, t[4] / k * 1e3, t[5] / k * 1e3,
            ts["capacity_density"], ts["capacity_acceleration"], ts["untiled_density"],
            ts["untiled_acceleration"]), flush=True)
        sph.close()
    ex = results["exact"]
    for name in ("fast",):
        r = results[name]
        rel = vec_rel(r["mAcceleration"], ex["mAcceleration"])
        drho = np.abs(r["mDensity"].astype(np.float64) - ex["mDensity"]) / term
        print("%9d %-11s vs exact after 4 steps: counts equal %s | acc vec_rel max %.3g p99.99 %.3g median %.3g, >1e-4: %d | "
              "density |d|/(k1 m h^6) max %.3g | pos max abs %.3g vel vec_rel max %.3g" % (
                  n, name, np.array_equal(r["mNeighborCount"], ex["mNeighborCount"]), rel.max(),
                  np.quantile(rel, 0.9999), np.median(rel), int((rel > 1e-4).sum()), drho.max(),
                  np.abs(r["mPosition"] - ex["mPosition"]).max(),
                  vec_rel(r["mVelocity"], ex["mVelocity"]).max()), flush=True)
