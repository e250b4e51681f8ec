"""sph_hip_run wall time per step of the dam-break column at several sizes on one GPU
(profiles/design_history_r1_r3.md section 5):  python tools/size_sweep.py [sizes...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smoothed_particle_hydrodynamics_amd as S
from smoothed_particle_hydrodynamics_amd import scenes

sizes = [int(a) for a in sys.argv[1:]] or [262144, 1048576, 4194304, 16777216, 67108864]
for n in sizes:
    box = (1.0, 1.0, 8.0) if n == 67108864 else (1.0, 1.0, 1.0)
    p, pos, vel, mass = scenes.dam_break(n, box)
    with S.SPH(n, p) as sph:
        sph.setParticles(pos, vel, mass)
        del pos, vel, mass
        sph.setTiming(S.TIMING_OFF)
        sph.run(10)
        sph.synchronize()
        k = 100 if n <= 4194304 else 20
        t0 = time.perf_counter()
        sph.run(k)
        sph.synchronize()
        dt = (time.perf_counter() - t0) / k
        ts = sph.tileStats()
        print("%9d particles, box %s: %8.3f ms/step = %6.2f G particle-steps/s  (tile caps %d/%d, largest tile %d)" % (
            n, "x".join("%g" % b for b in box), dt * 1e3, n / dt / 1e9, ts["capacity_density"],
            ts["capacity_acceleration"], ts["largest_tile"]), flush=True)
