"""The breaking 4M-particle dam (gravity + walls, tolerance mode) in windows of 20 steps from step 0 to 600:
ms per step in each window - what a change costs or gains over the whole transient, not only at the two
steps the bench quotes (SPH_HIP_LIBRARY selects the build)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smoothed_particle_hydrodynamics_amd as S
from smoothed_particle_hydrodynamics_amd import scenes
n = 4194304
p, pos, vel, mass = scenes.dam_break(n)
p.apply_gravity = 1
p.apply_walls = 1
p.gravity[0], p.gravity[1], p.gravity[2] = 0.0, -9.81, 0.0
out = []
with S.SPH(n, p, mode=S.MODE_FULL_FAST) as sph:
    sph.setParticles(pos, vel, mass)
    sph.setTiming(S.TIMING_OFF)
    for w in range(0, 600, 20):
        sph.synchronize()
        t0 = time.perf_counter()
        sph.run(20)
        sph.synchronize()
        out.append((time.perf_counter() - t0) / 20 * 1e3)
print(os.path.basename(os.environ.get("SPH_HIP_LIBRARY", "default")), "total %.1f ms |" % (sum(out) * 20),
      " ".join("%.2f" % x for x in out), flush=True)
