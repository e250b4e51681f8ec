import sys, numpy as np
sys.path.insert(0, '/root/repo')
import smoothed_particle_hydrodynamics_amd as S
from smoothed_particle_hydrodynamics_amd import scenes
p, pos, vel, mass = scenes.dam_break(150000)
p.apply_gravity = 1; p.apply_walls = 1
p.gravity[0], p.gravity[1], p.gravity[2] = 0.0, -9.81, 0.0
sph = S.SPH(mass.size, p); sph.setParticles(pos, vel, mass); sph.setTiming(S.TIMING_OFF)
for s in range(400):
    sph.run(1); sph.synchronize()
    ts = sph.tileStats()
    print(s, ts["capacity_density"], ts["capacity_acceleration"], ts["largest_tile"], ts["untiled_density"], ts["untiled_acceleration"], ts["wide_entries"], ts["list_capacity"], flush=True)
print("done")
