"""Timeline of a dam that actually breaks (apply_gravity + apply_walls, 4M particles): step time,
phase split, neighbour counts, tile statistics every 51 steps."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smoothed_particle_hydrodynamics_amd as S
from smoothed_particle_hydrodynamics_amd import scenes
n=4194304
p,pos,vel,mass=scenes.dam_break(n)
p.apply_gravity=1; p.apply_walls=1
p.gravity[0],p.gravity[1],p.gravity[2]=0.0,-9.81,0.0
sph=S.SPH(n,p); sph.setParticles(pos,vel,mass)
for blk in range(16):
    t0=time.perf_counter(); sph.run(50); sph.synchronize(); dt=(time.perf_counter()-t0)/50
    sph.resetTimings(); sph.step(); sph.synchronize(); t,k=sph.phaseTotals()
    part=sph.getParticles()
    c=part.mNeighborCount
    ts=sph.tileStats()
    print("steps %4d: %.3f ms/step [build %.2f dens %.2f acc %.2f]  nb mean %.1f max %d >254: %d >510: %d p50 %d p90 %d p99 %d | wg %d largest tile %d caps %d/%d untiled %d/%d" % ((blk+1)*51, dt*1e3, t[0],t[2],t[4], c.mean(), c.max(), (c>254).sum(), (c>510).sum(), int(np.percentile(c,50)), int(np.percentile(c,90)), int(np.percentile(c,99)), ts["workgroups"], ts["largest_tile"], ts["capacity_density"], ts["capacity_acceleration"], ts["untiled_density"], ts["untiled_acceleration"]) + (" wide" if ts["wide_entries"] else ""), flush=True)
