"""Which LDS tile capacities a build picks on the 4M column at rest and how many workgroups it leaves without
a tile, plus the phase times of ten instrumented steps (SPH_HIP_LIBRARY selects the build): the quick check
that a change to the kernels' LDS budget did not cost a capacity level (r4 notes 4c)."""
import os, sys
sys.path.insert(0, os.getcwd())
import smoothed_particle_hydrodynamics_amd as S
from smoothed_particle_hydrodynamics_amd import scenes
n = 4194304
p, pos, vel, mass = scenes.dam_break(n)
with S.SPH(n, p, mode=S.MODE_FULL_FAST) as sph:
    sph.setParticles(pos, vel, mass)
    sph.run(30); sph.synchronize()
    print(os.path.basename(os.environ.get("SPH_HIP_LIBRARY","default")), sph.tileStats(), flush=True)
    sph.setTiming(S.TIMING_PHASES)
    for _ in range(10): sph.step()
    sph.synchronize()
    t,k = sph.phaseTotals(); print([round(x/k*1e3,1) for x in t])
