"""Density / acceleration phase time of the 4M-particle column with the LDS tile capacity pinned
(SPH_HIP_TILE_CAP) from 2016 to 2752 entries: the steps in the times are the capacities at which a
pass loses a workgroup per CU, i.e. the LDS allocation granularity of the device (DESIGN.md 4.1,
csrc/sph_hip.hip tile_levels).  One process per capacity."""
import sys, time, os, subprocess
if os.environ.get("CAP_SWEEP_CHILD"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import smoothed_particle_hydrodynamics_amd as S
    from smoothed_particle_hydrodynamics_amd import scenes
    n=4194304
    p,pos,vel,mass=scenes.dam_break(n)
    sph=S.SPH(n,p); sph.setParticles(pos,vel,mass)
    sph.run(3); sph.synchronize(); sph.resetTimings()
    for s in range(10): sph.step()
    sph.synchronize()
    t,k=sph.phaseTotals()
    print("cap %s density %.1f accel %.1f us" % (os.environ.get("SPH_HIP_TILE_CAP"), t[2]/k*1e3, t[4]/k*1e3), flush=True)
else:
    for cap in [int(a) for a in sys.argv[1:]] or range(2016, 2760, 32):
        env=dict(os.environ, SPH_HIP_TILE_CAP=str(cap), CAP_SWEEP_CHILD="1")
        subprocess.run([sys.executable, __file__, "x"], env=env, timeout=200)
