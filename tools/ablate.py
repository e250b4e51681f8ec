"""Kernel time by ablation: build/variants/abl<N>.so are builds with -DSPH_ABLATE=N (csrc/full_tiled.h:
1 no SUM, 2 no TEST/append/SUM, 7 no {v,B} gather, 9 free append); prints the density and
acceleration time of one 4M-particle step for each."""
import sys, time, os, glob, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import smoothed_particle_hydrodynamics_amd as S
    from smoothed_particle_hydrodynamics_amd import scenes
    n=4194304
    p,pos,vel,mass=scenes.dam_break(n)
    sph=S.SPH(n,p); sph.setParticles(pos,vel,mass)
    # one step only per measurement: ablated kernels produce garbage states
    sph.step(); sph.synchronize(); sph.setParticles(pos,vel,mass); sph.resetTimings()
    sph.step(); sph.synchronize()
    t,k=sph.phaseTotals()
    print(os.path.basename(os.environ["SPH_HIP_LIBRARY"]), "density %.1f accel %.1f us"%(t[2]*1e3, t[4]*1e3), flush=True)
else:
    for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build', 'variants', 'abl*.so'))):
        for r in range(2):
            subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, SPH_HIP_LIBRARY=so), timeout=200)
