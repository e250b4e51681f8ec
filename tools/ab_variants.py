"""A/B timing of library builds on ONE box (boxes of the pool differ by several per cent).

Put the candidate libraries under build/variants/*.so (git-ignored, but shipped to the GPU box),
then on the GPU:  python tools/ab_variants.py  - one process per library (the library path is
fixed at first load), two interleaved rounds, sph_hip_run wall time per step at three sizes."""
import sys, time, os, glob, subprocess
# each variant in its own process (the library path is fixed at first load); two rounds, interleaved
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import smoothed_particle_hydrodynamics_amd as S
    from smoothed_particle_hydrodynamics_amd import scenes
    out=[]
    for n in (262144, 1048576, 4194304):
        p,pos,vel,mass=scenes.dam_break(n)
        sph=S.SPH(n,p); sph.setParticles(pos,vel,mass)
        sph.run(5); sph.synchronize()
        K=100
        t3=time.perf_counter(); sph.run(K); sph.synchronize(); t5=time.perf_counter()
        out.append("%d: %.1f us"%(n,(t5-t3)/K*1e6))
        sph.close()
    print(os.path.basename(os.environ.get("SPH_HIP_LIBRARY","default")), " | ".join(out), flush=True)
else:
    for rnd in range(2):
        for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build', 'variants', '*.so'))):
            env=dict(os.environ, SPH_HIP_LIBRARY=so)
            subprocess.run([sys.executable, __file__, "x"], env=env, timeout=200)
