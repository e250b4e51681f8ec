"""Kernel time by ablation: build/variants/abl<N>.so are builds with -DSPH_ABLATE=N
(tools/build_variant.sh abl9 WORK -DSPH_ABLATE=9; csrc/full_tiled.h: 1 no SUM, 2 no TEST/append/SUM,
7 no {v,B} gather, 21 acceleration pass without its pair loop (prologue + epilogue), 23 = 21 without the tile loads, 22 pair loop
without the pair arithmetic (lists, gathers, tile reads and distances stay), 9 no append loop at all, 10 append loop without its store, 11 every list store
to the lane's first word, 12 = 9 with four LDS reads per TEST step instead of six and eight more additions,
14 = 9 with three LDS reads and four packed operations + a byte permute in place of the eight v_alignbit); prints the density and acceleration time of one 4M-particle step for
each.  Ablated kernels compute garbage: only the times mean anything, and only for hooks that do
not change what the later phases have to do (3 / 4 / 5 - no tile load / no row ranges - do, by
starving TEST of candidates)."""
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
    print(os.path.basename(os.environ["SPH_HIP_LIBRARY"]), os.environ.get("SPH_HIP_ARITH", "exact"), "density %.1f accel %.1f us"%(t[2]*1e3, t[4]*1e3), flush=True)
else:
    for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build', 'variants', 'abl*.so'))):
        for r in range(2):
            subprocess.run([sys.executable, __file__, "x"],
                           env=dict(os.environ, SPH_HIP_LIBRARY=so, SPH_HIP_ALLOW_DIAGNOSTIC="1"), timeout=200)
