"""A/B timing of library builds on ONE box (boxes of the pool differ by several per cent).

Put the candidate libraries under build/variants/*.so (git-ignored, but shipped to the GPU box;
tools/build_variant.sh makes them), then on the GPU:  python tools/ab_variants.py [sizes...]
- one process per library (the library path is fixed at first load), two interleaved rounds:
sph_hip_run wall time per step at each size, plus the density / acceleration phase times (HIP
events, 10 instrumented steps) at the largest one."""
import glob
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import smoothed_particle_hydrodynamics_amd as S
    from smoothed_particle_hydrodynamics_amd import scenes
    sizes = [int(a) for a in sys.argv[2:]] or [262144, 1048576, 4194304]
    out = []
    for n in sizes:
        p, pos, vel, mass = scenes.dam_break(n)
        # AB_MODE=fast: the tolerance-mode arithmetic (bench.py's headline); default: bit-exact
        sph = S.SPH(n, p, mode=S.MODE_FULL_FAST if os.environ.get("AB_MODE") == "fast" else S.MODE_FULL)
        sph.setParticles(pos, vel, mass)
        sph.run(10)
        sph.synchronize()
        K = 100
        t3 = time.perf_counter()
        sph.run(K)
        sph.synchronize()
        t5 = time.perf_counter()
        out.append("%d: %.1f us" % (n, (t5 - t3) / K * 1e6))
        if n == sizes[-1]:
            sph.setTiming(S.TIMING_PHASES)
            for _ in range(10):
                sph.step()
            sph.synchronize()
            t, k = sph.phaseTotals()
            out.append("[build %.0f density %.0f accel %.0f integrate %.0f us]" % (
                t[0] / k * 1e3, t[2] / k * 1e3, t[4] / k * 1e3, t[5] / k * 1e3))
        sph.close()
    print("%-22s %-5s" % (os.path.basename(os.environ.get("SPH_HIP_LIBRARY", "default")),
                          os.environ.get("AB_MODE", "exact")), " | ".join(out), flush=True)
else:
    for rnd in range(2):
        for so in sorted(glob.glob(os.path.join(ROOT, "build", "variants", "*.so"))):
            env = dict(os.environ, SPH_HIP_LIBRARY=so)
            subprocess.run([sys.executable, __file__, "--one"] + sys.argv[1:], env=env, timeout=300)
