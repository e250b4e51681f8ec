"""What a workgroup per CU is worth to each pair kernel: the 4M column stepped with the LDS tile
capacity of ONE pass pinned to each capacity level (the other pass at its usual one), one process per
setting, on one box.  Feeds density_thr / accel_thr in csrc/sph_hip.hip (pick_tile_caps).
    python tools/occupancy_prices.py            (SPH_HIP_ARITH=fast for the tolerance-mode kernels)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import smoothed_particle_hydrodynamics_amd as S
    from smoothed_particle_hydrodynamics_amd import scenes
    n = 4194304
    p, pos, vel, mass = scenes.dam_break(n)
    sph = S.SPH(n, p)
    sph.setParticles(pos, vel, mass)
    sph.run(30)
    sph.synchronize()
    sph.setTiming(S.TIMING_PHASES)
    for _ in range(20):
        sph.step()
    sph.synchronize()
    t, k = sph.phaseTotals()
    ts = sph.tileStats()
    print("%s: density %.0f us (cap %d, %d untiled)  acceleration %.0f us (cap %d, %d untiled)" % (
        sys.argv[2], t[2] / k * 1e3, ts["capacity_density"], ts["untiled_density"], t[4] / k * 1e3,
        ts["capacity_acceleration"], ts["untiled_acceleration"]), flush=True)
else:
    density = [2176, 2624, 3360, 4416, 6784]          # 6 .. 2 workgroups per CU (12 B per entry)
    accel = [1952, 2496, 3296, 5056]                  # 5 .. 2 (16 B per entry)
    for rnd in range(2):
        for c in density:
            env = dict(os.environ, SPH_HIP_TILE_CAP=str(c), SPH_HIP_TILE_CAP_ACCEL="1952")
            subprocess.run([sys.executable, __file__, "--one", "density pinned to %d" % c], env=env, timeout=200)
        for c in accel:
            env = dict(os.environ, SPH_HIP_TILE_CAP=str(max(c, 2176)), SPH_HIP_TILE_CAP_DENSITY="2176",
                       SPH_HIP_TILE_CAP_ACCEL=str(c))
            subprocess.run([sys.executable, __file__, "--one", "acceleration pinned to %d" % c], env=env, timeout=200)
