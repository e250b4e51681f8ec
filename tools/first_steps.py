"""Why are the first steps after an upload slower?  Per-step trace of steps 1..16 of a fresh
context: wall time of each step run alone (sph_hip_run(1) + synchronize), then the same with the
phases timed (HIP events), the capacities each step was launched with and the tile statistics
the host had at that moment.   python tools/first_steps.py [particles]   (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smoothed_particle_hydrodynamics_amd as S  # noqa: E402
from smoothed_particle_hydrodynamics_amd import scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
p, pos, vel, mass = scenes.dam_break(n)
for rep in range(2):
    sph = S.SPH(n, p)
    sph.setParticles(pos, vel, mass)
    sph.setTiming(S.TIMING_OFF)
    sph.synchronize()
    ts = []
    for s in range(16):
        t0 = time.perf_counter()
        sph.run(1)
        sph.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("context %d, one step at a time (ms):" % rep, " ".join("%.3f" % t for t in ts), flush=True)
    # the same steps enqueued back to back, in groups of 4
    sph.setParticles(pos, vel, mass)
    sph.synchronize()
    ts = []
    for s in range(6):
        t0 = time.perf_counter()
        sph.run(4)
        sph.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3 / 4)
    print("context %d, groups of 4 (ms per step):" % rep, " ".join("%.3f" % t for t in ts), flush=True)
    sph.close()
sph = S.SPH(n, p)
sph.setParticles(pos, vel, mass)
sph.synchronize()
sph.setTiming(S.TIMING_PHASES)
for s in range(12):
    sph.resetTimings()
    sph.step()
    sph.synchronize()
    t, k = sph.phaseTotals()
    ts = sph.tileStats()
    print("step %2d: build %.3f density %.3f accel %.3f | caps %d/%d largest tile %d untiled %d/%d" % (
        s + 1, t[0], t[2], t[4], ts["capacity_density"], ts["capacity_acceleration"], ts["largest_tile"],
        ts["untiled_density"], ts["untiled_acceleration"]), flush=True)
