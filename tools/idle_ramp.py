import os, sys, time
sys.path.insert(0, '/root/repo')
import smoothed_particle_hydrodynamics_amd as S
from smoothed_particle_hydrodynamics_amd import scenes
n = 4194304
p, pos, vel, mass = scenes.dam_break(n)
sph = S.SPH(n, p); sph.setParticles(pos, vel, mass); sph.setTiming(S.TIMING_OFF)
def trace(label, k=12):
    ts = []
    for s in range(k):
        t0 = time.perf_counter(); sph.run(1); sph.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(label, " ".join("%.3f" % t for t in ts), flush=True)
trace("after upload      ")
sph.run(40); sph.synchronize()
trace("steady (no idle)  ")
time.sleep(0.05); trace("after 50 ms idle  ")
time.sleep(0.5); trace("after 500 ms idle ")
sph.run(40); sph.synchronize()
# host busy but GPU idle for ~100 ms
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1: pass
trace("after 100 ms spin ")
