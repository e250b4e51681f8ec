"""The breaking dam at a chosen step, in two processes, so that a profiler only sees the steps of
interest (a `rocprofv3 --pmc` run over the 400 steps that lead there serialises 3600 dispatches
with a counter read each and does not finish in a gpurun call: r4 notes, call 11):

    python3 tools/dense_state.py save 400 /tmp/dam400.npz          (no profiler)
    rocprofv3 --pmc ... -- python3 tools/dense_state.py run /tmp/dam400.npz 4

`save STEP FILE` steps the 4M-particle dam (gravity + walls, tolerance-mode arithmetic) to STEP and
writes positions and velocities; `run FILE K` uploads them, grows the lists as the scene needs
(4 untimed steps), and takes K more steps."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smoothed_particle_hydrodynamics_amd as S
from smoothed_particle_hydrodynamics_amd import scenes

n = int(os.environ.get("SPH_DENSE_N", "4194304"))
p, pos, vel, mass = scenes.dam_break(n)
p.apply_gravity = 1
p.apply_walls = 1
p.gravity[0], p.gravity[1], p.gravity[2] = 0.0, -9.81, 0.0
if sys.argv[1] == "save":
    with S.SPH(n, p, mode=S.MODE_FULL_FAST) as sph:
        sph.setParticles(pos, vel, mass)
        sph.setTiming(S.TIMING_OFF)
        sph.run(int(sys.argv[2]))
        part = sph.getParticles()
        np.savez(sys.argv[3], pos=part.mPosition, vel=part.mVelocity)
        print("saved step", sys.argv[2], "mean neighbours", float(part.mNeighborCount.mean()), flush=True)
else:
    st = np.load(sys.argv[2])
    with S.SPH(n, p, mode=S.MODE_FULL_FAST) as sph:
        sph.setParticles(st["pos"], st["vel"], mass)
        sph.setTiming(S.TIMING_OFF)
        sph.run(4)                     # the lists and tile capacities settle on the scene
        sph.synchronize()
        k = int(sys.argv[3])
        t0 = time.perf_counter()
        sph.run(k)
        sph.synchronize()
        print("%d steps: %.3f ms/step" % (k, (time.perf_counter() - t0) / k * 1e3), sph.tileStats(), flush=True)
        sph.setTiming(S.TIMING_PHASES)
        for _ in range(4):
            sph.step()
        sph.synchronize()
        t, kk = sph.phaseTotals()
        print("phases (ms): build %.3f density %.3f acceleration %.3f integrate %.3f" % (
            t[0] / kk, t[2] / kk, t[4] / kk, t[5] / kk), flush=True)
