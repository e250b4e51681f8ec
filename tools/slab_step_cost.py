"""What a slab step costs next to the single context, without a second GPU: `world` logical slabs
of an n-particle dam-break take turns on this GPU (LocalSlabGroup: messages handed over by pointer,
early exchange on a second stream unless `serial` is given), then the same scene steps in one
context.

    python tools/slab_step_cost.py 16777216 8 [serial]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import smoothed_particle_hydrodynamics_amd as S
from smoothed_particle_hydrodynamics_amd import scenes, slab as SL

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16777216
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
serial = "serial" in sys.argv[3:]
p, pos, vel, mass = scenes.dam_break(n)
z = pos.reshape(-1, 3)[:, 2]
cuts = SL.plan_cuts(p, z, world)
hist = np.bincount(SL.plane_of(p, z), minlength=p.full_cells_z)
stream = torch.cuda.Stream()
slabs = []
for r in range(world):
    cap, msg = SL.slab_capacities(hist, cuts, r, slack=1.5)
    s = SL.HipSlab(p, cuts[r], cuts[r + 1], cap, msg, device=0, has_left=r > 0,
                   has_right=r + 1 < world, stream=stream)
    s.upload(*SL.split_scene(p, cuts, r, pos, vel, mass), all_masses_equal=True)
    s.set_timing(S.TIMING_OFF)
    slabs.append(s)
group = SL.LocalSlabGroup(slabs, overlap=not serial,
                          exchange_stream=None if serial else torch.cuda.Stream(priority=-1))
for _ in range(5):
    group.step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    group.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
errors = [s.status()["errors"] for s in slabs]
print("%d slabs of %d particles, %s: %.1f us per step = %.1f us per slab (host enqueue %.1f us per "
      "slab), error bits %s" % (world, n, "serial" if serial else "early exchange",
                                 (t2 - t0) / K * 1e6, (t2 - t0) / K / world * 1e6,
                                 (t1 - t0) / K / world * 1e6, errors), flush=True)
print("tiles of slab 0:", slabs[0].tile_stats(), flush=True)
for s in slabs:
    s.close()
with S.SPH(n, p) as sph:
    sph.setParticles(pos, vel, mass)
    sph.run(5)
    sph.synchronize()
    t0 = time.perf_counter()
    sph.run(K)
    sph.synchronize()
    one = (time.perf_counter() - t0) / K * 1e6
print("single context: %.1f us per step = %.1f us per %d-th" % (one, one / world, world), flush=True)
