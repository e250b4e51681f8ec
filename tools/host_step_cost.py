"""What the HOST spends per slab step, with the device work made negligible (verdict r3, item 5a):
a tiny slab (2000 particles: every kernel of a step is a handful of workgroups), world 1 - no
neighbour, so no message crosses anything - stepped through
  * DistSlabStepper.step   the torch transport's per-step Python: ~6 ctypes calls, the (empty) P2P
                           batch, event record / wait, error poll every 16 steps;
  * NativeSlabStepper.run  the same step issued by sph_hip_slab_comm_run inside libsph_hip.so;
  * SPH.run                a context without the slab protocol (sph_hip_run), for scale.
Host time = wall time of the enqueueing loop with a device that keeps up (the kernels of a
2000-particle step take less than the host needs to enqueue them), reported as us per step; the
device-drained time beside it.  What this cannot measure on a one-GPU box: the host cost of
torch.distributed.batch_isend_irecv with a real peer (RCCL refuses two ranks on one device)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import smoothed_particle_hydrodynamics_amd as S
from smoothed_particle_hydrodynamics_amd import scenes, slab as SL

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
K = 2000
p, pos, vel, mass = scenes.dam_break(n)


def slab():
    s = SL.HipSlab(p, 0, p.full_cells_z, 4 * n, 1024, device=0, has_left=False, has_right=False)
    s.upload(np.arange(n, dtype=np.uint32), pos, vel, mass, all_masses_equal=True)
    s.set_timing(S.TIMING_OFF)
    return s


def timed(step, sync, label):
    for _ in range(50):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    t1 = time.perf_counter()
    sync()
    t2 = time.perf_counter()
    print("%-58s host %6.1f us per step   (device drained: %6.1f us per step)" % (
        label, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6), flush=True)


class NoPeers(SL.DistTransport):
    """DistTransport of a 1-rank world without a process group: the op list is empty"""
    def __init__(self):
        self.rank, self.world, self.group, self._comm = 0, 1, None, None
        self.dist = None

    def _run(self, ops):
        assert not ops


s = slab()
stepper = SL.DistSlabStepper(s, NoPeers(), overlap=True)
timed(stepper.step, s.synchronize, "DistSlabStepper.step (torch transport, early exchange)")
s.close()

s = slab()
stepper = SL.DistSlabStepper(s, NoPeers(), overlap=False)
timed(stepper.step, s.synchronize, "DistSlabStepper.step (torch transport, serial)")
s.close()

s = slab()
s.comm_init(SL.rccl_unique_id(), 0, 1)
timed(lambda: s.comm_run(1), s.synchronize, "sph_hip_slab_comm_run(1) per call (native loop)")
s.synchronize()
t0 = time.perf_counter()
s.comm_run(K)
t1 = time.perf_counter()
s.synchronize()
t2 = time.perf_counter()
print("%-58s host %6.1f us per step   (device drained: %6.1f us per step)" % (
    "sph_hip_slab_comm_run(%d) one call (native loop)" % K, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6), flush=True)
s.close()

with S.SPH(n, p) as sph:
    sph.setParticles(pos, vel, mass)
    sph.setTiming(S.TIMING_OFF)
    timed(lambda: sph.run(1), sph.synchronize, "sph_hip_run(1) per call (no slab protocol)")
    sph.synchronize()
    t0 = time.perf_counter()
    sph.run(K)
    t1 = time.perf_counter()
    sph.synchronize()
    t2 = time.perf_counter()
    print("%-58s host %6.1f us per step   (device drained: %6.1f us per step)" % (
        "sph_hip_run(%d) one call" % K, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6), flush=True)
