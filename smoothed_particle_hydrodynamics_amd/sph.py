"""Host-side mirror of the reference's solver interface over the HIP C ABI.

`SPH` keeps the public/protected method names of the reference class (reference
src/sph.h:20-139) so that code and tests written against the reference read the same here;
`Particle` keeps the reference container's member names and layouts (reference
src/particle.h:7-20: one object for all particles, xyz interleaved).  All computation
happens in libsph_hip.so on the GPU; this file only moves arrays and parameters.
"""
import ctypes as C

import numpy as np

from .lib import ARITH_EXACT, ARITH_FAST, MODE_FULL, MODE_FULL_FAST, MODE_REF, SphHipError, SphParams, default_params, load_library

__all__ = ["SPH", "Particle", "MODE_REF", "MODE_FULL", "MODE_FULL_FAST", "ARITH_EXACT", "ARITH_FAST"]


class Particle:
    """reference src/particle.h:7-20 — struct of arrays for ALL particles."""

    def __init__(self, num_particles):
        n = int(num_particles)
        self.mMass = np.zeros(n, np.float32)
        self.mDensity = np.zeros(n, np.float32)
        self.mPosition = np.zeros(3 * n, np.float32)
        self.mVelocity = np.zeros(3 * n, np.float32)
        self.mAcceleration = np.zeros(3 * n, np.float32)
        self.mNeighborCount = np.zeros(n, np.int32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class SPH:
    """The reference's `SPH` (src/sph.h:15-216) with the step executed on an MI355X.

    Differences that are part of the contract:
      * the particle count and the scene are arguments (the reference fixes N = M*1024 at
        compile time and always builds its sphere scene, src/sph.cpp:59,117);
      * `mode` selects the shipped sampled search (MODE_REF) or complete neighbourhoods
        (MODE_FULL; MODE_FULL_FAST = the same neighbourhoods and order with tolerance-mode pair
        arithmetic, see include/sph_hip.h);
      * the host `Particle` mirror is refreshed by `syncParticles()` / `getParticles()`
        rather than being written by every phase.
    """

    def __init__(self, particle_count, params=None, mode=MODE_FULL, device=0, capacity=None):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        self.mParticleCount = int(particle_count)
        self.mode = mode
        self._params = params.copy() if params is not None else default_params()
        cap = int(capacity) if capacity is not None else max(1, self.mParticleCount)
        rc = self._lib.sph_hip_create(C.byref(self._ctx), C.byref(self._params), cap, int(mode),
                                      int(device))
        if rc != 0:
            msg = self._lib.sph_hip_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise SphHipError("sph_hip_create failed (%d): %s" % (rc, msg))
        self.mSrcParticles = Particle(self.mParticleCount)
        self._mirror_fresh = False

    # ---- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.sph_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.sph_hip_last_error(self._ctx).decode()
            raise SphHipError("%s failed (%d): %s" % (what, rc, msg))

    # ---- scene ------------------------------------------------------------------------------
    def setParticles(self, position, velocity, mass):
        """Fill Particle::mPosition / mVelocity / mMass (what the reference's constructor and
        initParticlePolitionsSphere do, src/sph.cpp:105-108, 361-425) and upload."""
        pos = np.ascontiguousarray(position, np.float32).reshape(-1)
        vel = np.ascontiguousarray(velocity, np.float32).reshape(-1)
        m = np.ascontiguousarray(mass, np.float32).reshape(-1)
        n = m.size
        if pos.size != 3 * n or vel.size != 3 * n:
            raise ValueError("position/velocity must hold 3 floats per particle")
        if n != self.mParticleCount:
            self.mParticleCount = n
            self.mSrcParticles = Particle(n)
        self.mSrcParticles.mPosition[:] = pos
        self.mSrcParticles.mVelocity[:] = vel
        self.mSrcParticles.mMass[:] = m
        self._check(self._lib.sph_hip_upload(self._ctx, n, _ptr(pos), _ptr(vel), _ptr(m)),
                    "sph_hip_upload")
        self._mirror_fresh = False

    def syncParticles(self):
        """Refresh the host `Particle` mirror from the device."""
        p = self.mSrcParticles
        self._check(self._lib.sph_hip_download(self._ctx, _ptr(p.mPosition), _ptr(p.mVelocity),
                                               _ptr(p.mDensity), _ptr(p.mAcceleration),
                                               _ptr(p.mNeighborCount)), "sph_hip_download")
        self._mirror_fresh = True
        return p

    # ---- public getters (reference src/sph.cpp:1172-1289) -------------------------------------
    def getParticles(self):
        if not self._mirror_fresh:
            self.syncParticles()
        return self.mSrcParticles

    def getParticleCount(self):
        return self.mParticleCount

    def getGridCellCounts(self):
        return self._params.cells_x, self._params.cells_y, self._params.cells_z

    def getParticleBounds(self):
        return self._params.max_x, self._params.max_y, self._params.max_z

    def getInteractionRadius2(self):
        return self._params.hscaled2

    def getCellSize(self):
        return self._params.cell_size

    def getGrid(self):
        """Per-voxel occupancy: what callers take from getGrid()[i].count()
        (reference src/visualization.cpp:178-193)."""
        p = self._params
        if self.mode == MODE_REF:
            ncells = p.cells_x * p.cells_y * p.cells_z
        else:
            ncells = p.full_cells_x * p.full_cells_y * p.full_cells_z
        counts = np.zeros(ncells, np.int32)
        self._check(self._lib.sph_hip_download_grid_counts(self._ctx, _ptr(counts)),
                    "sph_hip_download_grid_counts")
        return counts

    def getParams(self):
        return self._params.copy()

    def _push(self):
        self._check(self._lib.sph_hip_set_params(self._ctx, C.byref(self._params)),
                    "sph_hip_set_params")

    def getGravity(self):
        return tuple(self._params.gravity)

    def setGravity(self, gravity):
        for c in range(3):
            self._params.gravity[c] = gravity[c]
        self._push()

    def getStiffness(self):
        return self._params.stiffness

    def setStiffness(self, stiffness):
        self._params.stiffness = stiffness
        self._push()

    def getViscosityScalar(self):
        return self._params.viscosity

    def setViscosityScalar(self, viscosity):
        self._params.viscosity = viscosity
        self._push()

    def getTimeStep(self):
        return self._params.time_step

    def setTimeStep(self, time_step):
        self._params.time_step = time_step
        self._push()

    def getDamping(self):
        return self._params.damping

    def setDamping(self, damping):
        self._params.damping = damping
        self._push()

    def getCflLimit(self):
        return self._params.cfl_limit

    def setCflLimit(self, cfl_limit):
        # reference src/sph.cpp:1237-1241: also refreshes the squared limit, in fp32
        self._params.cfl_limit = cfl_limit
        lim = np.float32(self._params.cfl_limit)
        self._params.cfl_limit2 = float(lim * lim)
        self._push()

    # ---- slots ---------------------------------------------------------------------------------
    def step(self):
        """SPH::step() (reference src/sph.cpp:190-304)."""
        self._check(self._lib.sph_hip_step(self._ctx), "sph_hip_step")
        self._mirror_fresh = False

    def run(self, steps):
        """`steps` steps queued back to back (SPH::run's loop body, src/sph.cpp:171-181)."""
        self._check(self._lib.sph_hip_run(self._ctx, int(steps)), "sph_hip_run")
        self._mirror_fresh = False

    def runToFiles(self, total_steps, outdir="out"):
        """SPH::run() (reference src/sph.cpp:149-187, 203, 232): `total_steps + 1` steps, one line
        per step in energy.txt / angularmomentum.txt / timing.txt / neighbors.txt with the
        reference's headers and column order.  Differences: times are fractional milliseconds
        (the reference truncates to int), KE/PE are summed in a fixed order in f64."""
        import os
        os.makedirs(outdir, exist_ok=True)
        with open(os.path.join(outdir, "energy.txt"), "w") as fe, \
                open(os.path.join(outdir, "angularmomentum.txt"), "w") as fl, \
                open(os.path.join(outdir, "timing.txt"), "w") as ft, \
                open(os.path.join(outdir, "neighbors.txt"), "w") as fn:
            fe.write("Step, Kinetic Energy, Potential Energy, Total Energy\n")
            fl.write("Step, Angular Momentum\n")
            ft.write("Step, Voxelize, Find Neighbors, Compute Density, Compute Pressure, "
                     "Compute Acceleration, Integrate\n")
            for s in range(int(total_steps) + 1):
                self.step()
                ke, pe = self.energy()
                fe.write("%d, %.9g, %.9g, %.9g\n" % (s, ke, pe, np.float32(ke) + np.float32(pe)))
                fl.write("%d, 0\n" % s)          # mAngularMomentumTotal is never accumulated
                ft.write("%d, %s\n" % (s, ", ".join("%.4f" % v for v in self.elapsed())))
                fn.write("%d, %d, %d\n" % self.neighborStats())

    def synchronize(self):
        self._check(self._lib.sph_hip_synchronize(self._ctx), "sph_hip_synchronize")

    def setArithmetic(self, arithmetic):
        """ARITH_EXACT / ARITH_FAST for the pair sums of a FULL-mode context (sph_hip_set_arithmetic)."""
        self._check(self._lib.sph_hip_set_arithmetic(self._ctx, int(arithmetic)), "sph_hip_set_arithmetic")
        self._mirror_fresh = False

    def getArithmetic(self):
        return self._lib.sph_hip_get_arithmetic(self._ctx)

    # ---- protected pipeline (reference src/sph.h:96-112) ----------------------------------------
    def voxelizeParticles(self):
        self._check(self._lib.sph_hip_voxelize(self._ctx), "sph_hip_voxelize")
        self._mirror_fresh = False

    def findNeighbors(self):
        self._check(self._lib.sph_hip_find_neighbors(self._ctx), "sph_hip_find_neighbors")
        self._mirror_fresh = False

    def computeDensity(self):
        self._check(self._lib.sph_hip_compute_density(self._ctx), "sph_hip_compute_density")
        self._mirror_fresh = False

    def computeAcceleration(self):
        self._check(self._lib.sph_hip_compute_acceleration(self._ctx),
                    "sph_hip_compute_acceleration")
        self._mirror_fresh = False

    def integrate(self):
        self._check(self._lib.sph_hip_integrate(self._ctx), "sph_hip_integrate")
        self._mirror_fresh = False

    # ---- diagnostics -----------------------------------------------------------------------------
    def elapsed(self):
        """The six numbers of SPH::updateElapsed (reference src/sph.cpp:292-299), in ms."""
        ms = (C.c_float * 6)()
        self._check(self._lib.sph_hip_get_timings(self._ctx, C.byref(ms)), "sph_hip_get_timings")
        return list(ms)

    def resetTimings(self):
        self._check(self._lib.sph_hip_reset_timings(self._ctx), "sph_hip_reset_timings")

    def setTiming(self, level):
        """Which intervals step() times: TIMING_PHASES (default, all six), TIMING_SUMS (density +
        acceleration as one interval, in slot 2), TIMING_OFF.  Resets the collected timings."""
        self._check(self._lib.sph_hip_set_timing(self._ctx, int(level)), "sph_hip_set_timing")

    def setTimingStride(self, every):
        """Record the timing events on every `every`-th step() only (sph_hip_set_timing_stride)."""
        self._check(self._lib.sph_hip_set_timing_stride(self._ctx, int(every)),
                    "sph_hip_set_timing_stride")

    def tileStats(self):
        """dict of the last step's LDS-tile statistics (sph_hip_get_tile_stats)."""
        out = (C.c_int32 * 20)()
        self._check(self._lib.sph_hip_get_tile_stats(self._ctx, C.byref(out)), "sph_hip_get_tile_stats")
        v = list(out)
        return {"over_level": v[0:12], "workgroups": v[12], "largest_tile": v[13],
                "untiled_density": v[14], "untiled_acceleration": v[15],
                "capacity_density": v[16], "capacity_acceleration": v[17], "wide_entries": v[18],
                "list_capacity": v[19]}

    def phaseTotals(self):
        """(sum of the six phase times in ms over the step() calls since resetTimings(), steps)"""
        ms = (C.c_double * 6)()
        k = C.c_int32()
        self._check(self._lib.sph_hip_get_phase_totals(self._ctx, C.byref(ms), C.byref(k)),
                    "sph_hip_get_phase_totals")
        return list(ms), k.value

    def energy(self):
        """(mKineticEnergyTotal, mPotentialEnergyTotal) of the last integrate."""
        ke, pe = C.c_float(), C.c_float()
        self._check(self._lib.sph_hip_get_energy(self._ctx, C.byref(ke), C.byref(pe)),
                    "sph_hip_get_energy")
        return ke.value, pe.value

    def neighborStats(self):
        """(avg, max, min) as written to out/neighbors.txt (reference src/sph.cpp:232)."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.sph_hip_get_neighbor_stats(self._ctx, C.byref(a), C.byref(b),
                                                         C.byref(c)), "sph_hip_get_neighbor_stats")
        return a.value, b.value, c.value

    def voxels(self):
        """(mVoxelCoords as 3 ints per particle, mVoxelIds) — REF mode."""
        n = self.mParticleCount
        coords = np.zeros(3 * n, np.int32)
        ids = np.zeros(n, np.int32)
        self._check(self._lib.sph_hip_download_voxels(self._ctx, _ptr(coords), _ptr(ids)),
                    "sph_hip_download_voxels")
        return coords, ids

    def neighborLists(self):
        """(mNeighbors, mNeighborDistancesScaled), row stride mExamineCount — REF mode."""
        m = self.mParticleCount * self._params.examine_count
        nb = np.zeros(m, np.uint32)
        nd = np.zeros(m, np.float32)
        self._check(self._lib.sph_hip_download_neighbor_lists(self._ctx, _ptr(nb), _ptr(nd)),
                    "sph_hip_download_neighbor_lists")
        return nb, nd
