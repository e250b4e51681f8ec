"""TEST INFRASTRUCTURE ONLY — ctypes front-ends for the parity oracle.

`Oracle`     wraps oracle/liboracle.so, the C restatement (sph_oracle.c).
`Reference`  wraps oracle/_ref/libsphref.so, the reference's own sph.cpp compiled by
             `make -C oracle ref` (only where /root/reference exists; the built .so travels).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package must never do so.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsphref.so")


class OracleParams(C.Structure):
    """Mirror of sph_oracle_params (oracle/sph_oracle.h)."""

    _fields_ = [
        ("cells_x", C.c_int32), ("cells_y", C.c_int32), ("cells_z", C.c_int32),
        ("cell_size", C.c_float),
        ("max_x", C.c_float), ("max_y", C.c_float), ("max_z", C.c_float),
        ("h", C.c_float), ("h2", C.c_float), ("hscaled", C.c_float), ("hscaled2", C.c_float),
        ("hscaled6", C.c_float), ("hscaled9", C.c_float), ("htimes2", C.c_float),
        ("htimes2inv", C.c_float), ("sim_scale", C.c_float), ("sim_scale_inv", C.c_float),
        ("kernel1", C.c_float), ("kernel2", C.c_float), ("kernel3", C.c_float),
        ("rho0", C.c_float), ("stiffness", C.c_float), ("viscosity", C.c_float),
        ("time_step", C.c_float), ("damping", C.c_float),
        ("cfl_limit", C.c_float), ("cfl_limit2", C.c_float),
        ("gravity", C.c_float * 3),
        ("grav_const", C.c_float), ("central_mass", C.c_float), ("central_pos", C.c_float * 3),
        ("softening", C.c_float),
        ("examine_count", C.c_int32),
        ("full_cells_x", C.c_int32), ("full_cells_y", C.c_int32), ("full_cells_z", C.c_int32),
        ("full_cell_inv", C.c_float),
        ("apply_gravity", C.c_int32), ("apply_walls", C.c_int32),
    ]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


def build(ref=None):
    """(Re)build liboracle.so and, if the reference tree is present, _ref/libsphref.so."""
    subprocess.run(["make", "-C", HERE, "-s"], check=True)
    if ref is None:
        ref = os.path.isdir("/root/reference/src")
    if ref:
        subprocess.run(["make", "-C", HERE, "-s", "ref"], check=True)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        self.lib = C.CDLL(ORACLE_SO)
        self.lib.oracle_full_build_lists.restype = C.c_int

    # ---- A0
    def params_for_h(self, h=0.1, cells=(32, 32, 32)):
        p = OracleParams()
        self.lib.oracle_params_for_h(C.byref(p), C.c_float(h), int(cells[0]), int(cells[1]),
                                     int(cells[2]))
        return p

    def init_sphere(self, p, n):
        pos = np.zeros(3 * n, np.float32)
        vel = np.zeros(3 * n, np.float32)
        self.lib.oracle_init_sphere(C.byref(p), n, _ptr(pos), _ptr(vel))
        return pos, vel

    # ---- REF mode phases
    def voxelize(self, p, pos):
        n = pos.size // 3
        ncells = p.cells_x * p.cells_y * p.cells_z
        coords = np.zeros(3 * n, np.int32)
        ids = np.zeros(n, np.int32)
        cs = np.zeros(ncells + 1, np.int32)
        ci = np.zeros(n, np.int32)
        self.lib.oracle_voxelize(C.byref(p), n, _ptr(pos), _ptr(coords), _ptr(ids), _ptr(cs),
                                 _ptr(ci))
        return coords, ids, cs, ci

    def find_neighbors(self, p, pos, coords, cs, ci):
        n = pos.size // 3
        cap = p.examine_count
        nb = np.zeros(n * cap, np.uint32)
        nd = np.zeros(n * cap, np.float32)
        cnt = np.zeros(n, np.int32)
        self.lib.oracle_find_neighbors(C.byref(p), n, _ptr(pos), _ptr(coords), _ptr(cs), _ptr(ci),
                                       _ptr(nb), _ptr(nd), _ptr(cnt))
        return nb, nd, cnt

    def neighbor_stats(self, cnt):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.oracle_neighbor_stats(cnt.size, _ptr(cnt), C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def density_lists(self, p, cap, nb, nd, cnt, mass):
        n = cnt.size
        rho = np.zeros(n, np.float32)
        self.lib.oracle_density_lists(C.byref(p), n, cap, _ptr(nb), _ptr(nd), _ptr(cnt),
                                      _ptr(mass), _ptr(rho))
        return rho

    def accel_lists(self, p, cap, nb, nd, cnt, pos, vel, mass, rho):
        n = cnt.size
        acc = np.zeros(3 * n, np.float32)
        self.lib.oracle_accel_lists(C.byref(p), n, cap, _ptr(nb), _ptr(nd), _ptr(cnt), _ptr(pos),
                                    _ptr(vel), _ptr(mass), _ptr(rho), _ptr(acc))
        return acc

    def boundary(self, p, position, vel, time_step, newpos):
        """oracle_boundary on every particle; returns (vel, newpos) copies."""
        position = _f32(position)
        vel = _f32(vel).copy()
        newpos = _f32(newpos).copy()
        for i in range(position.size // 3):
            self.lib.oracle_boundary(C.byref(p), _ptr(position[3 * i:]), _ptr(vel[3 * i:]),
                                     C.c_float(time_step), _ptr(newpos[3 * i:]))
        return vel, newpos

    def integrate(self, p, pos, vel, acc, mass):
        """in place on pos/vel; returns (ke, pe)"""
        n = mass.size
        ke, pe = C.c_float(), C.c_float()
        self.lib.oracle_integrate(C.byref(p), n, _ptr(pos), _ptr(vel), _ptr(acc), _ptr(mass),
                                  C.byref(ke), C.byref(pe))
        return ke.value, pe.value

    def step(self, p, pos, vel, mass, mode="ref"):
        """One whole step in place on pos/vel. Returns dict(rho, acc, ncount, ke, pe)."""
        n = mass.size
        rho = np.zeros(n, np.float32)
        acc = np.zeros(3 * n, np.float32)
        cnt = np.zeros(n, np.int32)
        ke, pe = C.c_float(), C.c_float()
        fn = self.lib.oracle_step_ref if mode == "ref" else self.lib.oracle_step_full
        fn(C.byref(p), n, _ptr(pos), _ptr(vel), _ptr(mass), _ptr(rho), _ptr(acc), _ptr(cnt),
           C.byref(ke), C.byref(pe))
        return dict(rho=rho, acc=acc, ncount=cnt, ke=ke.value, pe=pe.value)

    # ---- FULL mode
    def full_cells(self, p, pos):
        n = pos.size // 3
        ncells = p.full_cells_x * p.full_cells_y * p.full_cells_z
        ids = np.zeros(n, np.int32)
        cs = np.zeros(ncells + 1, np.int32)
        ci = np.zeros(n, np.int32)
        self.lib.oracle_full_cells(C.byref(p), n, _ptr(pos), _ptr(ids), _ptr(cs), _ptr(ci))
        return ids, cs, ci

    def full_build_lists(self, p, pos, cap):
        n = pos.size // 3
        nb = np.zeros(n * cap, np.uint32)
        nd = np.zeros(n * cap, np.float32)
        cnt = np.zeros(n, np.int32)
        worst = self.lib.oracle_full_build_lists(C.byref(p), n, _ptr(pos), cap, _ptr(nb),
                                                 _ptr(nd), _ptr(cnt))
        return nb, nd, cnt, worst

    def full_density(self, p, pos, mass, cs, ci):
        n = mass.size
        rho = np.zeros(n, np.float32)
        cnt = np.zeros(n, np.int32)
        self.lib.oracle_full_density(C.byref(p), n, _ptr(pos), _ptr(mass), _ptr(cs), _ptr(ci),
                                     _ptr(rho), _ptr(cnt))
        return rho, cnt

    def full_accel(self, p, pos, vel, mass, rho, cs, ci):
        n = mass.size
        acc = np.zeros(3 * n, np.float32)
        self.lib.oracle_full_accel(C.byref(p), n, _ptr(pos), _ptr(vel), _ptr(mass), _ptr(rho),
                                   _ptr(cs), _ptr(ci), _ptr(acc))
        return acc

    def full_accel_scale(self, p, pos, vel, mass, rho):
        """per particle: the magnitude sum of the terms its acceleration is made of (float64)"""
        n = mass.size
        _, cs, ci = self.full_cells(p, pos)
        scale = np.zeros(n, np.float64)
        self.lib.oracle_full_accel_scale(C.byref(p), n, _ptr(pos), _ptr(vel), _ptr(mass), _ptr(rho),
                                         _ptr(cs), _ptr(ci), scale.ctypes.data_as(C.c_void_p))
        return scale


def reference_available():
    if not os.path.exists(REF_SO):
        return False
    try:
        C.CDLL(REF_SO)
        return True
    except OSError:
        return False


class Reference:
    """The reference's own compiled sph.cpp behind oracle/ref_harness.cpp.

    The library holds ONE global SPH object (the reference's constructor is not cheap to
    repeat and never frees), so this wrapper is a singleton view; call `configure` before
    each use.
    """

    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        self.lib.ref_particle_count.restype = C.c_int
        self.lib.ref_examine_count.restype = C.c_int

    def configure(self, p, n):
        """Size for n particles and install every constant from OracleParams `p`."""
        L = self.lib
        pos0 = (C.c_float * 3)(*p.central_pos)
        L.ref_set_grid(p.cells_x, p.cells_y, p.cells_z, C.c_float(p.cell_size))
        L.ref_set_h(*[C.c_float(v) for v in (p.h, p.h2, p.hscaled, p.hscaled2, p.hscaled6,
                                             p.hscaled9, p.htimes2, p.htimes2inv, p.kernel1,
                                             p.kernel2, p.kernel3, p.softening)])
        L.ref_set_physics(*[C.c_float(v) for v in (p.rho0, p.stiffness, p.viscosity, p.time_step,
                                                   p.cfl_limit, p.grav_const, p.central_mass)],
                          pos0)
        L.ref_set_examine_count(p.examine_count)
        L.ref_set_damping(C.c_float(p.damping))
        L.ref_set_scale(C.c_float(p.sim_scale), C.c_float(p.sim_scale_inv))
        L.ref_resize(n)

    def constants(self, fresh=True):
        """fresh: from a newly constructed SPH (the constructor's own values)"""
        out = np.zeros(32, np.float32)
        self.lib.ref_get_constants2(_ptr(out), 1 if fresh else 0)
        return out

    def init_sphere(self):
        self.lib.ref_init_sphere()

    def set_state(self, pos=None, vel=None, mass=None):
        self.lib.ref_set_state(_ptr(None if pos is None else _f32(pos)),
                               _ptr(None if vel is None else _f32(vel)),
                               _ptr(None if mass is None else _f32(mass)))

    def set_density(self, rho):
        self.lib.ref_set_density(_ptr(_f32(rho)))

    def get_state(self):
        n = self.lib.ref_particle_count()
        out = dict(pos=np.zeros(3 * n, np.float32), vel=np.zeros(3 * n, np.float32),
                   mass=np.zeros(n, np.float32), rho=np.zeros(n, np.float32),
                   acc=np.zeros(3 * n, np.float32), ncount=np.zeros(n, np.int32))
        self.lib.ref_get_state(_ptr(out["pos"]), _ptr(out["vel"]), _ptr(out["mass"]),
                               _ptr(out["rho"]), _ptr(out["acc"]), _ptr(out["ncount"]))
        return out

    def get_voxels(self):
        n = self.lib.ref_particle_count()
        coords = np.zeros(3 * n, np.int32)
        ids = np.zeros(n, np.int32)
        self.lib.ref_get_voxels(_ptr(coords), _ptr(ids))
        return coords, ids

    def get_grid_counts(self, ncells):
        counts = np.zeros(ncells, np.int32)
        self.lib.ref_get_grid_counts(_ptr(counts))
        return counts

    def get_lists(self):
        n = self.lib.ref_particle_count()
        cap = self.lib.ref_examine_count()
        nb = np.zeros(n * cap, np.uint32)
        nd = np.zeros(n * cap, np.float32)
        self.lib.ref_get_lists(_ptr(nb), _ptr(nd))
        return nb, nd

    def set_lists(self, cap, nb, nd, cnt):
        self.lib.ref_set_lists(cap, _ptr(nb), _ptr(nd), _ptr(cnt))

    def boundary(self, position, vel, time_step, newpos):
        """SPH::handleBoundaryConditions on every particle; returns (vel, newpos) copies."""
        position = _f32(position)
        vel = _f32(vel).copy()
        newpos = _f32(newpos).copy()
        self.lib.ref_boundary(position.size // 3, _ptr(position), _ptr(vel),
                              C.c_float(time_step), _ptr(newpos))
        return vel, newpos

    def energy(self):
        ke, pe = C.c_float(), C.c_float()
        self.lib.ref_get_energy(C.byref(ke), C.byref(pe))
        return ke.value, pe.value

    def voxelize(self):
        self.lib.ref_voxelize()

    def find_neighbors(self):
        self.lib.ref_find_neighbors()

    def compute_density(self):
        self.lib.ref_compute_density()

    def compute_acceleration(self):
        self.lib.ref_compute_acceleration()

    def integrate(self):
        self.lib.ref_integrate()

    def step(self):
        self.lib.ref_step()
