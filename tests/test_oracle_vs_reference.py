"""The CPU restatement against the reference's own compiled sph.cpp, live (needs
oracle/_ref/libsphref.so: `make -C oracle ref` where /root/reference exists; the built library
travels with the tree).  Skipped where that library cannot be loaded — test_oracle_golden.py
carries the same pins as data."""
import numpy as np
import pytest

from helpers import live_mask


def test_constructor_constants(oracle, reference):
    """A0: every constant SPH::SPH() derives (reference src/sph.cpp:46-98)"""
    p = oracle.params_for_h(0.1)
    c = reference.constants()
    names = ["h", "h2", "hscaled", "hscaled2", "hscaled6", "hscaled9", "htimes2", "htimes2inv",
             "kernel1", "kernel2", "kernel3", "softening", "rho0", "stiffness", "viscosity",
             "time_step", "cfl_limit", "cfl_limit2", "grav_const", "central_mass"]
    for i, nm in enumerate(names):
        assert np.float32(getattr(p, nm)) == c[i], nm
    assert list(np.float32(p.central_pos)) == list(c[20:23])
    assert np.float32(p.cell_size) == c[23]
    assert [np.float32(p.max_x), np.float32(p.max_y), np.float32(p.max_z)] == list(c[24:27])
    assert np.float32(p.sim_scale) == c[27] and np.float32(p.damping) == c[28]
    assert [p.cells_x, p.cells_y, p.cells_z] == [int(v) for v in c[29:32]]


@pytest.mark.parametrize("m", [8, 32, 96])
def test_ref_phases_bit_exact(oracle, reference, m):
    """A0', A1-A6 phase by phase on the reference's default scene at N = m*1024"""
    p = oracle.params_for_h(0.1)
    n = m * 1024
    reference.configure(p, n)
    reference.init_sphere()
    s = reference.get_state()
    pos, vel = oracle.init_sphere(p, n)
    assert np.array_equal(pos, s["pos"]) and np.array_equal(vel, s["vel"])
    mass = s["mass"]
    cap = p.examine_count

    reference.voxelize()
    rc, ri = reference.get_voxels()
    oc, oi, cs, ci = oracle.voxelize(p, pos)
    assert np.array_equal(rc, oc) and np.array_equal(ri, oi)
    assert np.array_equal(reference.get_grid_counts(32 ** 3), np.diff(cs))

    reference.find_neighbors()
    rnb, rnd = reference.get_lists()
    rcnt = reference.get_state()["ncount"]
    onb, ond, ocnt = oracle.find_neighbors(p, pos, oc, cs, ci)
    assert np.array_equal(ocnt, rcnt)
    live = live_mask(ocnt, cap)
    assert np.array_equal(onb[live], rnb[live]) and np.array_equal(ond[live], rnd[live])

    reference.compute_density()
    orho = oracle.density_lists(p, cap, onb, ond, ocnt, mass)
    assert np.array_equal(orho, reference.get_state()["rho"])

    reference.compute_acceleration()
    oacc = oracle.accel_lists(p, cap, onb, ond, ocnt, pos, vel, mass, orho)
    assert np.array_equal(oacc, reference.get_state()["acc"])

    reference.integrate()
    s = reference.get_state()
    oke, ope = oracle.integrate(p, pos, vel, oacc, mass)
    assert np.array_equal(pos, s["pos"]) and np.array_equal(vel, s["vel"])
    assert (oke, ope) == reference.energy()


def test_ref_five_whole_steps_through_SPH_step(oracle, reference):
    """SPH::step() itself, five times, on a scene where the search finds many neighbours"""
    from test_oracle_golden import box_fill
    p = oracle.params_for_h(0.1)
    n = 20000
    pos = box_fill(n, (1.0, 1.0, 1.0), (2.4, 2.2, 2.3), 17)
    vel = box_fill(n, (-3.0,) * 3, (3.0,) * 3, 18)
    mass = np.ones(n, np.float32)
    reference.configure(p, n)
    reference.set_state(pos, vel, mass)
    for _ in range(5):
        reference.step()
        out = oracle.step(p, pos, vel, mass, mode="ref")
    s = reference.get_state()
    for a, b in ((pos, s["pos"]), (vel, s["vel"]), (out["rho"], s["rho"]), (out["acc"], s["acc"]),
                 (out["ncount"], s["ncount"])):
        assert np.array_equal(a, b)
    assert out["ncount"].max() > 20


@pytest.mark.parametrize("scale", [1.0, 0.5])
def test_full_mode_against_reference_pair_functions(oracle, reference, scale):
    """FULL mode: the oracle's lists (canonical order) fed to the reference's own
    computeDensity / computeAcceleration == the oracle's list-free FULL sums."""
    from test_oracle_golden import box_fill
    p = oracle.params_for_h(0.1)
    p.sim_scale = scale            # exercises the scaled branches (reference src/sph.cpp:668,847-849)
    p.sim_scale_inv = 1.0 / scale
    n = 15000
    pos = box_fill(n, (2.0, 2.0, 2.0), (3.3, 3.4, 3.2), 27)
    vel = box_fill(n, (-8.0,) * 3, (8.0,) * 3, 28)
    mass = (0.5 + box_fill(n, (0,) * 3, (1,) * 3, 29)[:n]).astype(np.float32)
    cap = 128
    nb, nd, cnt, worst = oracle.full_build_lists(p, pos, cap)
    assert worst <= cap and cnt.mean() > 20
    reference.configure(p, n)
    reference.set_state(pos, vel, mass)
    reference.set_lists(cap, nb, nd, cnt)
    reference.compute_density()
    reference.compute_acceleration()
    s = reference.get_state()
    ids, cs, ci = oracle.full_cells(p, pos)
    orho, ocnt = oracle.full_density(p, pos, mass, cs, ci)
    oacc = oracle.full_accel(p, pos, vel, mass, orho, cs, ci)
    assert np.array_equal(ocnt, cnt)
    assert np.array_equal(orho, s["rho"])
    assert np.array_equal(oacc, s["acc"])
    reference.integrate()
    s = reference.get_state()
    opos, ovel = pos.copy(), vel.copy()
    energies = oracle.integrate(p, opos, ovel, oacc, mass)
    assert np.array_equal(opos, s["pos"]) and np.array_equal(ovel, s["vel"])
    assert energies == reference.energy()
