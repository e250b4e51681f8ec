"""GPU parity on randomly drawn small scenes: particle counts from 1 to a few thousand, random
boxes (partly outside the grid), smoothing lengths, grids, mass distributions, simulation scales,
gravity / wall flags, clustered and duplicated positions - FULL and REF mode against the oracle,
bit for bit, two steps each.  The draws are seeded: a failure names its case."""
import os

import numpy as np
import pytest

from helpers import to_oracle_params

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("SPH_RANDOM_CASES", "24"))   # more for a one-off soak


def draw(case):
    from smoothed_particle_hydrodynamics_amd import default_params, scenes
    rng = np.random.default_rng(1000 + case)
    n = int(rng.choice([1, 2, 3, 17, 64, 65, 255, 256, 257, 700, 1500, 3000, 6000]))
    h = float(rng.choice([0.05, 0.1, 0.13, 0.25]))
    cells = [int(rng.integers(3, 20)) for _ in range(3)]
    p = default_params(h, cells)
    ext = np.array([c * 2.0 * h for c in cells], np.float32)
    lo = (rng.uniform(-0.3, 0.4, 3) * ext).astype(np.float32)
    hi = (lo + rng.uniform(0.05, 0.9, 3) * ext).astype(np.float32)
    pos = scenes.box_fill(n, lo, hi, seed=case).reshape(-1, 3)
    if n >= 64 and rng.random() < 0.5:            # a tight cluster: many neighbours, one cell
        k = n // 3
        pos[:k] = pos[0] + (rng.random((k, 3)).astype(np.float32) - 0.5) * np.float32(0.5 * h)
    if n >= 17 and rng.random() < 0.5:            # exact duplicates (d = 0)
        pos[1:9] = pos[9:17]
    if n >= 3 and rng.random() < 0.5:             # on cell faces of the FULL grid
        edge = np.float32(1.0) / np.float32(p.full_cell_inv)
        pos[-(n // 4 + 1):] = edge * np.round(pos[-(n // 4 + 1):] / edge)
    speed = float(rng.choice([0.0, 1.0, 30.0]))
    vel = scenes.box_fill(n, (-speed,) * 3, (speed + 1e-6,) * 3, seed=case + 77)
    mass = np.ones(n, np.float32)
    if rng.random() < 0.5:
        mass = (0.25 + 2.0 * rng.random(n)).astype(np.float32)
    if rng.random() < 0.4:
        s = float(rng.choice([0.5, 2.0, 0.01]))
        p.sim_scale, p.sim_scale_inv = s, 1.0 / s
    if rng.random() < 0.3:
        p.central_mass = 0.0
    p.apply_gravity = int(rng.random() < 0.5)
    p.apply_walls = int(rng.random() < 0.5)
    p.gravity[0], p.gravity[1], p.gravity[2] = 0.0, -9.81, 0.5
    return p, np.ascontiguousarray(pos.reshape(-1)), vel, mass


@pytest.mark.parametrize("mode", ["full", "ref"])
@pytest.mark.parametrize("case", range(CASES))
def test_random_scene(oracle, hiplib, case, mode):
    import smoothed_particle_hydrodynamics_amd as S
    p, pos, vel, mass = draw(case)
    op = to_oracle_params(p)
    opos, ovel = pos.copy(), vel.copy()
    with S.SPH(mass.size, p, mode=S.MODE_FULL if mode == "full" else S.MODE_REF) as sph:
        sph.setParticles(pos, vel, mass)
        for step in range(2):
            sph.step()
            ref = oracle.step(op, opos, ovel, mass, mode=mode)
            part = sph.getParticles()
            what = "case %d (%s, n=%d) step %d: " % (case, mode, mass.size, step)
            assert np.array_equal(part.mNeighborCount, ref["ncount"]), what + "neighbour counts"
            assert np.array_equal(part.mDensity, ref["rho"], equal_nan=True), what + "density"
            assert np.array_equal(part.mAcceleration, ref["acc"], equal_nan=True), what + "acceleration"
            assert np.array_equal(part.mPosition, opos, equal_nan=True), what + "position"
            assert np.array_equal(part.mVelocity, ovel, equal_nan=True), what + "velocity"


# cases the long soaks singled out (profiles/r3_notes.md, sections 7-8): 219 - one rim neighbour of next
# to no density carries the whole force; 305 - velocities of 1e-6 and a force that cancels; 1751 - a
# neighbour of density 3.5e-17: the reference's sum overflows and its clamp returns zeros; 594 (round 4,
# soak of 3000) - no point mass, and particles the reference lost to a NaN in the first step: the point-mass
# term the tolerance mode skips is NaN for them in the second
SOAK_FINDS = [c for c in (219, 305, 594, 1751) if c >= CASES]


@pytest.mark.parametrize("case", list(range(CASES)) + SOAK_FINDS)
def test_random_scene_tolerance_mode(oracle, hiplib, case):
    """the same draws with SPH_HIP_MODE_FULL_FAST, held to tests/test_gpu_full_fast.py's bar (every
    step started from the state the GPU started from) - the ONLY tests in which the cancellation
    clause of that bar may be taken (check_fast prints every particle that takes it)"""
    import smoothed_particle_hydrodynamics_amd as S
    from test_gpu_full_fast import check_fast, check_fast_velocity
    p, pos, vel, mass = draw(case)
    op = to_oracle_params(p)
    cur_pos, cur_vel = pos.copy(), vel.copy()
    with S.SPH(mass.size, p, mode=S.MODE_FULL_FAST) as sph:
        sph.setParticles(pos, vel, mass)
        for step in range(2):
            sph.step()
            part = sph.getParticles()
            opos, ovel = cur_pos.copy(), cur_vel.copy()
            ref = oracle.step(op, opos, ovel, mass, mode="full")
            what = "case %d (fast, n=%d) step %d" % (case, mass.size, step)
            finite = np.isfinite(ref["acc"]).reshape(-1, 3).all(axis=1) & np.isfinite(ref["rho"])
            assert np.array_equal(np.isfinite(part.mAcceleration).reshape(-1, 3).all(axis=1), finite), what

            class Part:
                pass
            sel = Part()
            sel.mNeighborCount = part.mNeighborCount
            sel.mDensity = np.where(finite, part.mDensity, 0).astype(np.float32)
            sel.mAcceleration = np.where(np.repeat(finite, 3), part.mAcceleration, 0).astype(np.float32)
            want = dict(ncount=ref["ncount"], rho=np.where(finite, ref["rho"], 0).astype(np.float32),
                        acc=np.where(np.repeat(finite, 3), ref["acc"], 0).astype(np.float32))
            _, allowed = check_fast(sel, want, p, mass, what,
                                    scale=lambda: np.maximum(oracle.full_accel_scale(op, cur_pos, cur_vel, mass, ref["rho"]), 1e-300))
            ok = np.repeat(finite, 3)
            check_fast_velocity(np.where(ok, part.mVelocity, 0), np.where(ok, ovel, 0), allowed, p.time_step, what)
            cur_pos, cur_vel = part.mPosition.copy(), part.mVelocity.copy()


@pytest.mark.parametrize("env", [{"SPH_HIP_UNTILED": "1"}, {"SPH_HIP_TILE_CAP": "512"}, {"SPH_HIP_LIST_CAP": "30"}],
                         ids=["untiled", "tile-cap-512", "list-cap-30"])
@pytest.mark.parametrize("case", range(CASES))
def test_random_scene_tolerance_mode_any_route(hiplib, case, env, monkeypatch):
    """the tolerance-mode arithmetic is written out operation by operation: the routes that do not
    split their neighbour loop (untiled, give-up workgroups, the walk of a particle without a list)
    produce the bits of the tiled one - which is what makes a slab's ghosts agree with their owner"""
    import smoothed_particle_hydrodynamics_amd as S
    p, pos, vel, mass = draw(case)
    out = []
    for e in ({}, env):
        for k in ("SPH_HIP_UNTILED", "SPH_HIP_TILE_CAP", "SPH_HIP_LIST_CAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in e.items():
            monkeypatch.setenv(k, v)
        with S.SPH(mass.size, p, mode=S.MODE_FULL_FAST) as sph:
            sph.setParticles(pos, vel, mass)
            sph.step()
            sph.step()
            part = sph.getParticles()
            out.append({k: getattr(part, k).copy() for k in ("mPosition", "mVelocity", "mDensity",
                                                            "mAcceleration", "mNeighborCount")})
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k], equal_nan=True), "case %d: %s" % (case, k)
