"""GPU parity, REF mode: the HIP path through the C ABI against the CPU restatement of the
reference's shipped pipeline (reference src/sph.cpp:190-304), phase by phase.

Bar: integer outputs (voxel coords/ids, per-voxel occupancy, neighbour counts, neighbour
lists) identical; fp32 outputs (distances, density, acceleration, position, velocity)
bit-identical as well — the kernels are compiled without FMA contraction and use correctly
rounded sqrt/divide, so anything weaker than equality would hide an ordering bug.  KE/PE: the
reference adds N fp32 terms serially (error ~sqrt(N)*2^-24): tolerance energy_rtol(N).
"""
import numpy as np
import pytest

from helpers import check_energy, live_mask, to_oracle_params

pytestmark = pytest.mark.gpu



def sphere_scene(oracle, m):
    from smoothed_particle_hydrodynamics_amd.lib import default_params
    p = default_params()
    n = m * 1024
    pos, vel = oracle.init_sphere(to_oracle_params(p), n)
    return p, pos, vel, np.ones(n, np.float32)


def dense_scene(n):
    from smoothed_particle_hydrodynamics_amd import scenes
    return scenes.dense_block(n)


def edge_scene():
    """particles on voxel index 0 (excluded by the shipped search), outside the box (clamped),
    exactly on voxel faces, duplicates (distance 0), and one crowded voxel"""
    from smoothed_particle_hydrodynamics_amd import scenes
    p, pos, vel, mass = scenes.dense_block(4096, lo=(-0.3, 0.0, 0.1), hi=(0.9, 0.7, 0.8), seed=3)
    pos = pos.reshape(-1, 3)
    pos[:64] = pos[64:128]                      # exact duplicates
    pos[128:192] = np.float32(0.2) * np.round(pos[128:192] / np.float32(0.2))  # on faces
    pos[192:200] = [7.5, 3.0, -2.0]             # outside the box
    pos[200:520] = np.float32([3.3, 3.3, 3.3]) + np.float32(0.19) * (pos[200:520] % 1.0)
    return p, np.ascontiguousarray(pos.reshape(-1)), vel, mass


SCENES = {
    "sphere_8k": lambda o: sphere_scene(o, 8),
    "sphere_32k": lambda o: sphere_scene(o, 32),
    "dense_16k": lambda o: dense_scene(16384),
    "edges_4k": lambda o: edge_scene(),
}


@pytest.mark.parametrize("scene", sorted(SCENES))
def test_ref_phases_match_oracle(oracle, hiplib, scene):
    import smoothed_particle_hydrodynamics_amd as S
    p, pos, vel, mass = SCENES[scene](oracle)
    op = to_oracle_params(p)
    n = mass.size
    cap = p.examine_count

    with S.SPH(n, p, mode=S.MODE_REF) as sph:
        sph.setParticles(pos, vel, mass)

        # A1 voxelize
        sph.voxelizeParticles()
        coords, ids = sph.voxels()
        ocoords, oids, cs, ci = oracle.voxelize(op, pos)
        assert np.array_equal(coords, ocoords)
        assert np.array_equal(ids, oids)
        assert np.array_equal(sph.getGrid(), np.diff(cs))

        # A2 findNeighbors
        sph.findNeighbors()
        nb, nd = sph.neighborLists()
        cnt = sph.syncParticles().mNeighborCount.copy()
        onb, ond, ocnt = oracle.find_neighbors(op, pos, ocoords, cs, ci)
        assert np.array_equal(cnt, ocnt)
        m = live_mask(ocnt, cap)
        assert np.array_equal(nb[m], onb[m])
        assert np.array_equal(nd[m], ond[m])
        assert sph.neighborStats() == oracle.neighbor_stats(ocnt)

        # A4 density
        sph.computeDensity()
        rho = sph.syncParticles().mDensity.copy()
        orho = oracle.density_lists(op, cap, onb, ond, ocnt, mass)
        assert np.array_equal(rho, orho)

        # A5 acceleration
        sph.computeAcceleration()
        acc = sph.syncParticles().mAcceleration.copy()
        oacc = oracle.accel_lists(op, cap, onb, ond, ocnt, pos, vel, mass, orho)
        assert np.array_equal(acc, oacc)

        # A6 integrate
        sph.integrate()
        part = sph.syncParticles()
        opos, ovel = pos.copy(), vel.copy()
        oke, ope = oracle.integrate(op, opos, ovel, oacc, mass)
        assert np.array_equal(part.mPosition, opos)
        assert np.array_equal(part.mVelocity, ovel)
        check_energy(sph.energy(), (oke, ope), part.mVelocity, mass)


@pytest.mark.parametrize("scene,steps", [("dense_16k", 5), ("sphere_32k", 3)])
def test_ref_multi_step_matches_oracle(oracle, hiplib, scene, steps):
    import smoothed_particle_hydrodynamics_amd as S
    p, pos, vel, mass = SCENES[scene](oracle)
    op = to_oracle_params(p)
    opos, ovel = pos.copy(), vel.copy()
    with S.SPH(mass.size, p, mode=S.MODE_REF) as sph:
        sph.setParticles(pos, vel, mass)
        for _ in range(steps):
            sph.step()
            out = oracle.step(op, opos, ovel, mass, mode="ref")
        part = sph.getParticles()
        assert np.array_equal(part.mNeighborCount, out["ncount"])
        assert np.array_equal(part.mDensity, out["rho"])
        assert np.array_equal(part.mAcceleration, out["acc"])
        assert np.array_equal(part.mPosition, opos)
        assert np.array_equal(part.mVelocity, ovel)
        ms = sph.elapsed()
        assert len(ms) == 6 and all(t >= 0 for t in ms) and ms[3] < 0.5  # pressure phase is empty


def test_ref_setters_take_effect_next_step(oracle, hiplib):
    """The six GUI setters (reference src/sph.cpp:1225-1289) applied between steps."""
    import smoothed_particle_hydrodynamics_amd as S
    p, pos, vel, mass = dense_scene(8192)
    opos, ovel = pos.copy(), vel.copy()
    with S.SPH(mass.size, p, mode=S.MODE_REF) as sph:
        sph.setParticles(pos, vel, mass)
        sph.step()
        oracle.step(to_oracle_params(p), opos, ovel, mass, mode="ref")
        sph.setStiffness(0.004)
        sph.setViscosityScalar(0.02)
        sph.setTimeStep(0.0005)
        sph.setCflLimit(50.0)
        sph.setDamping(0.5)
        sph.setGravity((0.0, -9.8, 0.0))  # stored, never applied — exactly like the reference
        q = sph.getParams()
        assert q.cfl_limit2 == 2500.0
        sph.step()
        out = oracle.step(to_oracle_params(q), opos, ovel, mass, mode="ref")
        part = sph.getParticles()
        assert np.array_equal(part.mAcceleration, out["acc"])
        assert np.array_equal(part.mPosition, opos)
        assert np.array_equal(part.mVelocity, ovel)


def test_run_to_files_writes_the_reference_outputs(oracle, hiplib, tmp_path):
    """SPH::run()'s four output files (reference src/sph.cpp:162-178, 232)"""
    import smoothed_particle_hydrodynamics_amd as S
    p, pos, vel, mass = dense_scene(8192)
    op = to_oracle_params(p)
    opos, ovel = pos.copy(), vel.copy()
    out = str(tmp_path / "out")
    with S.SPH(mass.size, p, mode=S.MODE_REF) as sph:
        sph.setParticles(pos, vel, mass)
        sph.runToFiles(2, out)
    lines = {f: open(out + "/" + f).read().strip().splitlines()
             for f in ("energy.txt", "angularmomentum.txt", "timing.txt", "neighbors.txt")}
    assert lines["energy.txt"][0] == "Step, Kinetic Energy, Potential Energy, Total Energy"
    assert lines["timing.txt"][0].startswith("Step, Voxelize, Find Neighbors, Compute Density")
    assert len(lines["energy.txt"]) == 4 and len(lines["neighbors.txt"]) == 3   # steps 0..2
    for s in range(3):
        ref = oracle.step(op, opos, ovel, mass, mode="ref")
        assert lines["neighbors.txt"][s] == "%d, %d, %d" % oracle.neighbor_stats(ref["ncount"])
        step, ke, pe, tot = [float(v) for v in lines["energy.txt"][s + 1].split(",")]
        assert step == s
        assert ke == pytest.approx(ref["ke"], rel=1e-4) and pe == pytest.approx(ref["pe"], rel=1e-4)
        assert len(lines["timing.txt"][s + 1].split(",")) == 7


@pytest.mark.parametrize("m", [8, 32])
def test_ref_500_steps_match_reference_golden(oracle, hiplib, m):
    """BASELINE configs[0]: the default scene, 500 steps (the reference's run loop does 1001,
    src/sph.cpp:69-71,171).  Expected values come from the reference's own compiled sph.cpp
    (tests/golden/make_golden.py): SHA-256 of every per-particle array at steps 100, 250 and 500.
    By step 500 hundreds of particles have left the box and are clamped into edge voxels
    (src/sph.cpp:456-463)."""
    import json
    import os
    import smoothed_particle_hydrodynamics_amd as S
    from helpers import sha
    golden = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                         "golden.json")))["ref_sphere_M%d_steps500" % m]
    p, pos, vel, mass = sphere_scene(oracle, m)
    assert sha(pos) == golden["input_sha256"]["pos"] and sha(vel) == golden["input_sha256"]["vel"]
    with S.SPH(mass.size, p, mode=S.MODE_REF) as sph:
        sph.setParticles(pos, vel, mass)
        done = 0
        for upto in (100, 250, 500):
            sph.run(upto - done)
            done = upto
            part = sph.getParticles()
            got = dict(pos=part.mPosition, vel=part.mVelocity, rho=part.mDensity,
                       acc=part.mAcceleration, ncount=part.mNeighborCount)
            want = golden["checkpoints"][str(upto)]
            for name, h in want["sha256"].items():
                assert sha(got[name]) == h, "M=%d %s at step %d" % (m, name, upto)
            assert int(part.mNeighborCount.sum()) == want["neighbors_total"]
            check_energy(sph.energy(), (want["ke"], want["pe"]), part.mVelocity, mass)
        x = part.mPosition.reshape(-1, 3)
        outside = ((x < 0) | (x >= np.float32([p.max_x, p.max_y, p.max_z]))).any(axis=1)
        assert int(outside.sum()) == golden["particles_outside_box"]
        assert sph.getGrid().sum() == mass.size       # everybody is in some voxel, clamped or not


def test_ref_100k_particles_clamped_into_one_voxel(oracle, hiplib):
    """100 000 particles outside the box: the reference clamps them all into one edge voxel
    (src/sph.cpp:456-463) and appends them to its list in index order.  The per-voxel ascending
    lists come from k_rank_big's sort (25 LDS chunks + merge by binary search) instead of every
    member scanning 100 000 entries; lists, sums and new state equal the oracle's, in bounded time."""
    import time
    import smoothed_particle_hydrodynamics_amd as S
    from smoothed_particle_hydrodynamics_amd import scenes
    p, far, _, _ = scenes.dense_block(100000, lo=(6.45, 6.5, 6.6), hi=(6.95, 7.0, 7.1), seed=31)
    _, box, _, _ = scenes.dense_block(12000, lo=(1.0, 1.0, 1.0), hi=(2.2, 2.2, 2.2), seed=32)
    pos = np.concatenate([far, box])
    n = pos.size // 3
    order = np.random.default_rng(5).permutation(n)
    pos = np.ascontiguousarray(pos.reshape(-1, 3)[order]).reshape(-1)
    vel = scenes.box_fill(n, (-1.0,) * 3, (1.0,) * 3, 33)
    mass = np.ones(n, np.float32)
    op = to_oracle_params(p)
    opos, ovel = pos.copy(), vel.copy()
    with S.SPH(n, p, mode=S.MODE_REF) as sph:
        sph.setParticles(pos, vel, mass)
        t0 = time.perf_counter()
        sph.step()
        sph.synchronize()
        assert time.perf_counter() - t0 < 20.0
        ref = oracle.step(op, opos, ovel, mass, mode="ref")
        part = sph.getParticles()
        assert np.array_equal(part.mNeighborCount, ref["ncount"])
        assert np.array_equal(part.mDensity, ref["rho"])
        assert np.array_equal(part.mAcceleration, ref["acc"])
        assert np.array_equal(part.mPosition, opos)
        assert np.array_equal(part.mVelocity, ovel)
        assert sph.getGrid().max() >= 100000
