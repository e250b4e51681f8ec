"""The CPU restatement (oracle/sph_oracle.c) against golden vectors produced by the reference's
own compiled sph.cpp (tests/golden/make_golden.py).  Runs anywhere — no reference tree, no GPU.

Every output array of every case must hash to the golden SHA-256 (bit-exact), and the stored
sample rows must match, which also gives readable known answers when a hash breaks.
"""
import json
import math
import os

import numpy as np
import pytest

from helpers import sha

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))
SAMPLES = np.load(os.path.join(HERE, "golden", "golden_samples.npz"))


def box_fill(n, lo, hi, seed):
    # numpy twin of scenes.box_fill, so this file needs neither the product library nor a GPU
    idx = np.arange(3 * n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = z + np.uint64(seed) * np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    u = ((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(n, 3)
    lo = np.asarray(lo, np.float32)
    ext = np.asarray(hi, np.float32) - lo
    return np.ascontiguousarray((lo + u * ext).astype(np.float32).reshape(-1))


def check(case, got):
    g = GOLDEN[case]
    for name, want in g["sha256"].items():
        a = got[name]
        per = a.size // got["ncount"].size if name != "ncount" else 1
        rows = np.ascontiguousarray(a.reshape(-1, per)[::g["stride"]])
        assert np.array_equal(rows, SAMPLES["%s/%s" % (case, name)]), "%s %s sample rows" % (case, name)
        assert sha(a) == want, "%s %s hash" % (case, name)


def inputs(oracle, case):
    g = GOLDEN[case]
    n = g["n"]
    if g.get("scene") == "sphere":
        p = oracle.params_for_h(0.1)
        pos, vel = oracle.init_sphere(p, n)
        assert [float(v) for v in pos[:3]] == g["first_particle"]["pos"]
        assert [float(v) for v in vel[:3]] == g["first_particle"]["vel"]
        mass = np.ones(n, np.float32)
    elif g.get("scene") == "dense_block":
        p = oracle.params_for_h(0.1)
        pos = box_fill(n, (1.0, 1.0, 1.0), (2.2, 2.2, 2.2), 7)
        vel = box_fill(n, (-0.5,) * 3, (0.5,) * 3, 8)
        mass = np.ones(n, np.float32)
    elif case == "full_dense_6000":
        p = oracle.params_for_h(0.1)
        pos = box_fill(n, (1.0, 1.0, 1.0), (2.2, 2.2, 2.2), 7)
        vel = box_fill(n, (-5.0,) * 3, (5.0,) * 3, 8)
        mass = np.ones(n, np.float32)
    elif case == "full_dambreak_8000_unequal_mass":
        p = oracle.params_for_h(g["h"], g["cells"])
        p.central_mass = 0.0
        pos = box_fill(n, (0, 0, 0), (0.1, 0.75, 1.0), 42)
        vel = np.zeros(3 * n, np.float32)
        mass = (0.5 + box_fill(n, (0,) * 3, (1,) * 3, 11)[:n]).astype(np.float32)
    else:
        raise KeyError(case)
    for name, want in g["input_sha256"].items():
        assert sha({"pos": pos, "vel": vel, "mass": mass}[name]) == want, "input " + name
    return p, pos, vel, mass


def test_first_particle_of_default_scene_matches_survey(oracle):
    """SURVEY.md §8(a) A0': first particle of the srand(42) sphere"""
    p = oracle.params_for_h(0.1)
    pos, vel = oracle.init_sphere(p, 8192)
    assert np.allclose(pos[:3], [1.4006387, 2.90813398, 3.31938601], rtol=0, atol=1e-7)
    assert np.allclose(vel[:3], [-0.96650362, 0.0369208753, -14.5669632], rtol=0, atol=1e-6)


@pytest.mark.parametrize("case", sorted(k for k, v in GOLDEN.items() if v["mode"] == "ref"))
def test_ref_mode_steps_match_reference_golden(oracle, case):
    g = GOLDEN[case]
    p, pos, vel, mass = inputs(oracle, case)
    marks = {int(k): v for k, v in g.get("checkpoints", {}).items()}
    for step in range(1, g["steps"] + 1):
        out = oracle.step(p, pos, vel, mass, mode="ref")
        if step in marks:   # long runs: the state on the way, so that a break can be dated
            got = dict(pos=pos, vel=vel, rho=out["rho"], acc=out["acc"], ncount=out["ncount"])
            for name, want in marks[step]["sha256"].items():
                assert sha(got[name]) == want, "%s %s at step %d" % (case, name, step)
            assert out["ke"] == marks[step]["ke"] and out["pe"] == marks[step]["pe"]
    if "particles_outside_box" in g:   # the clamp path of voxelizeParticles is really exercised
        x = pos.reshape(-1, 3)
        outside = ((x < 0) | (x >= np.float32([p.max_x, p.max_y, p.max_z]))).any(axis=1)
        assert int(outside.sum()) == g["particles_outside_box"] > 50
    check(case, dict(pos=pos, vel=vel, rho=out["rho"], acc=out["acc"], ncount=out["ncount"]))
    assert int(out["ncount"].sum()) == g["neighbors_total"]
    assert out["ke"] == g["ke"] and out["pe"] == g["pe"]     # same serial fp32 sums


def test_ref_mode_lists_match_reference_golden(oracle):
    case = "ref_dense_16384_steps1"
    g = GOLDEN[case]["lists_after_search"]
    p, pos, vel, mass = inputs(oracle, case)
    coords, ids, cs, ci = oracle.voxelize(p, pos)
    nb, nd, cnt = oracle.find_neighbors(p, pos, coords, cs, ci)
    live = (np.arange(p.examine_count)[None, :] < cnt[:, None]).ravel()
    assert int(cnt.max()) == g["count_max"] and g["count_max"] <= 28
    assert sha(nb[live]) == g["nb"]
    assert sha(nd[live]) == g["nd"]


def test_survey_neighbor_totals(oracle):
    """the shipped search finds 0 / 6121 neighbours at M=8 / M=32"""
    assert GOLDEN["ref_sphere_M8_steps1"]["neighbors_total"] == 0
    assert GOLDEN["ref_sphere_M32_steps1"]["neighbors_total"] == 6121


@pytest.mark.parametrize("case", sorted(k for k, v in GOLDEN.items() if v["mode"] == "full"))
def test_full_mode_matches_reference_pair_functions(oracle, case):
    """oracle FULL mode (its own 27-cell search + restated pair arithmetic) == the reference's
    computeDensity/computeAcceleration/integrate run on brute-force canonical lists"""
    g = GOLDEN[case]
    p, pos, vel, mass = inputs(oracle, case)
    out = oracle.step(p, pos, vel, mass, mode="full")
    check(case, dict(pos=pos, vel=vel, rho=out["rho"], acc=out["acc"], ncount=out["ncount"]))
    assert int(out["ncount"].max()) == g["neighbors_max"]
    assert out["ke"] == g["ke"] and out["pe"] == g["pe"]
