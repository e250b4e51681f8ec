"""GPU: the C++ drop-in (integration/sph_dropin.cpp = the reference's `SPH` class body over the
C ABI, compiled against the reference's UNCHANGED sph.h together with its own particle.cpp,
vec3.cpp and moc output) run as a headless program, like the reference's `./sph r`.

The host mirrors it leaves behind — Particle::mPosition/mVelocity/mDensity/mAcceleration/
mNeighborCount — must hash to the golden vectors produced by the reference itself for its
default scene (N = 32*1024, srand(42) sphere).  The binary is built by `make -C integration`
where /root/reference exists and travels with the tree; elsewhere the test is skipped."""
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import sha

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "integration", "_ref", "sph_dropin_demo")
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))


@pytest.mark.parametrize("steps", [1, 3])
def test_dropin_program_reproduces_reference_goldens(hiplib, tmp_path, steps):
    if not os.path.exists(DEMO):
        pytest.skip("integration/_ref/sph_dropin_demo not built (needs the reference tree)")
    out = str(tmp_path / "state.bin")
    env = dict(os.environ)
    env.pop("SPH_HIP_FULL", None)
    try:
        res = subprocess.run([DEMO, str(steps), out], capture_output=True, text=True, timeout=120,
                             env=env)
    except OSError as exc:
        pytest.skip("cannot execute the demo binary here: %s" % exc)
    if res.returncode != 0 and "error while loading shared libraries" in res.stderr:
        pytest.skip("demo binary's shared libraries are not present here: " + res.stderr.strip())
    assert res.returncode == 0, res.stdout + res.stderr
    raw = open(out, "rb").read()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    assert n == 32 * 1024
    f = np.frombuffer(raw[4:4 + 4 * 10 * n], np.float32)
    pos, vel, rho, acc = f[:3 * n], f[3 * n:6 * n], f[6 * n:7 * n], f[7 * n:10 * n]
    cnt = np.frombuffer(raw[4 + 40 * n:4 + 44 * n], np.int32)
    in_grid = int(np.frombuffer(raw[4 + 44 * n:], np.int64)[0])
    g = GOLDEN["ref_sphere_M32_steps%d" % steps]["sha256"]
    assert sha(cnt) == g["ncount"]
    assert sha(rho) == g["rho"]
    assert sha(acc) == g["acc"]
    assert sha(pos) == g["pos"]
    assert sha(vel) == g["vel"]
    assert in_grid == n          # getGrid()[i].count() mirrors sum to the particle count
