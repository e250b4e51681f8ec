"""GPU parity against the committed FULL-mode golden vectors directly.

tests/golden/golden.json holds SHA-256 hashes (and sampled rows) of what the REFERENCE's own
compiled computeDensity / computeAcceleration / integrate produce on brute-force canonical
neighbour lists (tests/golden/make_golden.py, generated from oracle/_ref/libsphref.so).  The
oracle is pinned to them on the CPU (tests/test_oracle_golden.py); here the HIP FULL path is held
to the same hashes without the oracle in between: unequal masses (the mass gathers), a dense block
in motion with the point mass on (the viscous sum, central gravity)."""
import numpy as np
import pytest

from helpers import to_product_params
from test_oracle_golden import GOLDEN, check, inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", sorted(k for k, v in GOLDEN.items() if v["mode"] == "full"))
@pytest.mark.parametrize("route", ["tiled", "untiled"])
def test_hip_full_mode_hashes_to_the_reference_goldens(oracle, hiplib, case, route, monkeypatch):
    import smoothed_particle_hydrodynamics_amd as S
    g = GOLDEN[case]
    op, pos, vel, mass = inputs(oracle, case)          # (oracle: parameter block + input hashes only)
    p = to_product_params(op)
    if route == "untiled":
        monkeypatch.setenv("SPH_HIP_UNTILED", "1")
    with S.SPH(mass.size, p, mode=S.MODE_FULL) as sph:
        sph.setParticles(pos, vel, mass)
        sph.step()
        part = sph.getParticles()
        check(case, dict(pos=part.mPosition, vel=part.mVelocity, rho=part.mDensity,
                         acc=part.mAcceleration, ncount=part.mNeighborCount))
        assert int(part.mNeighborCount.max()) == g["neighbors_max"]
        ke, pe = sph.energy()
        assert ke == pytest.approx(g["ke"], rel=1e-5, abs=1e-30)
        assert pe == pytest.approx(g["pe"], rel=1e-5, abs=1e-30)
