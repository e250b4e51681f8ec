import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import smoothed_particle_hydrodynamics_amd as S
from test_gpu_random_scenes import draw
from conftest import *  # noqa
import importlib
case = int(sys.argv[1]) if len(sys.argv) > 1 else 594
p, pos, vel, mass = draw(case)
print("n", mass.size, "h", p.h, "sim_scale", p.sim_scale, "central_mass", p.central_mass, "gravity", p.apply_gravity, "walls", p.apply_walls, "uniform", bool((mass == mass[0]).all()))
res = {}
for mode, name in ((S.MODE_FULL, "exact"), (S.MODE_FULL_FAST, "fast")):
    with S.SPH(mass.size, p, mode=mode) as sph:
        sph.setParticles(pos, vel, mass)
        sph.step()
        a = sph.getParticles()
        st1 = (a.mPosition.copy(), a.mVelocity.copy(), a.mDensity.copy(), a.mAcceleration.copy(), a.mNeighborCount.copy())
        sph.step()
        b = sph.getParticles()
        res[name] = (st1, (b.mPosition.copy(), b.mVelocity.copy(), b.mDensity.copy(), b.mAcceleration.copy(), b.mNeighborCount.copy()), sph.tileStats())
for step in (0, 1):
    ae = res["exact"][step][3].reshape(-1, 3); af = res["fast"][step][3].reshape(-1, 3)
    fe = np.isfinite(ae).all(axis=1); ff = np.isfinite(af).all(axis=1)
    bad = np.nonzero(fe != ff)[0]
    print("step", step, "finite exact", fe.sum(), "fast", ff.sum(), "differ at", bad[:10])
    for i in bad[:5]:
        print("  particle", i, "exact acc", ae[i], "fast acc", af[i], "rho", res["exact"][step][2][i], res["fast"][step][2][i], "count", res["exact"][step][4][i])
print(res["fast"][1][2])
