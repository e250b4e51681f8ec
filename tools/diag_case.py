import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smoothed_particle_hydrodynamics_amd as S
from oracle.oracle import Oracle
from helpers import to_oracle_params, vec_rel
from test_gpu_random_scenes import draw
case = int(sys.argv[1])
p, pos, vel, mass = draw(case)
op = to_oracle_params(p); orc = Oracle()
with S.SPH(mass.size, p, mode=S.MODE_FULL_FAST) as sph:
    sph.setParticles(pos, vel, mass); sph.step(); part = sph.getParticles()
    fa, fr = part.mAcceleration.copy(), part.mDensity.copy()
opos, ovel = pos.copy(), vel.copy()
ref = orc.step(op, opos, ovel, mass, mode="full")
T = orc.full_accel_scale(op, pos, vel, mass, ref["rho"])
rel = vec_rel(fa, ref["acc"])
bad = np.argsort(rel)[-4:]
for i in bad:
    rho = ref["rho"][i]; pi = (rho - p.rho0) * p.stiffness
    rinv = 1.0 / pi if pi > 0 else 1.0
    print("particle", i, "rel", rel[i], "cnt", ref["ncount"][i], "rho", rho, "p_i", pi, "s", p.viscosity * rinv,
          "A", pi * rinv * rinv, "acc fast", fa[3*i:3*i+3], "acc ref", ref["acc"][3*i:3*i+3], "T", T[i], "rho fast", fr[i])
