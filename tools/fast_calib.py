import sys, os, numpy as np, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import smoothed_particle_hydrodynamics_amd as S
from smoothed_particle_hydrodynamics_amd import scenes
from oracle.oracle import Oracle
from helpers import to_oracle_params, vec_rel
orc=Oracle()
for n,speed in ((262144,0.05),(262144,0.0),(1000000,0.05)):
    p,pos,vel,mass=scenes.dam_break(n,speed=speed)
    op=to_oracle_params(p)
    with S.SPH(n,p,mode=S.MODE_FULL_FAST) as sph:
        sph.setParticles(pos,vel,mass); sph.step(); part=sph.getParticles()
        opos,ovel=pos.copy(),vel.copy()
        ref=orc.step(op,opos,ovel,mass,mode="full")
        T=orc.full_accel_scale(op,pos,vel,mass,ref["rho"])
        err=np.linalg.norm(part.mAcceleration.astype(np.float64).reshape(-1,3)-ref["acc"].astype(np.float64).reshape(-1,3),axis=1)
        rel=vec_rel(part.mAcceleration,ref["acc"])
        an=np.linalg.norm(ref["acc"].astype(np.float64).reshape(-1,3),axis=1)
        c=err/np.maximum(T,1e-300)
        print(n,speed,"rel max %.3g >1e-4: %d | err/T max %.3g p99.9 %.3g median %.3g | |a|/T median %.3g min %.3g | among rel>1e-4: err/T max %.3g, |a|/T max %.3g"%(
            rel.max(),(rel>1e-4).sum(),c.max(),np.quantile(c,0.999),np.median(c),np.median(an/T),(an/T).min(),
            c[rel>1e-4].max() if (rel>1e-4).any() else 0,(an/T)[rel>1e-4].max() if (rel>1e-4).any() else 0),flush=True)
