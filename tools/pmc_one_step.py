"""One 4M-particle dam-break step, for rocprofv3 --pmc passes (profiles/r1_pmc_notes.md):
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d out -- python3 tools/pmc_one_step.py
SPH_HIP_LIBRARY selects a diagnostic build (-DSPH_ABLATE=n)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smoothed_particle_hydrodynamics_amd as S
from smoothed_particle_hydrodynamics_amd import scenes
n = 4*1024*1024
p, pos, vel, mass = scenes.dam_break(n)
# SPH_PMC_MODE=fast: the tolerance-mode arithmetic (bench.py's headline)
sph = S.SPH(n, p, mode=S.MODE_FULL_FAST if os.environ.get("SPH_PMC_MODE") == "fast" else S.MODE_FULL)
sph.setParticles(pos, vel, mass)
sph.setTiming(S.TIMING_OFF)
for _ in range(3):
    sph.step()
sph.synchronize()
