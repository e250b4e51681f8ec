"""MI355X-native SPH step: the hot path of DanielaCourel/smoothed_particle_hydrodynamics
(reference src/sph.cpp:190-304) as hand-written HIP kernels for gfx950 behind a C ABI
(include/sph_hip.h).  This package is the host-side mirror of the reference's `SPH` /
`Particle` interface; there is no CPU fallback — without the HIP library every entry
point raises."""
from .build import build_library, library_path  # noqa: F401
from .lib import SphParams, load_library, SphHipError, default_params  # noqa: F401
from .sph import SPH, Particle, MODE_REF, MODE_FULL, MODE_FULL_FAST, ARITH_EXACT, ARITH_FAST  # noqa: F401
from .lib import TIMING_OFF, TIMING_SUMS, TIMING_PHASES  # noqa: F401
from . import scenes  # noqa: F401
