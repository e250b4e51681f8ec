"""ctypes binding of include/sph_hip.h.  Fails loudly when the HIP library is absent."""
import ctypes as C
import os

from .build import library_path

MODE_REF = 0
MODE_FULL = 1
MODE_FULL_FAST = 2   # FULL with tolerance-mode pair arithmetic (include/sph_hip.h)
ARITH_EXACT, ARITH_FAST = 0, 1
ABI_DIAGNOSTIC = 0x4000
# sph_hip_set_timing levels (include/sph_hip.h)
TIMING_OFF, TIMING_SUMS, TIMING_PHASES = 0, 1, 2


class SphHipError(RuntimeError):
    pass


class SphParams(C.Structure):
    """Mirror of sph_hip_params (include/sph_hip.h) = the protected constants of the
    reference's SPH class (reference src/sph.h:149-210)."""

    _fields_ = [
        ("cells_x", C.c_int32), ("cells_y", C.c_int32), ("cells_z", C.c_int32),
        ("cell_size", C.c_float),
        ("max_x", C.c_float), ("max_y", C.c_float), ("max_z", C.c_float),
        ("h", C.c_float), ("h2", C.c_float), ("hscaled", C.c_float), ("hscaled2", C.c_float),
        ("hscaled6", C.c_float), ("hscaled9", C.c_float), ("htimes2", C.c_float),
        ("htimes2inv", C.c_float), ("sim_scale", C.c_float), ("sim_scale_inv", C.c_float),
        ("kernel1", C.c_float), ("kernel2", C.c_float), ("kernel3", C.c_float),
        ("rho0", C.c_float), ("stiffness", C.c_float), ("viscosity", C.c_float),
        ("time_step", C.c_float), ("damping", C.c_float),
        ("cfl_limit", C.c_float), ("cfl_limit2", C.c_float),
        ("gravity", C.c_float * 3),
        ("grav_const", C.c_float), ("central_mass", C.c_float), ("central_pos", C.c_float * 3),
        ("softening", C.c_float),
        ("examine_count", C.c_int32),
        ("full_cells_x", C.c_int32), ("full_cells_y", C.c_int32), ("full_cells_z", C.c_int32),
        ("full_cell_inv", C.c_float),
        ("apply_gravity", C.c_int32), ("apply_walls", C.c_int32),
    ]

    def copy(self):
        other = SphParams()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(SphParams))
        return other

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


# every symbol include/sph_hip.h declares: name -> (restype, argtypes)
_P = C.POINTER
_ctx = C.c_void_p
PROTOTYPES = {
    "sph_hip_params_default": (C.c_int, [_P(SphParams), C.c_float, C.c_int, C.c_int, C.c_int]),
    "sph_hip_create": (C.c_int, [_P(_ctx), _P(SphParams), C.c_int, C.c_int, C.c_int]),
    "sph_hip_destroy": (None, [_ctx]),
    "sph_hip_last_error": (C.c_char_p, [_ctx]),
    "sph_hip_set_params": (C.c_int, [_ctx, _P(SphParams)]),
    "sph_hip_get_params": (C.c_int, [_ctx, _P(SphParams)]),
    "sph_hip_set_arithmetic": (C.c_int, [_ctx, C.c_int]),
    "sph_hip_get_arithmetic": (C.c_int, [_ctx]),
    "sph_hip_upload": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sph_hip_download": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sph_hip_particle_count": (C.c_int, [_ctx]),
    "sph_hip_download_async": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, _P(C.c_int)]),
    "sph_hip_download_done": (C.c_int, [_ctx, C.c_int]),
    "sph_hip_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "sph_hip_host_unregister": (C.c_int, [C.c_void_p]),
    "sph_hip_step": (C.c_int, [_ctx]),
    "sph_hip_run": (C.c_int, [_ctx, C.c_int]),
    "sph_hip_voxelize": (C.c_int, [_ctx]),
    "sph_hip_find_neighbors": (C.c_int, [_ctx]),
    "sph_hip_compute_density": (C.c_int, [_ctx]),
    "sph_hip_compute_acceleration": (C.c_int, [_ctx]),
    "sph_hip_integrate": (C.c_int, [_ctx]),
    "sph_hip_synchronize": (C.c_int, [_ctx]),
    "sph_hip_get_timings": (C.c_int, [_ctx, _P(C.c_float * 6)]),
    "sph_hip_get_phase_totals": (C.c_int, [_ctx, _P(C.c_double * 6), _P(C.c_int32)]),
    "sph_hip_reset_timings": (C.c_int, [_ctx]),
    "sph_hip_set_timing": (C.c_int, [_ctx, C.c_int]),
    "sph_hip_set_timing_stride": (C.c_int, [_ctx, C.c_int]),
    "sph_hip_get_tile_stats": (C.c_int, [_ctx, _P(C.c_int32 * 20)]),
    "sph_hip_get_energy": (C.c_int, [_ctx, _P(C.c_float), _P(C.c_float)]),
    "sph_hip_get_neighbor_stats": (C.c_int, [_ctx, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "sph_hip_download_voxels": (C.c_int, [_ctx, C.c_void_p, C.c_void_p]),
    "sph_hip_download_grid_counts": (C.c_int, [_ctx, C.c_void_p]),
    "sph_hip_download_neighbor_lists": (C.c_int, [_ctx, C.c_void_p, C.c_void_p]),
    "sph_hip_stream": (C.c_void_p, [_ctx]),
    "sph_hip_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "sph_hip_create_slab": (C.c_int, [_P(_ctx), _P(SphParams), C.c_int, C.c_int, C.c_int, C.c_int]),
    "sph_hip_slab_upload": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int]),
    "sph_hip_slab_download": (C.c_int, [_ctx, C.c_int, _P(C.c_int32), C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sph_hip_slab_download_mass": (C.c_int, [_ctx, C.c_int, _P(C.c_int32), C.c_void_p]),
    "sph_hip_slab_export_records": (C.c_int, [_ctx, C.c_void_p, C.c_int, _P(C.c_int32)]),
    "sph_hip_slab_upload_records": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int]),
    "sph_hip_slab_message_bytes": (C.c_size_t, [C.c_int]),
    "sph_hip_slab_pack": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int]),
    "sph_hip_slab_unpack": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int]),
    "sph_hip_slab_step_begin": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "sph_hip_slab_step_end": (C.c_int, [_ctx]),
    "sph_hip_rccl_unique_id": (C.c_int, [C.c_void_p, C.c_int]),
    "sph_hip_slab_comm_init": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sph_hip_slab_comm_run": (C.c_int, [_ctx, C.c_int]),
    "sph_hip_slab_comm_trim": (C.c_int, [_ctx, C.c_float, C.c_int, _P(C.c_int32)]),
    "sph_hip_slab_comm_selftest": (C.c_int, [_ctx]),
    "sph_hip_slab_comm_exchange_check": (C.c_int, [_ctx]),
    "sph_hip_slab_comm_stats": (C.c_int, [_ctx, _P(C.c_int32)]),
    "sph_hip_slab_status": (C.c_int, [_ctx, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "sph_hip_slab_poll_errors": (C.c_int, [_ctx, _P(C.c_int32)]),
    "sph_hip_abi_version": (C.c_int, []),
    "sph_hip_selftest_sqrt": (C.c_int, [C.c_int, _P(C.c_uint64), _P(C.c_uint32)]),
}
ABI_VERSION = 6   # SPH_HIP_ABI_VERSION of the include/sph_hip.h these prototypes mirror

_LIB = None


def load_library(path=None):
    """Load libsph_hip.so and bind every entry point of include/sph_hip.h.

    There is deliberately no fallback: if the library has not been built (or cannot be
    loaded) this raises, and so does everything that depends on it.
    """
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    path = path or library_path()
    # Plumbing: when PyTorch is installed it must bring up ITS ROCm runtime before this library
    # touches HIP — two independently initialised HIP runtimes in one process leave the second
    # one without a device ("no ROCm-capable device is detected").  torch is only imported, never
    # used here; the C++ host (integration/) has no such concern.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise SphHipError(
            "libsph_hip.so is missing (%s): build it with "
            "`python -m smoothed_particle_hydrodynamics_amd.build`; there is no CPU fallback" % path)
    try:
        lib = C.CDLL(path)
    except OSError as exc:  # pragma: no cover - depends on the machine
        raise SphHipError("cannot load %s: %s" % (path, exc)) from exc
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    abi = lib.sph_hip_abi_version()
    if abi & ABI_DIAGNOSTIC and os.environ.get("SPH_HIP_ALLOW_DIAGNOSTIC") != "1":
        raise SphHipError("%s is a diagnostic build (profiling hooks that cut pieces out of the "
                          "kernels: its results are garbage by design); only tools/ablate.py, with "
                          "SPH_HIP_ALLOW_DIAGNOSTIC=1, may load one" % path)
    if abi & ~ABI_DIAGNOSTIC != ABI_VERSION:
        raise SphHipError("%s has ABI version %d, this binding expects %d: rebuild it" %
                          (path, abi & ~ABI_DIAGNOSTIC, ABI_VERSION))
    _LIB = lib
    return lib


def default_params(h=0.1, cells=(32, 32, 32)):
    """SPH::SPH()'s constants (reference src/sph.cpp:46-98) for smoothing length h."""
    lib = load_library()
    p = SphParams()
    rc = lib.sph_hip_params_default(C.byref(p), h, int(cells[0]), int(cells[1]), int(cells[2]))
    if rc != 0:
        raise SphHipError("sph_hip_params_default failed (%d)" % rc)
    return p
