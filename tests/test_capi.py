"""CPU-side checks of the C-ABI boundary: the library loads, exports exactly what
include/sph_hip.h declares, derives the reference's constants, and refuses to run without a
GPU instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import to_oracle_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sph_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sph_hip_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "sph_hip_step" in syms and "sph_hip_create" in syms and len(syms) >= 20


def test_library_exports_every_declared_symbol(hiplib):
    from smoothed_particle_hydrodynamics_amd.lib import PROTOTYPES
    syms = declared_symbols()
    for s in syms:
        assert hasattr(hiplib, s), "libsph_hip.so does not export %s" % s
    # the ctypes binding covers the whole header too
    assert sorted(PROTOTYPES) == syms


def test_param_struct_layout_matches_oracle(hiplib):
    from oracle.oracle import OracleParams
    from smoothed_particle_hydrodynamics_amd import SphParams
    assert C.sizeof(OracleParams) == C.sizeof(SphParams)
    assert [f[0] for f in OracleParams._fields_] == [f[0] for f in SphParams._fields_]


@pytest.mark.parametrize("h,cells", [(0.1, (32, 32, 32)), (0.05, (16, 24, 8)),
                                     (0.0051501622, (98, 98, 98)), (0.37, (5, 3, 9))])
def test_params_default_matches_oracle(hiplib, oracle, h, cells):
    """sph_hip_params_default == the restated constructor constants (reference
    src/sph.cpp:46-98), byte for byte."""
    from smoothed_particle_hydrodynamics_amd.lib import default_params
    p = default_params(h, cells)
    o = oracle.params_for_h(h, cells)
    assert bytes(p) == bytes(o)


def test_reference_default_constants(hiplib):
    """k1 = 1.56668134e9, k2 = -14323942 for h = 0.1"""
    from smoothed_particle_hydrodynamics_amd.lib import default_params
    p = default_params()
    assert np.float32(p.kernel1) == np.float32(1.56668134e9)
    assert np.float32(p.kernel2) == np.float32(-14323942.0)
    assert p.kernel3 == -p.kernel2
    assert (p.cells_x, p.cells_y, p.cells_z) == (32, 32, 32)
    assert np.float32(p.max_x) == np.float32(6.4)
    assert p.examine_count == 32


def test_invalid_arguments_are_reported(hiplib):
    from smoothed_particle_hydrodynamics_amd import SphParams
    p = SphParams()
    assert hiplib.sph_hip_params_default(C.byref(p), -1.0, 32, 32, 32) == -1
    assert hiplib.sph_hip_params_default(C.byref(p), 0.1, 0, 32, 32) == -1
    ctx = C.c_void_p()
    assert hiplib.sph_hip_create(C.byref(ctx), C.byref(p), 0, 0, 0) == -1
    assert hiplib.sph_hip_last_error(None)  # a message, never NULL


def test_no_gpu_means_loud_failure_not_fallback(hiplib):
    """Without a device the product refuses to construct; with one it constructs.  Either way
    nothing routes to the CPU oracle."""
    import smoothed_particle_hydrodynamics_amd as S
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        with S.SPH(1024) as s:
            assert s.getParticleCount() == 1024
    else:
        with pytest.raises(S.SphHipError, match="no usable HIP device"):
            S.SPH(1024)


def test_abi_version_is_exported_and_matches_the_header(hiplib):
    from smoothed_particle_hydrodynamics_amd import lib as L
    text = open(os.path.join(ROOT, "include", "sph_hip.h")).read()
    declared = int(re.search(r"#define\s+SPH_HIP_ABI_VERSION\s+(\d+)", text).group(1))
    assert hiplib.sph_hip_abi_version() == declared == L.ABI_VERSION


def test_missing_rccl_is_an_error_message_not_a_crash():
    """librccl cannot be opened (here: forced to a name that does not exist): the entry points
    that need it return an error with the loader's message.  (The branch used to call dlerror()
    twice - the second call returns NULL - and build a std::string from it.)  Own process: the
    library looks RCCL up once per process."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "from smoothed_particle_hydrodynamics_amd.lib import load_library\n"
        "lib = load_library()\n"
        "buf = (C.c_char * 128)()\n"
        "rc = lib.sph_hip_rccl_unique_id(buf, 128)\n"
        "print('rc', rc, lib.sph_hip_last_error(None).decode())\n" % ROOT)
    env = dict(os.environ, SPH_HIP_RCCL_LIBRARY="/nonexistent/librccl-not-here.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rc -1" in out.stdout and "cannot open librccl" in out.stdout
    assert "librccl-not-here" in out.stdout      # the loader's own text made it through


def test_slab_thinner_than_its_halos_is_refused(hiplib):
    """sph_hip_create_slab: a slab next to another needs 2 * SPH_HIP_SLAB_HALO planes (what
    slab.plan_cuts plans with); the check comes before any device is touched."""
    from smoothed_particle_hydrodynamics_amd.lib import default_params
    p = default_params(0.1, (8, 8, 8))
    ctx = C.c_void_p()
    nz = p.full_cells_z
    assert nz >= 16
    for lo, hi in ((4, 7), (0, 3), (nz - 2, nz)):
        rc = hiplib.sph_hip_create_slab(C.byref(ctx), C.byref(p), 1024, 0, lo, hi)
        msg = hiplib.sph_hip_last_error(None).decode()
        # without a GPU the device check comes first (-4); with one the thickness check answers
        assert rc in (-1, -4), (lo, hi, rc)
        if rc == -1:
            assert "2 * SPH_HIP_SLAB_HALO" in msg


def test_missing_library_is_loud(tmp_path):
    from smoothed_particle_hydrodynamics_amd.lib import SphHipError, load_library
    with pytest.raises(SphHipError, match="no CPU fallback"):
        load_library(str(tmp_path / "libsph_hip.so"))


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ in any way."""
    pkg = os.path.join(ROOT, "smoothed_particle_hydrodynamics_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), "%s mentions the oracle" % f


def test_reference_sphere_scene_is_bit_identical(hiplib, oracle):
    """scenes.reference_sphere (libc rand() through ctypes) == the restated / compiled
    reference initial condition (reference src/sph.cpp:361-425)"""
    from smoothed_particle_hydrodynamics_amd import scenes
    n = 2048
    p, pos, vel, mass = scenes.reference_sphere(n)
    opos, ovel = oracle.init_sphere(to_oracle_params(p), n)
    assert np.array_equal(pos, opos)
    assert np.array_equal(vel, ovel)
    assert np.all(mass == 1.0)


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/sph_hip.h compiles as strict C99, and a C program linked against the library can
    call it: the constants come out, and without a GPU sph_hip_create fails loudly with a message
    (no GPU is needed for this test; with one, creation simply succeeds)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import smoothed_particle_hydrodynamics_amd as S
    lib = S.library_path()
    if not os.path.exists(lib):
        S.build_library()
    src = tmp_path / "use_abi.c"
    src.write_text(r'''
#include <stdio.h>
#include "sph_hip.h"
int main(void)
{
   sph_hip_params p;
   sph_hip_context* ctx = NULL;
   if (sph_hip_params_default(&p, 0.1f, 32, 32, 32) != SPH_HIP_OK) return 2;
   printf("h2 %.9g examine %d abi %d\n", (double)p.h2, (int)p.examine_count, SPH_HIP_ABI_VERSION);
   int rc = sph_hip_create(&ctx, &p, 1024, SPH_HIP_MODE_FULL, 0);
   if (rc == SPH_HIP_OK) { sph_hip_destroy(ctx); printf("created\n"); return 0; }
   printf("create failed (%d): %s\n", rc, sph_hip_last_error(NULL));
   return ctx == NULL ? 0 : 3;
}
''')
    exe = tmp_path / "use_abi"
    inc = os.path.join(root, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c",
                    str(src), "-o", str(tmp_path / "use_abi.o")], check=True)
    subprocess.run(["gcc", str(tmp_path / "use_abi.o"), "-o", str(exe), lib,
                    "-Wl,-rpath," + os.path.dirname(lib), "-Wl,--allow-shlib-undefined"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "h2 0.01" in out.stdout and "examine 32" in out.stdout
    assert "created" in out.stdout or "create failed" in out.stdout
