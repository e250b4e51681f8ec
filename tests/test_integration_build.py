"""CPU: the C++ drop-in must compile against the reference's UNCHANGED header (where the
reference tree and the Qt5Core headers of this image are present)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dropin_compiles_against_reference_header():
    ref = "/root/reference/src"
    qt = "/opt/conda/include/qt"
    if not (os.path.isdir(ref) and os.path.isdir(qt)):
        pytest.skip("reference tree / Qt headers not present on this machine")
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-fPIC", "-D_USE_MATH_DEFINES", "-Wall",
           "-I" + ref, "-I" + qt, "-I" + qt + "/QtCore", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "integration", "sph_dropin.cpp")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_dropin_uses_only_declared_entry_points():
    import re
    text = open(os.path.join(ROOT, "integration", "sph_dropin.cpp")).read()
    header = open(os.path.join(ROOT, "include", "sph_hip.h")).read()
    used = set(re.findall(r"\b(sph_hip_[a-z_0-9]+)\s*\(", text))
    declared = set(re.findall(r"\b(sph_hip_[a-z_0-9]+)\s*\(", re.sub(r"/\*.*?\*/", "", header, flags=re.S)))
    assert used and used <= declared, used - declared
