"""Builds csrc/ into smoothed_particle_hydrodynamics_amd/libsph_hip.so with hipcc for gfx950.

The library is built IN-TREE so that it travels with a snapshot of the repository; it is
git-ignored.  hipcc cross-compiles without a GPU present.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["sph_hip.hip"]
HEADERS = ["sph_device.h", "cell_build.h", "pair_math.h", "full_kernels.h", "full_tiled.h",
           "ref_kernels.h", "common_kernels.h", "slab_kernels.h", "slab_rccl.h",
           os.path.join("..", "..", "include", "sph_hip.h")]

# -ffp-contract=off: the reference's x86-64 IEEE build has no FMA contraction; neighbour
# membership (d^2 < h^2) and the order-sensitive viscous sum must round identically.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-shared", "-Wall", "-ldl"]


def library_path():
    # SPH_HIP_LIBRARY points experiments at a diagnostic build; the product uses the in-tree one
    return os.environ.get("SPH_HIP_LIBRARY") or os.path.join(HERE, "libsph_hip.so")


def _code_only(text):
    """C++ source without comments, trailing blanks and empty lines (string and character literals
    kept as they are): what source_hash() hashes, so that a reworded comment does not orphan a
    committed profile."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    lines = (line.rstrip() for line in "".join(out).split("\n"))
    return "\n".join(line for line in lines if line)


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources and the C header, comments excluded:
    what a committed counter profile (profiles/) is stamped with, so that bench.py can tell whether
    it still describes the code being run."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        path = os.path.normpath(os.path.join(CSRC, name))
        h.update(os.path.basename(path).encode())
        with open(path, "r", encoding="utf-8", errors="replace") as f:
            h.update(_code_only(f.read()).encode())
    return h.hexdigest()[:16]


def _stale(out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile the HIP library if missing or older than its sources. Returns its path."""
    out = library_path()
    if not force and not _stale(out):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libsph_hip.so")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
