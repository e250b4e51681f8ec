"""Synthetic initial conditions for the SPH step.

The reference ships one scene (a rotating sphere from glibc rand(), src/sph.cpp:361-425) and
a commented-out box/dam init (src/sph.cpp:324-358).  The generators here use a counter-based
PRNG (SplitMix64 finaliser on the particle/component counter) so that any host — numpy here,
C elsewhere — produces bit-identical fp32 positions from (seed, index) without libc state.
"""
import math

import numpy as np

from .lib import default_params

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def uniform01(seed, counters):
    """24-bit uniforms in [0,1) as float32: SplitMix64 finaliser of (counter+1)*golden + seed*c."""
    with np.errstate(over="ignore"):
        z = (np.asarray(counters, np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = z + np.uint64(seed) * np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def box_fill(n, lo, hi, seed=42):
    """n points uniform in the axis-aligned box [lo, hi) (fp32), interleaved xyz."""
    idx = np.arange(3 * n, dtype=np.uint64)
    u = uniform01(seed, idx).reshape(n, 3)
    lo = np.asarray(lo, np.float32)
    ext = np.asarray(hi, np.float32) - lo
    pos = (lo + u * ext).astype(np.float32)
    return np.ascontiguousarray(pos.reshape(-1))


def dam_break_h(n, box=(1.0, 1.0, 1.0), fill=(0.1, 0.75, 1.0), neighbors=32.0):
    """Smoothing length giving ~`neighbors` particles inside radius h in the filled column."""
    vol = box[0] * fill[0] * box[1] * fill[1] * box[2] * fill[2]
    number_density = n / vol
    return (3.0 * neighbors / (4.0 * math.pi * number_density)) ** (1.0 / 3.0)


def box_fill_axis(n, lo, hi, axis, seed=42):
    """Coordinate `axis` of box_fill(n, lo, hi, seed)'s points, without the other two."""
    idx = np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(axis)
    u = uniform01(seed, idx)
    lo = np.float32(lo[axis])
    return (lo + u * (np.float32(hi[axis]) - lo)).astype(np.float32)


def box_fill_subset(ids, lo, hi, seed=42):
    """Rows `ids` of box_fill(n, lo, hi, seed) (any n > max(ids)), interleaved xyz."""
    ids = np.asarray(ids, np.uint64)
    idx = (ids[:, None] * np.uint64(3) + np.arange(3, dtype=np.uint64)[None, :]).reshape(-1)
    u = uniform01(seed, idx).reshape(-1, 3)
    lo = np.asarray(lo, np.float32)
    ext = np.asarray(hi, np.float32) - lo
    return np.ascontiguousarray((lo + u * ext).astype(np.float32).reshape(-1))


def dam_break_params(n, box=(1.0, 1.0, 1.0), fill=(0.1, 0.75, 1.0), neighbors=32.0):
    """(params, column extent) of dam_break(n, box, fill, neighbors) without any particle."""
    h = np.float32(dam_break_h(n, box, fill, neighbors))
    cells = [max(1, int(math.ceil(b / (2.0 * float(h))))) for b in box]
    p = default_params(float(h), cells)
    p.central_mass = 0.0
    return p, [box[c] * fill[c] for c in range(3)]


def dam_break(n, box=(1.0, 1.0, 1.0), fill=(0.1, 0.75, 1.0), neighbors=32.0, seed=42, speed=0.0):
    """Dam-break column modelled on the reference's commented-out init (src/sph.cpp:328-345):
    uniform random points in x<0.1*Lx, y<0.75*Ly, z<Lz, at rest, unit masses.  speed > 0: a seeded
    uniform random velocity in [-speed, speed)^3 per particle instead of rest (parity tests: with
    every particle at rest the viscous sum of src/sph.cpp:875-882 is identically zero).

    Returns (params, pos[3n], vel[3n], mass[n]).  The voxel grid (edge 2h) covers the box; the
    central point mass is switched off (it is the astrophysical part of the reference's
    default scene, not of a dam-break); everything else keeps the reference's defaults.
    """
    p, hi = dam_break_params(n, box, fill, neighbors)
    pos = box_fill(n, (0.0, 0.0, 0.0), hi, seed)
    vel = np.zeros(3 * n, np.float32)
    if speed > 0.0:
        vel = box_fill(n, (-speed,) * 3, (speed,) * 3, seed + 1)
    mass = np.ones(n, np.float32)
    return p, pos, vel, mass


def dense_block(n, lo=(1.0, 1.0, 1.0), hi=(2.2, 2.2, 2.2), seed=7, speed=0.5):
    """Reference default constants (h=0.1, 32^3 voxels) with n particles packed into a small
    block, so that the shipped sampled search actually finds neighbours (its stock sphere
    scene finds almost none).  Velocities: small uniform random."""
    p = default_params()
    pos = box_fill(n, lo, hi, seed)
    vel = (box_fill(n, (-speed,) * 3, (speed,) * 3, seed + 1)).astype(np.float32)
    mass = np.ones(n, np.float32)
    return p, pos, vel, mass


def reference_sphere(n, params=None):
    """The reference's default scene (src/sph.cpp:361-425): srand(42), rejection-sampled points
    within radius 2 of the box centre, tangential velocity 20*(dist + h/2)^-0.5 in the x-z plane,
    small random v_y.  Uses the C library's own rand()/atan2f/sinf/cosf/pow through ctypes, so on
    a glibc system the arrays are bit-identical to what `SPH::SPH()` builds with -DM=n/1024.

    Returns (params, pos[3n], vel[3n], mass[n]).  Not thread-safe (libc rand() state)."""
    import ctypes as C
    import ctypes.util
    libc = C.CDLL(ctypes.util.find_library("c") or "libc.so.6")
    libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    libc.rand.restype = C.c_int
    for name in ("atan2f",):
        getattr(libm, name).restype = C.c_float
        getattr(libm, name).argtypes = [C.c_float, C.c_float]
    for name in ("sinf", "cosf", "sqrtf"):
        getattr(libm, name).restype = C.c_float
        getattr(libm, name).argtypes = [C.c_float]
    libm.pow.restype = C.c_double
    libm.pow.argtypes = [C.c_double, C.c_double]
    p = params.copy() if params is not None else default_params()
    f32 = np.float32
    rand_max = f32(2147483647.0)           # (float)RAND_MAX
    ext = [f32(p.cells_x) * f32(p.htimes2), f32(p.cells_y) * f32(p.htimes2),
           f32(p.cells_z) * f32(p.htimes2)]
    cells = [f32(p.cells_x), f32(p.cells_y), f32(p.cells_z)]
    centre = [f32(p.max_x) * f32(0.5), f32(p.max_y) * f32(0.5), f32(p.max_z) * f32(0.5)]
    half_h = float(f32(p.hscaled)) * 0.5   # double, as in `mHScaled*0.5`
    pos = np.zeros(3 * n, f32)
    vel = np.zeros(3 * n, f32)
    libc.srand(42)
    for i in range(n):
        while True:
            c = []
            for a in range(3):
                v = f32(libc.rand()) / rand_max
                v = v * ext[a]
                if v == cells[a]:
                    v = v - f32(0.00001)
                c.append(f32(v))
            d = [f32(c[a] - centre[a]) for a in range(3)]
            dist = f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))
            dist = f32(libm.sqrtf(dist))
            if not dist > f32(2.0):
                break
        pos[3 * i:3 * i + 3] = c
        phi = f32(libm.atan2f(f32(c[2] - centre[2]), f32(c[0] - centre[0])))
        amp = 20.0 * libm.pow(float(dist) + half_h, -0.5)          # double
        vel[3 * i] = f32(amp * float(f32(-libm.sinf(phi))))
        vel[3 * i + 2] = f32(amp * float(f32(libm.cosf(phi))))
        vel[3 * i + 1] = f32(f32(f32(libc.rand()) / rand_max) * f32(0.5)) - f32(0.25)
    return p, pos, vel, np.ones(n, f32)
