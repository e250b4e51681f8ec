"""Where a workgroup of the two tiled pair kernels spends its life (verdict r3 item 2 asked for
instructions; this asks for TIME): a -DSPH_DIAGNOSTIC_BUILD -DSPH_PHASECLOCK build has thread 0 of every
workgroup read the constant-rate clock (100 MHz) at marks between the phases and accumulate the
differences (csrc/full_tiled.h: PHASE_MARK).  Residency of a workgroup = launch duration x workgroups
resident per CU / workgroups a CU processes; what the marks do not cover is the drain of the
workgroup's last stores and the dispatch of its successor.

    tools/build_variant.sh phase WORK -DSPH_DIAGNOSTIC_BUILD -DSPH_PHASECLOCK
    SPH_HIP_ALLOW_DIAGNOSTIC=1 SPH_HIP_LIBRARY=build/variants/phase.so python3 tools/phase_clock.py [n] [dam step]
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smoothed_particle_hydrodynamics_amd as S
from smoothed_particle_hydrodynamics_amd import scenes

DENSITY = ["descriptor, own position, row ranges", "tile", "TEST + append", "pad / particles without a list",
           "SUM", "results issued"]
ACCEL = ["ranges, flags, descriptor", "tile, own loads, first list block", "pressure loop (exact: -)",
         "viscous loop (exact: the pair loop)", "end of the sum, acceleration issued",
         "integrate, hash, count, energy sums"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 1024 * 1024
    dense_step = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = S.load_library()
    lib.sph_hip_diag_phases.restype = C.c_int
    p, pos, vel, mass = scenes.dam_break(n)
    if dense_step:
        p.apply_gravity = 1
        p.apply_walls = 1
        p.gravity[0], p.gravity[1], p.gravity[2] = 0.0, -9.81, 0.0
    out = (C.c_ulonglong * 32)()
    steps = 10
    with S.SPH(n, p, mode=S.MODE_FULL_FAST) as sph:
        sph.setParticles(pos, vel, mass)
        sph.run(max(dense_step, 20))
        sph.synchronize()
        lib.sph_hip_diag_phases(out, 32, 1)
        sph.setTiming(S.TIMING_PHASES)
        for _ in range(steps):
            sph.step()
        sph.synchronize()
        assert lib.sph_hip_diag_phases(out, 32, 1) == 0     # the LAST step's workgroups (each step overwrites)
        t, k = sph.phaseTotals()
        tiles = sph.tileStats()
    v = list(out)
    res = {"particles": n, "breaking_dam_step": dense_step, "steps": steps, "tiles": tiles,
           "clock": "wall_clock64, 100 MHz; thread 0 of every workgroup",
           "launch_ms": {"density": t[2] / k, "acceleration": t[4] / k}}
    for name, base, labels, count_at, ms in (("density", 0, DENSITY, 7, t[2] / k), ("acceleration", 16, ACCEL, 23, t[4] / k)):
        wgs = v[count_at]
        phases = {labels[i]: v[base + i] / wgs * 1e-2 for i in range(len(labels))}   # ticks of 10 ns -> us
        res[name] = {"workgroups_marked": wgs, "us_per_workgroup": phases,
                     "us_marked_total": sum(phases.values())}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
