"""Loop trip counts of the two pair kernels (verdict r3, item 2: "slots tested per particle and
append trips per particle"), from a diagnostic build with counters in the loops:

    tools/build_variant.sh trips WORK -DSPH_DIAGNOSTIC_BUILD -DSPH_TRIPCOUNT
    python tools/trip_counts.py [particles]            (on the GPU box)

"W" counters count once per WAVE whenever any lane takes the trip (what the SIMD issues), "L"
counters once per LANE that needs it (what the particles need); the ratio is the SIMD efficiency of
that loop.  One step of the dam-break column, tolerance-mode arithmetic (the bench's headline)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SPH_HIP_LIBRARY", os.path.join(ROOT, "build", "variants", "trips.so"))
os.environ["SPH_HIP_ALLOW_DIAGNOSTIC"] = "1"
import smoothed_particle_hydrodynamics_amd as S  # noqa: E402
from smoothed_particle_hydrodynamics_amd import scenes  # noqa: E402

NAMES = {0: "density waves", 1: "TEST chunks (32 slots) per wave", 2: "test8 steps issued (wave)",
         3: "test8 steps needed (lanes)", 4: "candidate slots in range (lanes)", 5: "append pops issued (wave)",
         6: "append pops needed (lanes) = screened candidates", 7: "append loop iterations (wave, 4 pops each)",
         8: "SUM trips (wave, 8 entries each)", 9: "SUM entry slots issued (wave)", 10: "SUM entries (lanes)",
         11: "density live lanes", 16: "acceleration waves", 17: "pressure trips (wave, 8 entries each)",
         18: "list entries (lanes)", 19: "viscous trips (wave, 4 entries each)", 20: "viscous entries (lanes)",
         21: "acceleration live lanes"}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 1024 * 1024
    # second argument: step of the BREAKING dam (gravity + walls) to take the counts at, e.g. 400 =
    # the compressed transient; default: the column at rest (the bench's headline scene)
    dense_step = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = S.load_library()
    lib.sph_hip_diag_trips.restype = C.c_int
    p, pos, vel, mass = scenes.dam_break(n)
    if dense_step:
        p.apply_gravity = 1
        p.apply_walls = 1
        p.gravity[0], p.gravity[1], p.gravity[2] = 0.0, -9.81, 0.0
    out = (C.c_ulonglong * 32)()
    with S.SPH(n, p, mode=S.MODE_FULL_FAST) as sph:
        sph.setParticles(pos, vel, mass)
        if dense_step:
            sph.run(dense_step)
        sph.step()
        sph.synchronize()
        lib.sph_hip_diag_trips(out, 32, 1)          # the upload's own sort + first step: discard
        sph.step()
        sph.synchronize()
        assert lib.sph_hip_diag_trips(out, 32, 1) == 0
        nb = float(sph.getParticles().mNeighborCount.mean())
        tiles = sph.tileStats()
    v = list(out)
    res = {NAMES[i]: v[i] for i in NAMES}
    dw, lanes = v[0], v[11]
    per = {
        "particles": n, "neighbours_mean": nb, "breaking_dam_step": dense_step, "tiles": tiles,
        "density: test8 steps issued per wave": v[2] / dw,
        "density: slots tested per particle as issued (8 x test8 per wave)": 8.0 * v[2] / dw,
        "density: slots tested per particle as needed (8 x test8 per lane)": 8.0 * v[3] / lanes,
        "density: candidate slots in range per particle": v[4] / lanes,
        "density: TEST SIMD efficiency (needed / issued)": v[3] / (64.0 * v[2]),
        "density: append pops issued per wave": v[5] / dw,
        "density: append pops needed per particle": v[6] / lanes,
        "density: append SIMD efficiency": v[6] / (64.0 * v[5]),
        "density: append loop iterations per wave": v[7] / dw,
        "density: chunks per wave": v[1] / dw,
        "density: SUM entry slots issued per wave": v[9] / dw,
        "density: SUM trips per wave": v[8] / dw,
        "density: SUM SIMD efficiency": v[10] / (64.0 * v[9]),
        "acceleration: pressure trips per wave": v[17] / v[16],
        "acceleration: pressure SIMD efficiency": v[18] / (64.0 * 8.0 * v[17]),
        "acceleration: viscous trips per wave": v[19] / v[16],
        "acceleration: viscous entries per particle": v[20] / v[21],
    }
    print(json.dumps({"raw": res, "per": per}, indent=1))


if __name__ == "__main__":
    main()
