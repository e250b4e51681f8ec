"""Summarise the FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes of tools/profile_round.sh.

    python3 tools/pmc_summary.py r2   ->  gpurun_out/r2_kernel_counters.json

The file is stamped with the hash of the kernel sources it was measured on
(smoothed_particle_hydrodynamics_amd.build.source_hash); bench.py reports its figures only while
that hash is the one of the code it runs.  Copy it to profiles/ to commit it.

Counter values are KiB per dispatch (rocprofv3 derives them from the TCC request counters);
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, FETCH doubled as MI355X_MICROARCH.md prescribes
for 16-byte-per-lane streaming reads on gfx950.  Means are over all dispatches of a kernel.
"""
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def means(pattern, counter):
    acc = {}
    for path in glob.glob(pattern, recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                s = acc.setdefault(short(row["Kernel_Name"]), [0.0, 0])
                s[0] += float(row["Counter_Value"])
                s[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
    out = "gpurun_out"
    fetch = means(os.path.join(out, tag + "_pmc_fetch", "**", "*counter_collection.csv"), "FETCH_SIZE")
    write = means(os.path.join(out, tag + "_pmc_write", "**", "*counter_collection.csv"), "WRITE_SIZE")
    valu = means(os.path.join(out, tag + "_pmc_valu", "**", "*counter_collection.csv"), "SQ_INSTS_VALU")
    waves = means(os.path.join(out, tag + "_pmc_valu", "**", "*counter_collection.csv"), "SQ_WAVES")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from smoothed_particle_hydrodynamics_amd.build import source_hash
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, (0.0, 0))
        w = write.get(k, (0.0, 0))
        kernels[k] = {"FETCH_SIZE_KiB": f[0], "WRITE_SIZE_KiB": w[0], "dispatches": max(f[1], w[1]),
                      "hbm_bytes": (2.0 * f[0] + w[0]) * 1024.0}
        if k in valu:
            kernels[k]["SQ_INSTS_VALU"] = valu[k][0]
            kernels[k]["SQ_WAVES"] = waves.get(k, (0.0, 0))[0]
    doc = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU SQ_WAVES "
                  "(separate passes), python3 bench.py --steps 4 --warmup 1 --cpu-sample 0 "
                  "--no-breaking-dam, 4194304-particle dam-break, MI355X (tools/profile_round.sh)",
        "csrc_sha16": source_hash(),
        "units": "counter values are KiB per dispatch (mean over dispatches); hbm_bytes = "
                 "(2*FETCH_SIZE + WRITE_SIZE)*1024 - FETCH_SIZE doubled as MI355X_MICROARCH.md "
                 "prescribes for 16-B/lane streaming reads on gfx950",
        "particles": int(os.environ.get("SPH_PROFILE_PARTICLES", 4 * 1024 * 1024)),
        "density_plus_acceleration_hbm_bytes": sum(
            v["hbm_bytes"] for k, v in kernels.items()
            if k.startswith("k_full_density") or k.startswith("k_full_accel")),
        "valu_wave_instructions_per_launch_pair": sum(
            v.get("SQ_INSTS_VALU", 0.0) for k, v in kernels.items()
            if k.startswith("k_full_density") or k.startswith("k_full_accel")),
        "kernels": kernels,
    }
    path = os.path.join(out, tag + "_kernel_counters.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote", path)
    for k, v in kernels.items():
        print("%-40s %12.1f MB" % (k[:40], v["hbm_bytes"] / 1e6))


if __name__ == "__main__":
    main()
