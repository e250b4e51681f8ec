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


def clocks(pattern):
    """shader clock a kernel ran at, GHz: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the
    dispatch's duration (MI355X_MICROARCH.md, DVFS give-back); reads high on dispatches shorter
    than ~0.3 ms"""
    acc = {}
    for path in glob.glob(pattern, recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != "GRBM_GUI_ACTIVE":
                    continue
                s = acc.setdefault(short(row["Kernel_Name"]), [0.0, 0.0])
                s[0] += float(row["Counter_Value"]) / 8.0
                s[1] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return {k: v[0] / v[1] for k, v in acc.items() if v[1] > 0}


VALU_CLASSES = ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_ADD_F64",
                "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
                "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]


def arithmetic_of(kernel):
    """the tiled pair kernels carry the FAST flag as their last template argument"""
    if "<" not in kernel:
        return None
    args = [a.strip() for a in kernel[kernel.index("<") + 1:kernel.rindex(">")].split(",")]
    return "fast" if args[-1] == "true" else "exact"


def ubench_calibration(out, tag):
    """per microbenchmark kernel (k<MODE>): class counters per SQ_INSTS_VALU - how the SQ counts
    each instruction form (a packed fp32 op as one FMA_F32? cmp/cndmask under INT32?)"""
    names = ["fma_f32", "mul_f32", "add_f32", "pk_fma_f32", "pk_add_f32", "pk_mul_f32", "add_u32", "and_b32",
             "lshl_or_b32", "bfe_u32", "alignbit_b32", "ffbl_b32", "cmp+cndmask", "rcp_f32", "rsq_f32", "sqrt_f32",
             "fma_f64", "mul_f64", "add_f64", "rcp_f64", "mov_b32", "fma_f32+alignbit"]
    table = {}
    for sub in ("_ubench_pmc1", "_ubench_pmc2"):
        pattern = os.path.join(out, tag + sub, "**", "*counter_collection.csv")
        total = means(pattern, "SQ_INSTS_VALU")
        for counter in VALU_CLASSES:
            vals = means(pattern, counter)
            for k, (v, _) in vals.items():
                if not k.startswith("k<") or k not in total or total[k][0] == 0:
                    continue
                mode = int(k[2:k.index(">")])
                table.setdefault(names[mode] if mode < len(names) else k, {})[counter] = round(v / total[k][0], 4)
    return table


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
    out = "gpurun_out"
    fetch = means(os.path.join(out, tag + "_pmc_fetch", "**", "*counter_collection.csv"), "FETCH_SIZE")
    write = means(os.path.join(out, tag + "_pmc_write", "**", "*counter_collection.csv"), "WRITE_SIZE")
    valu = means(os.path.join(out, tag + "_pmc_valu", "**", "*counter_collection.csv"), "SQ_INSTS_VALU")
    waves = means(os.path.join(out, tag + "_pmc_valu", "**", "*counter_collection.csv"), "SQ_WAVES")
    classes = {}
    for sub in ("_pmc_valu", "_pmc_valu2"):
        for counter in VALU_CLASSES:
            for k, (v, _) in means(os.path.join(out, tag + sub, "**", "*counter_collection.csv"), counter).items():
                classes.setdefault(k, {})[counter] = v
    ghz = clocks(os.path.join(out, tag + "_pmc_valu2", "**", "*counter_collection.csv"))
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
        kernels[k].update(classes.get(k, {}))
        if k in ghz:
            kernels[k]["shader_clock_ghz_while_profiled"] = ghz[k]
    doc = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU + class counters "
                  "(separate passes), python3 bench.py --steps 4 --warmup 1 --cpu-sample 0 "
                  "--no-breaking-dam --no-preheat (both pair arithmetics per run), 4194304-particle "
                  "dam-break, MI355X (tools/profile_round.sh)",
        "csrc_sha16": source_hash(),
        "units": "counter values are KiB per dispatch (mean over dispatches); hbm_bytes = "
                 "(2*FETCH_SIZE + WRITE_SIZE)*1024 - FETCH_SIZE doubled as MI355X_MICROARCH.md "
                 "prescribes for 16-B/lane streaming reads on gfx950",
        "particles": int(os.environ.get("SPH_PROFILE_PARTICLES", 4 * 1024 * 1024)),
        "arithmetic": {},
        "ubench_counter_calibration": ubench_calibration(out, tag),
        "kernels": kernels,
    }
    for arith in ("fast", "exact"):
        pair = {k: v for k, v in kernels.items()
                if (k.startswith("k_full_density") or k.startswith("k_full_accel")) and arithmetic_of(k) == arith}
        if not pair:
            continue
        entry = {"kernels": sorted(pair),
                 "density_plus_acceleration_hbm_bytes": sum(v["hbm_bytes"] for v in pair.values()),
                 "valu_wave_instructions_per_launch_pair": sum(v.get("SQ_INSTS_VALU", 0.0) for v in pair.values())}
        for counter in VALU_CLASSES:
            entry[counter] = sum(v.get(counter, 0.0) for v in pair.values())
        cl = [v["shader_clock_ghz_while_profiled"] for v in pair.values() if "shader_clock_ghz_while_profiled" in v]
        if cl:
            entry["shader_clock_ghz_while_profiled"] = sum(cl) / len(cl)
        doc["arithmetic"][arith] = entry
    path = os.path.join(out, tag + "_kernel_counters.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote", path)
    for k, v in kernels.items():
        print("%-40s %12.1f MB" % (k[:40], v["hbm_bytes"] / 1e6))


if __name__ == "__main__":
    main()
