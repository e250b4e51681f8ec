"""Per-kernel means of every counter found under a directory of rocprofv3 --pmc passes.

    python3 tools/pmc_table.py gpurun_out/pmc_r2 [kernel-substring ...]

Every *counter_collection.csv below the directory is read (one sub-directory per pass);
prints, per kernel, counter -> mean per dispatch and, where SQ_WAVES is known, per wave."""
import csv
import glob
import os
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    root = sys.argv[1]
    want = sys.argv[2:]
    acc = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if want and not any(w in k for w in want):
                    continue
                s = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
                s[0] += float(row["Counter_Value"])
                s[1] += 1
    for k in sorted(acc):
        c = {n: v[0] / v[1] for n, v in acc[k].items()}
        waves = c.get("SQ_WAVES", 0.0)
        print("== %s  (%d dispatches)" % (k, max(v[1] for v in acc[k].values())))
        for n in sorted(c):
            per = ("   per wave %12.1f" % (c[n] / waves)) if waves else ""
            print("   %-34s %16.1f%s" % (n, c[n], per))


if __name__ == "__main__":
    main()
