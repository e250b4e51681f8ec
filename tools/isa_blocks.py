"""Per-basic-block instruction census of one kernel in a hipcc -save-temps .s file.

usage: python tools/isa_blocks.py file.s kernel_name_substring [--dump]
Prints, for every basic block of the kernel: VALU / packed / f64 / transcendental / SALU / LDS /
VMEM / branch counts, so that loop bodies can be priced without a GPU.
"""
import re
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_") and ("_f64" in op):
        return "f64"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_cbranch", "s_branch")):
        return "br"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[_A-Za-z0-9]+:", l) and name in l and not l.startswith(".L"):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    # find last s_endpgm before .Lfunc_end
    fe = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = []
    cur = ["entry", {}, []]
    for l in lines[start + 1:fe]:
        s = l.strip()
        if not s or s.startswith((";", "//")):
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            blocks.append(cur)
            cur = [m.group(1), {}, []]
            continue
        if s.startswith("."):
            continue
        op = s.split()[0]
        c = classify(op)
        cur[1][c] = cur[1].get(c, 0) + 1
        cur[2].append(s.split(";")[0].rstrip())
    blocks.append(cur)
    keys = ["valu", "pk", "f64", "trans", "salu", "lds", "vmem", "wait", "br"]
    print("%-14s" % "block" + "".join("%7s" % k for k in keys) + "  last")
    tot = {}
    for b in blocks:
        n = sum(b[1].values())
        if n == 0:
            continue
        print("%-14s" % b[0] + "".join("%7d" % b[1].get(k, 0) for k in keys) + "  " + (b[2][-1] if b[2] else ""))
        for k in keys:
            tot[k] = tot.get(k, 0) + b[1].get(k, 0)
        if dump:
            for s in b[2]:
                print("      " + s)
    print("%-14s" % "TOTAL" + "".join("%7d" % tot.get(k, 0) for k in keys))


if __name__ == "__main__":
    main()
