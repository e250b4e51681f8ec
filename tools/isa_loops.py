"""Loops of one kernel in the device ISA (-save-temps .s): first/last line, VALU / LDS / global /
scalar instruction counts per loop body - to see what a source change did to the hot loops.
    python tools/isa_loops.py <file.s> <mangled kernel name prefix>"""
import re
import sys

t = open(sys.argv[1]).read().split("\n")
k = sys.argv[2]
start = next(i for i, l in enumerate(t) if l.startswith(k) and ":" in l.split()[0])
end = next(i for i in range(start, len(t)) if t[i].startswith(".Lfunc_end"))
body = t[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        seg = [x.strip() for x in body[a:i]]
        cnt = lambda p: sum(1 for x in seg if x.startswith(p))
        print("lines %6d-%6d  valu %4d  ds %3d  global %3d  salu %3d  waitcnt %3d  trans %2d" % (
            a, i, cnt("v_"), cnt("ds_"), cnt("global_") + cnt("buffer_"), cnt("s_") - cnt("s_waitcnt") - cnt("s_nop"),
            cnt("s_waitcnt"), sum(1 for x in seg if re.match(r"v_(rsq|rcp|sqrt|log|exp)", x))))
print("kernel: %d lines, %d valu" % (len(body), sum(1 for x in body if x.strip().startswith("v_"))))
