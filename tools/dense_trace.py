"""Per-kernel picture of the breaking dam at one point of its run.

   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dense -o run -- python3 tools/dense_trace.py run 400 20
   python3 tools/dense_trace.py report gpurun_out/dense 20

`run STEP K` steps the 4M-particle dam (gravity + walls, FAST unless SPH_DENSE_EXACT=1) to STEP and
then K more; `report DIR K` reads the kernel trace and prints, for the last K steps, every kernel's
mean duration, its share of the span, and how much of it ran beside another kernel (the chunked
density kernel runs on a stream of its own)."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(step, k):
    import smoothed_particle_hydrodynamics_amd as S
    from smoothed_particle_hydrodynamics_amd import scenes
    n = int(os.environ.get("SPH_DENSE_N", "4194304"))
    p, pos, vel, mass = scenes.dam_break(n)
    p.apply_gravity = 1
    p.apply_walls = 1
    p.gravity[0], p.gravity[1], p.gravity[2] = 0.0, -9.81, 0.0
    mode = S.MODE_FULL if os.environ.get("SPH_DENSE_EXACT") else S.MODE_FULL_FAST
    with S.SPH(n, p, mode=mode) as sph:
        sph.setParticles(pos, vel, mass)
        sph.run(step)
        sph.synchronize()
        import time
        t0 = time.perf_counter()
        sph.run(k)
        sph.synchronize()
        print("steps %d..%d: %.3f ms/step" % (step, step + k, (time.perf_counter() - t0) / k * 1e3), flush=True)
        print(sph.tileStats(), flush=True)


def report(d, k):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
    rows.sort()
    # a step ends with its acceleration launch(es); take the launches after the (k+1)-th last k_scan_reduce
    scans = [i for i, r in enumerate(rows) if r[2].startswith("k_scan_reduce")]
    first = scans[-k] if len(scans) >= k else 0
    sel = rows[first:]
    span = sel[-1][1] - sel[0][0]
    print("last %d steps: %.3f ms per step (span of the trace)" % (k, span / k * 1e-6))
    by = {}
    for s, e, name in sel:
        b = by.setdefault(name, [0, 0, 0])
        b[0] += 1
        b[1] += e - s
        # time this launch shared with any other launch
        ov = 0
        for s2, e2, n2 in sel:
            if (s2, e2, n2) != (s, e, name) and s2 < e and e2 > s:
                ov += min(e, e2) - max(s, s2)
        b[2] += ov
    for name, (c, t, ov) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print("%-60s %5d launches  %9.1f us each  %6.1f %% of span  overlapped %5.1f %%" % (
            name[:60], c, t / c * 1e-3, 100.0 * t / span, 100.0 * ov / max(t, 1)))
    # idle gaps
    busy_end = sel[0][0]
    idle = 0
    for s, e, _ in sel:
        if s > busy_end:
            idle += s - busy_end
        busy_end = max(busy_end, e)
    print("device idle between launches: %.1f %% of span" % (100.0 * idle / span))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]))
    else:
        report(sys.argv[2], int(sys.argv[3]))
