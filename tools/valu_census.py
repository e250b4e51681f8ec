"""Where the VALU instructions of the two pair kernels live, and what they cost to issue: one number
per phase, one `valu.frac` <= 1 per kernel (verdict r3, item 3).

Three measured inputs, no model of the code:
  1. DYNAMIC wave-instruction counts per phase: SQ_INSTS_VALU of the shipped kernels and of builds
     with one phase cut out (tools/pmc_census.sh: abl2 = density without TEST / append / SUM, abl15 =
     without append and SUM, abl1 = without SUM, abl21 = acceleration without its pair loops);
     phase = difference of two builds.  Loop trip counts from tools/trip_counts.py split TEST from
     the per-chunk bookkeeping.
  2. The opcode MIX of each phase: the shipped ISA compiled with -gline-tables-only (same code, plus
     .loc directives), every VALU instruction attributed to the phase whose source lines it came from
     (helpers inlined from sph_device.h / pair_math.h inherit the phase of the code around them;
     code that the 4M column never runs - untiled give-up bodies, the walk of a particle without a
     list, sqrtf's slow path - is left out of the mix; TEST + append are compiled twice, for workgroups
     that stage their appends in LDS and for the one in a hundred that cannot: both copies share
     their source lines, so the mix of those two phases is the average of the two).
  3. PRICES per opcode: tools/ubench/valu3.hip + valu5.hip, cycles per wave-instruction per SIMD
     with every SIMD saturated (profiles/r3_valu_prices.json, r4_valu_prices_more.json).
cycles(phase) = dynamic count x sum over opcodes of (share of the phase's static mix x price).
valu.frac(kernel) = 64 waves per SIMD x cycles per wave / (kernel duration x shader clock).

    python tools/valu_census.py gpurun_out/r4_census profiles/r4_trip_counts.json \\
           --density-us 541 --accel-us 278 --ghz 2.2 > profiles/r4_valu_census.md
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "smoothed_particle_hydrodynamics_amd", "csrc")
DENSITY = "k_full_density_tiledILb1ELb1ELb0ELb1E"      # <unit scale, uniform mass, narrow entries, FAST>
ACCEL = "k_full_accel_listsILb1ELb1ELb0ELb1E"


def line_table_asm():
    """the shipped translation unit compiled with -gline-tables-only -save-temps -> path of the .s"""
    out = os.path.join(ROOT, "build", "isa_g")
    os.makedirs(out, exist_ok=True)
    asm = os.path.join(out, "sph_hip-hip-amdgcn-amd-amdhsa-gfx950.s")
    src = os.path.join(CSRC, "sph_hip.hip")
    newest = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC))
    if not os.path.exists(asm) or os.path.getmtime(asm) < newest:
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                        "-shared", "-Wall", "-ldl", "-gline-tables-only", "-save-temps", "-o", "/dev/null", src],
                       check=True, cwd=out, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return asm


def source_spans():
    """phase -> list of (file, first line, last line) from markers in the sources as they are now"""
    def lines(name):
        return open(os.path.join(CSRC, name)).read().split("\n")

    def find(ls, text, start=0):
        for i in range(start, len(ls)):
            if text in ls[i]:
                return i + 1                      # 1-based
        raise SystemExit("marker not found: " + text)

    ft = lines("full_tiled.h")
    spans = collections.defaultdict(list)
    # ---- density kernel
    k0 = find(ft, "k_full_density_tiled(const float4* __restrict__ posm")
    k1 = find(ft, "// ---- density pass of the workgroups whose tile fits no capacity")
    spans["d_test"].append(("full_tiled.h", find(ft, "__device__ __forceinline__ f32x2 screen_pair"),
                            find(ft, "// ---- density pass: TILE + TEST")))
    row0 = find(ft, "for (int kk = 0; kk < 9; kk++) {", k0)
    chunk0 = find(ft, "for (int t0 = (ts < te) ? (ts & ~3) : te;", k0)
    keep0 = find(ft, "// keep only slots inside [ts, te)", k0)
    app0 = find(ft, "// append the set bits, ascending", k0)
    app1 = find(ft, "if (staged) test_and_append(std::true_type());", k0)
    walk0 = find(ft, "if (__any(overflowed) && overflowed) {", k0)
    sum0 = find(ft, "// SUM: one pass over the list, in canonical order", k0)
    sum1 = find(ft, "if (!overflowed && kept != count) {", k0)
    spans["d_chunk"].append(("full_tiled.h", row0, chunk0 - 1))
    spans["d_test"].append(("full_tiled.h", chunk0, keep0 - 1))
    spans["d_chunk"].append(("full_tiled.h", keep0, app0 - 1))
    spans["d_append"].append(("full_tiled.h", app0, app1 - 1))
    spans["d_rare"].append(("full_tiled.h", walk0, sum0 - 1))
    spans["d_sum"].append(("full_tiled.h", sum0, sum1 + 3))
    spans["d_pro"].append(("full_tiled.h", k0, row0 - 1))
    spans["d_pro"].append(("full_tiled.h", app1, walk0 - 1))
    spans["d_pro"].append(("full_tiled.h", sum1 + 4, k1 - 1))
    spans["d_pro"].append(("full_tiled.h", find(ft, "tile_desc_load(const TileDesc*"),
                           find(ft, "// TEST screens, SUM confirms.")))
    # ---- acceleration kernel
    a0 = find(ft, "k_full_accel_lists(const float4* __restrict__ posm")
    g0 = find(ft, "if ((int)blockIdx.x < tile_stats[TSTAT_GIVEUP_ACCEL]) {", a0)
    g1 = find(ft, "if (gave_up == 1u) return;", a0)
    p0 = find(ft, "if constexpr (FAST) {", a0)
    p1 = find(ft, "if (gave_up == 2u && __any(no_list) && no_list) {", a0)
    e0 = find(ft, "if (FAST) accel_fast_finish(k, s);", p1)
    spans["a_pro"].append(("full_tiled.h", a0, g0 - 1))
    spans["a_rare"].append(("full_tiled.h", g0, g1 - 1))
    spans["a_pro"].append(("full_tiled.h", g1, p0 - 1))
    spans["a_pair"].append(("full_tiled.h", p0, p1 - 1))
    spans["a_rare"].append(("full_tiled.h", p1, e0 - 1))
    spans["a_pro"].append(("full_tiled.h", e0, len(ft)))
    spans["a_pro"].append(("full_tiled.h", find(ft, "accel_part_has(int part"), find(ft, "// A workgroup whose tile fitted the density pass")))
    spans["a_rare"].append(("full_tiled.h", find(ft, "// A workgroup whose tile fitted the density pass"), a0 - 5))
    for phase in ("d_rare", "a_rare"):
        spans[phase].append(("full_kernels.h", 1, 100000))
    return spans


HELPERS = ("sph_device.h", "pair_math.h", "common_kernels.h", "slab_kernels.h", "cell_build.h")


def attribute(asm, kernel, prefix, spans):
    """phase -> Counter(opcode) over the kernel's VALU instructions"""
    text = open(asm).read().split("\n")
    files = {}
    for l in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(2))
    start = next(i for i, l in enumerate(text) if re.match(r"^[_A-Za-z0-9]+:", l) and kernel in l)
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))

    def phase_of(fname, line):
        for ph, sp in spans.items():
            if not ph.startswith(prefix):
                continue
            for f, a, b in sp:
                if f == fname and a <= line <= b:
                    return ph
        return None

    hist = collections.defaultdict(collections.Counter)
    cur, pending = prefix + "pro", None
    for l in text[start:end]:
        s = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            fname, line = files.get(int(m.group(1)), "?"), int(m.group(2))
            if line == 0:
                continue
            ph = phase_of(fname, line)
            if ph is not None:
                cur = ph
            elif fname not in HELPERS and not fname.startswith(("amd_", "__clang")) and fname != "sph_hip.hip":
                pass
            # helpers (sph_device.h, pair_math.h, the HIP headers) inherit the phase around them; a
            # slow path inside a helper is recognised by its opcodes below
            continue
        if s.startswith("v_"):
            op = s.split()[0]
            for suffix in ("_e32", "_e64", "_sdwa", "_dpp"):
                if op.endswith(suffix):
                    op = op[:-len(suffix)]
            hist[cur][op] += 1
    return hist


def load_prices():
    table = {}
    a = json.load(open(os.path.join(ROOT, "profiles", "r3_valu_prices.json")))["instructions"]
    for name, v in a.items():
        table[name] = v["cycles"][3]
    b = json.load(open(os.path.join(ROOT, "profiles", "r4_valu_prices_more.json")))["instructions"]
    for name, v in b.items():
        table[name] = v["cycles"]
    both = table.pop("v_cmp + v_cndmask")
    table["v_cndmask_b32"] = both           # (alone, with a VCC nobody writes, it measures a hazard, not the issue)
    table["v_fma_f32 + v_alignbit mix"] = table.get("v_fma_f32 + v_alignbit mix", 0.0)
    return table, both


def price(op, table, cmp_price):
    if op in table:
        return table[op]
    if op.startswith("v_cmp"):
        return cmp_price
    if op.startswith("v_pk_"):
        return table["v_pk_fma_f32"]
    if op.endswith("_f64") or "_u64" in op or "_i64" in op or op.endswith("_b64"):
        return table["v_fma_f64"]
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return table["v_rcp_f32"]
    if op in ("v_subrev_u32", "v_subrev_f32", "v_add_u32", "v_add_f32", "v_mul_f32", "v_fma_f32"):
        return table["v_add_u32"]
    if op.startswith(("v_readlane", "v_writelane", "v_mbcnt", "v_add_co", "v_sub_co", "v_subb_co", "v_addc_co", "v_subrev_co",
                      "v_bfi", "v_bfrev", "v_ffbh", "v_min3", "v_bitop3_b16", "v_lshlrev_b16", "v_ceil", "v_min_u32")):
        return table["v_lshl_or_b32"]
    return table["v_lshl_or_b32"]           # unknown: the wide price (an upper bound)


ARITH = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_pk_fma_f32", "v_pk_add_f32",
         "v_pk_mul_f32", "v_rsq_f32", "v_rcp_f32", "v_sqrt_f32", "v_max_f32", "v_div_scale_f32", "v_div_fmas_f32",
         "v_div_fixup_f32", "v_floor_f32", "v_log_f32", "v_ceil_f32")


def pmc(census_dir, variant, kernel_short):
    """counter -> mean per wave for one kernel of one variant"""
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(census_dir, variant, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if kernel_short in row["Kernel_Name"]:
                    a = acc[row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    c = {k: v[0] / v[1] for k, v in acc.items()}
    waves = c.get("SQ_WAVES", 1.0)
    return {k: v / waves for k, v in c.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("census")
    ap.add_argument("trips")
    ap.add_argument("--density-us", type=float, required=True)
    ap.add_argument("--accel-us", type=float, required=True)
    ap.add_argument("--ghz", type=float, required=True)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    table, cmp_price = load_prices()
    spans = source_spans()
    asm = line_table_asm()
    hd = attribute(asm, DENSITY, "d_", spans)
    ha = attribute(asm, ACCEL, "a_", spans)
    trips = json.load(open(a.trips))["per"]
    D = {v: pmc(a.census, v, "k_full_density_tiled<true, true, false, true>") for v in ("base", "abl1", "abl2", "abl15")}
    A = {v: pmc(a.census, v, "k_full_accel_lists<true, true, false, true>") for v in ("base", "abl21")}
    valu = lambda t, v: t[v]["SQ_INSTS_VALU"]
    # dynamic wave-instructions per wave, by phase (differences of builds)
    test8 = trips["density: test8 steps issued per wave"]
    # 9 rows x 4 steps, all alike - in each of the two copies of TEST + append (staged / direct appends)
    copies = 2 if "test_and_append(std::false_type())" in open(os.path.join(CSRC, "full_tiled.h")).read() else 1
    # (the screen_pair / test8 helpers above the kernel are inlined into both copies and attributed once per copy)
    static_test8 = sum(hd["d_test"].values()) / (36.0 * copies)
    test_chunk = valu(D, "abl15") - valu(D, "abl2")
    d_test = min(test8 * static_test8, test_chunk)
    dyn = collections.OrderedDict([
        ("density: prologue + epilogue (tile fill, ranges, results)", ("d_pro", valu(D, "abl2"))),
        ("density: TEST (%.1f 8-slot steps per wave)" % test8, ("d_test", d_test)),
        ("density: per-row / per-chunk bookkeeping (masks, counts)", ("d_chunk", test_chunk - d_test)),
        ("density: append (%.1f pops issued per wave for %.1f accepted per particle)" % (
            trips["density: append pops issued per wave"], trips["density: append pops needed per particle"]),
         ("d_append", valu(D, "abl1") - valu(D, "abl15"))),
        ("density: SUM (%.1f entry slots per wave)" % trips["density: SUM entry slots issued per wave"],
         ("d_sum", valu(D, "base") - valu(D, "abl1"))),
        ("acceleration: prologue + epilogue (tile fill, fused integrate + hash)", ("a_pro", valu(A, "abl21"))),
        ("acceleration: pair loops (%.1f pressure + %.1f viscous trips per wave)" % (
            trips["acceleration: pressure trips per wave"], trips["acceleration: viscous trips per wave"]),
         ("a_pair", valu(A, "base") - valu(A, "abl21"))),
    ])
    rows, tot = [], {"d": [0.0, 0.0, 0.0], "a": [0.0, 0.0, 0.0]}
    for label, (ph, n) in dyn.items():
        h = hd[ph] if ph.startswith("d_") else ha[ph]
        total = float(sum(h.values()))
        cyc_per = sum(c / total * price(op, table, cmp_price) for op, c in h.items())
        arith = sum(c for op, c in h.items() if op in ARITH) / total
        wide = sum(c for op, c in h.items() if price(op, table, cmp_price) > 3.0) / total
        top = ", ".join("%s %.0f%%" % (op, 100.0 * c / total) for op, c in h.most_common(5))
        rows.append((label, n, cyc_per, n * cyc_per, arith, wide, top, int(total)))
        t = tot[ph[0]]
        t[0] += n
        t[1] += n * cyc_per
        t[2] += n * (1.0 - arith)
    waves_per_simd = 65536.0 / 1024.0
    sys.path.insert(0, ROOT)
    from smoothed_particle_hydrodynamics_amd.build import source_hash
    out = {"source": "tools/valu_census.py: SQ_INSTS_VALU per phase (ablated builds) x opcode mix (line tables of "
                     "the shipped ISA) x price per opcode (tools/ubench/valu3.hip, valu5.hip)",
           "csrc_sha16": source_hash(), "particles": 4 * 1024 * 1024, "arithmetic": "fast", "ghz": a.ghz,
           "phases": [], "kernels": {}}
    print("# VALU census of the density + acceleration pair, 4M-particle column, tolerance-mode arithmetic\n")
    print(__doc__.split("\n\n")[1].replace("\n", " ") + "\n")
    print("| phase | wave-instructions per wave | cycles per instruction (priced mix) | issue cycles per wave | fp32 arithmetic | wide-class opcodes | static instructions attributed | largest opcodes |")
    print("|---|---|---|---|---|---|---|---|")
    for label, n, cp, cyc, arith, wide, top, st in rows:
        print("| %s | %.0f | %.2f | %.0f | %.0f %% | %.0f %% | %d | %s |" % (label, n, cp, cyc, 100 * arith, 100 * wide, st, top))
        out["phases"].append({"phase": label, "wave_instructions_per_wave": n, "cycles_per_instruction": cp,
                              "issue_cycles_per_wave": cyc, "arithmetic_share": arith, "wide_share": wide})
    print()
    pair_cycles = 0.0
    for key, name, us in (("d", "k_full_density_tiled", a.density_us), ("a", "k_full_accel_lists", a.accel_us)):
        n, cyc, nonarith = tot[key]
        have = us * 1e-6 * a.ghz * 1e9
        raw = waves_per_simd * cyc / have
        frac = min(1.0, raw)
        pair_cycles += waves_per_simd * cyc
        measured = valu(D if key == "d" else A, "base")
        print("**%s**: %.0f VALU wave-instructions per wave (SQ_INSTS_VALU: %.0f; attributed by phase: %.0f, residual "
              "%.1f %%), of them %.0f (%.0f %%) not fp32 arithmetic; %.0f issue cycles per wave x 64 waves per SIMD = "
              "%.2f M cycles of the %.2f M the SIMD has in %.0f us at %.2f GHz: **valu.frac = %.2f**%s\n" % (
                  name, measured, measured, n, 100.0 * (measured - n) / measured, nonarith, 100.0 * nonarith / n, cyc,
                  waves_per_simd * cyc * 1e-6, have * 1e-6, us, a.ghz, frac,
                  "" if raw <= 1.0 else " (the additive prices give %.2f: they are measured per opcode in isolation, and "
                  "mixed streams - tools/ubench/valu6.hip, profiles/r4_valu_mix.txt - deviate from their sum by several "
                  "per cent both ways; a kernel cannot use more than all of its issue cycles: the pass is VALU-bound)" % raw))
        out["kernels"][name] = {"wave_instructions_per_wave": measured, "issue_cycles_per_wave": cyc,
                                "non_arithmetic_per_wave": nonarith, "duration_us": us, "valu_frac": frac,
                                "valu_frac_additive_prices": raw}
    have = (a.density_us + a.accel_us) * 1e-6 * a.ghz * 1e9
    out["pair"] = {"issue_cycles_per_simd": pair_cycles, "valu_frac": min(1.0, pair_cycles / have),
                   "valu_frac_additive_prices": pair_cycles / have,
                   "wave_instructions_per_launch_pair": (tot["d"][0] + tot["a"][0]) * 65536.0,
                   "non_arithmetic_per_launch_pair": (tot["d"][2] + tot["a"][2]) * 65536.0}
    print("**pair**: valu.frac = %.2f; %.0f M wave-instructions per launch pair, of them %.0f M not fp32 arithmetic\n" % (
        out["pair"]["valu_frac"], out["pair"]["wave_instructions_per_launch_pair"] * 1e-6,
        out["pair"]["non_arithmetic_per_launch_pair"] * 1e-6))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
