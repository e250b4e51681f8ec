#!/bin/bash
# Memory-path counters of the pair kernels (TA busy, L2 request latency as the L1 sees it, SQ wait / active
# cycles): one rocprofv3 --pmc pass per group, the program itself after "--".
#   bash tools/pmc_latency.sh <tag>      (on the GPU box)  ->  gpurun_out/<tag>_latency/<pass>/ + table.txt
set -eo pipefail
tag=${1:-r4}
out=gpurun_out/${tag}_latency
mkdir -p $out
export TMPDIR=/tmp SPH_PMC_MODE=fast
# (one derived TA / TCP counter per pass: more than that and rocprofv3 aborts with "exceeds the capabilities of
# the hardware" and then never exits - r4 notes 7; every pass under its own timeout, progress to the log)
i=0
for p in "TA_BUSY_avr" "TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum" \
         "TCP_TCC_WRITE_REQ_sum" "TCP_TCP_LATENCY_sum" "TCP_TOTAL_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
         "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
   i=$((i+1))
   echo "pass $i: $p" >> $out/progress.log
   timeout -k 10 150 rocprofv3 --pmc $p --output-format csv -d $out/p$i -o run -- python3 tools/pmc_one_step.py > $out/p$i.log 2>&1 \
      || echo "pass $i FAILED ($p): see $out/p$i.log" | tee -a $out/progress.log
done
python3 - $out <<'PY' | tee $out/table.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        k = row["Kernel_Name"]
        if "k_full_density_tiled" in k or "k_full_accel_lists" in k or "k_rank_gather" in k:
            a = acc[k.split("(")[0]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print("   %-40s %16.1f  (mean of %d dispatches)" % (c, s / n, n))
PY
