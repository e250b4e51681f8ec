#!/bin/bash
# Evidence run for one round, on the GPU box:  bash tools/profile_round.sh r4
# Writes under gpurun_out/<tag>_*; tools/pmc_summary.py then turns the counter CSVs into
# gpurun_out/<tag>_kernel_counters.json (stamped with the hash of the kernel sources); that file,
# the kernel-stats CSV and the microbenchmark's price table are copied to profiles/ by hand.
# rocprofv3 gets the program itself after "--" (no env/bash hop: the profiler's preloaded
# library has already initialised the GPU), and the --pmc passes are separate from the trace pass.
# (every rocprofv3 call under a timeout: a counter set the hardware cannot collect makes it abort and then
# hang - profiles/r4_notes.md 7)
# Every bench.py run steps the scene with BOTH pair arithmetics (headline + other_arithmetic), so
# one pass yields the counters of the exact and the tolerance-mode kernels (the template argument
# in the kernel name tells them apart).
set -eo pipefail
tag=${1:-r4}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
args="bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-breaking-dam"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o run -- python3 $args > $out/${tag}_stats.log 2>&1
find $out/${tag}_stats -name "*kernel_trace.csv" -delete
pargs="bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-breaking-dam --no-preheat"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o run -- python3 $pargs > $out/${tag}_pmc_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o run -- python3 $pargs > $out/${tag}_pmc_write.log 2>&1
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $out/${tag}_pmc_valu -o run -- python3 $pargs > $out/${tag}_pmc_valu.log 2>&1
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmc_valu2 -o run -- python3 $pargs > $out/${tag}_pmc_valu2.log 2>&1
# how the counters count the microbenchmark's instruction classes (calibration of the class mapping)
if [ -x build/ubench/valu3 ]; then
   ./build/ubench/valu3 $out/${tag}_valu_prices.json > $out/${tag}_valu_prices.txt
   timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $out/${tag}_ubench_pmc1 -o run -- ./build/ubench/valu3 > $out/${tag}_ubench_pmc1.log 2>&1
   timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $out/${tag}_ubench_pmc2 -o run -- ./build/ubench/valu3 > $out/${tag}_ubench_pmc2.log 2>&1
fi
# mixed opcode streams: how far the per-opcode prices are from additive (tools/ubench/valu6.hip)
if [ -x build/ubench/valu6 ]; then
   timeout -k 10 120 ./build/ubench/valu6 $out/${tag}_valu_mix.json > $out/${tag}_valu_mix.txt || true
fi
python3 tools/pmc_summary.py $tag
# the bench line last, with the counters and prices just taken (they are committed under profiles/
# with these names; bench.py checks the stamp against the kernel sources it runs)
SPH_BENCH_COUNTERS=$out/${tag}_kernel_counters.json SPH_BENCH_PRICES=$out/${tag}_valu_prices.json python3 bench.py > $out/${tag}_bench.json
cat $out/${tag}_bench.json
