#!/bin/bash
# Everything a round's committed evidence is made from, in one gpurun call:
#   bash tools/profile_all.sh r4
# kernel stats + HBM / VALU counters + bench line (tools/profile_round.sh), the per-phase VALU counts
# on the ablated builds (tools/pmc_census.sh; build them first: see its header), the loop trip counts
# (tools/trip_counts.py, build/variants/trips.so).  Then, where hipcc is:
#   python tools/valu_census.py gpurun_out/<tag>_census gpurun_out/<tag>_trips.json --density-us .. --accel-us .. --ghz ..
set -eo pipefail
tag=${1:-r4}
bash tools/profile_round.sh $tag > gpurun_out/${tag}_profile_round.log 2>&1
bash tools/pmc_census.sh $tag > gpurun_out/${tag}_pmc_census.log 2>&1
python3 tools/trip_counts.py > gpurun_out/${tag}_trips.json 2> gpurun_out/${tag}_trips.err
tail -3 gpurun_out/${tag}_profile_round.log
