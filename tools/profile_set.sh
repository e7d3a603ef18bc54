#!/bin/bash
# GPU box: the round-3 measurement set behind profiles/<tag>_*.
#   1. the default bench run (with cpu_baseline + parity_check)                       -> bench.json
#   2. the same command under rocprofv3 --kernel-trace --stats
#   3. one rocprofv3 --pmc pass per counter group over the SAME default command (without the CPU legs and the close-up extra):
#      SQ issue counters, FETCH_SIZE, WRITE_SIZE, TCC hits / misses, SQ wait / LDS / VMEM groups
#   4. (with C5 as 2nd argument) kernel stats + the same groups for --config C5
# usage: tools/profile_r03.sh <tag> [C5]      then here: tools/summarize_prof.py <tag>
[ -n "$1" ] || { echo "usage: $0 <tag> [C5]"; exit 2; }
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
O="$R/gpurun_out/$tag"
rm -rf "$O"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
SQ="SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"
SQ2="SQ_WAIT_ANY SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM"
timeout -k 10 600 python3 "$R/bench.py" > "$O/bench.json" 2> "$O/bench.err"
tail -c 300 "$O/bench.json"; echo
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --no-cpu-baseline --no-extras > "$O/bench_profiled.json" 2> "$O/stats.log"
echo "stats done"
i=0
for grp in "$SQ" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "$SQ2"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$O/pmc_g$i" -- python3 "$R/bench.py" --no-cpu-baseline --no-extras > "$O/pmc_g$i.log" 2>&1 || echo "pmc group $i failed"
  echo "pmc group $i done"
done
if [ "$2" = "C5" ]; then
  C5ARGS="--config C5 --steps 2 --warmup 1 --spp-per-step 64 --no-cpu-baseline --no-extras"
  timeout -k 10 600 python3 "$R/bench.py" --config C5 --steps 2 --warmup 1 --spp-per-step 64 --no-extras > "$O/c5_bench.json" 2> "$O/c5_bench.err"
  tail -c 300 "$O/c5_bench.json"; echo
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/c5_stats" -- python3 "$R/bench.py" $C5ARGS > "$O/c5_bench_profiled.json" 2> "$O/c5_stats.log"
  i=0
  for grp in "$SQ" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$O/c5_pmc_g$i" -- python3 "$R/bench.py" $C5ARGS > "$O/c5_pmc_g$i.log" 2>&1 || echo "c5 pmc group $i failed"
    echo "c5 pmc group $i done"
  done
fi
# keep the merge-back small: the per-dispatch CSVs are reduced on the box
python3 "$R/tools/summarize_prof.py" "$tag" --reduce-only || echo "reduce failed"
