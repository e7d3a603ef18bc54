#!/bin/bash
# Development helper (GPU box): one rocprofv3 --pmc pass per counter group over a short bench run.
# usage: tools/pmc_passes.sh <tag> <spp-per-step> "<group1 counters>" "<group2 counters>" ...
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "$2" ] || { echo "usage: $0 <tag> <spp-per-step> \"<counters>\" ..."; exit 2; }
tag=$1; spp=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  out=$R/gpurun_out/pmc_${tag}_g$i
  rm -rf $out
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --steps 1 --warmup 0 --spp-per-step $spp --no-cpu-baseline > $out.log 2>&1 || { echo "group $i failed"; tail -5 $out.log; }
  echo "group $i done: $grp"
done
