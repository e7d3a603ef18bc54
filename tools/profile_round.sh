#!/bin/bash
# GPU box: the measurement set behind profiles/<tag>_*: the default bench run, the same command under
# rocprofv3 --kernel-trace --stats, and one rocprofv3 --pmc pass per counter group (HBM bytes, L2 hits)
# over one default step.  usage: tools/profile_round.sh <tag>     then: tools/summarize_pmc.py <tag> ...
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
[ -n "$tag" ] || { echo "usage: $0 <tag>"; exit 2; }
O="$R/gpurun_out/$tag"
rm -rf "$O"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json; echo
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py > $O/bench_profiled.json 2> $O/stats.log
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $grp | tr ' ' '_')
  timeout -k 10 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$n.log 2>&1 || echo "pmc $grp failed"
  echo "pmc $grp done"
done
