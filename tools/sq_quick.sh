#!/bin/bash
# GPU box, development: the SQ issue counters of one short bench run per library variant.  usage: tools/sq_quick.sh <tag> [variant suffixes...]
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "$1" ] || { echo "usage: $0 <tag> [variant suffixes...]"; exit 2; }
tag=$1; shift
O="$R/gpurun_out/$tag"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
SQ="SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU"
for v in "$@"; do
  [ "$v" = "base" ] && lib="" || lib=$v
  JADE_HIP_LIB=$R/jaderaytracerendering_amd/lib/libjade_hip$lib.so timeout -k 10 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc$lib -- python3 $R/bench.py --steps 1 --warmup 1 --spp-per-step 256 --no-cpu-baseline --no-extras > $O/pmc$lib.log 2>&1
  python3 - <<PY
import csv,glob,collections
fs=glob.glob("$O/pmc$lib/*/*counter_collection.csv")
agg=collections.defaultdict(float); ns=collections.defaultdict(int); seen=set()
for r in csv.DictReader(open(fs[0])):
    k=r["Kernel_Name"].split("(")[0]
    if k not in ("k_trace","k_shade","k_shade_binned","k_shade_lean"): continue
    agg[(k,r["Counter_Name"])]+=float(r["Counter_Value"])
    if (k,r["Dispatch_Id"]) not in seen:
        seen.add((k,r["Dispatch_Id"])); ns[k]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k in ("k_trace","k_shade","k_shade_binned","k_shade_lean"):
    if not ns[k]: continue
    g=lambda c: agg[(k,c)]
    print("$v %-13s ms %.1f VALU %.4g SALU %.4g lanes/inst %.1f valu_busy(4cyc) %.3f wait_any %.3f wait_inst %.3f" % (k, ns[k]*1e-6, g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_THREAD_CYCLES_VALU")/g("SQ_INSTS_VALU"), 4*g("SQ_ACTIVE_INST_VALU")/1024/(g("SQ_BUSY_CU_CYCLES")/256), g("SQ_WAIT_ANY")/g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY")/g("SQ_WAVE_CYCLES")))
PY
  rm -rf $O/pmc$lib
done
