#!/bin/bash
# GPU box, development: where does k_trace's time go?  (1) occupancy sweep, (2) SQ / TA / TCP counter groups on a short run.
# usage: tools/diag_r02.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "$1" ] || { echo "usage: $0 <tag> [bench args...]"; exit 2; }
tag=$1; shift
O="$R/gpurun_out/$tag"
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --spp-per-step 256 --no-cpu-baseline --no-extras $@"
for k in 2 3 4 5 6; do
  JADE_TRACE_BLOCKS_PER_CU=$k timeout -k 10 300 python3 $R/bench.py $ARGS > $O/occ_$k.json 2> $O/occ_$k.err
  python3 -c "
import json,sys
d=json.loads([l for l in open('$O/occ_$k.json') if l.startswith('{')][-1])
print('blocks/CU $k: %.0f Mray/s, k_trace %.2f ms/launch x %d, trace share %.3f' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], d['roofline']['trace_share_of_step_time']))"
done
i=0
for grp in "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_g$i -- python3 $R/bench.py $ARGS > $O/pmc_g$i.log 2>&1 || echo "pmc group $i failed: $grp"
  python3 - <<PY
import csv,glob,collections
fs=glob.glob("$O/pmc_g$i/*/*counter_collection.csv")
if fs:
    agg=collections.defaultdict(float); ns=collections.defaultdict(int); seen=set()
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_"): continue
        agg[(k,r["Counter_Name"])]+=float(r["Counter_Value"])
        if (k,r["Dispatch_Id"]) not in seen:
            seen.add((k,r["Dispatch_Id"])); ns[k]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    for (k,c),v in sorted(agg.items()):
        if k in ("k_trace","k_shade","k_shade_lean"): print("%-13s %-38s %.5g   (kernel ns %.4g)"%(k,c,v,ns[k]))
PY
  rm -rf $O/pmc_g$i
done
