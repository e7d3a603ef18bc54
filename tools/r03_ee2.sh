#!/bin/bash
set -e
mkdir -p gpurun_out/r03_ee2
O=gpurun_out/r03_ee2
timeout -k 10 400 python3 tools/ab_variants.py "" _hq32 _hq16 _cn150 _cn220 _sp2 > $O/ab_c3.log 2>&1 && cat $O/ab_c3.log
AB_CLOSEUP=1 timeout -k 10 400 python3 tools/ab_variants.py "" _hq32 _hq16 _cn150 _cn220 _sp2 > $O/ab_close.log 2>&1 && cat $O/ab_close.log
