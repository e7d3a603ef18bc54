#!/bin/bash
mkdir -p gpurun_out/r03_p2
O=gpurun_out/r03_p2
AB_ROUNDS=3 timeout -k 10 400 python3 tools/ab_variants.py "" _pw5 > $O/ab_c3.log 2>&1 && cat $O/ab_c3.log
bash tools/r03_ee7.sh
