#!/bin/bash
set -e
mkdir -p gpurun_out/r03_w3
O=gpurun_out/r03_w3
AB_CONFIG=C5 AB_ROUNDS=3 timeout -k 10 400 python3 tools/ab_variants.py "" _ww4 > $O/ab_c5_a.log 2>&1 && cat $O/ab_c5_a.log
AB_CONFIG=C5 AB_ROUNDS=3 timeout -k 10 400 python3 tools/ab_variants.py _wsp _ww4sp > $O/ab_c5_b.log 2>&1 && cat $O/ab_c5_b.log
