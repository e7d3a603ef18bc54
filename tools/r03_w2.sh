#!/bin/bash
set -e
mkdir -p gpurun_out/r03_w2
O=gpurun_out/r03_w2
export JADE_WIDE=1
AB_ROUNDS=3 timeout -k 10 400 python3 tools/ab_variants.py "" _wsp _wsp2 _w4w > $O/ab_c3.log 2>&1 && cat $O/ab_c3.log
AB_CONFIG=C5 AB_ROUNDS=2 timeout -k 10 400 python3 tools/ab_variants.py "" _wsp _wsp2 _w4w > $O/ab_c5.log 2>&1 && cat $O/ab_c5.log
