#!/bin/bash
set -e
mkdir -p gpurun_out/r03_ee6
O=gpurun_out/r03_ee6
AB_ROUNDS=4 timeout -k 10 400 python3 tools/ab_variants.py "" _shnt > $O/ab_c3.log 2>&1 && cat $O/ab_c3.log
AB_ROUNDS=3 AB_CLOSEUP=1 timeout -k 10 400 python3 tools/ab_variants.py "" _shnt > $O/ab_close.log 2>&1 && cat $O/ab_close.log
AB_CONFIG=C5 AB_ROUNDS=3 timeout -k 10 400 python3 tools/ab_variants.py "" _shnt > $O/ab_c5.log 2>&1 && cat $O/ab_c5.log
