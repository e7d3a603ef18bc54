#!/bin/bash
# early-exit first look: parity smoke + A/B walk 0 vs 1
set -e
mkdir -p gpurun_out/r03_ee1
O=gpurun_out/r03_ee1
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 300 python3 tools/ab_variants.py "@0" "@1" > $O/ab_c3.log 2>&1 && cat $O/ab_c3.log
AB_CLOSEUP=1 timeout -k 10 300 python3 tools/ab_variants.py "@0" "@1" > $O/ab_close.log 2>&1 && cat $O/ab_close.log
AB_CONFIG=C5 AB_ROUNDS=2 timeout -k 10 400 python3 tools/ab_variants.py "@0" "@1" > $O/ab_c5.log 2>&1 && cat $O/ab_c5.log
