#!/bin/bash
set -e
mkdir -p gpurun_out/r03_ee4
O=gpurun_out/r03_ee4
timeout -k 10 400 python3 tools/ab_variants.py "" _top144 _rm12 _rm24 _hq24 _hq48 > $O/ab_c3.log 2>&1 && cat $O/ab_c3.log
AB_CLOSEUP=1 timeout -k 10 400 python3 tools/ab_variants.py "" _top144 _rm12 _rm24 _hq24 _hq48 > $O/ab_close.log 2>&1 && cat $O/ab_close.log
