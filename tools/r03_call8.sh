#!/bin/bash
# round 3, GPU call 8: k_trace variants against the build before the merged candidate entries
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/r03_c8"; mkdir -p "$O"; cd "$R"
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "trace_rays or golden or small_configs or c1_cornell or c5_deep or stack_spill" > "$O/pytest.log" 2>&1
rc=$?; tail -3 "$O/pytest.log"
if [ $rc -ne 0 ]; then echo "parity failed: stopping"; tail -30 "$O/pytest.log"; exit $rc; fi
AB_ROUNDS=3 timeout -k 10 500 python3 tools/ab_variants.py _old "" _notop _hq32 > "$O/ab1.log" 2>&1; tail -5 "$O/ab1.log"
AB_ROUNDS=3 timeout -k 10 500 python3 tools/ab_variants.py "" _rf8 _rf24 _sp3 _sp6 > "$O/ab2.log" 2>&1; tail -6 "$O/ab2.log"
AB_CLOSEUP=1 AB_ROUNDS=2 timeout -k 10 500 python3 tools/ab_variants.py _old "" _notop > "$O/ab3.log" 2>&1; tail -4 "$O/ab3.log"
AB_CONFIG=C5 AB_ROUNDS=2 timeout -k 10 500 python3 tools/ab_variants.py _old "" _notop > "$O/ab4.log" 2>&1; tail -4 "$O/ab4.log"
