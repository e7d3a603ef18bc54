#!/bin/bash
# Round-4 GPU-box entry point: one script, one target per kind of call (the round-3 one-shots tools/r03_*.sh are in the git history).
#   tools/r04.sh tests [pytest args]        the GPU suite (default: all of tests -m gpu) -> gpurun_out/r04_tests/pytest.log
#   tools/r04.sh quick <tag> [pytest files] a subset of the suite, then the default bench line (side runs give the walk A/B) and C5
#   tools/r04.sh bench <tag> [bench args]   one bench line -> gpurun_out/<tag>/bench.json
#   tools/r04.sh env <tag> VAR=a,b,c [bench args]   the default bench once per value of an environment switch (same box)
#   tools/r04.sh profile <tag> [C5]         tools/profile_set.sh: bench + kernel stats + PMC passes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
what=$1; shift
line() {  # one summary line of a bench JSON
  python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k = d["kernels"]
    rw, ew, oc = d.get("reference_walk") or {}, d.get("early_exit_walk") or {}, d.get("occluder_cache") or {}
    cu = d.get("statue_closeup") or {}
    gl = d.get("glass_statue") or {}
    print("%-14s %7.0f Mray/s  ms/step %6.1f  k_trace %6.1f (%.0f Mray/s, V %.1f T %.1f)  first pass %6.1f  rest %6.1f  flush %.0f ms | ref walk %s  early %s  cache answered %s | close-up %s  glass %s | parity %s" % (
        sys.argv[2], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["k_trace"]["Mray_per_s"] or 0, k["k_trace"]["nodes_per_ray"], k["k_trace"]["tris_per_ray"],
        k["k_light"]["ms_per_step"], k["rest_ms_per_step"], d["final_flush"]["ms"],
        "%.0f (k_trace %.1f)" % (rw["value"], rw["k_trace_ms_per_step"]) if rw else "-", "%.0f (k_trace %.1f)" % (ew["value"], ew["k_trace_ms_per_step"]) if ew else "-",
        "%.2f" % oc["share_of_shadow_and_env_rays"] if oc else "-", "%.0f" % cu["value"] if cu else "-",
        "%.0f (parity %s)" % (gl["value"], (gl.get("parity_check") or {}).get("ok")) if gl else "-", (d.get("parity_check") or {}).get("ok")))
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
case "$what" in
  tests)
    O="$R/gpurun_out/r04_tests"; mkdir -p "$O"
    timeout -k 10 1100 python3 -m pytest ${@:-tests} -m gpu -x -q > "$O/pytest.log" 2>&1; rc=$?
    tail -6 "$O/pytest.log"; [ $rc -eq 0 ] || { tail -60 "$O/pytest.log"; exit $rc; } ;;
  quick)
    tag=$1; shift; O="$R/gpurun_out/$tag"; mkdir -p "$O"
    timeout -k 10 900 python3 -m pytest ${@:-tests/test_gpu_early_exit.py tests/test_gpu_packet.py tests/test_gpu_parity.py} -m gpu -x -q > "$O/pytest.log" 2>&1; rc=$?
    tail -4 "$O/pytest.log"; [ $rc -eq 0 ] || { tail -80 "$O/pytest.log"; exit $rc; }
    timeout -k 10 600 python3 bench.py > "$O/bench.json" 2> "$O/bench.err" && line "$O/bench.json" C3 || { tail -20 "$O/bench.err"; exit 1; }
    timeout -k 10 600 python3 bench.py --config C5 --steps 2 --warmup 1 --spp-per-step 64 --no-cpu-baseline > "$O/c5_bench.json" 2> "$O/c5_bench.err" && line "$O/c5_bench.json" C5 || { tail -20 "$O/c5_bench.err"; exit 1; } ;;
  bench)
    tag=$1; shift; O="$R/gpurun_out/$tag"; mkdir -p "$O"
    timeout -k 10 900 python3 bench.py "$@" > "$O/bench.json" 2> "$O/bench.err" && line "$O/bench.json" "$tag" || { tail -20 "$O/bench.err"; exit 1; } ;;
  env)
    tag=$1; var=${2%%=*}; vals=${2#*=}; shift 2; O="$R/gpurun_out/$tag"; mkdir -p "$O"
    for v in ${vals//,/ }; do
      f=$(basename "$v")  # (a value may be a path: JADE_HIP_LIB=.../libjade_hip_x.so)
      env "$var=$v" timeout -k 10 600 python3 bench.py --no-cpu-baseline "$@" > "$O/${var}_$f.json" 2> "$O/${var}_$f.err" && line "$O/${var}_$f.json" "$var=$f" || { tail -20 "$O/${var}_$f.err"; exit 1; }
    done ;;
  profile)
    bash tools/profile_set.sh "$@" ;;
  *) echo "usage: $0 tests|quick|bench|env|profile ..."; exit 2 ;;
esac
