#!/bin/bash
# the bench lines with the re-stamped counter file (rooflines populated): C3 default, the driver's form (20 steps), C5
mkdir -p gpurun_out/r03_f2
O=gpurun_out/r03_f2
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; echo
timeout -k 10 600 python3 bench.py --config C5 --steps 2 --warmup 1 --spp-per-step 64 > $O/c5_bench.json 2> $O/c5_bench.err; tail -c 200 $O/c5_bench.json; echo
timeout -k 10 600 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_20.json 2> $O/bench_20.err; tail -c 200 $O/bench_20.json; echo
timeout -k 10 600 python3 bench.py --virtual-ranks 8 --no-cpu-baseline --no-extras > $O/bench_v8.json 2> $O/bench_v8.err; tail -c 200 $O/bench_v8.json; echo
