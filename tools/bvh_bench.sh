R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
show() { python3 -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], '%.0f Mray/s  ms/step %.1f  V/ray %.1f T/ray %.1f  build_s %.2f dev_ms %s nodes %d depth %d' % (d['value'], d['ms_per_step'], d['nodes_per_ray'], d['tris_per_ray'], d['scene_build_s'], d['device_bvh_ms'], d['config']['bvh_nodes'], d['config']['bvh_depth']))" $1 "$2"; }
A="--no-cpu-baseline --no-extras --steps 2 --warmup 1"
timeout -k 10 200 python bench.py $A > gpurun_out/bb_sah.json 2>/dev/null; show gpurun_out/bb_sah.json "C3 sah"
timeout -k 10 200 python bench.py $A --bvh ploc > gpurun_out/bb_ploc3.json 2>/dev/null; show gpurun_out/bb_ploc3.json "C3 ploc leaf3"
timeout -k 10 200 python bench.py $A --bvh ploc --bvh-leaf 8 > gpurun_out/bb_ploc8.json 2>/dev/null; show gpurun_out/bb_ploc8.json "C3 ploc leaf8"
timeout -k 10 200 python bench.py $A --bvh ploc --bvh-leaf 2 > gpurun_out/bb_ploc2.json 2>/dev/null; show gpurun_out/bb_ploc2.json "C3 ploc leaf2"
timeout -k 10 200 python bench.py $A --bvh lbvh > gpurun_out/bb_lbvh.json 2>/dev/null; show gpurun_out/bb_lbvh.json "C3 lbvh"
C="--config C5 --steps 2 --warmup 1 --spp-per-step 64 --no-cpu-baseline --no-extras"
timeout -k 10 300 python bench.py $C > gpurun_out/bb5_sah.json 2>/dev/null; show gpurun_out/bb5_sah.json "C5 sah"
timeout -k 10 300 python bench.py $C --bvh ploc > gpurun_out/bb5_ploc3.json 2>/dev/null; show gpurun_out/bb5_ploc3.json "C5 ploc leaf3"
timeout -k 10 300 python bench.py $C --bvh ploc --bvh-leaf 8 > gpurun_out/bb5_ploc8.json 2>/dev/null; show gpurun_out/bb5_ploc8.json "C5 ploc leaf8"
