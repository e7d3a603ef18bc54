#!/usr/bin/env python3
"""Reduces the rocprofv3 CSVs written by tools/profile_set.sh and cuts the tracked summaries from them.

  summarize_prof.py <tag> --reduce-only   on the GPU box: gpurun_out/<tag>/**.csv -> gpurun_out/<tag>/reduced.json
                                          (per kernel: dispatches, total ns, counter sums; the big per-dispatch CSVs
                                          are deleted afterwards so the merge-back stays small)
  summarize_prof.py <tag>                 here: reduced.json -> profiles/<tag>_sq_summary.json, <tag>_pmc_summary.json,
                                          <tag>_kernel_stats.csv, <tag>_bench*.json, and the file bench.py reads:
                                          profiles/k_trace_counters.json, stamped with a hash of csrc/ (bench.py withholds
                                          the roofline fractions when the tree's hash differs)

Units: FETCH_SIZE / WRITE_SIZE in KiB.  FETCH_SIZE x 1024 = TCC_EA0_RDREQ x 64 B: exact for the 64-B gathers k_trace and
k_shade make (profiles/fetch_calibration.json, factor 1.0), half the bytes of a coalesced streaming read (factor 2,
MI355X_MICROARCH.md).  SQ_ACTIVE_INST_* and SQ_BUSY_CU_CYCLES as rocprofv3 reports them (ACTIVE_INST in quad-cycles summed
over SIMDs, BUSY_CU_CYCLES summed over CUs); one counter group per rocprofv3 pass, never combined with a trace domain
other than --kernel-trace."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
O = os.path.join(ROOT, "gpurun_out", tag)
KERNELS = ("k_trace", "k_tail", "k_light", "k_light_packet", "k_shade", "k_shade_lean", "k_arm", "k_resolve", "k_init", "k_heavy_scan", "k_heavy_pack")
GATHER_FETCH_FACTOR = 1.0  # profiles/fetch_calibration.json: FETCH_SIZE counts the 64-B sectors a gather moves exactly
N_CU, N_SIMD = 256, 1024
VALU_PEAK_LANE_OPS = N_SIMD * 32 * 2.4e9  # one wave64 VALU instruction per SIMD per 2 clocks (bench.py, VALU_PEAK_TLANEOPS)


def reduce_dir(d):
    """per kernel: {dispatches, ns, counters{name: sum}} of the newest counter_collection.csv under d"""
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not files:
        return None
    f = max(files, key=os.path.getmtime)
    seen = collections.defaultdict(set)
    out = collections.defaultdict(lambda: {"dispatches": 0, "ns": 0, "counters": collections.defaultdict(float)})
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        k = {"k_trace_wide": "k_trace"}.get(k, k)  # the wide-walk form of k_trace (launched for renders with early exits) counts as k_trace
        if k not in KERNELS:
            continue
        o = out[k]
        o["counters"][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"])
            o["dispatches"] += 1
            o["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return {k: {"dispatches": v["dispatches"], "ns": v["ns"], "counters": dict(v["counters"])} for k, v in out.items()}


def bench_line(path):
    if not os.path.exists(path):
        return None
    for line in reversed(open(path, errors="replace").read().splitlines()):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    return None


def reduce_all():
    red = {}
    for pre in ("", "c5_"):
        for i in (1, 2, 3, 4, 5):
            d = os.path.join(O, "%spmc_g%d" % (pre, i))
            if os.path.isdir(d):
                red["%sg%d" % (pre, i)] = {"kernels": reduce_dir(d), "bench": bench_line(d + ".log")}
                shutil.rmtree(d, ignore_errors=True)
        sd = os.path.join(O, pre + "stats")
        if os.path.isdir(sd):
            ks = glob.glob(os.path.join(sd, "*", "*kernel_stats.csv"))
            if ks:
                shutil.copy(max(ks, key=os.path.getmtime), os.path.join(O, pre + "kernel_stats.csv"))
            shutil.rmtree(sd, ignore_errors=True)
    json.dump(red, open(os.path.join(O, "reduced.json"), "w"), indent=1)
    print("reduced:", sorted(red))


def sq_block(kern, ns_override=None):
    c = kern["counters"]
    insts, thr, act, busy = c.get("SQ_INSTS_VALU"), c.get("SQ_THREAD_CYCLES_VALU"), c.get("SQ_ACTIVE_INST_VALU"), c.get("SQ_BUSY_CU_CYCLES")
    if not insts:
        return None
    lanes = thr / insts
    # MI355X_MICROARCH.md: SQ_ACTIVE_INST_* count quad-cycles (summed over the SIMDs); BUSY_CU_CYCLES is summed over CUs
    valu_busy = 4.0 * act / N_SIMD / (busy / N_CU) if busy else None
    secs = (ns_override or kern["ns"]) * 1e-9
    return {"dispatches": kern["dispatches"], "total_ms_profiled": kern["ns"] * 1e-6, "SQ_INSTS_VALU": insts,
            "SQ_INSTS_SALU": c.get("SQ_INSTS_SALU"), "lanes_per_valu_inst": lanes, "valu_busy": valu_busy,
            "lane_ops": thr, "lane_ops_per_s_profiled": thr / secs if secs else None,
            "frac_of_valu_peak_profiled": thr / secs / VALU_PEAK_LANE_OPS if secs else None,
            "wave_cycles": c.get("SQ_WAVE_CYCLES"), "wait_inst_any": c.get("SQ_WAIT_INST_ANY"), "active_inst_any": c.get("SQ_ACTIVE_INST_ANY")}


def cut(pre, red, label, command):
    g1, g2, g3, g4 = (red.get(pre + "g%d" % i) for i in (1, 2, 3, 4))
    g5 = red.get(pre + "g5")
    P = os.path.join(ROOT, "profiles")
    res = {}
    if g1 and g1["kernels"]:
        b = g1["bench"] or {}
        sq = {"command": "rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU "
                         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -- python3 bench.py " + command,
              "derived": "lanes_per_valu_inst = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU (of 64); valu_busy = 4 * SQ_ACTIVE_INST_VALU / 1024 SIMDs "
                         "/ (SQ_BUSY_CU_CYCLES / 256 CUs); lane_ops = SQ_THREAD_CYCLES_VALU (one per active lane per VALU instruction); "
                         "VALU peak = 1024 SIMDs x 32 lanes/clock x 2.4 GHz = 78.6 T lane-ops/s; valu_busy is in 4-clock units (1 quad-cycle per instruction), 2.0 = that peak",
              "rays_of_profiled_run": b.get("rays"), "kernels": {}}
        for k, v in g1["kernels"].items():
            blk = sq_block(v)
            if blk:
                sq["kernels"][k] = blk
        kt = sq["kernels"].get("k_trace")
        if kt and b.get("rays"):
            # rays of the TIMED steps vs counters of the whole process (warm-up included): scale by the launches' share
            rl = b["roofline"]
            share = rl["launches"] / max(rl.get("launches_incl_warmup", rl["launches"]), 1)
            kt["note"] = "counters cover every launch of the process (warm-up included); per-ray figures use all rays of the process"
            rays_all = b.get("rays_k_trace_incl_warmup_this_rank") or b.get("rays_incl_warmup_this_rank") or b["rays"] / b["steps"] * (b["steps"] + b["warmup"])
            kt["valu_lane_ops_per_ray"] = kt["lane_ops"] / rays_all
            kt["valu_wave_insts_per_ray"] = kt["SQ_INSTS_VALU"] / rays_all
            kt["timed_launch_share"] = share
        if g5 and g5["kernels"]:
            # second SQ pass: where the waves' time goes (quad-cycles summed over waves; WAIT_ANY = parked on s_waitcnt / barrier)
            sq["wait_group_command"] = "rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM ..."
            for k, v in g5["kernels"].items():
                if k in sq["kernels"]:
                    sq["kernels"][k]["second_pass"] = dict(v["counters"], dispatches=v["dispatches"], total_ms_profiled=v["ns"] * 1e-6)
        json.dump(sq, open(os.path.join(P, "%s_%ssq_summary.json" % (tag, pre)), "w"), indent=1)
        res["sq"] = sq
    if g2 and g3 and g4 and g2["kernels"] and g3["kernels"] and g4["kernels"]:
        out = {}
        for g in (g2, g3, g4):
            for k, v in g["kernels"].items():
                for cn, s in v["counters"].items():
                    out.setdefault(k, {})[cn] = {"dispatches": v["dispatches"], "sum": s, "total_ms_profiled": v["ns"] * 1e-6}
        t = out.get("k_trace", {})
        b = g2["bench"] or {}
        summ = {"build": tag, "command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py " + command +
                "   (one pass per group: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum)",
                "units": "FETCH_SIZE/WRITE_SIZE in KiB; bytes = value*1024; FETCH_SIZE taken as it reads (factor 1.0: these kernels gather 64-B sectors, profiles/fetch_calibration.json)",
                "counters": out}
        if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
            n = t["FETCH_SIZE"]["dispatches"]
            hbm = (GATHER_FETCH_FACTOR * t["FETCH_SIZE"]["sum"] + t["WRITE_SIZE"]["sum"]) * 1024
            secs = t["FETCH_SIZE"]["total_ms_profiled"] * 1e-3
            summ["k_trace_hbm_bytes_total"] = hbm
            summ["k_trace_hbm_bytes_per_launch"] = hbm / n
            summ["k_trace_hbm_GBps_profiled"] = hbm / secs / 1e9
            summ["k_trace_l2_hit_rate"] = t["TCC_HIT_sum"]["sum"] / (t["TCC_HIT_sum"]["sum"] + t["TCC_MISS_sum"]["sum"])
            summ["k_trace_l2_requests_total"] = t["TCC_HIT_sum"]["sum"] + t["TCC_MISS_sum"]["sum"]
            summ["k_trace_l2_request_GBps_profiled"] = summ["k_trace_l2_requests_total"] * 64 / (t["TCC_HIT_sum"]["total_ms_profiled"] * 1e-3) / 1e9
            if b.get("roofline"):
                rays_all = b.get("rays_k_trace_incl_warmup_this_rank") or b.get("rays_incl_warmup_this_rank") or b["rays"] / b["steps"] * (b["steps"] + b["warmup"])
                summ["k_trace_hbm_bytes_per_ray"] = hbm / rays_all
                summ["k_trace_l2_requests_per_ray"] = summ["k_trace_l2_requests_total"] / rays_all
                summ["rays_of_profiled_process"] = rays_all
        for k in ("k_shade", "k_shade_lean", "k_light", "k_light_packet"):
            tt = out.get(k, {})
            if "FETCH_SIZE" in tt and "WRITE_SIZE" in tt:
                summ[k + "_hbm_bytes_total"] = (GATHER_FETCH_FACTOR * tt["FETCH_SIZE"]["sum"] + tt["WRITE_SIZE"]["sum"]) * 1024
                summ[k + "_hbm_GBps_profiled"] = summ[k + "_hbm_bytes_total"] / (tt["FETCH_SIZE"]["total_ms_profiled"] * 1e-3) / 1e9
        json.dump(summ, open(os.path.join(P, "%s_%spmc_summary.json" % (tag, pre)), "w"), indent=1)
        res["pmc"] = summ
    for name in ("bench.json", "bench_profiled.json", "kernel_stats.csv"):
        src = os.path.join(O, pre + name)
        if os.path.exists(src) and os.path.getsize(src):
            shutil.copy(src, os.path.join(P, "%s_%s%s" % (tag, pre, name)))
    return res


def main():
    if "--reduce-only" in sys.argv:
        reduce_all()
        return
    red = json.load(open(os.path.join(O, "reduced.json")))
    c3 = cut("", red, "C3", "--no-cpu-baseline --no-extras")
    c5 = cut("c5_", red, "C5", "--config C5 --steps 2 --warmup 1 --spp-per-step 64 --no-cpu-baseline")
    P = os.path.join(ROOT, "profiles")

    def load(name):
        try:
            d = json.load(open(os.path.join(P, name)))
            return d if isinstance(d.get("C3", {}), dict) and "source" not in d else {}
        except Exception:
            return {}

    sys.path.insert(0, ROOT)
    from bench import csrc_hash
    ctr = {"csrc_sha": csrc_hash(), "tag": tag,
           "note": "per-ray counter figures of k_trace from the rocprofv3 --pmc passes of `python3 bench.py --no-cpu-baseline --no-extras` "
                   "(C5: --config C5 --steps 2 --warmup 1 --spp-per-step 64); bench.py multiplies them by the rays and divides by the k_trace time of ITS run, "
                   "and withholds the fractions when csrc_sha is not the tree's"}
    for key, pre, res in (("C3", "", c3), ("C5", "c5_", c5)):
        e = {}
        if "sq" in res and "k_trace" in res["sq"]["kernels"]:
            kt = res["sq"]["kernels"]["k_trace"]
            e.update({"source": "profiles/%s_%ssq_summary.json + %s_%spmc_summary.json" % (tag, pre, tag, pre),
                      "valu_lane_ops_per_ray": kt.get("valu_lane_ops_per_ray"), "valu_wave_insts_per_ray": kt.get("valu_wave_insts_per_ray"),
                      "lanes_per_valu_inst": kt["lanes_per_valu_inst"], "valu_busy": kt["valu_busy"]})
            sp = kt.get("second_pass") or {}
            if sp.get("SQ_ACTIVE_INST_VALU2") and kt.get("SQ_INSTS_VALU"):
                # quad-cycles in which two waves' VALU instructions were in execution together, per VALU instruction (1.007 quad-cycles each)
                e["valu2_share_of_valu_time"] = sp["SQ_ACTIVE_INST_VALU2"] / (kt["SQ_INSTS_VALU"] * 1.007)
        if "pmc" in res and "k_trace_hbm_bytes_per_ray" in res["pmc"]:
            e.update({"hbm_bytes_per_ray": res["pmc"]["k_trace_hbm_bytes_per_ray"], "fetch_size_factor": GATHER_FETCH_FACTOR,
                      "l2_requests_per_ray": res["pmc"].get("k_trace_l2_requests_per_ray"), "l2_hit_rate": res["pmc"]["k_trace_l2_hit_rate"]})
        if e:
            ctr[key] = e
    json.dump(ctr, open(os.path.join(P, "k_trace_counters.json"), "w"), indent=1)
    print("wrote profiles/%s_*" % tag)


if __name__ == "__main__":
    main()
