#!/usr/bin/env python3
"""Condenses gpurun_out/mix_<tag>/ (scripts/valu_mix.sh) into profiles/<tag>_valu_mix.{md,json}: the sampler kernel's
vector-instruction mix per chain, the clock the chip held, and the VALU-time floor that mix implies with the issue
costs measured by scripts/micro/valu_rates.hip (profiles/r02_valu_rates.txt)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02b"
kernel = sys.argv[2] if len(sys.argv) > 2 else "gibbs_lean_kernel"
src = os.path.join(ROOT, "gpurun_out", f"mix_{tag}")

counters, durs, waves = {}, [], None
# (rocprofv3 also traces bench.py's cold-start children, which run small products of their own: per counter pass keep the
# largest csv, the bench process itself)
paths = []
for passdir in sorted(glob.glob(os.path.join(src, "*", ""))):
    files = glob.glob(os.path.join(passdir, "*", "*counter_collection.csv"))
    if files:
        paths.append(max(files, key=os.path.getsize))
for path in paths:
    acc = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel not in row["Kernel_Name"]:
                continue
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            if row["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"):
                durs.append((row["Counter_Name"], float(row["End_Timestamp"]) - float(row["Start_Timestamp"])))
    for k, v in acc.items():
        v = sorted(v)[: max(1, len(v) - 1)] if len(v) > 3 else v  # (the one table-build launch is the largest)
        counters[k] = sum(v) / len(v)

line = json.loads(open(os.path.join(src, "bench_f64.json")).read().strip().splitlines()[-1])
nchains = line["config"]["nout_per_gpu"]
per = {k: v / nchains for k, v in counters.items()}
clk_dur = [d for n, d in durs if n == "GRBM_GUI_ACTIVE"]
clk_dur = sorted(clk_dur)[len(clk_dur) // 2] if clk_dur else None
ghz = counters.get("GRBM_GUI_ACTIVE", 0) / 8.0 / clk_dur if clk_dur else None  # cycles summed over 8 XCDs / ns

# issue cost in shader cycles per wave-instruction and SIMD (profiles/r02_valu_rates.txt, 4 wavefronts per SIMD)
COST = {"f64": 4.3, "trans64": 16.3, "f32": 2.7, "trans32": 8.2, "other": 3.4}
f64 = per.get("SQ_INSTS_VALU_ADD_F64", 0) + per.get("SQ_INSTS_VALU_MUL_F64", 0) + per.get("SQ_INSTS_VALU_FMA_F64", 0)
t64 = per.get("SQ_INSTS_VALU_TRANS_F64", 0)
f32 = per.get("SQ_INSTS_VALU_ADD_F32", 0) + per.get("SQ_INSTS_VALU_MUL_F32", 0) + per.get("SQ_INSTS_VALU_FMA_F32", 0)
t32 = per.get("SQ_INSTS_VALU_TRANS_F32", 0)
total = per.get("SQ_INSTS_VALU", 0)
other = total - f64 - t64 - f32 - t32
cycles_per_chain = f64 * COST["f64"] + t64 * COST["trans64"] + f32 * COST["f32"] + t32 * COST["trans32"] + other * COST["other"]
simds = 1024
chains_per_simd = nchains / simds
out = {
    "tag": tag, "kernel": kernel, "chains": nchains, "per_chain": per, "clock_ghz": ghz,
    "mix_per_chain": {"fp64 add/mul/fma": f64, "fp64 transcendental": t64, "fp32 add/mul/fma": f32,
                      "fp32 transcendental": t32, "other (int, mov, cmp, select, dpp, lane)": other, "total": total},
    "issue_cost_cycles": COST,
    "valu_cycles_per_chain": cycles_per_chain,
    "valu_floor_ms": cycles_per_chain * chains_per_simd / (ghz * 1e6) if ghz else None,
    "bench_line_under_profiler": {k: line.get(k) for k in ("value", "ms_per_step")},
    "kernel_ms_under_profiler": line.get("roofline", {}).get("kernel_ms"),
}
with open(os.path.join(ROOT, "profiles", f"{tag}_valu_mix.json"), "w") as f:
    json.dump(out, f, indent=1)
with open(os.path.join(ROOT, "profiles", f"{tag}_valu_mix.md"), "w") as f:
    f.write(f"# vector-instruction mix `{tag}` ({kernel}, {nchains} chains)\n\n")
    f.write("`scripts/valu_mix.sh` (rocprofv3 PMC passes of their own) condensed by `scripts/summarize_mix.py`.\n\n")
    f.write("| class | instructions per chain | issue cost (cycles per wave-instruction and SIMD) |\n|---|---|---|\n")
    for (name, n), c in zip(list(out["mix_per_chain"].items())[:5], COST.values()):
        f.write(f"| {name} | {n:.0f} | {c} |\n")
    f.write(f"| total | {total:.0f} | |\n\n")
    f.write(f"clock held (GRBM_GUI_ACTIVE / 8 / kernel time): {ghz:.2f} GHz\n\n" if ghz else "")
    f.write(f"VALU issue cycles per chain: {cycles_per_chain:.0f}; with {chains_per_simd:g} chains per SIMD the vector pipe alone "
            f"needs {out['valu_floor_ms']:.3f} ms per launch at that clock.\n\n")
    f.write("other counters per chain: " + ", ".join(f"{k}={v:.0f}" for k, v in sorted(per.items())) + "\n")
# bench.py reads the committed profile of its workload (profiles/traffic_<config>.json): attach the floor to it
wl = line["config"]["workload"].split(":")[0]
for name in [f"traffic_{wl}.json"] + (["traffic_latest.json"] if wl == "c3" else []):
    tl = os.path.join(ROOT, "profiles", name)
    try:
        t = json.load(open(tl))
        if t.get("workload") == wl:
            t["valu_floor"] = {"cycles_per_chain": cycles_per_chain, "clock_ghz": ghz, "simds": simds,
                               "mix_per_chain": out["mix_per_chain"], "issue_cost_cycles": COST,
                               "source": f"profiles/{tag}_valu_mix.json + profiles/r02_valu_rates.txt"}
            # what the counters say the vector pipe was doing: quads (4 cycles) in which a wavefront's vector instruction
            # executed, against the quads the wavefront was resident (bench.py: x wavefronts per SIMD = occupancy of the pipe)
            if per.get("SQ_ACTIVE_INST_VALU") and per.get("SQ_WAVE_CYCLES"):
                t["valu_busy"] = {"active_inst_valu_quads_per_chain": per["SQ_ACTIVE_INST_VALU"],
                                  "wave_quads_per_chain": per["SQ_WAVE_CYCLES"], "chains_profiled": nchains,
                                  "source": f"profiles/{tag}_valu_mix.json"}
            json.dump(t, open(tl, "w"), indent=1)
    except (OSError, ValueError, KeyError):
        pass
