#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (written by scripts/profile_gpu.sh) into the tracked
files profiles/<tag>_rocprof_summary.md / .json and profiles/traffic_latest.json.

Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KiB of memory-side L2 traffic;
on gfx950 FETCH_SIZE under-reports wide (16 B/lane) coalesced reads by exactly 2x; other widths are
uncalibrated, so `fetch_calibration` (measured for this kernel's 8 B/lane loads by
scripts/calibrate_fetch.hip when available, else 2.0 as the conservative upper bound) multiplies it.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def modal_grid(rows, key):
    """The sampling launches all have the same grid; the one-off table-build launch of the same kernel
    (one wavefront per table row) is left out of the per-launch averages."""
    c = defaultdict(int)
    for r in rows:
        c[r[key]] += 1
    return max(c, key=c.get) if c else None


def main_process(files):
    """rocprofv3 also traces child processes (bench.py's cold-start children run small products of their own): one set of
    csv files per process id -- keep the largest, the bench process itself."""
    files = list(files)
    return [max(files, key=os.path.getsize)] if files else []


def counters(d, sub, match):
    acc = defaultdict(list)
    for f in main_process(glob.glob(os.path.join(d, sub, "*", "*_counter_collection.csv"))):
        rows = [r for r in csv.DictReader(open(f)) if match in r["Kernel_Name"]]
        grid = modal_grid(rows, "Grid_Size")
        for r in rows:
            if r["Grid_Size"] == grid:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc["_vgpr"].append(float(r["VGPR_Count"]))
                acc["_sgpr"].append(float(r["SGPR_Count"]))
                acc["_lds"].append(float(r["LDS_Block_Size"]))
                acc["_grid"].append(float(r["Grid_Size"]))
                acc["_wg"].append(float(r["Workgroup_Size"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    tag = sys.argv[1]
    match = sys.argv[2] if len(sys.argv) > 2 else "gibbs_lean_kernel"
    workload = sys.argv[3] if len(sys.argv) > 3 else "c3"
    calib = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    stats = []
    for f in main_process(glob.glob(os.path.join(d, "stats", "*", "*_kernel_stats.csv"))):
        stats += list(csv.DictReader(open(f)))
    krow = next(r for r in stats if match in r["Name"])
    # per-dispatch durations from the kernel trace, sampling launches only (see modal_grid)
    trace = []
    for f in main_process(glob.glob(os.path.join(d, "stats", "*", "*_kernel_trace.csv"))):
        trace += [r for r in csv.DictReader(open(f)) if match in r["Kernel_Name"]]
    gkey = "Grid_Size" if trace and "Grid_Size" in trace[0] else ("Grid_Size_X" if trace and "Grid_Size_X" in trace[0] else None)
    durs = []
    if gkey:
        grid = modal_grid(trace, gkey)
        durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in trace if r[gkey] == grid]
    if not durs:
        durs = [float(krow["AverageNs"])]
    out = {"tag": tag, "workload": workload, "kernel": krow["Name"], "calls": len(durs),
           "calls_incl_table_build": int(krow["Calls"]),
           "avg_ns": sum(durs) / len(durs), "min_ns": min(durs), "max_ns": max(durs),
           "pct_of_gpu_time": float(krow["Percentage"])}
    pmc = {}
    for sub in ("fetch", "write", "sq", "sq2", "sq3"):
        if os.path.isdir(os.path.join(d, sub)):
            m, _ = counters(d, sub, match)
            pmc.update(m)
    out["pmc_avg_per_launch"] = {k: v for k, v in pmc.items() if not k.startswith("_")}
    out["launch"] = {"vgpr": pmc.get("_vgpr"), "sgpr": pmc.get("_sgpr"), "lds_bytes": pmc.get("_lds"),
                     "grid_threads": pmc.get("_grid"), "workgroup": pmc.get("_wg")}
    fetch_kib, write_kib = pmc.get("FETCH_SIZE"), pmc.get("WRITE_SIZE")
    if fetch_kib is not None and write_kib is not None:
        out["hbm_bytes_per_launch_raw"] = (fetch_kib + write_kib) * 1024.0
        out["fetch_calibration"] = calib
        out["hbm_bytes_per_launch"] = (fetch_kib * calib + write_kib) * 1024.0
    # The binding resource is the fp64 VALU pipe, not HBM: share of SIMD-cycles with a VALU instruction
    # in flight (SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; 1024 SIMDs; cycles from the
    # kernel duration at the 2.4 GHz maximum clock, i.e. a lower bound on the utilisation).
    if "SQ_ACTIVE_INST_VALU" in pmc:
        simd_cycles = out["avg_ns"] * 1e-9 * 2.4e9 * 1024
        out["valu_pipe_busy_frac_lower_bound"] = pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles
        if "SQ_INSTS_VALU" in pmc and "SQ_WAVES" in pmc:
            out["valu_insts_per_chain"] = pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"]
    # Issue-slot view (what binds the kernel): a wavefront issues at most one instruction per 4-cycle slot of its
    # SIMD; SQ_WAVE_CYCLES and SQ_ACTIVE_INST_* count those slots ("quads") summed over wavefronts.
    if all(k in pmc for k in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")):
        nw = pmc["SQ_WAVES"]
        out["issue"] = {
            "valu_insts_per_chain": pmc["SQ_INSTS_VALU"] / nw,
            "salu_insts_per_chain": pmc["SQ_INSTS_SALU"] / nw,
            "lds_insts_per_chain": pmc["SQ_INSTS_LDS"] / nw,
            "issue_slots_per_chain": pmc["SQ_WAVE_CYCLES"] / nw,
            "slots_issuing_frac": pmc.get("SQ_ACTIVE_INST_ANY", float("nan")) / pmc["SQ_WAVE_CYCLES"],
            "slots_waiting_for_data_frac": pmc.get("SQ_WAIT_ANY", float("nan")) / pmc["SQ_WAVE_CYCLES"],
            "slots_waiting_to_issue_frac": pmc.get("SQ_WAIT_INST_ANY", float("nan")) / pmc["SQ_WAVE_CYCLES"],
            "unit": "4-cycle issue slots of one wavefront (SQ 'quad' cycles); tag " + tag,
        }
    bench = {}
    try:
        bench = json.loads(open(os.path.join(d, "bench_stats.json")).read().strip().splitlines()[-1])
        out["bench_line_under_profiler"] = {"kernel_ms": bench["roofline"]["kernel_ms"], "value": bench["value"],
                                            "roofline_frac": bench["roofline"]["frac"]}
    except Exception as e:  # noqa: BLE001
        out["bench_line_under_profiler"] = f"unavailable: {e}"
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    if "hbm_bytes_per_launch" in out:  # what bench.py --config <workload> replays next to its live numbers
        import datetime
        import subprocess
        try:
            commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        except OSError:
            commit = ""
        names = [f"traffic_{workload}.json"] + (["traffic_latest.json"] if workload == "c3" else [])
        for name in names:
            tl = os.path.join(ROOT, "profiles", name)
            keep = {}
            try:  # the vector-pipe floor (scripts/summarize_mix.py) stays attached until it is re-measured
                prev = json.load(open(tl))
                if prev.get("workload") == workload and "valu_floor" in prev:
                    keep["valu_floor"] = prev["valu_floor"]
            except (OSError, ValueError):
                pass
            with open(tl, "w") as f:
                json.dump({"workload": workload, "tag": tag, "commit": commit, "date": datetime.date.today().isoformat(),
                           "kernel": out["kernel"], "kernel_avg_ns_under_rocprof": out["avg_ns"],
                           "hbm_bytes_per_launch": out["hbm_bytes_per_launch"],
                           "raw_bytes": out["hbm_bytes_per_launch_raw"], "fetch_calibration": calib,
                           "issue": out.get("issue"), **keep}, f, indent=1)
    lines = [f"# rocprofv3 summary `{tag}` ({workload})", "",
             "Command (on the MI355X box): `scripts/profile_gpu.sh " + tag + "` = `rocprofv3 --kernel-trace --stats -- python3 bench.py "
             "--steps 20 --warmup 3 --no-cpu-baseline` plus separate `--pmc` passes.", "",
             "## kernel stats (`--kernel-trace --stats`)", "", "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
    lines.append(f"| `{out['kernel'][:90]}` sampling launches only | {out['calls']} | {out['avg_ns']:.0f} | {out['min_ns']:.0f} | {out['max_ns']:.0f} | |")
    for r in stats[:6]:
        lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
    lines += ["", "## PMC counters, average per launch of the product kernel", "", "| counter | value |", "|---|---|"]
    for k, v in sorted(out["pmc_avg_per_launch"].items()):
        lines.append(f"| {k} | {v:.6g} |")
    lines += ["", f"launch: {out['launch']}", ""]
    if "hbm_bytes_per_launch" in out:
        lines += [f"HBM bytes per launch: raw (FETCH_SIZE+WRITE_SIZE)*1024 = {out['hbm_bytes_per_launch_raw']:.4g}; "
                  f"with FETCH_SIZE x {calib} = {out['hbm_bytes_per_launch']:.4g}", ""]
    if "issue" in out:
        i = out["issue"]
        lines += [f"issue slots per chain: {i['issue_slots_per_chain']:.0f}; instructions per chain: VALU {i['valu_insts_per_chain']:.0f}, "
                  f"SALU {i['salu_insts_per_chain']:.0f}, LDS {i['lds_insts_per_chain']:.0f}; slots issuing {i['slots_issuing_frac']:.3f}, "
                  f"waiting for data {i['slots_waiting_for_data_frac']:.3f}, waiting to issue {i['slots_waiting_to_issue_frac']:.3f}", ""]
    if "valu_pipe_busy_frac_lower_bound" in out:
        lines += [f"fp64 VALU pipe busy (lower bound, 2.4 GHz): {out['valu_pipe_busy_frac_lower_bound']:.3f}; "
                  f"VALU instructions per chain: {out.get('valu_insts_per_chain', float('nan')):.0f}", ""]
    lines += [f"bench line under the profiler: {out['bench_line_under_profiler']}", ""]
    with open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.md"), "w") as f:
        f.write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
