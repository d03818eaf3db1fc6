#!/usr/bin/env python3
"""bench.py -- Gibbs-product throughput of the HIP path on MI355X (BASELINE.json metric).

A "step" is ONE COMPLETE `prodAppxMSGibbsS`-equivalent call (reference src/MSGibbs01.jl:645-703; SURVEY.md 8d: "Nout /
wall time of one prodAppxMSGibbsS-equivalent call") on the headline workload (BASELINE config 3: 6-D, 4 densities x
1000 points, Nout = 2048 per GPU, Niter = 10, fp64): the input densities -- the reference's flat BallTreeDensity
arrays -- are resident in HBM when the clock starts (uploaded once, `kdehip_density_upload`); every step re-lays them
out into level tiles on the GPU, builds the conditional tables, draws its Nout chains (`kdehip_prod_philox_device`) and,
with N > 1 ranks, all-gathers the product samples over RCCL.  Nothing is reused between steps but the densities
themselves; outputs stay in HBM.  Weak scaling: per-GPU work is fixed.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Beside `value` it carries, each measured in the same run: `call_inclusive` = the same
call with HOST buffers in and out (pack on the host, PCIe both ways: what a drop-in `ccall` from Julia sees; never
`value`), `resident_plan` = repeated sampling of ONE packed plan (no re-layout, no table build), and `roofline` =
the sampling kernel alone (HIP events on the launch stream).  The path is bound by vector-ALU issue, not by memory
(the working set, 832 KB at config 3, is LDS/L2 resident; PMC-measured HBM traffic is ~10 MB per launch): `roofline`
reports the algorithmic fp64 (fp32) flops of the kernel evaluations -- E * (6D+4) per sample, E = (Niter+1) * sum_j
sum_l n_{j,l} -- per second of kernel time against the MI355X vector peak, and `traffic` carries the
PMC-measured HBM bytes per launch of the committed profile.  SURVEY.md 8(d)'s "algorithmic bytes / 8 TB/s" figure is
kept as `normalised_hbm` (a throughput normalisation, not a roofline: it exceeds 1).
--strong splits the configuration's TOTAL chain count over the ranks instead of giving every rank a full batch.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    # name: (D, M, N, Nout per GPU, Niter, precision, config_id)
    "c2": (2, 3, 200, 256, 5, 64, 2),
    "c3": (6, 4, 1000, 2048, 10, 64, 3),
    "c4": (3, 8, 5000, 2048, 10, 64, 4),      # 16384 over 8 GPUs
    "c5": (6, 4, 10000, 8192, 20, 32, 5),     # 65536 over 8 GPUs, fp32
}
# chains of the whole job as BASELINE.json states the configs (--strong splits these over the ranks)
TOTAL_NOUT = {"c2": 256, "c3": 2048, "c4": 16384, "c5": 65536}
VALU_PEAK_TFLOPS = {64: 78.6, 32: 157.3}  # MI355X vector peaks: 256 CU x 4 SIMD x 16 (32) lanes x 2 flop x 2.4 GHz


def synth_inputs(kdehip, D, M, N, config_id):
    """density j = N points from a 3-component Gaussian mixture (centres U(-2,2)^D,
    std 0.5, equal mixing), uniform weights, Silverman bandwidth; all draws from the library's host
    Philox with key 0x4B44452D48495000 + config_id, stream = j."""
    key = 0x4B44452D48495000 + config_id
    pts_all, bw_all = [], []
    for j in range(M):
        u, n = kdehip.philox_streams(key, j, 1, 3 * D + N, N * D)
        centres = (4.0 * u[: 3 * D] - 2.0).reshape(3, D)
        comp = np.minimum((u[3 * D:] * 3.0).astype(np.int64), 2)
        pts = centres[comp] + 0.5 * n.reshape(N, D)
        pts = np.ascontiguousarray(pts.T)
        bw = pts.std(axis=1, ddof=1) * (4.0 / ((D + 2.0) * N)) ** (1.0 / (D + 4.0))
        pts_all.append(pts)
        bw_all.append(bw)
    return pts_all, bw_all


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def load_traffic(workload):
    """The committed PMC profile of this workload (profiles/traffic_latest.json): HBM bytes per launch and the
    issue-slot counters, or an empty dict."""
    for name in (f"traffic_{workload}.json", "traffic_latest.json"):
        p = os.path.join(ROOT, "profiles", name)
        try:
            with open(p) as f:
                t = json.load(f)
            if t.get("workload") == workload:
                t["_file"] = "profiles/" + name
                return t
        except (OSError, ValueError):
            pass
    return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spin-up", action="store_true",
                    help="skip the 40 ms of resident-plan launches that bring the device's clock up before the warm-up steps")
    ap.add_argument("--spin-up-calls", type=int, default=48,
                    help="untimed calls of the measured path after the resident-plan launches of the spin-up (0 = none)")
    ap.add_argument("--spin-up-ms", type=float, default=40.0,
                    help="how long the device is kept busy with launches of one resident plan before the warm-up steps")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--nout", type=int, default=0, help="override chains per GPU (experiments; 0 = the config's)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: the configuration's TOTAL chain count split over the ranks")
    ap.add_argument("--inproc-gpus", type=int, default=0,
                    help="ONE process driving G GPUs through the C ABI's multi-device plans (kdehip_product_multi_*: the route a "
                         "Julia host takes; all-gather fused into the kernel epilogue as peer stores) instead of one process "
                         "per GPU + RCCL.  With KDEHIP_ALIAS_DEVICES=1 the G logical devices wrap around the visible ones.")
    ap.add_argument("--batch", type=int, default=0,
                    help="B independent products of the configuration's shape per step in ONE library call "
                         "(kdehip_prod_philox_batch: the serving pattern of many small products), with the same B products "
                         "enqueued one by one beside it")
    ap.add_argument("--mul", action="store_true",
                    help="with --batch B: a step = B complete `*` (product + kde!(pGM): LOOCV bandwidth + tree; reference "
                         "src/MSGibbs01.jl:707-726, Np = round(mean Npts), Niter = 5) in ONE kdehip_mul_device_batch call, "
                         "with the same B `*` as B kdehip_mul_device calls beside it")
    ap.add_argument("--frow", choices=["loocv", "evaluate", "tree"], default=None,
                    help="the callers either side of the product (SURVEY.md 8f) at the measurement bar: loocv = kde!(points)'s "
                         "bandwidth search on the product's output shape (D x Nout of --config), evaluate = evaluateDualTree(bd, pos) "
                         "of one density of --config at 65,536 positions, tree = kde!(points, ks) of D x Nout points, host builder "
                         "beside the GPU builder; each with roofline (kernel time by HIP events inside the library) and cpu_baseline")
    args = ap.parse_args()
    if args.frow:
        return frow_mode(args)
    if args.inproc_gpus > 0:
        return inproc_multi(args)
    if args.batch > 0 and args.mul:
        return mul_batch_mode(args)
    if args.batch > 0:
        return batch_mode(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    # The JSON line must be the only thing on stdout: RCCL prints a version banner through C stdio (flushed at
    # exit, i.e. AFTER a Python print) and torchrun merges the stdout of all ranks.  File descriptor 1 points at
    # stderr while the job runs and is restored on rank 0 only around the final print.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # one process per GPU; if the launcher narrowed the visible devices per rank, take what is visible
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or os.environ.get("KDEHIP_FORCE_DIST") == "1"  # (the latter: 1-rank test of the RCCL path)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    import kdehip
    from kdehip.sharded import ShardedProduct

    D, M, N, Nout, Niter, prec, cid = CONFIGS[args.config]
    if args.nout > 0:
        Nout = args.nout
    if args.strong:
        Np_total = TOTAL_NOUT[args.config] if args.nout <= 0 else args.nout
        Nout = (Np_total + world - 1) // world   # chains of the largest shard (what one launch processes)
    else:
        Np_total = Nout * world
    workload = (f"{args.config}: {D}-D, {M} densities x {N} pts, Nout={Np_total} total over {world} GPU(s), Niter={Niter}, fp{prec}"
                if args.strong else f"{args.config}: {D}-D, {M} densities x {N} pts, Nout={Nout}/GPU, Niter={Niter}, fp{prec}")
    pts_all, bw_all = synth_inputs(kdehip, D, M, N, cid)
    trees = [kdehip.kde(p, b) for p, b in zip(pts_all, bw_all)]
    plan = kdehip.ProductPlan(trees, precision=prec, device=dev_index)
    if args.variant:
        plan.set_variant(args.variant)
    sp = ShardedProduct(plan, dev)
    seed = 20260101

    # the densities live in HBM from here on (the timed region starts with resident inputs)
    dd = [kdehip.DeviceDensity(t, device=dev_index) for t in trees]
    stream = torch.cuda.current_stream(dev)
    lo, hi = (Np_total * rank) // world, (Np_total * (rank + 1)) // world
    # two buffer slots: the all-gather of step i (asynchronous collective on RCCL's stream) overlaps the work of step
    # i+1; a slot is written again only after its previous gather has been waited for
    slots = [sp._buffers(Np_total, 0), sp._buffers(Np_total, 1)]

    gather_wait = [0.0]  # host seconds this rank spent waiting for all-gathers (N > 1)

    def wait_gather(bufs):
        if bufs["pending"] is not None:
            tw = time.perf_counter()
            bufs["pending"].wait()
            gather_wait[0] += time.perf_counter() - tw
            bufs["pending"] = None

    def one_call(i):
        """ONE prodAppxMSGibbsS-equivalent call: tiles packed on the GPU from the resident densities + conditional
        tables + sampling of this rank's chains; then (N > 1) the single all-gather of [pGM | labels]."""
        bufs = slots[i & 1]
        wait_gather(bufs)
        if hi > lo:
            kdehip.prodAppxMSGibbsS_device(dd, bufs["pts"], bufs["ind"], Np=hi - lo, Niter=Niter, seed=seed,
                                           sample_offset=i * Np_total + lo, precision=prec, stream=stream.cuda_stream)
        if use_dist:
            bufs["pending"] = sp.gather(bufs, async_op=True)

    def drain():
        for bufs in slots:  # every product is complete (gathered) before the clock stops
            wait_gather(bufs)
        torch.cuda.synchronize()

    def timed_pass(first):
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; seconds of the K steps."""
        for i in range(args.warmup):
            one_call(first + i)
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for i in range(args.steps):
            one_call(first + args.warmup + i)
        drain()
        if use_dist:
            dist.barrier()
        return time.perf_counter() - ta

    no_spin_elapsed = None  # (measured AFTER the headline: see below)

    # The device idled while the host built the inputs, and its power manager takes ~20 ms of continuous work to bring the
    # clock back up (scripts/step_transient.py: 700 -> 590 us per config-3 call over the first 35 calls after 2 s of idle):
    # a SHORT run (--steps 20 --warmup 5 = 15 ms) would sit on that ramp.  So the device is first kept busy for 40 ms with
    # launches of ONE resident plan -- not warm-up steps of the measured call path, and reported as `spin_up`.
    spin_up = {"launches": 0, "ms": 0.0}
    if not args.no_spin_up:
        ts = time.perf_counter()
        while hi > lo and time.perf_counter() - ts < args.spin_up_ms * 1e-3:
            for _ in range(4):
                plan.sample_philox_device(hi - lo, Niter, seed, lo, True, slots[0]["pts"], slots[0]["ind"], None, stream.cuda_stream)
            torch.cuda.synchronize()
            spin_up["launches"] += 4
        # ... and the first pass of a PROCESS through the measured entry is slower for its first ~40 calls whatever the
        # clock does (525 -> 490 us per call; a pass of a DIFFERENT, small product beforehand removes it just as well: the HIP
        # runtime's event / signal / stream pools warming up, not this product's data -- scripts/first_pass_transient.py,
        # profiles/r06l_first_pass_transient.txt).  Rounds 3-4 measured inside that transient, round 5 behind an extra pass by
        # accident; now it is run here, untimed and counted: `spin_up.calls`.
        # (every rank runs them, also one without chains of its own: with N > 1 a call ends in the collective)
        for i in range(args.spin_up_calls):
            one_call(16 * (args.warmup + args.steps) + i)
        drain()
        spin_up["calls"] = args.spin_up_calls
        spin_up["ms"] = (time.perf_counter() - ts) * 1e3
    spin_up["what"] = ("untimed, before the W warm-up steps: launches of one resident plan for spin_up_ms (the device's clock is "
                       "back at its sustained value), then `calls` calls of the measured path (the first ~40 calls of a process "
                       "through the entry are ~5 % slower: HIP runtime pools); --no-spin-up: without; "
                       "ms_per_step_no_spin_up = the same W + K steps cold")

    elapsed = timed_pass(0)
    gather_wait_s = gather_wait[0]
    # The kernel's duration in THIS loop: the same calls once more, now with the sampling launch of every call bracketed
    # by timing events on its stream (inside the library: the launch is not visible from here).  A pass of its own because
    # the timestamps cost 2 % of a step (0.648 -> 0.661 ms at config 3) -- they are kept out of `value`.
    import ctypes as _C
    kdehip._clib.kdehip_profile_sampler(1)
    for i in range(args.steps):
        one_call(args.warmup + args.steps + i)
    drain()
    _ms, _n = _C.c_double(0.0), _C.c_int64(0)
    kdehip._clib.kdehip_profile_sampler_read(dev_index, _C.c_void_p(stream.cuda_stream), _C.byref(_ms), _C.byref(_n))
    kdehip._clib.kdehip_profile_sampler(0)
    kern_region_ms = _ms.value / _n.value if _n.value > 0 else None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- outside the timed region: the sampling kernel alone, on ONE resident plan (HIP events on the launch stream) ----
    nk = max(5, min(args.steps, 100))
    for i in range(3):
        plan.sample_philox_device(hi - lo, Niter, seed, lo, True, slots[0]["pts"], slots[0]["ind"], None, stream.cuda_stream)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nk)]
    tr0 = time.perf_counter()
    for i in range(nk):
        ev[i][0].record(stream)
        plan.sample_philox_device(hi - lo, Niter, seed, (i + 1) * Np_total + lo, True, slots[0]["pts"], slots[0]["ind"], None,
                                  stream.cuda_stream)
        ev[i][1].record(stream)
    torch.cuda.synchronize()
    resident_ms = (time.perf_counter() - tr0) / nk * 1e3
    kern_resident_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # the kernel's duration: the launches of the timed region itself (next to them the GPU prepares the following call);
    # the launches of one resident plan, alone on the device, beside it
    kern_ms = kern_region_ms if kern_region_ms is not None else kern_resident_ms
    # For comparability between rounds: the same W + K steps once WITHOUT the spin-up, on the clock ramp
    # of a device that has idled -> ms_per_step_no_spin_up.  ADVICE round 5: this pass runs AFTER the headline (whose warm
    # state it must not change), behind 0.5 s of idle, on sample offsets of its own; `passes` in the line records the order.
    if not args.no_spin_up and args.steps <= 100:
        time.sleep(0.5)
        no_spin_elapsed = timed_pass(4 * (args.warmup + args.steps))
        if use_dist:
            tt = torch.tensor([no_spin_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            no_spin_elapsed = float(tt.item())
    per_rank = None
    if use_dist:
        # what tells a slow kernel from a straggling rank or link when this first runs on N real GPUs: every rank's kernel
        # time (min / max over ranks) and the host time its calls spent waiting for all-gathers
        mine = torch.tensor([kern_ms, gather_wait_s * 1e3 / max(args.steps, 1)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ks = [float(a[0].item()) for a in allr]
        gw = [float(a[1].item()) for a in allr]
        per_rank = {"kernel_ms_min": min(ks), "kernel_ms_max": max(ks), "kernel_ms": ks,
                    "gather_wait_ms_per_step_max": max(gw), "gather_wait_ms_per_step": gw}
        kern_ms = max(ks)
    try:
        screen = plan.screen_stats()
    except Exception:  # noqa: BLE001  (a diagnostic)
        screen = None

    if rank == 0:
        import shutil
        E = plan.evals_per_sample(Niter)
        B = plan.bytes_per_eval
        flops_per_eval = 6 * D + 4
        alg_bytes = float(hi - lo) * E * B
        alg_flops = float(hi - lo) * E * flops_per_eval
        achieved_tf = alg_flops / (kern_ms * 1e-3) / 1e12
        peak_tf = VALU_PEAK_TFLOPS[prec]
        prof = load_traffic(args.config)
        # The physical floor of the vector pipe: the committed instruction mix of this workload's kernel
        # (profiles/*_valu_mix.json) priced with the issue costs measured by scripts/micro/valu_rates.hip, at the
        # clock the chip held in that profile; frac = floor / the kernel time measured live above.
        vf = prof.get("valu_floor")
        valu_floor = None
        if vf:
            floor_ms = vf["cycles_per_chain"] * float(hi - lo) / vf["simds"] / (vf["clock_ghz"] * 1e6)
            valu_floor = {"ms": floor_ms, "frac": floor_ms / kern_ms, "clock_ghz": vf["clock_ghz"],
                          "valu_cycles_per_chain": vf["cycles_per_chain"], "mix_per_chain": vf["mix_per_chain"],
                          "issue_cost_cycles": vf["issue_cost_cycles"], "source": vf["source"]}
        # Occupancy of the vector pipe by the counters of the same committed profile: SQ_ACTIVE_INST_VALU of one wavefront x
        # the wavefronts a SIMD holds (chains per workgroup / 4, one workgroup per CU: the LDS pool) / SQ_WAVE_CYCLES.
        vb = prof.get("valu_busy")
        valu_busy = None
        if vb and vb.get("chains_profiled") == hi - lo:  # (the counters of another chain count are another occupancy)
            geo = plan.launch_geometry(hi - lo)
            wps = geo.get("waves", 0) / 4.0 if isinstance(geo, dict) else 0.0
            if wps > 0:
                valu_busy = {"frac": vb["active_inst_valu_quads_per_chain"] * wps / vb["wave_quads_per_chain"],
                             "wavefronts_per_simd": wps, "source": vb["source"],
                             "what": "SQ_ACTIVE_INST_VALU x wavefronts per SIMD / SQ_WAVE_CYCLES (replayed from the committed profile)"}
        out = {
            "metric": "gibbs_product_samples_per_sec",
            "value": Np_total * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "spin_up": spin_up,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_no_spin_up": (no_spin_elapsed / args.steps * 1e3) if no_spin_elapsed is not None else None,
            "passes": "spin-up (resident launches, then untimed calls), W + K (value), K with kernel timestamps, resident plan, 0.5 s idle, W + K without spin-up",
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f64" if prec == 64 else "f32",
            "data": "synthetic",
            "julia": "present" if shutil.which("julia") else "absent",
            "value_is": "one complete prodAppxMSGibbsS-equivalent call per step, input densities resident in HBM: tiles packed "
                        "on the GPU + conditional tables + sampling (+ all-gather); outputs left in HBM",
            "config": {"workload": workload, "ndims": D, "ndens": M, "npts": N, "nout_per_gpu": Nout,
                       "nout_total": Np_total, "niter": Niter, "rng": "device philox4x32-10",
                       "evals_per_sample": E, "bytes_per_eval": B, "flops_per_eval": flops_per_eval,
                       "parallelism": f"chains sharded over {world} GPU(s), 1 all-gather"},
            # The binding resource: vector-ALU issue.  achieved = algorithmic flops of the kernel evaluations (exp, rsqrt,
            # scans and selection not counted) per second of kernel time; peak = the MI355X vector peak of the dtype.
            "roofline": {"bound": "valu", "achieved": achieved_tf, "peak": peak_tf, "unit": "TFLOP/s",
                         "frac": achieved_tf / peak_tf, "traffic": prof.get("hbm_bytes_per_launch"),
                         "kernel": plan.kernel_name(hi - lo), "kernel_ms": kern_ms,
                         "kernel_ms_is": ("average over the sampling launches of a repeat of the timed loop (HIP events on the launch "
                                          "stream inside the library, kdehip_profile_sampler; the following call is being "
                                          "prepared beside them)" if kern_region_ms is not None else
                                          "average over launches of one resident plan"),
                         "kernel_ms_resident_plan": kern_resident_ms,
                         "algorithmic_flops_per_launch": alg_flops,
                         # issue-slot view from the committed PMC profile of this workload (profiles/): instructions the
                         # wavefronts issued per chain against the 4-cycle issue slots of their lifetime
                         "issue": prof.get("issue"),
                         "valu_floor": valu_floor,
                         "valu_busy": valu_busy,
                         # traffic / issue / valu_floor above are REPLAYED from a committed rocprofv3 --pmc profile of this
                         # workload (counters cannot be collected inside a timed run); kernel_ms and frac are live
                         "profile_replayed_from": ({"file": prof.get("_file"), "tag": prof.get("tag"), "commit": prof.get("commit"),
                                                    "date": prof.get("date")} if prof else None),
                         # `frac` counts the reference's fp64-equivalent kernel evaluations (E per sample x (6 D + 4) flops)
                         # against the fp64 vector peak WHATEVER arithmetic certified them: on the screened levels of an fp64
                         # plan they execute as packed fp32 (csrc/screen_device.hpp), so frac is a normalised throughput, not
                         # the utilisation of a pipe.  The physical figures are executed_mix_frac (= valu_floor.frac: the
                         # EXECUTED instruction mix priced at measured issue costs / kernel time) and valu_busy.
                         "executed_mix_frac": valu_floor["frac"] if valu_floor else None,
                         "note": ("working set is LDS/L2 resident (HBM traffic ~0.1 % of peak). frac = fp64-EQUIVALENT evaluations "
                                  "per second against the fp64 vector peak; part of them execute as packed fp32 (the screen), so "
                                  "frac is a normalised throughput -- executed_mix_frac / valu_busy are the pipe's physical "
                                  "utilisation; the rest of the time is the dependent chain of a step (levels 1-8) and barriers")},
            # SURVEY.md 8(d)'s figure, kept for continuity: algorithmic bytes / kernel time against 8 TB/s.  NOT a roofline
            # (the bytes never come from HBM; the ratio exceeds 1).
            "normalised_hbm": {"achieved_GBps": alg_bytes / (kern_ms * 1e-3) / 1e9, "peak_GBps": HBM_PEAK_GBS,
                               "ratio": alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_launch": alg_bytes},
            "kernel_samples_per_sec": (hi - lo) / (kern_ms * 1e-3),
            "resident_plan": {"ms_per_launch": resident_ms, "samples_per_sec": (hi - lo) / (resident_ms * 1e-3),
                              "what": "repeated sampling of ONE packed plan: no re-layout, no table build (round-1/2 headline)"},
            "fast_math_path": plan.fast_math_path,
            # fp32 screening of the deep levels with fp64 certification (csrc/screen_device.hpp): levels screened, label
            # draws that went through the screen on the resident plan's launches above, draws repeated in fp64
            "screen": screen,
            "per_rank": per_rank,
        }
        if world == 1:
            out["call_inclusive"] = call_inclusive(kdehip, trees, plan, D, M, Nout, Niter, seed, prec)
            out["cold_start"] = cold_start(args.config)
        if world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baseline_and_parity(kdehip, plan, pts_all, bw_all, D, M, N, Nout, Niter, seed, args.warmup))
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)  # whatever C libraries buffered for fd 1 goes to stderr now
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)


def inproc_multi(args):
    """bench.py --inproc-gpus G: the multi-device route behind the C ABI, one process.  Same step as the default mode
    (every device draws its share of the chains, afterwards EVERY device holds the complete [pGM | indices]), but the
    gather is the kernel epilogue's peer stores over xGMI instead of an RCCL all-gather, and there is one host thread."""
    import torch
    import kdehip

    G = args.inproc_gpus
    D, M, N, Nout, Niter, prec, cid = CONFIGS[args.config]
    if args.nout > 0:
        Nout = args.nout
    Np_total = (TOTAL_NOUT[args.config] if args.nout <= 0 else args.nout) if args.strong else Nout * G
    alias = os.environ.get("KDEHIP_ALIAS_DEVICES") == "1"
    nvis = torch.cuda.device_count()
    if not alias and nvis < G:
        raise SystemExit(f"--inproc-gpus {G} but {nvis} visible device(s) (KDEHIP_ALIAS_DEVICES=1 wraps logical devices around them)")
    pts_all, bw_all = synth_inputs(kdehip, D, M, N, cid)
    trees = [kdehip.kde(p, b) for p, b in zip(pts_all, bw_all)]
    mp = kdehip.MultiProductPlan(trees, precision=prec, first_device=0, ngpus=G)
    devs = [torch.device("cuda", g % nvis) for g in range(G)]
    Ps = [torch.zeros(D * Np_total, dtype=torch.float64, device=d) for d in devs]
    Is = [torch.zeros(M * Np_total, dtype=torch.int64, device=d) for d in devs]
    sts = [torch.cuda.Stream(device=d) for d in devs]
    handles = [s.cuda_stream for s in sts]
    seed = 20260101

    def sync_all():
        for s in sts:
            s.synchronize()
    for i in range(args.warmup):
        mp.sample_philox_device(Np_total, Niter, seed, i * Np_total, True, Ps, Is, handles)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        mp.sample_philox_device(Np_total, Niter, seed, (args.warmup + i) * Np_total, True, Ps, Is, handles)
    sync_all()
    elapsed = time.perf_counter() - t0
    # per device: the duration of its sampling launch and when its slice had arrived everywhere, relative to the first device
    # to get there (a pass of its own: the timing events stay out of `value`) -- what the first run on N real GPUs needs to
    # tell a slow kernel from a straggling device or link
    kdehip._clib.kdehip_profile_sampler(1)
    kms, dms = [], []
    for i in range(min(args.steps, 20)):
        mp.sample_philox_device(Np_total, Niter, seed, (args.warmup + args.steps + i) * Np_total, True, Ps, Is, handles)
        k, d = mp.timing()
        kms.append(k)
        dms.append(d)
    sync_all()
    kdehip._clib.kdehip_profile_sampler(0)
    kms, dms = np.array(kms), np.array(dms)
    # every device must hold the same complete result
    ref_p, ref_i = Ps[0].cpu(), Is[0].cpu()
    same = all(torch.equal(ref_p, P.cpu()) and torch.equal(ref_i, I.cpu()) for P, I in zip(Ps[1:], Is[1:]))
    out = {
        "metric": "gibbs_product_samples_per_sec", "value": Np_total * args.steps / elapsed, "unit": "samples/s",
        "n_gpus": G, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
        "dtype": "f64" if prec == 64 else "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {D}-D, {M} densities x {N} pts, Nout={Np_total} total over {G} device(s), Niter={Niter}, fp{prec}",
                   "parallelism": f"ONE process, {G} device(s) through kdehip_product_multi_*; all-gather = kernel-epilogue peer stores",
                   "aliased_devices": alias, "copy_engine_transfers_per_device_per_product": mp.transfers_per_product},
        "all_devices_hold_the_same_result": bool(same),
        "per_device": {"kernel_ms": [float(x) for x in kms.mean(axis=0)], "done_after_first_ms": [float(x) for x in dms.mean(axis=0)],
                       "kernel_ms_skew": float((kms.max(axis=1) - kms.min(axis=1)).mean()),
                       "done_skew_ms": float((dms.max(axis=1) - dms.min(axis=1)).mean()),
                       "what": "mean over products of a repeat pass with timing events around every device's sampling launch "
                               "(kdehip_product_multi_timing): kernel duration per device; host time at which a device's slice "
                               "had arrived on every device, relative to the first"},
    }
    mp.close()
    print(json.dumps(out), flush=True)


def batch_mode(args):
    """bench.py --config c2 --batch 64: a step = B independent products of the configuration's shape (different densities,
    different seeds) in ONE kdehip_prod_philox_batch call from HBM-resident densities; `back_to_back` = the same B products
    as B kdehip_prod_philox_device calls on one stream.  roofline = algorithmic flops of all B products over the duration of
    the call's device work (HIP events on the stream around the call: descriptor upload + tile gather + the sampling launch)."""
    import torch
    import kdehip
    D, M, N, Nout, Niter, prec, cid = CONFIGS[args.config]
    if args.nout > 0:
        Nout = args.nout
    B = args.batch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream(dev)
    dds, outs = [], []
    for b in range(B):
        pts_all, bw_all = synth_inputs(kdehip, D, M, N, 1000 * cid + b)
        trees = [kdehip.kde(p, w) for p, w in zip(pts_all, bw_all)]
        dds.append([kdehip.DeviceDensity(t) for t in trees])
        outs.append((torch.zeros(D * Nout, dtype=torch.float64, device=dev), torch.zeros(M * Nout, dtype=torch.int64, device=dev)))
        if b == 0:
            plan = kdehip.ProductPlan(trees, precision=prec)
            E, Bytes = plan.evals_per_sample(Niter), plan.bytes_per_eval
            kname = plan.kernel_name(Nout)
            plan.close()
    seed = 20260101

    pb = kdehip.ProductBatch([dict(trees=dds[b], d_points=outs[b][0], d_indices=outs[b][1], Np=Nout, Niter=Niter, seed=seed + b)
                              for b in range(B)], precision=prec)   # (the argument block is marshalled once)

    def batched(i):
        pb.enqueue(stream=stream.cuda_stream, sample_offset=i * Nout)

    def one_by_one(i):
        for b in range(B):
            kdehip.prodAppxMSGibbsS_device(dds[b], outs[b][0], outs[b][1], Np=Nout, Niter=Niter, seed=seed + b,
                                           sample_offset=i * Nout, precision=prec, stream=stream.cuda_stream)

    def timed(f, steps, warmup):
        for i in range(warmup):
            f(i)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            ev[i][0].record(stream)
            f(warmup + i)
            ev[i][1].record(stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # parity of the batched call with the single calls, product by product (bit for bit)
    batched(0)
    torch.cuda.synchronize()
    got = [(p.clone(), q.clone()) for p, q in outs]
    one_by_one(0)
    torch.cuda.synchronize()
    identical = all(torch.equal(g[0], o[0]) and torch.equal(g[1], o[1]) for g, o in zip(got, outs))
    ms_b, dev_b = timed(batched, args.steps, args.warmup)
    ms_s, dev_s = timed(one_by_one, max(3, args.steps // 4), max(1, args.warmup // 2))
    flops = float(B) * Nout * E * (6 * D + 4)
    peak_tf = VALU_PEAK_TFLOPS[prec]
    ach = flops / (dev_b * 1e-3) / 1e12
    out = {
        "metric": "gibbs_product_samples_per_sec", "value": B * Nout / (ms_b * 1e-3), "unit": "samples/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_b, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64" if prec == 64 else "f32", "data": "synthetic",
        "value_is": f"{B} independent prodAppxMSGibbsS-equivalent calls per step in ONE kdehip_prod_philox_batch call, densities "
                    "resident in HBM, outputs left in HBM",
        "config": {"workload": f"{args.config} x {B}: {B} products of {D}-D, {M} densities x {N} pts, Nout={Nout} each, Niter={Niter}, fp{prec}",
                   "batch": B, "evals_per_sample": E, "bytes_per_eval": Bytes},
        "roofline": {"bound": "valu", "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach / peak_tf, "traffic": None,
                     "kernel": kname + " (BATCH instantiation, 16 chains per workgroup)", "kernel_ms": dev_b,
                     "kernel_ms_is": "HIP events on the stream around the whole batched call: descriptor upload + tile gather + "
                                     "ONE sampling launch",
                     "algorithmic_flops_per_launch": flops},
        "back_to_back": {"ms_per_step": ms_s, "device_ms": dev_s, "samples_per_sec": B * Nout / (ms_s * 1e-3),
                         "what": f"the same {B} products as {B} kdehip_prod_philox_device calls on one stream",
                         "batched_speedup": ms_s / ms_b},
        "batched_equals_single_calls_bit_for_bit": bool(identical),
    }
    for dd in dds:
        for d in dd:
            d.close()
    print(json.dumps(out), flush=True)


def frow_mode(args):
    """SURVEY.md 8(f) rows 1-3 to the same measurement bar as the product: a step = one blocking library call with host
    buffers in and out (what the reference's caller has); kernel time = HIP events around the launches INSIDE the library
    (kdehip_profile_phase_read: the entries run on the calling thread's own stream); roofline against the fp64 vector peak
    with the REFERENCE's flop count per kernel value (evalDirect, src/DualTree01.jl:130-162: per dimension subtract,
    square, divide, add; then exp, scale, weight = 4 D + 3); cpu_baseline = the oracle's restatement of the same function
    on one host core (a bounded sample for evaluate)."""
    import ctypes as C
    import torch
    import kdehip
    from oracle import oracle
    D, M, N, Nout, Niter, prec, cid = CONFIGS[args.config]
    if args.nout > 0:
        Nout = args.nout
    torch.cuda.set_device(0)
    steps, warmup = min(args.steps, 100), min(args.warmup, 10)
    clib = kdehip._clib
    peak_tf = VALU_PEAK_TFLOPS[64]

    def phase(which):
        ms, n = C.c_double(0.0), C.c_int64(0)
        clib.kdehip_profile_phase_read(which, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def timed(f):
        for _ in range(warmup):
            f()
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            f()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.mean(ts)), float(np.median(ts))

    def timed_kernel(f, which):
        clib.kdehip_profile_sampler(1)
        phase(which)
        for _ in range(steps):
            f()
        ms, n = phase(which)
        clib.kdehip_profile_sampler(0)
        return ms / steps, n / steps

    base = {"n_gpus": 1, "steps": steps, "warmup": warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic"}
    if args.frow == "loocv":
        # the matrix kde!(pGM) sees after a product of --config: D x Nout points of a mixture (SURVEY 8d's generator)
        pts = synth_inputs(kdehip, D, 1, Nout, 100 * cid + 7)[0][0]
        bw, ne = kdehip.auto_bandwidth(pts, return_evals=True)
        obw, one = oracle.auto_bandwidth(pts) if Nout <= 4096 else (None, None)
        mean_ms, med_ms = timed(lambda: kdehip.auto_bandwidth(pts))
        k_ms, k_n = timed_kernel(lambda: kdehip.auto_bandwidth(pts), 0)
        values = float(ne) * Nout * Nout                    # the reference evaluates every marginal at N x N pairs per likelihood
        flops = values * (4 * 1 + 3)
        ach = flops / (k_ms * 1e-3) / 1e12
        cpu = None
        if not args.no_cpu_baseline and Nout <= 4096:
            t0 = time.perf_counter()
            oracle.auto_bandwidth(pts)
            tc = time.perf_counter() - t0
            cpu = {"value": 1.0 / tc, "unit": "searches/s", "cores": 1, "kind": "port", "ms": tc * 1e3,
                   "sample": f"one complete search of the same {D} x {Nout} matrix by oracle/kde_oracle.c (okde_auto_bandwidth)"}
        out = dict(base, metric="loocv_bandwidth_searches_per_sec", value=1e3 / mean_ms, unit="searches/s", ms_per_step=mean_ms,
                   ms_per_step_median=med_ms,
                   value_is="one blocking kdehip_auto_bandwidth call (kde!(points)'s per-dimension golden-section LOOCV search, "
                            "src/KDE01.jl:3-27, src/CrossValidation.jl:15-120), host matrix in, D bandwidths out",
                   config={"workload": f"loocv: {D} x {Nout} points (the output shape of a {args.config} product)", "ndims": D,
                           "npts": Nout, "likelihood_evaluations": int(ne), "pair_values_per_evaluation": Nout * Nout,
                           "flops_per_value": 7},
                   roofline={"bound": "valu", "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach / peak_tf,
                             "traffic": None, "kernel": "loocv_prep_kernel + loo_round_*_kernel rounds + loo_finalize_kernel",
                             "kernel_ms": k_ms, "phases_per_search": k_n,
                             "kernel_ms_is": "HIP events on the search's stream around preparation + every batch of rounds, inside the "
                                             "library (kdehip_profile_phase_read(0)); includes the ~3.5 us gaps between the dependent launches",
                             "algorithmic_flops_per_launch": flops,
                             "note": "values counted as the reference computes them (N x N per evaluation); the kernels form each unordered "
                                     "pair ONCE (half the exps). A round of a search is latency: ~12 us fixed of ~18 us"},
                   parity={"bandwidths_equal_oracle_1e-9": (bool(np.allclose(bw, obw, rtol=1e-9, atol=0)) if obw is not None else None),
                           "evaluation_counts_equal": (int(ne) == int(one) if one is not None else None)},
                   cpu_baseline=cpu)
    elif args.frow == "evaluate":
        pts, bws = synth_inputs(kdehip, D, 1, N, 100 * cid + 8)
        bd = kdehip.kde(pts[0], bws[0])
        Nq = 65536
        pos = synth_inputs(kdehip, D, 1, Nq, 100 * cid + 9)[0][0]
        p = kdehip.evaluateDualTree(bd, pos)
        mean_ms, med_ms = timed(lambda: kdehip.evaluateDualTree(bd, pos))
        k_ms, _ = timed_kernel(lambda: kdehip.evaluateDualTree(bd, pos), 1)
        values = float(N) * Nq
        flops = values * (4 * D + 3)
        ach = flops / (k_ms * 1e-3) / 1e12
        cpu, par = None, None
        nqc = min(Nq, max(64, int(2.0e8 / N)))             # ~2e8 kernel values: a few seconds on one core
        od = oracle.OracleDensity(pts[0], bws[0])
        t0 = time.perf_counter()
        po = oracle.eval_direct(od, pos[:, :nqc])
        tc = time.perf_counter() - t0
        par = {"max_rel_err_vs_oracle": float(np.max(np.abs(p[:nqc] - po) / np.maximum(np.abs(po), 1e-300))), "queries_compared": nqc}
        if not args.no_cpu_baseline:
            cpu = {"value": N * nqc / tc, "unit": "kernel values/s", "cores": 1, "kind": "port",
                   "sample": f"okde_eval_direct of the same density at the first {nqc} of the {Nq} positions ({tc:.2f} s)"}
        out = dict(base, metric="kde_evaluation_kernel_values_per_sec", value=values / (mean_ms * 1e-3), unit="kernel values/s",
                   ms_per_step=mean_ms, ms_per_step_median=med_ms,
                   value_is="one blocking kdehip_evaluate call (evaluateDualTree(bd, pos) with FORCE_EVAL_DIRECT, "
                            "src/DualTree01.jl:130-162,303-346,370-446), host buffers in and out (PCIe both ways inside)",
                   config={"workload": f"evaluate: {D}-D density of {N} points at {Nq} positions", "ndims": D, "npts": N, "nq": Nq,
                           "flops_per_value": 4 * D + 3},
                   roofline={"bound": "valu", "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach / peak_tf, "traffic": None,
                             "kernel": f"eval_partial_kernel<{D}> + eval_finish_kernel", "kernel_ms": k_ms,
                             "kernel_ms_is": "HIP events around the two launches inside the library (kdehip_profile_phase_read(1))",
                             "kernel_values_per_sec": values / (k_ms * 1e-3), "algorithmic_flops_per_launch": flops},
                   parity=par, cpu_baseline=cpu)
    else:
        pts, bws = synth_inputs(kdehip, D, 1, Nout, 100 * cid + 7)
        x, ks = pts[0], bws[0]
        bd = kdehip.kde(x, ks)
        L = int(math.floor(math.log(Nout) / math.log(2.0))) + 1
        mean_ms, med_ms = timed(lambda: kdehip.kde(x, ks))
        gpu_ok = bool(clib.kdehip_make_density_device_supported(D, Nout))
        g1 = g16 = gk1 = gk16 = None
        same = None
        if gpu_ok:
            g = kdehip.kde_batch([(x, ks)])[0]
            same = all(np.array_equal(getattr(g.bt, a), getattr(bd.bt, a)) for a in
                       ("centers", "ranges", "weights", "left_child", "right_child", "lowest_leaf", "highest_leaf", "permutation")) and \
                all(np.array_equal(getattr(g, a), getattr(bd, a)) for a in ("means", "bandwidth"))
            g1, _ = timed(lambda: kdehip.kde_batch([(x, ks)]))
            gk1, _ = timed_kernel(lambda: kdehip.kde_batch([(x, ks)]), 2)
            many = [(synth_inputs(kdehip, D, 1, Nout, 100 * cid + 20 + b)[0][0], ks) for b in range(16)]
            g16, _ = timed(lambda: kdehip.kde_batch(many))
            gk16, _ = timed_kernel(lambda: kdehip.kde_batch(many), 2)
        cpu = None
        if not args.no_cpu_baseline:
            t0 = time.perf_counter()
            for _ in range(5):
                oracle.OracleDensity(x, ks)
            tc = (time.perf_counter() - t0) / 5
            cpu = {"value": 1.0 / tc, "unit": "trees/s", "cores": 1, "kind": "port", "ms": tc * 1e3,
                   "sample": "5 builds of the same density by oracle/kde_oracle.c (okde_build: the reference's sequential quick-select)"}
        # every level's quick-select passes read and swap whole points: ~2 passes x (D + 2) doubles x N per level
        alg_bytes = 2.0 * L * Nout * (D + 2) * 8
        best_ms = med_ms
        # (value from the MEDIAN build: this process hosts torch, whose OpenMP workers spin after its parallel regions -- one
        # build in ~50 that lands in that window takes ~80 ms, profiles/r06f_host_tree_in_torch.txt; the mean is kept beside it)
        out = dict(base, metric="kde_tree_builds_per_sec", value=1e3 / med_ms, unit="trees/s", ms_per_step=med_ms, ms_per_step_mean=mean_ms,
                   ms_per_step_median=med_ms,
                   value_is="one kdehip_make_density call (kde!(points, ks): buildTree! + calcStats, src/BallTree01.jl:223-463, "
                            "src/BallTreeDensity01.jl:141-231) on the library's pooled HOST builder -- the builder every `*` path uses",
                   config={"workload": f"tree: {D} x {Nout} points", "ndims": D, "npts": Nout, "levels": L,
                           "host_threads": os.environ.get("KDEHIP_HOST_THREADS", "default (<= 15 workers)")},
                   roofline={"bound": "hbm", "achieved": alg_bytes / (best_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg_bytes / (best_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "kernel": "host: balltree.cpp (pooled); GPU: tree_build_kernel (one workgroup per density)",
                             "kernel_ms": gk1,
                             "note": "a tree build is a chain of L dependent selections per subtree: latency, not bandwidth -- which is why "
                                     "ONE density builds faster on the host pool than in one workgroup of the GPU; the GPU builder pays "
                                     "when many densities are built at once (gpu_builder.batch16)"},
                   gpu_builder=({"supported": True, "one_density_call_ms": g1, "one_density_kernel_ms": gk1,
                                 "batch16_call_ms": g16, "batch16_kernel_ms": gk16, "batch16_ms_per_density": g16 / 16.0,
                                 "arrays_identical_to_host_builder": bool(same)} if gpu_ok else {"supported": False}),
                   cpu_baseline=cpu)
    print(json.dumps(out), flush=True)


def mul_batch_mode(args):
    """bench.py --config c2 --batch 64 --mul: a step = B complete `*` of the configuration's input shape -- the reference's
    serving pattern (src/MSGibbs01.jl:707-726: Np = round(mean Npts), Niter = 5, then kde!(pGM): LOOCV bandwidth,
    src/CrossValidation.jl:44-120, + ball tree, src/BallTree01.jl:415-434) -- in ONE blocking kdehip_mul_device_batch call
    on HBM-resident densities, results left in HBM as B new resident densities; `back_to_back` = the same B `*` as B blocking
    kdehip_mul_device calls.  Wall-clock per call (the calls are blocking: host and device work are both inside)."""
    import torch
    import kdehip
    D, M, N, _, _, prec, cid = CONFIGS[args.config]
    B = args.batch
    torch.cuda.set_device(0)
    dds = []
    for b in range(B):
        pts_all, bw_all = synth_inputs(kdehip, D, M, N, 1000 * cid + b)
        dds.append([kdehip.DeviceDensity(kdehip.kde(p, w)) for p, w in zip(pts_all, bw_all)])
    seeds = [20260101 + b for b in range(B)]

    def batched(i):
        outs = kdehip.mul_device_batch(dds, seeds=[s + 1000 * i for s in seeds])
        for o in outs:
            o.close()

    def one_by_one(i):
        for b in range(B):
            kdehip.mul_device(dds[b], seed=seeds[b] + 1000 * i).close()

    def timed(f, steps, warmup):
        for i in range(warmup):
            f(i)
        torch.cuda.synchronize()
        ts = []
        for i in range(steps):
            t0 = time.perf_counter()
            f(warmup + i)
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.mean(ts)), float(np.median(ts)), float(np.min(ts))
    # parity: every array of every density, bandwidths and evaluation counts, against the single calls
    outs = kdehip.mul_device_batch(dds, seeds=seeds)
    identical = True
    for b in range(B):
        with kdehip.mul_device(dds[b], seed=seeds[b]) as ref:
            x, y = outs[b].download(), ref.download()
            identical &= bool(np.array_equal(outs[b].bw, ref.bw)) and outs[b].nevals == ref.nevals
            for name in ("centers", "ranges", "weights", "left_child", "right_child", "lowest_leaf", "highest_leaf", "permutation"):
                identical &= bool(np.array_equal(getattr(x.bt, name), getattr(y.bt, name)))
            for name in ("means", "bandwidth", "bandwidthMin", "bandwidthMax"):
                identical &= bool(np.array_equal(getattr(x, name), getattr(y, name)))
    nevals = [o.nevals for o in outs]
    for o in outs:
        o.close()
    steps, warmup = min(args.steps, 100), min(args.warmup, 10)
    mean_b, med_b, min_b = timed(batched, steps, warmup)
    mean_s, med_s, min_s = timed(one_by_one, max(3, steps // 4), max(1, warmup // 2))
    out = {
        "metric": "star_operator_calls_per_sec", "value": B / (mean_b * 1e-3), "unit": "products/s", "n_gpus": 1,
        "steps": steps, "warmup": warmup, "ms_per_step": mean_b, "ms_per_step_median": med_b, "ms_per_step_min": min_b,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "value_is": f"{B} complete `*` (product, LOOCV bandwidth search, ball tree; results = {B} new HBM-resident densities) per "
                    "step in ONE blocking kdehip_mul_device_batch call; wall clock, host work included",
        "config": {"workload": f"{args.config} x {B} `*`: {B} products of {D}-D, {M} densities x {N} pts, Np={N}, Niter=5, then kde!(pGM)",
                   "batch": B, "loocv_evaluations_per_product": float(np.mean(nevals))},
        "back_to_back": {"ms_per_step": mean_s, "ms_per_step_median": med_s, "ms_per_step_min": min_s,
                         "what": f"the same {B} `*` as {B} blocking kdehip_mul_device calls", "batched_speedup": mean_s / mean_b},
        "batched_equals_single_calls_bit_for_bit": bool(identical),
    }
    for dd in dds:
        for d in dd:
            d.close()
    print(json.dumps(out), flush=True)


def call_inclusive(kdehip, trees, plan, D, M, Nout, Niter, seed, prec):
    """What a drop-in caller sees (wall time of one prodAppxMSGibbsS-equivalent call, host buffers in and
    out): median of 15 calls each of (a) prodAppxMSGibbsS with the device Philox stream = pack + upload + [table build]
    + kernel + copy back (kdehip_prod_philox), (b) the gibbs1 drop-in with caller-supplied randU/randN, which
    additionally uploads the streams over PCIe (kdehip_gibbs1), (c) a run of the resident plan + copy back."""
    K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    randU, randN = kdehip.philox_streams(seed, 0, Nout, K, R)

    def med(f, n=15, busy_s=0.040):
        # (the calls are blocking, so the device idles between them and before the first: the same calls are first repeated
        # for 40 ms, which brings the device's clock to what a caller that issues products back to back sees -- `spin_up`
        # in main(); `from_idle_ms` below is the other end: the first calls after half a second of idle)
        f()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < busy_s:
            f()
        ts = []
        for _ in range(n):
            t = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t)
        return float(np.median(ts)) * 1e3

    def from_idle(f, n=15):
        time.sleep(0.5)
        ts = []
        for _ in range(n):
            t = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t)
        return float(np.median(ts)) * 1e3
    prod = lambda: kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Nout, seed=seed, precision=prec)  # noqa: E731
    t_prod = med(prod)
    t_prod_idle = from_idle(prod)
    t_g1 = med(lambda: kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Nout, randU=randU, randN=randN)) \
        if prec == 64 else None
    t_res = med(lambda: plan.sample(Nout, Niter=Niter, seed=seed))
    dd = [kdehip.DeviceDensity(t) for t in trees]
    t_dev = med(lambda: kdehip.prodAppxMSGibbsS_resident(dd, Np=Nout, Niter=Niter, seed=seed, precision=prec))
    for d in dd:
        d.close()
    return {"ms": t_prod, "samples_per_sec": Nout / (t_prod * 1e-3),
            "what": "prodAppxMSGibbsS, device Philox: pack + H2D + kernel + D2H, host buffers in and out (median of 15 "
                    "blocking calls after 40 ms of the same calls)",
            "from_idle_ms": t_prod_idle,   # median of the first 15 calls after 0.5 s of idle: the clock ramp included
            "gibbs1_caller_streams_ms": t_g1, "randU_MB_over_pcie": randU.nbytes / 1e6,
            "densities_resident_host_outputs_ms": t_dev,   # kdehip_prod_philox_resident: GPU re-layout, one copy back
            "resident_plan_run_plus_d2h_ms": t_res}


_COLD_CHILD = r"""
import ctypes, json, os, sys, time
root, config = sys.argv[1], sys.argv[2]
t0 = time.perf_counter()
lib = ctypes.CDLL(os.path.join(root, "kerneldensityestimate.jl_amd", "libkdehip.so"))   # what a Julia `ccall` pays to bind
t_dl = time.perf_counter()
lib.kdehip_device_count.restype = ctypes.c_int
lib.kdehip_device_count()          # hipGetDeviceCount: the HIP runtime initialises itself (not this library's time)
t_init = time.perf_counter()
sys.path.insert(0, root)
import numpy as np
import kdehip
import bench
t_py = time.perf_counter()
D, M, N, Nout, Niter, prec, cid = bench.CONFIGS[config]
pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
t1 = time.perf_counter()
kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Nout, seed=1, precision=prec)   # code objects of this shape, first launches
t2 = time.perf_counter()
kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Nout, seed=2, precision=prec)
t3 = time.perf_counter()
so = os.path.getsize(kdehip.LIB_PATH)
ms = lambda a, b: (b - a) * 1e3
print(json.dumps({"dlopen_ms": ms(t0, t_dl), "hip_runtime_init_ms": ms(t_dl, t_init), "first_call_ms": ms(t1, t2),
                  "second_call_ms": ms(t2, t3), "library_ms": ms(t0, t_dl) + ms(t1, t2),
                  "cold_start_ms": ms(t0, t_init) + ms(t1, t2), "python_imports_ms": ms(t_init, t_py), "libkdehip_bytes": so}))
"""


def cold_start(config):
    """What a fresh process pays before its first product is back, in a child process without torch: `dlopen` of
    libkdehip.so; the HIP runtime's own initialisation (the first HIP call of any process: device enumeration -- 140-280 ms
    on the pool's boxes with or without this library, scripts/cold_pieces.py); the first `prodAppxMSGibbsS` one-shot call
    (unpacking and loading the code objects of this shape's kernels, first launches) against the second.  `library_ms` =
    dlopen + first call: the part that is this library's; `cold_start_ms` adds the runtime's initialisation.  For the
    benched configuration and for config 2 (a small product).  None if the child fails."""
    import subprocess
    out = {}
    for cfg in dict.fromkeys([config, "c2"]):
        try:
            r = subprocess.run([sys.executable, "-c", _COLD_CHILD, ROOT, cfg], capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out[cfg] = json.loads(line[-1]) if line else None
        except Exception:  # noqa: BLE001  (diagnostic extra: never fails the bench)
            out[cfg] = None
    first = out.get(config)
    if first:   # (flat keys of the benched configuration, as in earlier rounds' lines)
        out.update({k: first[k] for k in ("dlopen_ms", "hip_runtime_init_ms", "first_call_ms", "second_call_ms", "library_ms",
                                          "cold_start_ms", "libkdehip_bytes")})
    return out


def cpu_baseline_and_parity(kdehip, plan, pts_all, bw_all, D, M, N, Nout, Niter, seed, warmup):
    """Times the CPU oracle (a faithful single-thread port of gibbs1; all host cores via OpenMP over
    samples) on a bounded sample of the same workload and checks the GPU output against it."""
    from oracle import oracle
    otrees = [oracle.OracleDensity(p, b) for p, b in zip(pts_all, bw_all)]
    K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    cores = usable_cores()
    # bounded sample, ~50 core-seconds in total (config 3 costs ~3-6 ms per sample per core): the
    # all-cores run takes the first `nall` chains of the workload's Philox stream (more than one
    # GPU batch when the host has many cores), the single-core run the first `n1`.
    per_sample_evals = plan.evals_per_sample(Niter)
    nall = int(max(cores * 8, min(16384, 1.5e9 / per_sample_evals)))
    n1 = int(max(16, min(1024, 1.0e8 / per_sample_evals)))
    randU, randN = kdehip.philox_streams(seed, 0, nall, K, R)
    t = time.perf_counter()
    oracle.gibbs1(otrees, n1, Niter, randU[: n1 * K], randN[: n1 * R])
    t_one = time.perf_counter() - t
    t = time.perf_counter()
    o_pts, o_ind = oracle.gibbs1(otrees, nall, Niter, randU, randN, nthreads=cores)
    t_all = time.perf_counter() - t
    g_pts, g_ind = plan.sample(nall, Niter=Niter, seed=seed, sample_offset=0)
    mism = int((g_pts.shape != o_pts.shape) or (g_ind != o_ind).sum())

    def ks_two_sample(a, b):  # sup |F_a - F_b| of two equal-size samples (identical samples give 1/n: ties)
        both = np.concatenate([a, b])
        order = np.argsort(both, kind="stable")
        steps = np.where(order < a.size, 1.0, -1.0)
        return float(np.abs(np.cumsum(steps)).max() / a.size)
    return {
        "cpu_baseline": {"value": nall / t_all, "unit": "samples/s", "cores": cores, "kind": "port",
                         "sample": f"first {nall} chains of the same workload and Philox stream (C oracle, OpenMP over samples, {cores} threads)",
                         "single_core_value": n1 / t_one, "single_core_sample": f"first {n1} samples"},
        "parity": {"samples_checked": nall, "label_mismatches": mism,
                   "max_abs_point_diff": float(np.abs(g_pts - o_pts).max()),
                   "moment_mean_diff": float(np.abs(g_pts.mean(axis=1) - o_pts.mean(axis=1)).max()),
                   "moment_var_diff": float(np.abs(g_pts.var(axis=1) - o_pts.var(axis=1)).max()),
                   # two-sample Kolmogorov-Smirnov statistic per dimension, GPU pGM vs oracle pGM (north_star gate)
                   "ks_max": max(ks_two_sample(g_pts[d], o_pts[d]) for d in range(g_pts.shape[0]))},
    }


if __name__ == "__main__":
    main()
