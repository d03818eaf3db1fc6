set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
L=kerneldensityestimate.jl_amd
python scripts/ab_libs.py --libs $L/libkdehip_old.so $L/libkdehip_new.so $L/libkdehip_ilp.so $L/libkdehip_bias0.so --configs c3 --rounds 9 --steps 20 > $O/ab.txt 2>&1
tail -12 $O/ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $R/$O/ic_a -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/$O/ic_c -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/ic_d -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
for d in ["ic_a","ic_c","ic_d"]:
    fs = glob.glob("$O/" + d + "/*/*counter_collection.csv")
    if not fs: print(d, "no csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "gibbs_lean_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} {sum(v)/len(v):14.0f} per launch, {sum(v)/len(v)/2048:10.1f} per chain  (n={len(v)})")
PY
rm -rf $O/ic_a $O/ic_c $O/ic_d
