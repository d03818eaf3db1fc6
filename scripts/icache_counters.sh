#!/bin/bash
# Instruction-fetch and scalar-cache behaviour of the sampler (default bench workload); GPU box.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $OUT/a -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/b -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/c -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for d in "abc":
    f = glob.glob("$OUT/" + d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gibbs_product_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 131072:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} {sum(v)/len(v):14.0f} per launch, {sum(v)/len(v)/2048:10.1f} per chain")
PY
