#!/bin/bash
# Per-level instruction and wait profile of the config-3 sampler (runs on the GPU box).  Needs a diagnostic library
# built beforehand with -DKDEHIP_EXPERIMENTS (level cut-offs; e.g. `scripts/dev_lean.sh exp -DKDEHIP_EXPERIMENTS`):
#   scripts/level_profile.sh kerneldensityestimate.jl_amd/libkdehip_exp.so [kernel name substring] > profiles/<tag>_level_insts.txt
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$(realpath $1); KNAME=${2:-gibbs_lean_kernel}
OUT=$REPO/gpurun_out/linst_$$
mkdir -p $OUT
export KDEHIP_LIB=$LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc -- python3 $REPO/scripts/level_insts.py > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 $REPO/scripts/level_insts.py >> $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
def load(d):
    f = glob.glob("$OUT/" + d + "/*/*counter_collection.csv")[0]
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "$KNAME" not in r["Kernel_Name"]: continue
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [v for k, v in sorted(rows.items())]
a, b = load("pmc"), load("pmc2")
NL, NCH, NST = int('${KDEHIP_LEVELS:-10}'), int('${KDEHIP_CHAINS:-2048}'), int('${KDEHIP_STEPS_PER_LEVEL:-44}')
a, b = a[-NL:], b[-NL:]   # the cut-off launches are the last L
prev = {}
print("level | VALU SALU LDS SMEM per chain-step | wave-quads active wait_any wait_inst per chain-step")
tot = collections.Counter()
for k, (x, y) in enumerate(zip(a, b), 1):
    z = dict(x); z.update(y)
    d = {c: (z[c] - prev.get(c, 0.0)) / NCH / NST for c in z}
    prev = z
    print(f"{k:5d} | {d['SQ_INSTS_VALU']:6.0f} {d['SQ_INSTS_SALU']:6.0f} {d['SQ_INSTS_LDS']:5.0f} {d['SQ_INSTS_SMEM']:5.0f} | "
          f"{d['SQ_WAVE_CYCLES']:7.0f} {d['SQ_ACTIVE_INST_ANY']:7.0f} {d['SQ_WAIT_ANY']:7.0f} {d['SQ_WAIT_INST_ANY']:7.0f}")
print("per chain, whole run: " + " ".join(f"{c}={prev[c]/NCH:.0f}" for c in sorted(prev)))
PY
rm -rf $OUT
