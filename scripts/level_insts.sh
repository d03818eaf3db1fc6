#!/bin/bash
# Instructions per level of the sampler (diagnostic build with level cut-offs); runs on the GPU box.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/linst
mkdir -p $OUT
cd $REPO/kerneldensityestimate.jl_amd/csrc
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950"
for f in balltree.cpp pack_levels.cpp gibbs_dispatch.cpp devmem.cpp; do $CXX -x hip -c $f -o $OUT/$f.o & done
for f in product.hip evaluate.hip; do $CXX -c $f -o $OUT/$f.o & done
for d in 1 2 3 4 5 6 7 8; do $CXX -DKDEHIP_EXPERIMENTS -mllvm -disable-vector-combine -DKDEHIP_DIM=$d -c gibbs_kernel.hip -o $OUT/gibbs_kernel_d$d.o & done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libkdehip_exp.so $OUT/*.o
rm -f $OUT/*.o
export KDEHIP_LIB=$OUT/libkdehip_exp.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc -- python3 $REPO/scripts/level_insts.py > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 $REPO/scripts/level_insts.py >> $OUT/run.log 2>&1
rm -f $OUT/libkdehip_exp.so
python3 - <<PY
import csv, glob, collections
def load(d):
    f = glob.glob("$OUT/" + d + "/*/*counter_collection.csv")[0]
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "gibbs_product_kernel" not in r["Kernel_Name"]: continue
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [v for k, v in sorted(rows.items())]
a, b = load("pmc"), load("pmc2")
a, b = a[-10:], b[-10:]   # the ten cut-off launches are the last ten
prev = {}
print("level | VALU SALU LDS SMEM per chain-step | wave-quads active wait_any wait_inst per chain-step")
for k, (x, y) in enumerate(zip(a, b), 1):
    z = dict(x); z.update(y)
    d = {c: (z[c] - prev.get(c, 0.0)) / 2048 / 44 for c in z}
    prev = z
    print(f"{k:5d} | {d['SQ_INSTS_VALU']:6.0f} {d['SQ_INSTS_SALU']:6.0f} {d['SQ_INSTS_LDS']:5.0f} {d['SQ_INSTS_SMEM']:5.0f} | "
          f"{d['SQ_WAVE_CYCLES']:7.0f} {d['SQ_ACTIVE_INST_ANY']:7.0f} {d['SQ_WAIT_ANY']:7.0f} {d['SQ_WAIT_INST_ANY']:7.0f}")
PY
