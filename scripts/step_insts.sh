#!/bin/bash
# Instruction cost of the parts of one single-row step (ablation flags of the diagnostic build); GPU box.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sinst
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
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc -- python3 $REPO/scripts/step_insts.py > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 $REPO/scripts/step_insts.py >> $OUT/run.log 2>&1
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
a, b = load("pmc")[-12:], load("pmc2")[-12:]
names = ["all", "const uniform (1)", "no set_particle (2)", "no LOO product (4)", "no draw (8)", "none of 2,4,8"]
print("level-6 step (B=1, resident, FAST), per chain-step: VALU SALU LDS | wave-quads active wait_any wait_inst")
for i, nm in enumerate(names):
    z5 = dict(a[2 * i]); z5.update(b[2 * i]); z6 = dict(a[2 * i + 1]); z6.update(b[2 * i + 1])
    d = {c: (z6[c] - z5[c]) / 2048 / 44 for c in z6}
    print(f"{nm:22s} | {d['SQ_INSTS_VALU']:6.0f} {d['SQ_INSTS_SALU']:6.0f} {d['SQ_INSTS_LDS']:5.0f} | "
          f"{d['SQ_WAVE_CYCLES']:7.0f} {d['SQ_ACTIVE_INST_ANY']:7.0f} {d['SQ_WAIT_ANY']:7.0f} {d['SQ_WAIT_INST_ANY']:7.0f}")
PY
