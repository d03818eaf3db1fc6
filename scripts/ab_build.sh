#!/bin/bash
# Builds an alternative libkdehip on the GPU box with extra compiler flags and compares it with the
# in-tree library on several workloads:  scripts/ab_build.sh "-DKDEHIP_NO_PREFETCH16" [bench args...]
set -e
FLAGS="$1"; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ab
mkdir -p $OUT
cd $REPO/kerneldensityestimate.jl_amd/csrc
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 $FLAGS"
for f in balltree.cpp pack_levels.cpp gibbs_dispatch.cpp devmem.cpp; do $CXX -x hip -c $f -o $OUT/$f.o & done
for f in product.hip evaluate.hip; do $CXX -c $f -o $OUT/$f.o & done
for d in 1 2 3 4 5 6 7 8; do $CXX -mllvm -disable-vector-combine -DKDEHIP_DIM=$d -c gibbs_kernel.hip -o $OUT/gibbs_kernel_d$d.o & done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libkdehip_ab.so $OUT/*.o
rm -f $OUT/*.o
cd $REPO
run() {  # label, lib, bench args
  local lib=$2; local label=$1; shift 2
  KDEHIP_LIB=$lib python bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', '$*', 'samples/s', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],4))"
}
for rep in 1 2; do
  for args in "--config c3" "--config c3 --nout 16384" "--config c4" "--config c4 --nout 16384" "--config c5"; do
    run base $REPO/kerneldensityestimate.jl_amd/libkdehip.so $args
    run alt  $OUT/libkdehip_ab.so $args
  done
done
rm -f $OUT/libkdehip_ab.so
