#!/bin/bash
# Development build of the lean sampler: compiles ONE instantiation of gibbs_lean.hip (default: config 3's -- fp64,
# D = 6, M = 4, 8 chains per workgroup) with extra flags and links it with the objects of the regular build into
# kerneldensityestimate.jl_amd/libkdehip_<tag>.so, for interleaved A/B runs with scripts/ab_libs.py.
#   [DIM=6] scripts/dev_lean.sh <tag> [extra hipcc flags...]
#   other instantiations: -DKDEHIP_LEAN_DEV_F32, -DKDEHIP_LEAN_DEV_M=<densities>, -DKDEHIP_LEAN_DEV_W=<chains per workgroup>
set -e
TAG=$1; shift
# which of the regular build's translation units the development object replaces (same entry point name)
REPL="gibbs_lean_d${DIM:-6}.o"; EXTRA=""
case " $* " in
  *KDEHIP_LEAN_DEV_F32*) REPL="gibbs_lean_f32_d${DIM:-6}.o"; EXTRA="-DKDEHIP_LEAN_F32" ;;
  *KDEHIP_LEAN_DEV_M=[5-8]*) REPL="gibbs_lean_hi_d${DIM:-6}.o"; EXTRA="-DKDEHIP_LEAN_HI" ;;
esac
cd "$(dirname "$0")/../kerneldensityestimate.jl_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wextra -Wno-unused-parameter --offload-arch=gfx950 \
  -munsafe-fp-atomics -mllvm -disable-vector-combine -DKDEHIP_DIM=${DIM:-6} -DKDEHIP_LEAN_DEV $EXTRA "$@" -c gibbs_lean.hip -o build/dev_lean_$TAG.o
if [ -n "$MINI" ]; then
  # MINI=1: a small library (a few MB instead of ~100) holding only this dimension count's sampler units, stubs for the
  # rest: for experiments that ship several development libraries to the GPU box at once
  D=${DIM:-6}
  /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -DKDEHIP_DIM=$D -c ../../scripts/dev_stubs.cpp -o build/dev_stubs_$D.o
  OBJS=$(ls build/*.o | grep -v "dev_lean_\|dev_stubs_\|gibbs_lean\|gibbs_kernel_d")
  OBJS="$OBJS build/gibbs_kernel_d$D.o build/dev_stubs_$D.o"
  for u in gibbs_lean_d$D.o gibbs_lean_hi_d$D.o gibbs_lean_f32_d$D.o; do [ "$u" != "$REPL" ] && OBJS="$OBJS build/$u"; done
else
  OBJS=$(ls build/*.o | grep -v "/$REPL" | grep -v "dev_lean_\|dev_stubs_")
fi
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libkdehip_$TAG.so $OBJS build/dev_lean_$TAG.o
ls -la ../libkdehip_$TAG.so
