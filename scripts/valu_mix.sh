#!/bin/bash
# Runs on the GPU box (through gpurun): the vector-instruction mix of the default bench workload's sampler kernel
# (rocprofv3 PMC passes of their own, kernel-trace only) and the clock the chip held (GRBM_GUI_ACTIVE).
# Usage: scripts/valu_mix.sh <tag> [bench args...]   ->  gpurun_out/mix_<tag>/
set -u
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/mix_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline $*"
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 $BENCH > "$OUT/bench_$name.json" 2> "$OUT/$name.err"
}
pass f64 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
pass f32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32
pass int SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS
pass act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass lds SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass clk GRBM_GUI_ACTIVE SQ_WAVES
find "$OUT" -type f -size +8M -delete
du -sh "$OUT"
