#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace stats and HBM PMC counters of the
# default bench workload.  Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do
# not fit one pass on gfx950).  Usage: scripts/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $BENCH > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 $BENCH > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 $BENCH > "$OUT/bench_write.json" 2> "$OUT/write.err"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/sq" -- python3 $BENCH > "$OUT/bench_sq.json" 2> "$OUT/sq.err"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/sq2" -- python3 $BENCH > "$OUT/bench_sq2.json" 2> "$OUT/sq2.err"
find "$OUT" -name "*.csv" | head -50
# keep only the small CSVs (stats + counter collection); drop anything large
find "$OUT" -type f -size +8M -delete
du -sh "$OUT"
