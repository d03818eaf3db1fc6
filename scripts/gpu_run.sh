#!/bin/bash
# The ONE runner for everything that goes to the GPU box (replaces the per-experiment gpu_r0*.sh scripts of rounds 4-5):
#
#   gpurun --timeout 1800 -- 'bash scripts/gpu_run.sh <tag> <step> [<step> ...]'
#
# Every step writes under gpurun_out/<tag>/ (merged back by gpurun); a step is `name` or `name:arg1,arg2,...`
# (commas become spaces).  Steps:
#   tests[:pytest args]        python -m pytest tests -m gpu -x -q [args]      -> tests.txt
#   pytest:<file or -k expr>   python -m pytest <args> -x -q -m gpu            -> pytest_<n>.txt
#   bench:<bench.py args>      one bench line                                  -> bench_<n>.json (+ .err)
#   stats:<bench.py args>      rocprofv3 --kernel-trace --stats of that bench  -> stats_<n>/ + stats_<n>.txt (per-kernel table)
#   pmc:<bench.py args>        the HBM + SQ counter passes (separate --pmc runs, kernel trace only) -> pmc_<n>/
#   prof:<ptag>[,bench args]   scripts/profile_gpu.sh (stats + FETCH/WRITE/SQ passes) -> gpurun_out/prof_<ptag>/ ; then HERE:
#                              python scripts/summarize_profile.py <ptag> <kernel match> <workload>
#   mix:<ptag>[,bench args]    scripts/valu_mix.sh (instruction-mix passes) -> gpurun_out/mix_<ptag>/ ; then summarize_mix.py
#   ab:<lib,lib,...;args>      scripts/ab_libs.py on in-tree development libraries (interleaved A/B) -> ab_<n>.txt
#   py:[VAR=VALUE,]<script,args>  python <script> <args> (with the environment assignments)  -> py_<n>.txt
#   soak                       the round's soak list on the final build        -> soak.txt
# Steps run in order; a failing step does not stop the following ones (its exit code is in steps.txt).
set -u
TAG=${1:?tag}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/$TAG
mkdir -p "$O"
cd "$REPO"
export TMPDIR=/tmp
n=0
kernel_table() {  # <dir> -> the per-kernel table of a rocprofv3 --stats run
  python3 - "$1" <<'PY'
import csv, glob, os, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not fs:
    print("no kernel_stats.csv under", sys.argv[1]); sys.exit(0)
for row in csv.DictReader(open(max(fs, key=os.path.getsize))):
    print("%-90s calls %7s  avg %11.1f ns  total %13.0f ns  %6s %%" % (row["Name"][:90], row["Calls"], float(row["AverageNs"]),
                                                                     float(row["TotalDurationNs"]), row["Percentage"]))
PY
}
for step in "$@"; do
  n=$((n + 1))
  name=${step%%:*}
  arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  arg=${arg//,/ }
  t0=$(date +%s)
  case "$name" in
    tests)  timeout 2400 python3 -m pytest tests -m gpu -x -q $arg > "$O/tests.txt" 2>&1; rc=$? ;;
    pytest) timeout 1800 python3 -m pytest $arg -x -q -m gpu > "$O/pytest_$n.txt" 2>&1; rc=$? ;;
    bench)  timeout 900 python3 bench.py $arg > "$O/bench_$n.json" 2> "$O/bench_$n.err"; rc=$? ;;
    stats)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$n" -- python3 "$REPO/bench.py" --no-cpu-baseline $arg \
         > "$O/stats_${n}_bench.json" 2> "$O/stats_$n.err"); rc=$?
      kernel_table "$O/stats_$n" > "$O/stats_$n.txt"
      find "$O/stats_$n" -type f -size +2M -delete ;;
    prof)   # prof:<ptag>[,bench args] -> gpurun_out/prof_<ptag>/ (scripts/profile_gpu.sh; condensed here by summarize_profile.py)
      ptag=${arg%% *}; rest=""; [ "$arg" != "$ptag" ] && rest=${arg#* }
      bash scripts/profile_gpu.sh $ptag $rest > "$O/prof_$ptag.txt" 2>&1; rc=$? ;;
    mix)    # mix:<ptag>[,bench args]  -> gpurun_out/mix_<ptag>/  (scripts/valu_mix.sh; condensed by summarize_mix.py)
      ptag=${arg%% *}; rest=""; [ "$arg" != "$ptag" ] && rest=${arg#* }
      bash scripts/valu_mix.sh $ptag $rest > "$O/mix_$ptag.txt" 2>&1; rc=$? ;;
    pmc)
      rc=0
      for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
                 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
        d=$(echo "$set" | cut -d' ' -f1 | tr 'A-Z' 'a-z')
        (cd /tmp && timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pmc_$n/$d" -- python3 "$REPO/bench.py" --no-cpu-baseline --steps 20 --warmup 3 $arg \
           > "$O/pmc_${n}_$d.json" 2> "$O/pmc_${n}_$d.err") || rc=$?
      done
      find "$O/pmc_$n" -type f -size +8M -delete ;;
    ab)
      libs=${arg%%;*}; rest=""; [ "$arg" != "$libs" ] && rest=${arg#*;}
      timeout 1500 python3 scripts/ab_libs.py --libs $libs $rest > "$O/ab_$n.txt" 2>&1; rc=$? ;;
    py)     # (leading VAR=VALUE words are environment assignments for this step)
      envs=""; rest=""
      for w in $arg; do if [ -z "$rest" ] && [[ "$w" == *=* ]] && [[ "$w" != -* ]]; then envs="$envs $w"; else rest="$rest $w"; fi; done
      timeout 1800 env $envs python3 $rest > "$O/py_$n.txt" 2>&1; rc=$? ;;
    soak)
      {
        KDEHIP_FUZZ_N=1500 timeout 1500 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu
        timeout 900 python3 scripts/soak_screen.py 600
        timeout 900 python3 scripts/soak_threads.py 16 600
        timeout 900 python3 scripts/soak_multi.py 1200 --resident
        timeout 600 python3 scripts/soak_chunked.py 400
        timeout 600 python3 scripts/soak_fp32.py 400
        timeout 600 python3 scripts/soak_determinism.py 800 6
        timeout 900 python3 scripts/soak_callers.py 800
      } > "$O/soak.txt" 2>&1; rc=$? ;;
    *) echo "unknown step $name" >&2; rc=64 ;;
  esac
  echo "$n $step rc=$rc $(( $(date +%s) - t0 ))s" >> "$O/steps.txt"
done
cat "$O/steps.txt"
