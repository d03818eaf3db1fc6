# round 5 (q): does the sampler miss the instruction cache?  (the kernel is ~450 KB of code, every step's code exists M times)
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05q; mkdir -p $O
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "[A-Z_0-9]*\(ICACHE\|IFETCH\|INST_CACHE\|INSTR\)[A-Z_0-9]*" | sort -u > $O/counters.txt
cat $O/counters.txt
for cfg in c3 c4; do
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $O/ic_$cfg -- python3 $REPO/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_ic_$cfg.json 2> $O/ic_$cfg.err
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/if_$cfg -- python3 $REPO/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_if_$cfg.json 2> $O/if_$cfg.err
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get('O', '/tmp')
for d in sorted(glob.glob(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r05q/i*_c*'))):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'gibbs_lean_kernel' in row['Kernel_Name']:
                a = acc[row['Counter_Name']]; a[0] += float(row['Counter_Value']); a[1] += 1
    print(os.path.basename(d), {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()})
PY
find $O -type f -size +4M -delete
