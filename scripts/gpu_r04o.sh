# round 4 (o): tile and chunk copies by the older wavefronts -- GPU suite, fp32 soak, bench c5 / c3 / c4
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_tests.txt 2>&1; tail -n 6 $O/gpu_tests.txt
timeout 600 python scripts/soak_fp32.py 800 2>&1 | tail -1 > $O/soak.txt
timeout 600 python scripts/soak_chunked.py 400 2>&1 | tail -1 >> $O/soak.txt
cat $O/soak.txt
python bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench.err
python bench.py --steps 400 --warmup 20 --no-cpu-baseline > $O/bench_c3.json 2>> $O/bench.err
python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench.err
python -c "
import json
for f in ['bench_c5','bench_c3','bench_c4']:
    d=json.load(open('$O/'+f+'.json')); print(f, round(d['ms_per_step'],4), d['roofline']['kernel_ms'], (d.get('parity') or {}).get('label_mismatches'))
"
python bench.py --nout 16384 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_16k.json 2>> $O/bench.err
python bench.py --config c2 --batch 64 --steps 20 --warmup 3 > $O/bench_c2_batch64.json 2>> $O/bench.err
python -c "
import json
for f in ['bench_c3_16k','bench_c2_batch64']:
    d=json.load(open('$O/'+f+'.json')); print(f, round(d['ms_per_step'],4), d['roofline']['kernel_ms'])
"
