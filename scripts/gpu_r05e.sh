# round 5 (e): full GPU suite + soaks + bench lines on the screened build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_tests.txt 2>&1; tail -n 6 $O/gpu_tests.txt
KDEHIP_FUZZ_N=400 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -n 2 > $O/soak.txt
timeout 600 python scripts/soak_chunked.py 200 2>&1 | tail -1 >> $O/soak.txt
timeout 600 python scripts/soak_determinism.py 300 4 2>&1 | tail -1 >> $O/soak.txt
timeout 600 python scripts/soak_multi.py 400 --resident 2>&1 | tail -1 >> $O/soak.txt
cat $O/soak.txt
python bench.py --steps 400 --warmup 20 > $O/bench_c3.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3_steps20.json 2>> $O/bench.err
python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench.err
python bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err
python bench.py --nout 16384 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_16k.json 2>> $O/bench.err
python bench.py --config c2 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c2.json 2>> $O/bench.err
tail -3 $O/bench.err
python -c "
import json
for f in ['bench_c3','bench_c3_steps20','bench_c4','bench_c5','bench_c3_16k','bench_c2']:
    try:
        d=json.load(open('$O/'+f+'.json')); print(f, round(d['ms_per_step'],4), d['roofline']['kernel_ms'], d['roofline'].get('frac'), (d.get('parity') or {}).get('label_mismatches'))
    except Exception as e: print(f, 'failed', e)
"
