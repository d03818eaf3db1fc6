# round 5 (a): fp32 screening, first GPU run -- screen tests + lean/parity suites, bench c3 with and without the screen
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_screen.py -q -m gpu -x > $O/screen_tests.txt 2>&1; tail -n 15 $O/screen_tests.txt
timeout 1200 python -m pytest tests/test_gpu_lean.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/lean_tests.txt 2>&1; tail -n 6 $O/lean_tests.txt
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench.err
KDEHIP_SCREEN=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c3_noscreen.json 2>> $O/bench.err
python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench.err
KDEHIP_SCREEN=0 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4_noscreen.json 2>> $O/bench.err
python bench.py --nout 16384 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_16k.json 2>> $O/bench.err
KDEHIP_SCREEN=0 python bench.py --nout 16384 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_16k_noscreen.json 2>> $O/bench.err
tail -5 $O/bench.err
python -c "
import json
for f in ['bench_c3','bench_c3_noscreen','bench_c4','bench_c4_noscreen','bench_c3_16k','bench_c3_16k_noscreen']:
    try:
        d=json.load(open('$O/'+f+'.json')); print(f, round(d['ms_per_step'],4), d['roofline']['kernel_ms'], (d.get('parity') or {}).get('label_mismatches'))
    except Exception as e: print(f, 'failed', e)
"
