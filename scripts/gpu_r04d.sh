set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/tests.log; cat $O/tests.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench.err
python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench.err
python bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err
python bench.py --config c2 --batch 64 --steps 20 --warmup 3 > $O/bench_c2_batch64.json 2>> $O/bench.err
python bench.py --config c2 --batch 8 --steps 20 --warmup 3 > $O/bench_c2_batch8.json 2>> $O/bench.err
python bench.py --config c3 --batch 8 --nout 256 --steps 10 --warmup 2 > $O/bench_c3_batch8.json 2>> $O/bench.err
python scripts/cold_pieces.py > $O/cold_pieces.json 2>> $O/bench.err
python scripts/cold_pieces.py c2 c2 c3 c3 > $O/cold_pieces_c2first.json 2>> $O/bench.err
cat $O/cold_pieces.json $O/cold_pieces_c2first.json
python scripts/loocv_timing.py 20 > $O/loocv.txt 2>&1; tail -3 $O/loocv.txt
python -c "
import json
for f in ['bench_c3','bench_c4','bench_c5','bench_c2_batch64','bench_c2_batch8','bench_c3_batch8']:
    try:
        d=json.load(open('$O/'+f+'.json'))
        print(f, round(d['value']), round(d['ms_per_step'],4), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('kernel_ms'), (d.get('back_to_back') or {}).get('batched_speedup'), d.get('batched_equals_single_calls_bit_for_bit'))
    except Exception as e: print(f, 'ERR', e)
"
