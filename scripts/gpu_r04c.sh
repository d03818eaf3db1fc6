set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_device_density.py tests/test_gpu_bandwidth.py -x -q 2>&1 | tail -12 > $O/tests.log; cat $O/tests.log
python bench.py --steps 100 --warmup 10 > $O/bench_c3.json 2> $O/bench.err
python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench.err
python bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err
python bench.py --config c2 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2>> $O/bench.err
KDEHIP_ALIAS_DEVICES=1 python bench.py --inproc-gpus 4 --steps 10 --warmup 2 > $O/bench_inproc4.json 2>> $O/bench.err
python scripts/cold_pieces.py > $O/cold_pieces.json 2>> $O/bench.err
python scripts/cold_pieces.py c2 c2 c3 c3 > $O/cold_pieces_c2first.json 2>> $O/bench.err
cat $O/cold_pieces.json $O/cold_pieces_c2first.json
bash scripts/profile_gpu.sh r04c3 > $O/prof_c3.log 2>&1
bash scripts/profile_gpu.sh r04c4 --config c4 > $O/prof_c4.log 2>&1
bash scripts/profile_gpu.sh r04c5 --config c5 --steps 6 > $O/prof_c5.log 2>&1
bash scripts/valu_mix.sh r04c3 > $O/mix_c3.log 2>&1
bash scripts/valu_mix.sh r04c4 --config c4 > $O/mix_c4.log 2>&1
bash scripts/valu_mix.sh r04c5 --config c5 --steps 6 > $O/mix_c5.log 2>&1
python -c "
import json
for f in ['bench_c3','bench_c4','bench_c5','bench_c2','bench_inproc4']:
    try:
        d=json.load(open('$O/'+f+'.json'))
        print(f, round(d['value']), round(d['ms_per_step'],4), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('kernel_ms'), (d.get('call_inclusive') or {}).get('ms'), d.get('per_device'))
    except Exception as e: print(f, 'ERR', e)
"
