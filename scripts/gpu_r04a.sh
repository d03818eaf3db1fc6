set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_batch.py -x -q 2>&1 | tail -25 > gpurun_out/r04a/new_tests.log
cat gpurun_out/r04a/new_tests.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04a/gpu_suite.log
cat gpurun_out/r04a/gpu_suite.log
python bench.py --steps 100 --warmup 10 > gpurun_out/r04a/bench_c3.json 2> gpurun_out/r04a/bench_c3.err
python bench.py --config c2 --batch 64 --steps 20 --warmup 3 > gpurun_out/r04a/bench_c2_batch64.json 2> gpurun_out/r04a/bench_c2_batch.err
python bench.py --config c2 --batch 8 --steps 20 --warmup 3 > gpurun_out/r04a/bench_c2_batch8.json 2>> gpurun_out/r04a/bench_c2_batch.err
python scripts/chain_timing.py c3 10 > gpurun_out/r04a/chain_c3.txt 2>&1
python scripts/chain_timing.py c3 10 2048 >> gpurun_out/r04a/chain_c3.txt 2>&1
python scripts/pipeline_timing.py c3 > gpurun_out/r04a/pipeline_c3.txt 2>&1
tail -3 gpurun_out/r04a/chain_c3.txt gpurun_out/r04a/pipeline_c3.txt
python -c "
import json
for f in ['bench_c3','bench_c2_batch64','bench_c2_batch8']:
    try:
        d=json.load(open('gpurun_out/r04a/'+f+'.json'))
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d.get('back_to_back'), d.get('batched_equals_single_calls_bit_for_bit'), d.get('call_inclusive',{}).get('ms'), (d.get('parity') or {}).get('label_mismatches'))
    except Exception as e: print(f, 'ERR', e)
"
