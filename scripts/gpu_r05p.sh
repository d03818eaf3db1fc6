# evidence run of round 5 (GPU box): suite, soaks, bench lines, rocprofv3 summaries, per-level table
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r05p}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/tests.log; cat $O/tests.log
python bench.py --steps 400 --warmup 20 > $O/bench_c3.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver_form.json 2>> $O/bench.err
python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench.err
python bench.py --config c4 --strong --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_c4_strong.json 2>> $O/bench.err
python bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err
python bench.py --config c5 --strong --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_strong.json 2>> $O/bench.err
python bench.py --config c2 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2>> $O/bench.err
python bench.py --config c2 --batch 64 --steps 20 --warmup 3 > $O/bench_c2_batch64.json 2>> $O/bench.err
python bench.py --nout 16384 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_16k.json 2>> $O/bench.err
KDEHIP_ALIAS_DEVICES=1 python bench.py --inproc-gpus 4 --steps 10 --warmup 2 > $O/bench_inproc4.json 2>> $O/bench.err
python scripts/chain_timing.py c3 10 > $O/chain.txt 2>&1
python scripts/chain_timing.py c3 10 2048 >> $O/chain.txt 2>&1
python scripts/pipeline_timing.py c3 > $O/pipeline.txt 2>&1
python scripts/loocv_timing.py 20 > $O/loocv.txt 2>&1
python scripts/screen_rate.py --chains 64 --weighted > $O/screen_rate.txt 2>&1
python scripts/screen_rate.py --config c4 --chains 8 --weighted >> $O/screen_rate.txt 2>&1
( NP=1100 timeout 300 python scripts/check_screen_chunk.py 6 4 2048 4096 8000 3000
  NP=1100 timeout 300 python scripts/check_screen_chunk.py 3 8 5000 10000 2048
  NP=1100 timeout 300 python scripts/check_screen_chunk.py 2 3 20000 9000 ) 2>&1 | grep -v amdgpu.ids > $O/check_screen_chunk.txt
( KDEHIP_FUZZ_N=1500 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
  timeout 600 python scripts/soak_threads.py 16 1500 2>&1 | tail -1
  timeout 600 python scripts/soak_multi.py 3000 --resident 2>&1 | tail -1
  timeout 600 python scripts/soak_chunked.py 800 2>&1 | tail -1
  timeout 600 python scripts/soak_fp32.py 800 2>&1 | tail -1
  timeout 600 python scripts/soak_determinism.py 1500 6 2>&1 | tail -1
  timeout 600 python scripts/soak_callers.py 1500 2>&1 | tail -1 ) > $O/soaks.txt 2>&1
cat $O/soaks.txt
bash scripts/profile_gpu.sh ${TAG}3 > $O/prof_c3.log 2>&1
bash scripts/profile_gpu.sh ${TAG}4 --config c4 > $O/prof_c4.log 2>&1
bash scripts/profile_gpu.sh ${TAG}5 --config c5 --steps 6 > $O/prof_c5.log 2>&1
bash scripts/valu_mix.sh ${TAG}3 > $O/mix_c3.log 2>&1
bash scripts/valu_mix.sh ${TAG}4 --config c4 > $O/mix_c4.log 2>&1
bash scripts/valu_mix.sh ${TAG}5 --config c5 --steps 6 > $O/mix_c5.log 2>&1
if [ -f kerneldensityestimate.jl_amd/libkdehip_exp.so ]; then
  bash scripts/level_profile.sh kerneldensityestimate.jl_amd/libkdehip_exp.so > $O/level_insts.txt 2> $O/level_insts.err
  KDEHIP_LIB=$PWD/kerneldensityestimate.jl_amd/libkdehip_exp.so python scripts/level_timing2.py c3 0 > $O/level_timing.txt 2>&1
fi
if [ -f kerneldensityestimate.jl_amd/libkdehip_exp4.so ]; then
  KDEHIP_LIB=$PWD/kerneldensityestimate.jl_amd/libkdehip_exp4.so python scripts/level_timing2.py c4 0 > $O/level_timing_c4.txt 2>&1
fi
python -c "
import json
for f in ['bench_c3','bench_c3_driver_form','bench_c4','bench_c4_strong','bench_c5','bench_c5_strong','bench_c2','bench_c2_batch64','bench_c3_16k','bench_inproc4']:
    try:
        d=json.load(open('$O/'+f+'.json'))
        print(f, round(d['value']), round(d['ms_per_step'],4), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('kernel_ms'), (d.get('call_inclusive') or {}).get('ms'), (d.get('parity') or {}).get('label_mismatches'), ((d.get('roofline') or {}).get('valu_floor') or {}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
"
