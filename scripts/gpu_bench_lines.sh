# the bench lines of the evidence runs alone (after the traffic files of the same build have been regenerated)
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r05t}; O=gpurun_out/$TAG; mkdir -p $O
python bench.py --steps 400 --warmup 20 > $O/bench_c3.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver_form.json 2>> $O/bench.err
python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench.err
python bench.py --config c4 --strong --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_c4_strong.json 2>> $O/bench.err
python bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err
python bench.py --config c5 --strong --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_strong.json 2>> $O/bench.err
python bench.py --nout 16384 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_16k.json 2>> $O/bench.err
python -c "
import json
for f in ['bench_c3','bench_c3_driver_form','bench_c4','bench_c4_strong','bench_c5','bench_c5_strong','bench_c3_16k']:
    d=json.load(open('$O/'+f+'.json')); r=d['roofline']; print(f, round(d['value']), round(d['ms_per_step'],4), r['frac'], r['kernel_ms'], (r.get('valu_floor') or {}).get('frac'), d.get('screen'))
"
