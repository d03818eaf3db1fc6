#!/bin/bash
# Throughput of the headline shape vs chains per GPU and workgroup width (variant 8 / 16 waves).
for n in 2048 4096 8192 16384; do for v in 8 16; do
  timeout 300 python bench.py --nout $n --variant $v --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['nout_per_gpu'], 'waves/WG', $v, 'samples/s', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],3))"
done; done
