#!/bin/bash
# A/B of kernel variants on the headline workload, interleaved: scripts/ab_variant.sh [--nout N] 0 1 0 1
NOUT=""
if [ "$1" = "--nout" ]; then NOUT="--nout $2"; shift 2; fi
for v in "$@"; do
  python bench.py --variant $v $NOUT --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant', $v, 'samples/s', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],4))"
done
