#!/bin/bash
# usage: scratch/ab.sh ENVVAR A B [reps]: interleaved bench runs with ENVVAR=A and ENVVAR=B on one box; prints ms per step
v=$1; a=$2; b=$3; n=${4:-3}
for i in $(seq $n); do
  for x in $a $b; do
    env $v=$x python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-proposals 0 --no-ism 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v=$x', round(d['ms_per_step'],3))"
  done
done
