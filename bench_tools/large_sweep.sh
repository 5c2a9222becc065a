#!/bin/bash
# Developer tool: step time of the 256 x 256 unparameterized path (bench.py --leg config4) under the tuning
# environment variables of spectral_large.hip; one line per setting.
for v in "$@"; do
  printf "%-60s " "$v"
  env $v python bench.py --leg config4 --steps 100 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['config4']; print('%.4f ms/step  frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done
