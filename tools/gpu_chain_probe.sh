#!/bin/bash
# step time of the fused kernel against the number of waves per CU that pull tiles (GPNERF_WAVE_CAP), and the chained
# early-termination frame (512x512x128)
cd /root/repo
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3))"; }
for w in 1 2 3 4 5 6 7 8; do
  GPNERF_WAVE_CAP=$w python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | ms "fp32 cap$w"
done
for w in 2 4 6 8; do
  GPNERF_WAVE_CAP=$w python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --split-f16 2>&1 | tail -1 | ms "split cap$w"
done
python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras 2>&1 | tail -1 | ms "ET128 chain"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | ms "headline"
