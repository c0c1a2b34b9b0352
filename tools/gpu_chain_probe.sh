#!/bin/bash
cd /root/repo
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; }
for c in 8 24 40 56 72 96 136; do
GPNERF_QUEUE_CHUNK=$c GPNERF_CHAIN_SEG=16 timeout 120 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras 2>&1 | tail -1 | ms "ET128 item=16 chunk=$c"
done
for seg in 8 16 32; do
GPNERF_QUEUE_CHUNK=24 GPNERF_CHAIN_SEG=$seg timeout 120 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras 2>&1 | tail -1 | ms "ET128 item=$seg chunk=24"
done
