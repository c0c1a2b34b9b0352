#!/bin/bash
source tools/diag_env.sh   # the lab library: launcher experiment knobs exist only there (csrc/diag/)
# early-termination frame (512x512x128): item length (GPNERF_CHAIN_SEG) x queue chunk (GPNERF_QUEUE_CHUNK)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; }
for seg in 16 32; do for c in 32 64 72 96; do
GPNERF_QUEUE_CHUNK=$c GPNERF_CHAIN_SEG=$seg timeout 120 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras 2>&1 | tail -1 | ms "ET128 item=$seg chunk=$c"
done; done
for c in 32 64 96 0; do GPNERF_QUEUE_CHUNK=$c timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | ms "headline chunk=$c"; done
