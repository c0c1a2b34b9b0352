# experiment: start offset between the two waves of a SIMD (GPNERF_STAGGER x 64 cycles), split form without scheduling pins
run() { GPNERF_LIB_PATH=$PWD/$1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$1', '$2', 'stagger=$GPNERF_STAGGER', round(j['roofline']['kernel_ms'],3), 'ms')"; }
for st in 0 32 64 128 256 384 512 768; do GPNERF_STAGGER=$st run build/ab/stag.so --split-f16; done
for st in 0 128 256; do GPNERF_STAGGER=$st run build/ab/stag.so; done
GPNERF_STAGGER=0 run build/ab/nopin.so --split-f16
GPNERF_STAGGER=256 run build/ab/nopin.so --split-f16
