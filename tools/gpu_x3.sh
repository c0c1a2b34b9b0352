# split form with / without the scheduling pins (-DGPNERF_X_NOPIN_S), guarded and unguarded
run() { GPNERF_LIB_PATH=$PWD/$1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --split-f16 $2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$1', '$2', round(j['roofline']['kernel_ms'],3), 'ms')"; }
for l in build/ab/cur.so build/ab/nopin.so; do run $l ""; run $l "--no-guard"; done
for l in build/ab/cur.so build/ab/nopin.so; do run $l "--samples 128 --early-term"; done
