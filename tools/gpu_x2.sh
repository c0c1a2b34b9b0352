# step time of a wave against the number of waves per CU that pull tiles (GPNERF_WAVE_CAP), both forms
run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $1 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); ms=j['roofline']['kernel_ms']; cap=int('$GPNERF_WAVE_CAP'); print('$1 cap', cap, round(ms,3), 'ms; step of a wave', round(ms*1e3*256*cap/524288,2), 'us')"; }
for c in 1 2 3 4 5 6 7 8; do GPNERF_WAVE_CAP=$c run --split-f16; done
for c in 4 8; do GPNERF_WAVE_CAP=$c run; done
