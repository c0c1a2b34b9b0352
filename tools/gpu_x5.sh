for f in 0.25 0.5 0.75 1 1.5 2; do GPNERF_CHAIN_PFILL=$f timeout 120 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('fill $f', round(j['ms_per_step'],3))"; done
