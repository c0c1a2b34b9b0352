# configs[2] frame: library A/B (GPNERF_LIB_PATH), fp32 and split forms
for l in "$@"; do for f in "" "--split-f16"; do GPNERF_LIB_PATH=$PWD/$l timeout 120 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras $f 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$l', '$f', round(j['ms_per_step'],3))"; done; done
