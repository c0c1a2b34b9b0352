# round-2 GPU check: full GPU test suite, headline bench, configs[2] bench, static-vs-queue launch comparison
export GPNERF_DEBUG=1   # the experiment knobs / GPNERF_LIB_PATH below are honoured only under this switch
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_default.json 2> gpurun_out/r2a/bench_default.err; python - <<'PY'
import json; j=json.load(open("gpurun_out/r2a/bench_default.json")); print("default", j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], {k:v.get("kernel_ms") for k,v in j["beside_headline"].items() if isinstance(v,dict)}, j["beside_headline"]["renderer_api"])
PY
GPNERF_DYNAMIC=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r2a/bench_static.json 2>/dev/null; python -c "
import json; j=json.load(open('gpurun_out/r2a/bench_static.json')); print('static', j['value'], j['ms_per_step'])"
for dyn in 1 0; do
GPNERF_DYNAMIC=$dyn python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras > gpurun_out/r2a/bench_c3_dyn$dyn.json 2> gpurun_out/r2a/bench_c3.err; python -c "
import json; j=json.load(open('gpurun_out/r2a/bench_c3_dyn$dyn.json')); print('c3 dyn$dyn', j['value'], j['ms_per_step'], j['early_term'])"
GPNERF_DYNAMIC=$dyn python bench.py --steps 10 --warmup 3 --occ-cull --occupancy 0.1 --no-cpu-baseline --no-extras --patch 4x8 > gpurun_out/r2a/bench_cull_dyn$dyn.json 2> gpurun_out/r2a/bench_cull.err; python -c "
import json; j=json.load(open('gpurun_out/r2a/bench_cull_dyn$dyn.json')); print('cull10 dyn$dyn', j['value'], j['ms_per_step'])"
done
tail -3 gpurun_out/r2a/*.err
