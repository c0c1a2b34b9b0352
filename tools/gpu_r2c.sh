# full GPU suite + headline + configs[2] bench on the current library (each leg under its own timeout)
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2c/bench_default.json 2> gpurun_out/r2c/bench_default.err; python - <<'PY'
import json; j=json.load(open("gpurun_out/r2c/bench_default.json")); print("default", j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], {k:v.get("kernel_ms") for k,v in j["beside_headline"].items() if isinstance(v,dict)}, j["beside_headline"]["renderer_api"])
PY
timeout 200 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras > gpurun_out/r2c/bench_c3.json 2> gpurun_out/r2c/bench_c3.err; python -c "
import json; j=json.load(open('gpurun_out/r2c/bench_c3.json')); print('c3', j['value'], j['ms_per_step'], j['early_term'])"
timeout 200 python bench.py --steps 10 --warmup 3 --samples 128 --early-term --split-f16 --no-cpu-baseline --no-extras > gpurun_out/r2c/bench_c3s.json 2> gpurun_out/r2c/bench_c3.err; python -c "
import json; j=json.load(open('gpurun_out/r2c/bench_c3s.json')); print('c3 split', j['value'], j['ms_per_step'], j['early_term'])"
tail -3 gpurun_out/r2c/*.err
