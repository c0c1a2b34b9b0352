# quick GPU check: parity tests of the fused kernel + headline bench (no CPU baseline)
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -8
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err; python - <<'PY'
import json; j=json.load(open("gpurun_out/quick_bench.json")); print("default", round(j["value"]), j["ms_per_step"], j["roofline"]["kernel_ms"], {k:round(v.get("kernel_ms"),3) for k,v in j["beside_headline"].items() if isinstance(v,dict) and "kernel_ms" in v})
PY
tail -2 gpurun_out/quick_bench.err
