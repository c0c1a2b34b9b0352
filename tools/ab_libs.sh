# A/B of differently built libraries on the headline frame: tools/ab_libs.sh build/ab/a.so build/ab/b.so ... (diagnostic)
for lib in "$@"; do
  for outs in light api; do
    GPNERF_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --outputs $outs 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$lib', '$outs', round(j['roofline']['kernel_ms'],3), 'ms')"
  done
done
