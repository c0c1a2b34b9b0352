# A/B of differently built libraries on the headline frame (diagnostic):
export GPNERF_DEBUG=1   # the experiment knobs / GPNERF_LIB_PATH below are honoured only under this switch
#   tools/ab_libs.sh [--split] build/ab/a.so build/ab/b.so ...
extra=""; if [ "$1" = "--split" ]; then extra="--split-f16"; shift; fi
for lib in "$@"; do
  for outs in light api; do
    GPNERF_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --outputs $outs $extra 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$lib', '$outs', '$extra', round(j['roofline']['kernel_ms'],3), 'ms')"
  done
done
