source tools/diag_env.sh   # the lab library: launcher experiment knobs exist only there (csrc/diag/)
for c in 64 0 8 16 256 1024; do
  for fill in survey full; do
    GPNERF_DEBUG=1 GPNERF_QUEUE_CHUNK=$c python bench.py --fill $fill --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('chunk $c', '$fill', round(j['roofline']['kernel_ms'],3), round(j['roofline']['frac'],3))"
  done
done
