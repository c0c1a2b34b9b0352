source tools/diag_env.sh   # the lab library: launcher experiment knobs exist only there (csrc/diag/)
run() { python bench.py $2 --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['roofline']['kernel_ms'],3), round(j['roofline']['frac'],3))"; }
run "survey default" "--fill survey"; run "256 default" "--size 256"
for st in 4 8 16 32 64 128; do
  GPNERF_STAGGER=$st run "survey stagger=$st" "--fill survey"
  GPNERF_STAGGER=$st run "256 stagger=$st" "--size 256"
done
run "survey default" "--fill survey"; run "256 default" "--size 256"
