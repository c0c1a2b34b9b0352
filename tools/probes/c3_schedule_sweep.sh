# configs[2] (512x512x128, early termination): segment SCHEDULES (GPNERF_CHAIN_SCHEDULE, experiment knob) -- shorter segments where most
# rays die (samples 16..48 on the bench frame: 62 % / 26 % / 11 % alive after 16 / 32 / 48), longer ones where few are left.
# The default is run first, in the middle and last: a fresh box's first bench is slow, and boxes differ by several per cent.
source tools/diag_env.sh   # the lab library: launcher experiment knobs exist only there (csrc/diag/)
run() { python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],3), round(j['early_term']['samples_evaluated_frac'],4))"; }
run warmup-default; run default
for s in "16,16,16,16,32,32" "16,16,16,16,64" "16,16,16,32,48" "16,16,16,16,16,48" "16,16,16,16,16,16,32" "16,16,16,48,32" "16,16,32,64" "16,16,16,16,32,16,16"; do
  GPNERF_CHAIN_SCHEDULE=$s run "schedule=$s"
done
run default
for s in "16,16,8,8,16,32,32" "16,16,8,8,16,64" "16,8,8,16,16,32,32" "16,16,16,16,32,32"; do
  GPNERF_CHAIN_SCHEDULE=$s run "schedule=$s"
done
run default
