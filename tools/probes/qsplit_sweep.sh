source tools/diag_env.sh   # the lab library: launcher experiment knobs exist only there (csrc/diag/)
# frames of one to two rounds of wavefronts: persistent workgroups on a queue of (tile, sample segment) units (GPNERF_QSPLIT, experiment)
# against the default plan (whole tiles + eight-samples-per-step remainder units); GPNERF_QSPLIT_SEGMAJOR=1: all tiles' first segment first
for args in "--fill survey" "--size 256" "--size 272" "--size 320" "--size 360" "--fill survey --fold"; do
  for sm in 0 1; do for q in 0 2 4 8; do
    [ $q = 0 ] && [ $sm = 1 ] && continue
    r=$(GPNERF_DEBUG=1 GPNERF_QSPLIT=$q GPNERF_QSPLIT_SEGMAJOR=$sm python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $args 2>/dev/null | python -c "import sys,json; j=json.load(sys.stdin); print(round(j['ms_per_step'],3), 'ms', round(j['roofline']['frac'],4))")
    echo "$args qsplit=$q segmajor=$sm: $r"
  done; done
done
