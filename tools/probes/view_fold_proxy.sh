# Per-view fold of base_fc.0, measured by proxies on the headline frame (results of the SKIP builds are wrong on purpose):
#   SKIP   : 84 of the 108 per-view MFMAs of a step are not issued (a real fold drops 96)       -> what the matrix pipe saves
#   GATHER : the folded table's taps are loaded and accumulated (64 values per texel, 32 per lane, 4 taps x 3 views), nothing dropped -> what it costs
#   BOTH   : both at once = a lower bound for the real thing (which also has to keep 8 tap offsets / weights per view alive, or re-project)
export GPNERF_DEBUG=1 GPNERF_X_VIEWTAB=1
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],3), 'ms')"; }
run product
for v in SKIP GATHER BOTH; do GPNERF_LIB_PATH=$PWD/build/ab/libvf_$v.so run $v; done
