# A/B of the chained instantiations' level-0 software prefetch (-DGPNERF_PREFETCH=0 builds the library without it) on ONE box:
# the ZJU-sized survey frame, 272x272 / 300x300 full-fill frames (just over one round), configs[2], and the headline (untouched instantiation)
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/pf
C=gp-nerf_amd/csrc
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -DGPNERF_PREFETCH=0 -c -o /tmp/pf/k0.o $C/gpnerf_kernels.hip &&
hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/pf/lib_nopf.so /tmp/pf/k0.o $C/gpnerf_volume.o $C/gpnerf_conv.o || { echo "build failed"; exit 1; }
run() { out=$(python bench.py $2 --no-extras --no-cpu-baseline --steps 10 2>/dev/null) || { echo "$1 failed"; return; }
        echo "$out" | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['roofline']['kernel_ms'],3), round(j['roofline']['frac'],3))"; }
for rep in 1 2; do
for cfg in "survey:--fill survey" "s272:--size 272" "s300:--size 300" "c3:--samples 128 --early-term" "headline:"; do
  n=${cfg%%:*}; a=${cfg#*:}
  run "prefetch   $n" "$a"
  GPNERF_DEBUG=1 GPNERF_LIB_PATH=/tmp/pf/lib_nopf.so run "no-prefetch $n" "$a"
done
done
