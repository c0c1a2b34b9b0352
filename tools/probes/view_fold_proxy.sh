# Per-view fold of base_fc.0, measured by proxies on the headline frame (results of the SKIP builds are wrong on purpose):
#   SKIP   : 84 of the 108 per-view MFMAs of a step are not issued (a real fold drops 96)       -> what the matrix pipe saves
#   GATHER : the folded table's taps are loaded and accumulated (64 values per texel, 32 per lane, 4 taps x 3 views), nothing dropped -> what it costs
#   BOTH   : both at once = a lower bound for the real thing (which also has to keep 8 tap offsets / weights per view alive, or re-project)
# Builds its three diagnostic libraries ON the GPU box (hipcc is in the image; ~6 minutes of compiling on its cores) and stops at
# the first one that is missing -- round 3's version expected prebuilt libraries and hid a failed load behind 2>/dev/null.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/vf
C=gp-nerf_amd/csrc
build() { # name, flags
  hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $2 -c -o /tmp/vf/k_$1.o $C/gpnerf_kernels.hip &&
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/vf/libvf_$1.so /tmp/vf/k_$1.o $C/gpnerf_volume.o $C/gpnerf_conv.o; }
build SKIP "-DGPNERF_X_VIEWFOLD_SKIP" & build GATHER "-DGPNERF_X_VIEWFOLD_GATHER" & build BOTH "-DGPNERF_X_VIEWFOLD_SKIP -DGPNERF_X_VIEWFOLD_GATHER" & wait
for v in SKIP GATHER BOTH; do [ -s /tmp/vf/libvf_$v.so ] || { echo "view_fold_proxy: /tmp/vf/libvf_$v.so was not built"; exit 1; }; done
export GPNERF_DEBUG=1 GPNERF_X_VIEWTAB=1
run() { out=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras) || { echo "$1: bench.py failed"; exit 1; }
        echo "$out" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],3), 'ms')"; }
run product
for v in SKIP GATHER BOTH; do GPNERF_LIB_PATH=/tmp/vf/libvf_$v.so run $v; done
