# A/B of single layers on ONE box: the tree's library (A) against csrc/gpnerf_conv_b.hip (B, scratch; may be an older variant that lacks
# the encoder's newer entry points: only conv_layer_time.py is run against it)
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/ab
C=gp-nerf_amd/csrc
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -Igp-nerf_amd/csrc/nodiag -c -o /tmp/ab/conv_b.o $C/gpnerf_conv_b.hip &&
hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/ab/lib_b.so $C/gpnerf_kernels.o $C/gpnerf_volume.o /tmp/ab/conv_b.o || exit 1
for v in A B A B; do
  if [ $v = A ]; then L=$PWD/$C/libgpnerf_hip.so; else L=/tmp/ab/lib_b.so; fi
  echo "== $v"; GPNERF_DEBUG=1 GPNERF_LIB_PATH=$L python tools/probes/conv_layer_time.py 2>&1 | tail -2 | cut -c1-330
done
