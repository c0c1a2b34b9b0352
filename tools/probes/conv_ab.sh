# A/B of two builds of gpnerf_conv.hip on one box: the tree's (A) against csrc/gpnerf_conv_b.hip if present (B, scratch), encoder + layers
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/ab
C=gp-nerf_amd/csrc
run() { echo "== $1"; for i in 1 2; do GPNERF_DEBUG=1 GPNERF_LIB_PATH=$2 python tools/probes/encoder_time.py 2>&1 | tail -1 | cut -c1-60; done; GPNERF_DEBUG=1 GPNERF_LIB_PATH=$2 python tools/probes/conv_layer_time.py 2>&1 | tail -2 | cut -c1-330; }
if [ -f $C/gpnerf_conv_b.hip ]; then
  hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -Igp-nerf_amd/csrc/nodiag -c -o /tmp/ab/conv_b.o $C/gpnerf_conv_b.hip &&
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/ab/lib_b.so $C/gpnerf_kernels.o $C/gpnerf_volume.o /tmp/ab/conv_b.o
  run B /tmp/ab/lib_b.so
fi
run A $PWD/$C/libgpnerf_hip.so
[ -f /tmp/ab/lib_b.so ] && run B /tmp/ab/lib_b.so
run A $PWD/$C/libgpnerf_hip.so
