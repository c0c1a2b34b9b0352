# scheduling variants of the 3x3 kernel's tap loop, each timed on the whole encoder and on single layers (builds on the GPU box)
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/ab
C=gp-nerf_amd/csrc
build() { hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $2 -c -o /tmp/ab/conv_$1.o $C/gpnerf_conv.hip &&
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/ab/lib_$1.so $C/gpnerf_kernels.o $C/gpnerf_volume.o /tmp/ab/conv_$1.o; }
run() { echo "== $1"; GPNERF_DEBUG=1 GPNERF_LIB_PATH=/tmp/ab/lib_$1.so python tools/probes/encoder_time.py 2>&1 | tail -1 | cut -c1-60; GPNERF_DEBUG=1 GPNERF_LIB_PATH=/tmp/ab/lib_$1.so python tools/probes/conv_layer_time.py 2>&1 | tail -2 | cut -c1-330; }
build all "" & build nosched "-DGPNERF_X_CONV_NOSCHED" & build nogroups "-DGPNERF_X_CONV_NOGROUPS" & build neither "-DGPNERF_X_CONV_NOSCHED -DGPNERF_X_CONV_NOGROUPS" & wait
build look1 "-DGPNERF_CONV_LOOK=1" & build look1_neither "-DGPNERF_CONV_LOOK=1 -DGPNERF_X_CONV_NOSCHED -DGPNERF_X_CONV_NOGROUPS" & build look1_nogroups "-DGPNERF_CONV_LOOK=1 -DGPNERF_X_CONV_NOGROUPS" & wait
for v in all nosched nogroups neither look1 look1_nogroups look1_neither; do run $v; done
