# What a channel block of the 3x3 kernel spends its time on: diagnostic builds of gpnerf_conv.hip (results wrong on purpose), each
# timed by conv_layer_time.py's cin sweep (slope = time per block).  Runs on the GPU box (builds there: hipcc is in the image).
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/ab
C=gp-nerf_amd/csrc
build() { # name, flags
  hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $2 -c -o /tmp/ab/conv_$1.o $C/gpnerf_conv.hip &&
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/ab/lib_$1.so $C/gpnerf_kernels.o $C/gpnerf_volume.o /tmp/ab/conv_$1.o; }
run() { echo "== $1"; GPNERF_DEBUG=1 GPNERF_LIB_PATH=/tmp/ab/lib_$1.so GPNERF_CONV_KSPLIT_MAXWG=${KS:-256} python tools/probes/conv_layer_time.py 2>&1 | tail -2; }
build product "" & build nomfma "-DGPNERF_X_CONV_NOMFMA" & build nopark "-DGPNERF_X_CONV_NOPARK" & build nofetch "-DGPNERF_X_CONV_NOFETCH -DGPNERF_X_CONV_NOPARK" & wait
build nosync "-DGPNERF_X_CONV_NOSYNC" & build mfmaonly "-DGPNERF_X_CONV_NOFETCH -DGPNERF_X_CONV_NOPARK -DGPNERF_X_CONV_NOSYNC" & wait
for v in product nomfma nopark nofetch nosync mfmaonly; do KS=0 run $v; done
echo "---- with the K split"; for v in product nomfma mfmaonly; do run $v; done
