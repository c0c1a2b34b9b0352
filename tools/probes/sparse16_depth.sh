# prefetch depth of conv_mfma16_kernel (registers vs waves per SIMD): frame phase of the survey frame per build (builds on the GPU box)
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/ab
C=gp-nerf_amd/csrc
for d in 2 3 4; do
  ( hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -Igp-nerf_amd/csrc/nodiag -DGPNERF_SPARSE16_DEPTH=$d -Rpass-analysis=kernel-resource-usage -c -o /tmp/ab/vol_$d.o $C/gpnerf_volume.hip 2>&1 | grep -A2 "conv_mfma16_kernelILb0ELi2E" | grep "VGPRs:" | sed "s/.*remark://; s/^/depth $d:/" ;
    hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/ab/lib_d$d.so $C/gpnerf_kernels.o $C/gpnerf_conv.o /tmp/ab/vol_$d.o ) &
done; wait
for d in 2 3 4 2 3 4; do echo "== depth $d"; GPNERF_DEBUG=1 GPNERF_LIB_PATH=/tmp/ab/lib_d$d.so python tools/probes/render_phases.py 2>&1 | grep "frame:" | tr '\n' ' '; echo; done
