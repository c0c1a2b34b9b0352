# source tools/diag_env.sh : the LAB's environment for the probe scripts that turn launcher experiment knobs (GPNERF_WAVES,
# GPNERF_QSPLIT, GPNERF_CHAIN_*, GPNERF_CONV_*, ...).  The product library has no such knobs (gp-nerf_amd/csrc/nodiag/): this builds
# gp-nerf_amd/csrc/diag/libgpnerf_hip_diag.so (the same sources with the knob hooks filled in) when it is missing and points the
# binding at it -- GPNERF_LIB_PATH is honoured only together with GPNERF_DEBUG=1 (gp-nerf_amd/_lib.py).
_root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
[ -f "$_root/gp-nerf_amd/csrc/diag/libgpnerf_hip_diag.so" ] || make -s -C "$_root/gp-nerf_amd/csrc/diag" libgpnerf_hip_diag.so
export GPNERF_DEBUG=1 GPNERF_LIB_PATH="$_root/gp-nerf_amd/csrc/diag/libgpnerf_hip_diag.so"
