cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6k; rm -rf $o; mkdir -p $o
timeout 300 python tools/probes/render_glue.py profile > $o/glue.txt 2>&1
grep -v "amdgpu.ids" $o/glue.txt | cut -c1-190 | head -70
