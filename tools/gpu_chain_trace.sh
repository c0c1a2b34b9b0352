#!/bin/bash
# per-launch durations of the chained early-termination segments (512x512x128)
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/chain; mkdir -p /root/repo/gpurun_out/chain
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/chain -o chain -- python3 /root/repo/bench.py --steps 3 --warmup 1 --samples 128 --early-term --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
cd /root/repo
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/chain/**/*kernel_trace.csv',recursive=True))[-1]
rows=[r for r in csv.DictReader(open(f)) if 'render_fused' in r['Kernel_Name']]
n=len(rows)//4
last=rows[-(len(rows)//4 if len(rows)%4 else 4):] if False else rows[-8:]
t0=int(last[0]['Start_Timestamp'])
for r in last:
    print(round((int(r['Start_Timestamp'])-t0)/1e6,3), '+', round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,3), 'ms')
PY
