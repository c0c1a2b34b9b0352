source tools/diag_env.sh   # the lab library: launcher experiment knobs exist only there (csrc/diag/)
run() { python bench.py --steps 10 --warmup 3 --samples 128 --early-term --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],3))"; }
run default
for s in 8 12 24 32; do GPNERF_CHAIN_SEG=$s run seg=$s; done
for t in 1 2 8; do GPNERF_CHAIN_TAILP=$t run tailp=$t; done
for f in 0.5 2; do GPNERF_CHAIN_PFILL=$f run pfill=$f; done
for c in 32 128; do GPNERF_QUEUE_CHUNK=$c run chunk=$c; done
