# per-wavefront pixel block shape of the dense launch (bench.py --patch) on the ZJU-sized survey frame and the headline frame
for p in 32x8 8x4 4x8 16x2 16x4 32x1 32x4 64x8; do
  for fill in survey full; do
    python bench.py --fill $fill --patch $p --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('patch $p', '$fill', round(j['roofline']['kernel_ms'],3), round(j['roofline']['frac'],3))"
  done
done
