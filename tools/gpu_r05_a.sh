#!/bin/bash
# round 5, first GPU pass: the reference-order fp32 form against the trained-like fixtures and its CPU twin; its cost on the headline frame
set -x
O=gpurun_out/r05a; mkdir -p $O
python tools/trained_like_report.py > $O/trained_like.txt 2>&1
python tools/parity_report.py > $O/parity_report.txt 2>&1
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extras --steps 20 > $O/bench_fold_$i.json 2>$O/bench_fold_$i.err
python bench.py --no-cpu-baseline --no-extras --steps 20 --no-fold > $O/bench_ref_$i.json 2>$O/bench_ref_$i.err
done
python bench.py --no-cpu-baseline --no-extras --steps 10 --no-fold --fill survey > $O/bench_ref_survey.json 2>$O/bench_ref_survey.err
python bench.py --no-cpu-baseline --no-extras --steps 10 --fill survey > $O/bench_fold_survey.json 2>$O/bench_fold_survey.err
tail -n 30 $O/trained_like.txt
cat $O/bench_*.json | cut -c1-400
