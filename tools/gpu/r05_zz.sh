#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r05zz; mkdir -p $O
S=$(date +%s)
timeout 170 python bench.py > $O/bench_noflags.json 2> $O/bench_noflags.err
echo "rc=$? wall $(( $(date +%s) - S )) s, lines $(wc -l < $O/bench_noflags.json)"
python -c "
import json; d=json.load(open('$O/bench_noflags.json')); print(d['metric'], d['value'], d['unit'], d['n_gpus'], d['steps'], d['warmup'], round(d['ms_per_step'],1), d['roofline']['bound'], round(d['roofline']['frac'],3), d['roofline']['traffic'], d['cpu_baseline']['value'], d['fallback'], d['missing_values'])"
