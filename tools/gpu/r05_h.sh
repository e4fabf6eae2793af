#!/bin/bash
# round 5, trip H: the two regressions, same box.  (a) C2 one-shot bed_prodVec / bed_cprodVec calls:
# round 3's build, round 4's build and this build alternated three times; (b) bed_cor / bed_ld_scores at C5.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $O; : > $O/summary.txt
for rep in 1 2 3; do
  for tree in ab/r03 ab/r04 .; do
    tag=$(basename $tree); [ "$tree" = "." ] && tag=r05
    (cd $tree && timeout 300 python bench.py --workload matvec --steps 200 --warmup 20 --no-cpu-baseline > $O/c2_${tag}_$rep.json 2> $O/c2_${tag}_$rep.err)
    python -c "
import json; d=json.load(open('$O/c2_${tag}_$rep.json')); print('C2 one-shot calls, build $tag, repetition $rep: %.4f ms per call, whole-call fraction of HBM peak %.3f' % (d['ms_per_call'], d['roofline']['frac']))" | tee -a $O/summary.txt
  done
done
export PYTHONPATH="$GRAFT_REPO_ROOT"
timeout 600 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld_bench.json 2> $O/ld_bench.err
python -c "
import json; d=json.load(open('$O/ld_bench.json')); print("ld bench:", {k: d[k] for k in d if k.endswith("_ms") or k in ("value","ms_per_step")}, d["roofline"]["frac"])" | tee -a $O/summary.txt
python tools/probe_ld.py 2>&1 | tee -a $O/summary.txt
