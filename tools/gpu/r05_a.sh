#!/bin/bash
# round 5, trip A: first contact of the three-column-block kernels and the precision schedule
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
for f in tests/test_gpu_matvec.py tests/test_gpu_svd.py tests/test_gpu_smaj.py tests/test_gpu_comm.py; do
  timeout 600 python -m pytest $f -m gpu -q -x > $O/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' $O/$(basename $f .py).log | tail -1)" | tee -a $O/summary.txt
done
grep -n "FAILED\|^E " $O/*.log | head -20
# NB = 3 against the same sums in two launches: bit-identical d, u, v
cat > /tmp/ab.py <<'P'
import sys, numpy as np, bigsnpr_amd as ba
gb = ba.bed.synthetic(100000, 300000)
r = ba.bed_randomSVD(gb, k=20, slices=3, block=16, verbose=True)
np.save(sys.argv[1], np.concatenate([r["d"], r["u"].ravel(), r["v"].ravel()]))
print(r["niter"], r["n_wide_cprod"], r["n_wide_prod"], r["n_cprod"], r["n_prod"], r["wide_cprod_ms"], r["wide_prod_ms"])
P
python /tmp/ab.py /tmp/a.npy 2> $O/ab_verbose.txt; BSN_NO_NB3=1 python /tmp/ab.py /tmp/b.npy 2>/dev/null
python -c "
import numpy as np; a=np.load('/tmp/a.npy'); b=np.load('/tmp/b.npy'); print('NB3 vs split launches: identical =', bool(np.array_equal(a,b)), 'max diff', float(np.abs(a-b).max()))" | tee -a $O/summary.txt
python tools/probe_vectors.py > $O/vectors_c3.txt 2> $O/vectors_c3.err
python - <<'P' | tee -a $O/summary.txt
import json
for l in open('gpurun_out/r05a/vectors_c3.txt'):
    d = json.loads(l)
    if 'cfg' in d: print(d['cfg'], d['ms'], 'ms niter', d['niter'], 'wide', d['wide_steps'], d['n_wide'], d['wide_ms'], d['narrow_ms'], 'u lead/all %.1e %.1e v %.1e %.1e' % (d['u_lead'], d['u_all'], d['v_lead'], d['v_all']), 'resid', d['resid'])
    else: print(d)
P
python - <<'P' 2> $O/verbose_default.txt
import bigsnpr_amd as ba
gb = ba.bed.synthetic(400000, 1000000)
r = ba.bed_randomSVD(gb, k=20, verbose=True)
P
grep "bit products\|max rel resid" $O/verbose_default.txt | tee -a $O/summary.txt
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ingest --no-wide > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('bench default:', round(d['ms_per_step'],2), 'ms', d['roofline']['frac'], {k:round(v['avg_ms'],2) for k,v in d['roofline']['other'].items()})" | tee -a $O/summary.txt
