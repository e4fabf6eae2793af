#!/bin/bash
# round 6: the k = 10 solve of snp_autoSVD at 400K x 1M — vectors per pass: the default (8) against 10, 12, 16
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06blk; mkdir -p $O
for blk in 0 10 12 16 0; do
  timeout 600 python bench.py --k 10 --block $blk --steps 6 --warmup 2 --no-cpu-baseline --no-ingest --no-wide --no-cold --no-autosvd > $O/k10_block_$blk.json 2> $O/k10_block_$blk.err
  python - <<P
import json
d=json.loads(open('$O/k10_block_$blk.json').read().strip().splitlines()[-1]); r=d['roofline']
print('block $blk: %.1f ms' % d['ms_per_step'], 'passes', d['config'].get('passes'), 'block', d['config'].get('block'), 'niter', d['config'].get('niter'),
      {k:(round(v['avg_ms'],2), v['launches'], v['column_blocks']) for k,v in r['other'].items()},
      'u/v lead', d.get('accuracy',{}).get('u_leading_half'), d.get('accuracy',{}).get('v_leading_half'), 'all', d.get('accuracy',{}).get('u_all'))
P
done
