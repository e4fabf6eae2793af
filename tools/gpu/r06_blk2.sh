#!/bin/bash
# round 6: the k = 10 solve (8 vectors per pass) on the sample-major copy (the default since round 5 whenever wide panels MAY
# come), on the tiled copy (BSN_NO_SMAJ=1: round 4's layout for one-block kernels), and on the plain image alone
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06blk2; mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py --k 10 --steps 6 --warmup 2 --no-cpu-baseline --no-ingest --no-wide --no-cold --no-autosvd > $O/k10_$tag.json 2> $O/k10_$tag.err
  python - <<P
import json
d=json.loads(open('$O/k10_$tag.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$tag: %.1f ms' % d['ms_per_step'], {k:(round(v['avg_ms'],2), v['launches'], v['column_blocks']) for k,v in r['other'].items()},
      'u/v lead', d.get('accuracy',{}).get('u_leading_half'), d.get('accuracy',{}).get('v_leading_half'), [v.split('(')[0][-40:] for v in r['kernels_launched'].values()])
P
}
run smaj X=1
run tiled BSN_NO_SMAJ=1
run plain BSN_NO_SMAJ=1 BSN_NO_TILED=1
run smaj2 X=1
