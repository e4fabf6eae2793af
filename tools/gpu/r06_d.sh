#!/bin/bash
# round 6, trip D: slabs of k_prodT on the column shards of a 2 / 4 / 8-GPU run — profiling build, BSN_KY_T
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06d; mkdir -p $O
export BSN_LIB_PATH=$GRAFT_REPO_ROOT/bigsnpr_amd/libbigsnpr_hip_abl.so
for N in 8 4 2; do
  for KY in 0 2 3 4 5 6 8 10 12 14; do
    if [ $KY = 0 ]; then unset BSN_KY_T; else export BSN_KY_T=$KY; fi
    timeout 200 python bench.py --steps 6 --warmup 2 --shard-of $N --no-cpu-baseline --no-ingest --no-wide --no-accuracy 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
o=d['roofline']['other']
print('N=$N ky=$KY ms/solve %.2f  wide_prod %.3f prod %.3f wide_cprod %.3f cprod %.3f stats %.3f' % (d['ms_per_step'], o['wide_prod']['avg_ms'], o['prod']['avg_ms'], o['wide_cprod']['avg_ms'], o['cprod']['avg_ms'], o['cprod_stats']['avg_ms']))
" >> $O/slab_sweep.txt
  done
done
cat $O/slab_sweep.txt
