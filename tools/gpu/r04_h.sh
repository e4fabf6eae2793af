#!/bin/bash
# round 4, trip H: LD kernel with an explicit MFMA : VALU schedule (BSN_LD_SGB = 0 / 1 / 2), C5 workload; out-of-core at 8 GB
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04h; mkdir -p $O
for g in 0 1 2 0 1; do BSN_LD_SGB=$g timeout 300 python bench.py --workload ld --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('BSN_LD_SGB=$g', round(d['ms_per_step'],1), 'ms', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('value','stats_kernel_ms','kernel','mfma_frac')}, json.dumps(d.get('roofline',{}))[:300])"; done | tee $O/ld_sgb.txt
timeout 600 python tools/probe_ooc.py --gb 8 --budget-gb 2 2>&1 | grep "^{" | tee $O/ooc.json
