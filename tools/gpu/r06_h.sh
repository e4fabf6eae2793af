#!/bin/bash
# round 6, trip H: out-of-core tests again; HIP API trace of the cold first solve (what its ~ 50 ms beyond a warm solve are)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_out_of_core.py tests/test_gpu_smaj.py tests/test_gpu_ld.py tests/test_gpu_sct.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
BSN_TIMING=1 BSN_ALLOC_TRACE=1 timeout 300 python tools/probe_cold.py > $O/cold_1.txt 2> $O/cold_1.err
cd /tmp
BSN_TIMING=1 timeout 300 rocprofv3 --hip-trace --output-format csv -d $O/trace -o cold -- python $R/tools/probe_cold.py --solves 2 > $O/cold_traced.txt 2> $O/cold_traced.err
cd $R
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r06h/trace/**/*hip_api_trace.csv', recursive=True)
print(f)
rows=list(csv.DictReader(open(f[0])))
print(len(rows), rows[0].keys())
t0=min(int(r['Start_Timestamp']) for r in rows)
# find the solve windows: roughly by the big gaps; print the slowest 40 calls with their start offsets
rows.sort(key=lambda r:int(r['End_Timestamp'])-int(r['Start_Timestamp']), reverse=True)
for r in rows[:45]:
    print("%-34s start %9.1f ms  dur %8.2f ms" % (r['Function'], (int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
PY
grep -h "solve_ms" $O/cold_1.txt $O/cold_traced.txt | cut -c1-160
grep -h "helper thread\|host wall" $O/cold_1.err $O/cold_traced.err | cut -c1-250
