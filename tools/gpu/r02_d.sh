#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
BSN_BENCH_NO_TORCH=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest > $O/b1.json 2> $O/b1.err; echo "notorch rc=$?"; tail -8 $O/b1.err
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest > $O/b2.json 2> $O/b2.err; echo "torch rc=$?"; tail -8 $O/b2.err
BSN_BENCH_NO_TORCH=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest --force-dist > $O/b3.json 2> $O/b3.err; echo "notorch dist rc=$?"; tail -8 $O/b3.err
BSN_BENCH_NO_TORCH=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest --n 40000 --m 100000 > $O/b4.json 2> $O/b4.err; echo "small rc=$?"; tail -4 $O/b4.err
timeout 600 python -m pytest tests/test_gpu_ld.py tests/test_gpu_complete_data.py tests/test_gpu_sct.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -5
BSN_BENCH_NO_TORCH=1 timeout 300 python bench.py --workload ld --steps 3 --warmup 1 > $O/ld.json 2> $O/ld.err; echo "ld rc=$?"; cat $O/ld.json | cut -c1-1500; tail -3 $O/ld.err
