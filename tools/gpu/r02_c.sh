#!/bin/bash
# round 2, trip C: parity suite with the sample-block / RCCL backend, bench with and without the communicator
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -12 $O/pytest.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench.json
timeout 600 python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline --no-ingest > $O/bench_dist.json 2> $O/bench_dist.err; echo "bench dist rc=$?"; tail -c 600 $O/bench_dist.json; tail -5 $O/bench_dist.err
BSN_TIMING=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest --verbose 2 > $O/bench_t.json 2> $O/bench_t.err; grep -v "step " $O/bench_t.err | tail -20
