#!/bin/bash
# usage: tools/gpu/run.sh <timeout-seconds> <script>   — builds both libraries HERE first (the snapshot
# carries the .so files to the GPU box), then runs the script there
set -e
cd "$(dirname "$0")/../.."
python -m bigsnpr_amd.build > /dev/null
python -m bigsnpr_amd.build --ablation > /dev/null
(cd oracle && make -s)
python -c "import sys; sys.path.insert(0, 'tests/native'); import build_native; build_native.build(); build_native.build_mock_rccl()"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "bash $2"
