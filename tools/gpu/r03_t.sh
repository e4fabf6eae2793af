cd "$GRAFT_REPO_ROOT"; timeout 600 python -m pytest tests/test_gpu_svd.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15
