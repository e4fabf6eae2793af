"""Worker of tests/test_gpu_sharded_svd.py: one rank of a column-sharded bed_randomSVD.
All ranks share GPU 0 (the test box has one GPU, and RCCL refuses two ranks on one device), so the
collectives of the sample-block layout (reduce-scatter of the panel, small Gram all-reduces, all-gather
of the finished basis block) go through the library's `allreduce` hook with a gloo all-reduce on a host
copy; with >= 2 GPUs the same code path runs on RCCL inside the library (bench.py --gpus N,
tests/test_gpu_comm.py).  Usage (via torch.distributed.run): worker.py n m k out.json"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist


def main():
    n, m, k, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    tol = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-9
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bigsnpr_amd as ba
    from bigsnpr_amd import _lib
    L = _lib.load()
    j0, j1 = (m * rank) // world, (m * (rank + 1)) // world
    gb = ba.bed.synthetic(n, j1 - j0, seed=31, j_begin=j0)

    def allreduce(ptr, count):
        host = np.empty(count)
        _lib.check(L.bsn_memcpy_d2h(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), count * 8))
        t = torch.from_numpy(host)
        dist.all_reduce(t)
        _lib.check(L.bsn_memcpy_h2d(C.c_void_p(ptr), host.ctypes.data_as(C.c_void_p), count * 8))

    res = ba.bed_randomSVD(gb, k=k, tol=tol, allreduce=allreduce, rank=rank, world=world, m_total=m)
    # every rank must have taken the same decisions and hold the same d and u
    d_all = [None] * world
    dist.all_gather_object(d_all, (res["d"].tolist(), res["niter"], float(np.abs(res["u"]).sum())))
    v_all = [None] * world
    dist.all_gather_object(v_all, res["v"])
    if rank == 0:
        json.dump(dict(d=res["d"].tolist(), niter=res["niter"], warm_launches=res["warm_launches"],
                       same=all(x == d_all[0] for x in d_all),
                       v=np.concatenate(v_all, axis=0).tolist() if m <= 50000 else [],
                       u0=res["u"][:, 0].tolist()), open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
