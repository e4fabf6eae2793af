"""Worker of tests/test_gpu_comm.py::test_ranks_over_the_collective_calls: one rank (= one GPU) of a column-sharded
bed_randomSVD whose exchange runs inside the library over RCCL.  torch.distributed (gloo) only carries
the unique id.  Usage (via torch.distributed.run): rccl_svd_worker.py n m k out.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (first: the library binds to the HIP runtime torch brings)
import torch.distributed as dist


def main():
    n, m, k, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bigsnpr_amd as ba
    from bigsnpr_amd import _lib
    # one GPU per rank; with BSN_RCCL_LIBRARY naming the shared-memory stand-in the ranks share what there is
    _lib.check(_lib.load().bsn_set_device(int(os.environ.get("LOCAL_RANK", rank)) % ba.device_count()))
    uid = [ba.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    comm = ba.Comm(uid[0], rank, world)
    j0, j1 = (m * rank) // world, (m * (rank + 1)) // world
    gb = ba.bed.synthetic(n, j1 - j0, seed=31, j_begin=j0)
    # BSN_TEST_BLOCK / BSN_TEST_TOL: the configuration of the segmented product pass test (16-vector blocks)
    res = ba.bed_randomSVD(gb, k=k, tol=float(os.environ.get("BSN_TEST_TOL", "1e-9")), comm=comm, m_total=m,
                           block=int(os.environ.get("BSN_TEST_BLOCK", "0")))
    d_all = [None] * world
    dist.all_gather_object(d_all, (res["d"].tolist(), res["niter"], float(np.abs(res["u"]).sum())))
    # one-shot product of the column shards: x is the same seeded vector on every rank, each takes its slice
    x = np.random.default_rng(7).normal(size=m)
    y = ba.bed_prodVec(gb, x[j0:j1], center=res["center"], scale=res["scale"], comm=comm)
    if rank == 0:
        json.dump(dict(d=res["d"].tolist(), niter=res["niter"], same=all(x_ == d_all[0] for x_ in d_all),
                       y=y.tolist(), usum=float(np.abs(res["u"]).sum()), vsum=float(np.abs(res["v"]).sum()),
                       segmented_passes=res["segmented_passes"], compact_gathers=res["compact_gathers"], tiled=res["tiled"]), open(out, "w"))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
