"""Worker of tests/test_gpu_comm.py::test_first_contact_negotiation: one rank of bigsnpr_amd.comm.negotiate (the
first-contact probe of a sharded solve's exchange) followed by a sharded solve in the mode it chose.
Usage (via torch.distributed.run): negotiate_worker.py n m k timeout_ms out.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
import torch.distributed as dist


def main():
    n, m, k, timeout_ms, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bigsnpr_amd as ba
    from bigsnpr_amd import _lib
    _lib.check(_lib.load().bsn_set_device(int(os.environ.get("LOCAL_RANK", rank)) % ba.device_count()))

    def bcast(obj):
        box = [obj]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def agree_min(i):
        t = torch.tensor([int(i)])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    comm, report = ba.negotiate_exchange(rank, world, bcast, agree_min, timeout_ms=timeout_ms)
    j0, j1 = (m * rank) // world, (m * (rank + 1)) // world
    gb = ba.bed.synthetic(n, j1 - j0, seed=31, j_begin=j0)
    res = ba.bed_randomSVD(gb, k=k, comm=comm, m_total=m, block=16, exchange_timing=True, exchange_timeout_ms=60000)
    if rank == 0:
        json.dump(dict(report=report, d=res["d"].tolist(), usum=float(np.abs(res["u"]).sum()), mode=res["exchange_mode"],
                       exchange_ms=res["exchange_ms"], n_exchange=res["n_exchange"],
                       env={v: os.environ.get(v) for v in ("BSN_NO_OVERLAP", "BSN_NO_SEGMENTS")}), open(out, "w"))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
