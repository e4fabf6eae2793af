"""Comm — the RCCL communicator of a column-sharded solve (include/bigsnpr_hip.h, bsn_comm_*).

One process per GPU.  Rank 0 draws the unique id, the host program carries its 128 bytes to the
other ranks (here: any callable, e.g. a torch.distributed / MPI broadcast), every rank calls
Comm(id, rank, world) after selecting its device; bed_randomSVD(..., comm=comm, m_total=...) then
exchanges its panels inside the library.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, u8p, vp


class Comm:
    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = np.zeros(Comm.ID_BYTES, dtype=np.uint8)
        check(_lib.load().bsn_comm_unique_id(buf.ctypes.data_as(u8p)))
        return buf.tobytes()

    def __init__(self, uid, rank, world):
        buf = np.frombuffer(bytes(uid), dtype=np.uint8).copy()
        if buf.size != Comm.ID_BYTES:
            raise ValueError("the unique id has %d bytes" % Comm.ID_BYTES)
        h = vp()
        check(_lib.load().bsn_comm_init(buf.ctypes.data_as(u8p), int(rank), int(world), C.byref(h)))
        self.handle, self.rank, self.world = h, int(rank), int(world)

    def allreduce(self, dev_array):
        """in-place sum of a DeviceArray over the ranks (blocking)"""
        check(_lib.load().bsn_comm_allreduce(self.handle, dev_array.ptr, dev_array.rows * dev_array.cols))

    def close(self):
        if self.handle is not None and self.handle.value:
            _lib.load().bsn_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- choosing the exchange of a sharded solve on first contact with a transport (round 5) ---------------------------
# The product pass of a sharded bed_randomSVD can exchange its panel in three ways, fastest first:
#   "overlap"     segments of sample blocks, each segment's reduce-scatter on a second stream while the next computes
#   "one_stream"  the same segments, every collective on the solve's own stream            (BSN_NO_OVERLAP=1)
#   "whole"       the whole pass, then ONE reduce-scatter: the pattern of rounds 2 - 3      (BSN_NO_SEGMENTS=1)
# All three add the same numbers in the same order: d, u, v are bit-identical.  Which of them a given RCCL / driver /
# topology sustains is only known once it has run there, so negotiate() runs a MINIATURE sharded solve in each mode
# under the library's watchdog (bsn_svd_options.exchange_timeout_ms: a collective that never completes ends the call
# with an error and an aborted communicator instead of a hang), compares it bit for bit with the most conservative
# mode, lets the ranks agree through the caller's host channel, and leaves the environment set to the fastest mode that
# passed on EVERY rank — with a fresh communicator whenever one was aborted.
EXCHANGE_MODES = (("overlap", {}), ("one_stream", {"BSN_NO_OVERLAP": "1"}), ("whole", {"BSN_NO_SEGMENTS": "1"}))
_MODE_VARS = ("BSN_NO_OVERLAP", "BSN_NO_SEGMENTS")


def set_exchange_mode(name):
    """sets the environment switches of one of EXCHANGE_MODES (read by the library at every product pass)"""
    import os
    env = dict(EXCHANGE_MODES)[name]
    for var in _MODE_VARS:
        if var in env:
            os.environ[var] = env[var]
        else:
            os.environ.pop(var, None)


def negotiate(rank, world, bcast, agree_min, modes=None, timeout_ms=30000, samples_per_rank=4096,
              variants_per_rank=8192, k=20, log=None):
    """Returns (comm, report).  `bcast(obj)` returns rank 0's object on every rank; `agree_min(i)` returns the minimum
    of the ranks' integers (both over a host channel that works: torch.distributed / gloo, MPI, a socket).  The ranks
    must call this collectively, each with its own device selected.  report = {"mode": chosen, "tried": [{"mode", "ok",
    "ms", "error"} ...]}; raises RuntimeError when not even the whole-pass exchange works on every rank."""
    import time
    from .bed import bed as _bed_class
    from .svd import bed_randomSVD

    log = log or (lambda msg: None)

    def fresh():
        # every rank reaches every agreement point, whatever fails where: a rank that raised early would leave the
        # others waiting for it in the host channel
        uid, err = None, None
        if rank == 0:
            try:
                uid = Comm.unique_id()
            except Exception as e:      # RCCL cannot be loaded
                err = str(e)
        uid, err = bcast((uid, err))
        comm = None
        if uid is not None:
            try:
                comm = Comm(uid, rank, world)
            except Exception as e:
                err = str(e)
        if not agree_min(1 if comm is not None else 0):
            if comm is not None:
                comm.close()
            raise RuntimeError("no RCCL communicator over %d ranks: %s" % (world, err or "another rank failed"))
        return comm

    n, m_loc = samples_per_rank * world, variants_per_rank
    gb = _bed_class.synthetic(n, m_loc, seed=77, j_begin=rank * m_loc)

    def attempt(comm, mode):
        set_exchange_mode(mode)
        t0 = time.perf_counter()
        try:
            r = bed_randomSVD(gb, k=k, block=16, comm=comm, m_total=m_loc * world, exchange_timeout_ms=timeout_ms)
            sig = (r["d"].tobytes(), float(np.abs(r["u"]).sum()), float(np.abs(r["v"]).sum()), r["exchange_mode"])
            return sig, None, 1e3 * (time.perf_counter() - t0)
        except Exception as e:   # BsnError: a collective failed, or the watchdog aborted the communicator
            return None, str(e), 1e3 * (time.perf_counter() - t0)

    order = [mname for mname, _ in EXCHANGE_MODES if modes is None or mname in modes]
    if "whole" not in order:
        order.append("whole")
    tried = []
    comm = fresh()
    try:
        # the reference: the plainest pattern.  If that does not work on every rank nothing will.
        ref, err, ms = attempt(comm, "whole")
        ok = agree_min(1 if ref is not None else 0)
        tried.append(dict(mode="whole", ok=bool(ok), ms=ms, error=err))
        if not ok:
            raise RuntimeError("the sharded solve does not run over this communicator even with the whole-pass exchange: %s" % err)
        chosen = "whole"
        for mode in order:
            if mode == "whole":
                break
            sig, err, ms = attempt(comm, mode)
            same = sig is not None and sig[:3] == ref[:3]
            if sig is not None and not same:
                err = "results differ from the whole-pass exchange"
            ok = agree_min(1 if same else 0)
            tried.append(dict(mode=mode, ok=bool(ok), ms=ms, error=err, took=None if sig is None else sig[3]))
            log("exchange mode %-10s %s (%.0f ms)%s" % (mode, "ok" if ok else "FAILED", ms, "" if not err else ": " + err[:200]))
            if ok:
                chosen = mode
                break
            # a rank whose watchdog fired holds a dead communicator; the others must not wait for it in a collective:
            # everybody starts over with a new one
            comm.close()
            comm = fresh()
        set_exchange_mode(chosen)
        return comm, dict(mode=chosen, tried=tried, timeout_ms=timeout_ms)
    except Exception:
        comm.close()
        raise
    finally:
        gb.close()
