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
