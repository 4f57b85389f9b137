"""One process per GPU: communicator plumbing for the row-partitioned engine.

The reference scales through PETSc's MPI communicator (``PCDKSP(comm)``,
``fenapack/field_split.py:46-57``; explicit collectives at
``SubfieldBC.h:110-111,138-140,178-181``).  Here the communicator is RCCL over
xGMI inside the engine; ``torch.distributed`` only carries the 128-byte
``ncclUniqueId`` from rank 0 to the others at start-up (plumbing)."""

import os


class Comm(object):
    """Rank/size of this process and the shared RCCL unique id."""

    def __init__(self, rank=0, size=1, unique_id=None, thread_group=None,
                 host_transport=None, stream=None):
        self.rank, self.size, self._id = int(rank), int(size), unique_id
        #: ``ctypes.c_void_p`` shared by R engines of ONE process: the ranks
        #: are threads on one GPU (``pcd_comm_init_threads``; tests only)
        self.thread_group = thread_group
        #: host transport (``pcd_comm_init_host``): an object with
        #: ``allreduce(array)`` / ``exchange(sends, recvs)`` - the bootstrap
        #: of the peer-write protocol where there is no RCCL communicator
        self.host_transport = host_transport
        #: raw HIP stream the engine of this rank launches on (thread ranks
        #: that use the peer protocol need one each)
        self.stream = stream

    def unique_id(self):
        return self._id

    @classmethod
    def world(cls):
        """From an initialised ``torch.distributed`` group (backend nccl on
        the GPU box); a single process gives the trivial communicator."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) \
                or dist.get_world_size() == 1:
            if os.environ.get("PCD_FORCE_COMM") == "1":
                from . import _cabi
                return cls(0, 1, _cabi.comm_unique_id())
            return cls(0, 1, None)
        from . import _cabi
        rank, size = dist.get_rank(), dist.get_world_size()
        box = [_cabi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(rank, size, box[0])


def local_device():
    return int(os.environ.get("LOCAL_RANK", "0"))


class TorchHostTransport(object):
    """``torch.distributed`` (gloo) as the host transport of
    ``pcd_comm_init_host``: the collectives of set-up; with the peer protocol
    on, nothing of the solve goes through it."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self._group = torch, dist, group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)

    def allreduce(self, a):
        t = self._torch.from_numpy(a)
        self._dist.all_reduce(t, group=self._group)

    def exchange(self, sends, recvs):
        dist, torch = self._dist, self._torch
        ops, keep = [], []
        for peer, a in recvs:
            if a.size:
                ops.append(dist.P2POp(dist.irecv, torch.from_numpy(a), peer,
                                      self._group))
        for peer, a in sends:
            if a.size:
                t = torch.from_numpy(a)
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, peer, self._group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
