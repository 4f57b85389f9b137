"""One process per GPU: communicator plumbing for the row-partitioned engine.

The reference scales through PETSc's MPI communicator (``PCDKSP(comm)``,
``fenapack/field_split.py:46-57``; explicit collectives at
``SubfieldBC.h:110-111,138-140,178-181``).  Here the communicator is RCCL over
xGMI inside the engine; ``torch.distributed`` only carries the 128-byte
``ncclUniqueId`` from rank 0 to the others at start-up (plumbing)."""

import os


class Comm(object):
    """Rank/size of this process and the shared RCCL unique id."""

    def __init__(self, rank=0, size=1, unique_id=None, thread_group=None):
        self.rank, self.size, self._id = int(rank), int(size), unique_id
        #: ``ctypes.c_void_p`` shared by R engines of ONE process: the ranks
        #: are threads on one GPU (``pcd_comm_init_threads``; tests only)
        self.thread_group = thread_group

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
