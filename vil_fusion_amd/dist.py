"""Multi-GPU: independent window snapshots / sequence segments shard over ranks (one process per GPU); no intra-solve
communication. The only exchange is the gather of newest-frame poses [stamp x y z qx qy qz qw] (64 B per solved window) that
feeds the global_fusion pose graph (src/global_fusion/poseGraphOptimization.cpp:116-121 expects position + quaternion + stamp).
Backend "nccl" is RCCL on ROCm (xGMI point-to-point links; the message is latency-bound), "gloo" on CPU for tests.
"""
import numpy as np


def shard_range(n_units, rank, world_size):
    """contiguous block assignment: units [lo, hi) of rank; sizes differ by at most one, every unit assigned exactly once"""
    base, rem = divmod(n_units, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_counts(n_units, world_size):
    return [shard_range(n_units, r, world_size)[1] - shard_range(n_units, r, world_size)[0] for r in range(world_size)]


def gather_poses(local_poses, n_units=None, group=None):
    """All-gather of per-rank pose rows (torch tensor [n_local, 8], float64, on the backend's device).
    Equal shard sizes use one all_gather_into_tensor (single RCCL call); ragged shards pad to the largest shard.
    Returns a [n_total, 8] tensor ordered by rank then local index (= global unit order under shard_range)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n_local = local_poses.shape[0]
    if n_units is None:
        out = torch.empty((world * n_local, 8), dtype=local_poses.dtype, device=local_poses.device)
        dist.all_gather_into_tensor(out, local_poses.contiguous(), group=group)
        return out
    counts = shard_counts(n_units, world)
    m = max(counts)
    padded = torch.zeros((m, 8), dtype=local_poses.dtype, device=local_poses.device)
    padded[:n_local] = local_poses
    out = torch.empty((world * m, 8), dtype=local_poses.dtype, device=local_poses.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * m: r * m + counts[r]] for r in range(world)], dim=0)


def newest_poses_rows(results, stamps):
    """[stamp x y z qx qy qz qw] rows from solved windows (host side, for CPU tests): newest frame = last frame of the window."""
    from .synth import R_to_q
    rows = np.zeros((len(results), 8))
    for i, r in enumerate(results):
        rows[i, 0] = stamps[i]
        rows[i, 1:4] = r.Ps[-1]
        rows[i, 4:8] = R_to_q(r.Rs[-1])
    return rows


class RcclPoseGather:
    """ctypes mirror of the C-ABI collective (include/vilfusion.h: vilf_comm_* / vilf_gather_poses): what a C++ / ROS estimator process per GPU calls instead
    of torch.distributed. rank 0 creates the id (ncclGetUniqueId) and the launcher hands it to the other ranks (here: any byte channel)."""

    def __init__(self, world_size, rank, device=0, unique_id=None):
        import ctypes as C
        from . import lib as vlib
        self._L = vlib.lib()
        self._L.vilf_comm_last_error.restype = C.c_char_p
        self._L.vilf_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        self._L.vilf_comm_destroy.argtypes = [C.c_void_p]
        self._L.vilf_gather_poses.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self._L.vilf_gather_poses_handle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self._L.vilf_comm_ranks.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        if unique_id is None:
            unique_id = self.unique_id()
        self.world_size, self.rank = world_size, rank
        self._c = C.c_void_p()
        rc = self._L.vilf_comm_create(bytes(unique_id), world_size, rank, device, C.byref(self._c))
        if rc != 0:
            raise RuntimeError(f"vilf_comm_create: {rc}: {self._L.vilf_comm_last_error().decode()}")

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import lib as vlib
        L = vlib.lib()
        buf = (C.c_ubyte * 128)()
        L.vilf_comm_last_error.restype = C.c_char_p
        rc = L.vilf_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"vilf_comm_unique_id: {rc}: {L.vilf_comm_last_error().decode()}")
        return bytes(buf)

    def gather(self, local_dev_ptr, n_local, out_dev_ptr, stream=None):
        rc = self._L.vilf_gather_poses(self._c, stream, local_dev_ptr, n_local, out_dev_ptr)
        if rc != 0:
            raise RuntimeError(f"vilf_gather_poses: {rc}: {self._L.vilf_comm_last_error().decode()}")

    def gather_handle(self, solver, local_dev_ptr, n_local, out_dev_ptr):
        """vilf_gather_poses_handle: the all-gather on the stream `solver` (a BackendSolver) enqueues its work on — ordered behind newest_poses_to_device and an
        asynchronous batch_solve of that handle whatever stream it owns (the library's own streams are non-blocking: the NULL stream does not wait for them)."""
        rc = self._L.vilf_gather_poses_handle(self._c, solver._h, local_dev_ptr, n_local, out_dev_ptr)
        if rc != 0:
            raise RuntimeError(f"vilf_gather_poses_handle: {rc}: {self._L.vilf_comm_last_error().decode()}")

    def ranks(self):
        """(world size, rank) as RCCL reports them for this communicator (ncclCommCount / ncclCommUserRank)"""
        import ctypes as C
        n, me = C.c_int(0), C.c_int(0)
        rc = self._L.vilf_comm_ranks(self._c, C.byref(n), C.byref(me))
        if rc != 0:
            raise RuntimeError(f"vilf_comm_ranks: {rc}: {self._L.vilf_comm_last_error().decode()}")
        return n.value, me.value

    def close(self):
        if self._c:
            self._L.vilf_comm_destroy(self._c)
            self._c = None
