"""Host-staged collectives over torch.distributed (gloo) for `trc_group_set_collectives` -- harness code for tests and
for `bench.py --gpus N` on a box with fewer than N GPUs (RCCL refuses two ranks on one device).  The library stages the
device buffer through pinned host memory and hands these callbacks a host pointer; they run the collective in place.

Nothing here computes anything of the path: reduce(sum) with zeros outside a rank's own tiles is a gather, min / max of
order-preserving keys are exact, the all-gather moves bytes.
"""
import ctypes as C

import numpy as np

from . import abi

from .socket_group import (ALLGATHER_FN, ALLREDUCE_FN, ALLTOALL_FN, GATHER_FN, REDUCE_FN, Collectives,  # noqa: F401  (one ctypes mirror
                           _NP)                                                                       # of trc_collectives)


class GlooCollectives:
    """table = GlooCollectives(group=None); tracer.set_collectives(table, world, rank).  Keeps the ctypes callbacks
    alive; `calls` counts what the library asked for (tests assert the collective program)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self._group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.calls = {"reduce": 0, "allreduce": 0, "allgather": 0, "alltoall": 0, "gather": 0}
        self._cb = (REDUCE_FN(self._reduce), ALLREDUCE_FN(self._allreduce), ALLGATHER_FN(self._allgather),
                    ALLTOALL_FN(self._alltoall), GATHER_FN(self._gather))
        self.table = Collectives(None, 1, 0, *self._cb)

    def _global(self, root):
        """torch.distributed's dst= / src= are GLOBAL ranks; the table speaks of ranks of ITS group"""
        return root if self._group is None else self._dist.get_global_rank(self._group, root)

    def _view(self, buf, count, dtype):
        return np.ctypeslib.as_array(C.cast(buf, C.POINTER(np.ctypeslib.as_ctypes_type(_NP[dtype]))), shape=(count,))

    def _op(self, op):
        R = self._dist.ReduceOp
        return {abi.OP_SUM: R.SUM, abi.OP_MAX: R.MAX, abi.OP_MIN: R.MIN}[op]

    def _tensor(self, a):
        # gloo has no unsigned 32-bit reductions: widen (values are exact in int64), narrow on the way back
        if a.dtype == np.uint32:
            return self._torch.from_numpy(a.astype(np.int64)), True
        return self._torch.from_numpy(a), False

    def _reduce(self, user, buf, count, dtype, op, root, stream):
        try:
            a = self._view(buf, count, dtype)
            t, widened = self._tensor(a)
            self._dist.reduce(t, dst=self._global(root), op=self._op(op), group=self._group)
            if widened and self.rank == root:
                a[:] = t.numpy().astype(a.dtype)
            self.calls["reduce"] += 1
            return 0
        except Exception as e:      # never let an exception cross the C boundary
            print(f"GlooCollectives.reduce: {e!r}", flush=True)
            return 1

    def _allreduce(self, user, buf, count, dtype, op, stream):
        try:
            a = self._view(buf, count, dtype)
            t, widened = self._tensor(a)
            self._dist.all_reduce(t, op=self._op(op), group=self._group)
            if widened:
                a[:] = t.numpy().astype(a.dtype)
            self.calls["allreduce"] += 1
            return 0
        except Exception as e:
            print(f"GlooCollectives.allreduce: {e!r}", flush=True)
            return 1

    def _allgather(self, user, buf, bytes_per_rank, stream):
        try:
            a = self._view(buf, bytes_per_rank * self.world, abi.DT_U8)
            mine = self._torch.from_numpy(a[self.rank * bytes_per_rank:(self.rank + 1) * bytes_per_rank].copy())
            parts = [self._torch.empty(bytes_per_rank, dtype=self._torch.uint8) for _ in range(self.world)]
            self._dist.all_gather(parts, mine, group=self._group)
            for r, p in enumerate(parts):
                a[r * bytes_per_rank:(r + 1) * bytes_per_rank] = p.numpy()
            self.calls["allgather"] += 1
            return 0
        except Exception as e:
            print(f"GlooCollectives.allgather: {e!r}", flush=True)
            return 1

    def _alltoall(self, user, buf, bytes_per_rank, stream):
        try:
            a = self._view(buf, bytes_per_rank * self.world, abi.DT_U8)
            send = [self._torch.from_numpy(a[r * bytes_per_rank:(r + 1) * bytes_per_rank].copy()) for r in range(self.world)]
            recv = [self._torch.empty(bytes_per_rank, dtype=self._torch.uint8) for _ in range(self.world)]
            # gloo has no all_to_all: one all_gather per destination slice would move N times the data; every pair's send and
            # receive are POSTED (isend / irecv) before anything is waited for, so no order of the ranks can deadlock
            reqs = []
            for p in range(self.world):
                if p == self.rank:
                    recv[p] = send[p]
                    continue
                reqs.append(self._dist.isend(send[p], dst=self._global(p), group=self._group))
                reqs.append(self._dist.irecv(recv[p], src=self._global(p), group=self._group))
            for q in reqs:
                q.wait()
            for r, t in enumerate(recv):
                a[r * bytes_per_rank:(r + 1) * bytes_per_rank] = t.numpy()
            self.calls["alltoall"] += 1
            return 0
        except Exception as e:
            print(f"GlooCollectives.alltoall: {e!r}", flush=True)
            return 1

    def _gather(self, user, buf, bytes_per_rank, root, stream):
        try:
            a = self._view(buf, bytes_per_rank * self.world, abi.DT_U8)
            mine = self._torch.from_numpy(a[self.rank * bytes_per_rank:(self.rank + 1) * bytes_per_rank].copy())
            parts = [self._torch.empty(bytes_per_rank, dtype=self._torch.uint8) for _ in range(self.world)] if self.rank == root else None
            self._dist.gather(mine, parts, dst=self._global(root), group=self._group)
            if self.rank == root:
                for r, t in enumerate(parts):
                    a[r * bytes_per_rank:(r + 1) * bytes_per_rank] = t.numpy()
            self.calls["gather"] += 1
            return 0
        except Exception as e:
            print(f"GlooCollectives.gather: {e!r}", flush=True)
            return 1
