"""The per-iteration exchanges of the shared-point records as grouped ncclSend / ncclRecv issued ON THE STREAM THE ENGINE
COMPUTES ON (the RCCL of the running torch, bound with ctypes) -- the arrangement of the C++ front-end
(csrc/host/smoothMesh_main.cpp) for the Python driver.

Why not torch.distributed.all_to_all_single for these: ProcessGroupNCCL runs every collective on a stream of its own and
orders it against the caller's stream with two events.  On MI355X each of those cross-stream dependencies is ~14 us of idle
GPU time behind the 9 us collective kernel (rocprofv3 kernel trace of scripts/probe_rank_of_8.py), twice per iteration:
a third of the multi-rank overhead of a 1 M-point sub-domain.  A send / recv group on the engine's own stream is ordered by
the stream itself.

The communicator is created from a unique id broadcast through the existing process group.  Before it is used, one exchange
of a rank-specific pattern is compared with all_to_all_single on every rank (`self_check`); any difference, or any RCCL
error, makes all ranks keep the torch path.  SMOOTHMESH_EXCHANGE=torch skips this module altogether.
"""
import ctypes as C
import os

NCCL_INT8 = 0      # ncclDataType_t: counts below are bytes


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * 128)]      # (a c_char array field would read back truncated at the first NUL)


def _find_library(torch):
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so"]
    for c in cands:
        if os.path.sep not in c or os.path.exists(c):
            try:
                return C.CDLL(c)
            except OSError:
                continue
    raise OSError("librccl.so not found")


class RcclDirect:
    def __init__(self, torch, dist, device):
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.lib = lib = _find_library(torch)
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclGetErrorString.restype = C.c_char_p
        for f in ("ncclGetUniqueId", "ncclCommInitRank", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd", "ncclCommDestroy"):
            getattr(lib, f).restype = C.c_int
        uid = _UniqueId()
        if self.rank == 0:
            self._ok(lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        box = [C.string_at(C.byref(uid), 128)] if self.rank == 0 else [None]
        dist.broadcast_object_list(box, src=0)
        C.memmove(C.byref(uid), box[0], 128)
        self.comm = C.c_void_p()
        with torch.cuda.device(device):
            self._ok(lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
        # what the communicator itself says about its size (reported by bench.py next to the transport: a line measured with
        # fewer ranks in the communicator than GPUs in the job would be a different experiment)
        lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.ncclCommCount.restype = C.c_int
        n = C.c_int(-1)
        self._ok(lib.ncclCommCount(self.comm, C.byref(n)), "ncclCommCount")
        self.ranks_seen = int(n.value)
        if self.ranks_seen != self.world:
            raise RuntimeError(f"ncclCommCount = {self.ranks_seen}, world size = {self.world}")

    def _ok(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.ncclGetErrorString(rc).decode()}")

    def exchange(self, recv_ptr, send_ptr, counts, elem_bytes, stream):
        """records of elem_bytes each, counts[r] of them to and from rank r, packed in rank order in both buffers"""
        self.exchange_many([(recv_ptr, send_ptr, elem_bytes)], counts, stream)

    def exchange_many(self, buffers, counts, stream):
        """several record sets (recv_ptr, send_ptr, elem_bytes) with the same per-rank counts in ONE send / recv group
        (exchange L and exchange A leave at the same point of the iteration: one collective kernel instead of two)"""
        lib = self.lib
        self._ok(lib.ncclGroupStart(), "ncclGroupStart")
        for recv_ptr, send_ptr, elem_bytes in buffers:
            off = 0
            for r, c in enumerate(counts):
                if c:
                    nb = c * elem_bytes
                    self._ok(lib.ncclSend(send_ptr + off, nb, NCCL_INT8, r, self.comm, stream), "ncclSend")
                    self._ok(lib.ncclRecv(recv_ptr + off, nb, NCCL_INT8, r, self.comm, stream), "ncclRecv")
                    off += nb
        self._ok(lib.ncclGroupEnd(), "ncclGroupEnd")

    def self_check(self, counts, device):
        """one exchange of a rank-specific pattern both ways; True on every rank or on none"""
        torch, dist = self.torch, self.dist
        n = int(sum(counts))
        good = 1
        try:
            send = (torch.arange(max(n, 1) * 3, dtype=torch.float64, device=device) * (self.rank + 1) + 0.25 * self.rank).reshape(-1, 3)[:n]
            ref = torch.zeros_like(send)
            got = torch.full_like(send, -1.0)
            if n or self.world > 1:      # (a collective: ranks without shared points take part with zero counts)
                dist.all_to_all_single(ref, send.contiguous(), counts, counts)
            s = torch.cuda.current_stream(device)
            self.exchange(got.data_ptr(), send.data_ptr(), counts, 24, s.cuda_stream)
            s.synchronize()
            if n and not torch.equal(ref, got):
                good = 0
        except Exception:      # noqa: BLE001 -- any failure means: keep the torch path
            good = 0
        flag = torch.tensor([good], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def close(self):
        if getattr(self, "comm", None):
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass
