"""Multi-GPU decomposition of the prover: point-range sharding of the five MSM bases + one exchange step.

One process per GPU (torchrun / torch.distributed gives rank, world size and the rendezvous).  Every rank
builds a cache holding only its range of each base array — the witness slice it uploads itself, [rank·⌈L/W⌉, (rank+1)·⌈L/W⌉), when that
leaves no rank empty, else [rank·L/W, (rank+1)·L/W) — (C++: build_cache in
csrc/prover/cache.cpp — `shard_range` below is the same arithmetic), runs the replicated QAP/NTT front end
and its five partial MSMs, then all ranks all-gather their 576-byte commitment blocks and sum them with
the group law.  The data-path collective is RCCL over xGMI (csrc/comm/rccl_comm.cpp) driven by this
library's own HIP runtime; torch.distributed (gloo) is only the control plane that broadcasts the
ncclUniqueId and provides barriers.  `GlooExchange` is the CPU stand-in used by the world_size-2 tests.
"""
from __future__ import annotations

import ctypes as C
import os

from . import binding as K

_HERE = os.path.dirname(os.path.abspath(__file__))
RCCL_LIB_PATH = os.path.join(_HERE, "lib", "libicicle_snark_rccl.so")


_rccl_lib = None


def preload_rccl():
    """Load libicicle_snark_rccl.so (and with it /opt/rocm's librccl.so.1, which sits on this library's HIP
    runtime).  MUST run before `import torch`: the torch wheel bundles its own librccl with the same soname,
    built for its own bundled HIP runtime; whichever is loaded first serves both, and ours must win."""
    global _rccl_lib
    if _rccl_lib is None:
        if not os.path.exists(RCCL_LIB_PATH):
            raise ImportError(f"{RCCL_LIB_PATH} is missing: run `make`")
        import sys
        if "torch" in sys.modules:
            maps = open("/proc/self/maps").read()
            if "torch/lib/librccl" in maps:
                raise RuntimeError("torch (and its bundled librccl) was imported before preload_rccl(); "
                                   "call icicle-snark_amd.parallel.preload_rccl() first")
        _rccl_lib = C.CDLL(RCCL_LIB_PATH)
        _rccl_lib.icicle_snark_rccl_last_error.restype = C.c_char_p
    return _rccl_lib


def shard_range(total: int, rank: int, world: int):
    """range of rank `rank` of `world` (csrc/prover/cache.cpp: build_cache): the slice of ⌈total / world⌉ elements the rank uploads
    itself when that leaves no rank empty, else the even split"""
    s = (total + world - 1) // world
    if world > 1 and s * (world - 1) < total:
        return min(total, s * rank), min(total, s * (rank + 1))
    return total * rank // world, total * (rank + 1) // world


class LocalExchange:
    """world size 1"""
    world, rank = 1, 0

    def allgather(self, block: bytes) -> bytes:
        return block

    def max(self, x: float) -> float:
        return x

    def allgather_device(self, ptr: int, slice_bytes: int):
        pass

    def barrier(self):
        pass

    def close(self):
        pass


class GlooExchange:
    """torch.distributed (any initialised backend) on host tensors — CPU tests and control plane."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def allgather(self, block: bytes) -> bytes:
        import torch
        t = torch.frombuffer(bytearray(block), dtype=torch.uint8)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return b"".join(bytes(o.numpy()) for o in outs)

    def max(self, x: float) -> float:
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def alltoall_rows(self, send_ptr: int, recv_ptr: int, rows: int, row_bytes: int, chunk_bytes: int):
        """stand-in for the RCCL all-to-all on device buffers: staged through the host (gloo has no all_to_all: every rank
        gathers every send buffer and keeps the chunks addressed to it)"""
        import torch
        mine = torch.frombuffer(bytearray(K.raw_to_host(send_ptr, rows * row_bytes)), dtype=torch.uint8)
        outs = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(outs, mine)
        recv = bytearray(rows * row_bytes)
        for q in range(rows):
            for p in range(self.world):
                src = q * row_bytes + self.rank * chunk_bytes          # what rank p addressed to this rank
                dst = q * row_bytes + p * chunk_bytes
                recv[dst:dst + chunk_bytes] = bytes(outs[p][src:src + chunk_bytes].numpy())
        K.raw_to_device(recv_ptr, bytes(recv))

    def allgather_device(self, ptr: int, slice_bytes: int):
        """stand-in for the in-place RCCL all-gather of a device buffer (world × slice_bytes): staged through the host"""
        import torch
        mine = torch.frombuffer(bytearray(K.raw_to_host(ptr + self.rank * slice_bytes, slice_bytes)), dtype=torch.uint8)
        outs = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(outs, mine)
        K.raw_to_device(ptr, b"".join(bytes(o.numpy()) for o in outs))

    def barrier(self):
        self.dist.barrier()

    def close(self):
        pass


class RcclExchange:
    """RCCL all-gather on this library's HIP runtime; the id travels over the torch.distributed control plane."""

    def __init__(self, device_id: int, max_bytes: int = 4096):
        self.lib = preload_rccl()          # before torch: see preload_rccl()
        import torch.distributed as dist
        self.dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        ident = [None]
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            if self.lib.icicle_snark_rccl_unique_id(buf) == 0:
                ident = [bytes(buf)]
        dist.broadcast_object_list(ident, src=0)      # None tells every rank that rank 0 could not create the id
        if ident[0] is None:
            raise RuntimeError("rccl unique_id: " + self.lib.icicle_snark_rccl_last_error().decode())
        self.comm = C.c_void_p()
        self._check(self.lib.icicle_snark_rccl_init(ident[0], self.rank, self.world, device_id, C.c_size_t(max_bytes), C.byref(self.comm)), "init")
        # one round trip before anything relies on the communicator
        got = self.allgather(bytes([self.rank & 0xff]) * 8)
        if got != b"".join(bytes([r & 0xff]) * 8 for r in range(self.world)):
            raise RuntimeError("rccl all-gather self-test returned wrong data")
        # … and one all-to-all on device buffers (grouped ncclSend / ncclRecv: the transport of the distributed QAP front
        # end), two rows of 256-byte chunks: rank r sends (r, peer, row) patterns and must receive (peer, r, row)
        chunk, rows = 256, 2
        row_bytes = chunk * self.world
        send = b"".join(bytes([self.rank, p, q, 0xA5]) * (chunk // 4) for q in range(rows) for p in range(self.world))
        d_send, d_recv = K.DeviceVec(rows * row_bytes), K.DeviceVec(rows * row_bytes)
        try:
            K.raw_to_device(d_send.ptr, send)
            self.alltoall_rows(d_send.ptr, d_recv.ptr, rows, row_bytes, chunk)
            got = K.raw_to_host(d_recv.ptr, rows * row_bytes)
        finally:
            d_send.free(); d_recv.free()
        want = b"".join(bytes([p, self.rank, q, 0xA5]) * (chunk // 4) for q in range(rows) for p in range(self.world))
        if got != want:
            raise RuntimeError("rccl all-to-all self-test returned wrong data")
        # … and one in-place all-gather of a device buffer (the witness distribution of a sharded prove)
        sl = 512
        d_buf = K.DeviceVec(sl * self.world)
        try:
            K.raw_to_device(d_buf.ptr + self.rank * sl, bytes([self.rank & 0xff, 0x5A]) * (sl // 2))
            self.allgather_device(d_buf.ptr, sl)
            got = K.raw_to_host(d_buf.ptr, sl * self.world)
        finally:
            d_buf.free()
        if got != b"".join(bytes([r & 0xff, 0x5A]) * (sl // 2) for r in range(self.world)):
            raise RuntimeError("rccl in-place device all-gather self-test returned wrong data")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"rccl {what}: {self.lib.icicle_snark_rccl_last_error().decode()}")

    def allgather(self, block: bytes) -> bytes:
        out = (C.c_uint8 * (len(block) * self.world))()
        self._check(self.lib.icicle_snark_rccl_allgather(self.comm, block, C.c_size_t(len(block)), out), "allgather")
        return bytes(out)

    def max(self, x: float) -> float:
        v = C.c_double(x)
        self._check(self.lib.icicle_snark_rccl_allreduce_max(self.comm, C.byref(v)), "allreduce_max")
        return v.value

    def alltoall_rows(self, send_ptr: int, recv_ptr: int, rows: int, row_bytes: int, chunk_bytes: int):
        """device-to-device all-to-all over xGMI (grouped ncclSend / ncclRecv, csrc/comm/rccl_comm.cpp)"""
        self._check(self.lib.icicle_snark_rccl_alltoall_rows(self.comm, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), rows, C.c_size_t(row_bytes),
                                                             C.c_size_t(chunk_bytes)), "alltoall_rows")

    def allgather_device(self, ptr: int, slice_bytes: int):
        """in-place all-gather of a device buffer over xGMI: this rank's slice sits at ptr + rank·slice_bytes"""
        self._check(self.lib.icicle_snark_rccl_allgather_device(self.comm, C.c_void_p(ptr), C.c_size_t(slice_bytes)), "allgather_device")

    def barrier(self):
        self.dist.barrier()

    def close(self):
        if self.comm:
            self.lib.icicle_snark_rccl_destroy(self.comm)
            self.comm = None


def sharded_commitments(cm, key: str, wtns, exch, distributed_qap: bool = True, shard_witness: bool | None = None):
    """this rank's five partial commitments.  With 2, 4 or 8 ranks (H sharded by residue class) the QAP front end is
    distributed too: every rank transforms 1/world of the rows and two all-to-alls move the blocks (tests/dist_qap_model.py has the algebra); otherwise
    (or with distributed_qap=False, or when the witness is already resident: wtns=None) it is replicated, communication-free.
    shard_witness (default: on with more than one rank; ICICLE_SNARK_SHARD_WITNESS=0 turns it off): every rank uploads
    1/world of the witness over PCIe and an in-place all-gather over the exchange completes it on every device."""
    if shard_witness is None:
        shard_witness = exch.world > 1 and os.environ.get("ICICLE_SNARK_SHARD_WITNESS", "1") != "0"
    if shard_witness and wtns is not None and exch.world > 1:
        ptr, slice_bytes = cm.upload_witness_slice(key, wtns)
        exch.allgather_device(ptr, slice_bytes)
        cm.witness_ready(key)
        wtns = None
        if distributed_qap and cm.dist_supported(key):
            send, recv, rows, rb, cb = cm.dist_stage1(key, None)
            exch.alltoall_rows(send, recv, rows, rb, cb)
            send, recv = cm.dist_stage2(key)
            exch.alltoall_rows(send, recv, rows, rb, cb)
            cm.dist_exchange_done(key)
        return cm.commitments(key, None)
    if distributed_qap and wtns is not None and exch.world > 1 and cm.dist_supported(key):
        send, recv, rows, rb, cb = cm.dist_stage1(key, wtns)
        exch.alltoall_rows(send, recv, rows, rb, cb)
        send, recv = cm.dist_stage2(key)
        exch.alltoall_rows(send, recv, rows, rb, cb)
        cm.dist_exchange_done(key)
        return cm.commitments(key, None)
    return cm.commitments(key, wtns)


def sharded_prove(cm, key: str, wtns, exch, r=None, s=None, wtns_for_public=None):
    """one prove across exch.world GPUs: partial commitments → all-gather → group sum → blinding + JSON."""
    blk, tm = sharded_commitments(cm, key, wtns, exch)
    if exch.world > 1:
        blk = K.sum_commitments(exch.allgather(blk), exch.world)
    proof, public = cm.assemble(key, wtns_for_public if wtns is None else wtns, blk, r, s)
    return proof, public, tm
