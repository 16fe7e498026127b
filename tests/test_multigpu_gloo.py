"""The N > 1 path on CPU: two processes, gloo backend, world_size 2.  Each rank computes the partial
commitments of its point-range shard (with the CPU oracle standing in for the GPU kernels), the blocks are
exchanged with the same `Exchange` interface the RCCL path implements, summed with the product library's
groth16_sum_commitments (host code) and compared with the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    import importlib
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    K = importlib.import_module("icicle-snark_amd")
    P = importlib.import_module("icicle-snark_amd.parallel")
    exch = P.GlooExchange()
    assert (exch.world, exch.rank) == (world, rank)
    rng = np.random.default_rng(5)          # same inputs on every rank
    n = 301
    sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    b1 = O.fixed_base_mul("g1", O.ec_to_affine("g1", O.ec_generator("g1")), sc[::-1].copy())
    b2 = O.fixed_base_mul("g2", O.ec_to_affine("g2", O.ec_generator("g2")), sc[::-1].copy())

    def block(lo, hi):
        g1 = O.msm("g1", sc[lo:hi], b1[lo:hi]) if hi > lo else O.ec_zero("g1")
        g2 = O.msm("g2", sc[lo:hi], b2[lo:hi]) if hi > lo else O.ec_zero("g2")
        return g1.tobytes() + g1.tobytes() + g2.tobytes() + g1.tobytes() + g1.tobytes()   # A B1 B2 C H layout

    lo, hi = P.shard_range(n, rank, world)
    blocks = exch.allgather(block(lo, hi))
    assert len(blocks) == world * K.COMMITMENTS_BYTES
    total = K.sum_commitments(blocks, world)
    want = block(0, n)
    ok = True
    for off, grp, size in ((0, "g1", 96), (96, "g1", 96), (192, "g2", 192), (384, "g1", 96), (480, "g1", 96)):
        got_p = np.frombuffer(total[off:off + size], dtype=np.uint64).reshape(-1, 4)
        want_p = np.frombuffer(want[off:off + size], dtype=np.uint64).reshape(-1, 4)
        ok &= bool(np.array_equal(O.ec_to_affine(grp, got_p), O.ec_to_affine(grp, want_p)))
    mx = exch.max(float(rank + 1))
    exch.barrier()
    q.put((rank, ok, mx, (lo, hi)))
    dist.destroy_process_group()


def test_two_rank_sharded_commitments_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [2.0, 2.0]
    assert res[0][3] == (0, 151) and res[1][3] == (151, 301)      # contiguous, disjoint, covering: rank r = the witness slice it uploads (⌈301 / 2⌉ wires)


def test_shard_range_covers_everything():
    import importlib
    sys.path.insert(0, ROOT)
    P = importlib.import_module("icicle-snark_amd.parallel")
    for total in (0, 1, 7, 100002, 2097152):
        for world in (1, 2, 3, 4, 8):
            edges = [P.shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            # every rank but the last takes ⌈total / world⌉ (its witness slice), the last one the rest: within `world` of each other
            assert max(h - l for l, h in edges) - min(h - l for l, h in edges) < max(world, 2)
            assert all(h > l for l, h in edges) or total < world * world
