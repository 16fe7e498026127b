"""Distributed QAP front end (tests/dist_qap_model.py): the three-stage decomposition with two all-to-alls equals the
oracle's construct_r1cs, (a) simulated in one process for G = 2, 4, 8 and (b) run by two gloo processes that really exchange
their blocks.  CPU only — the HIP kernels of the same stages are compared with the single-GPU prover in tests/test_gpu_dist.py."""
import importlib
import os
import random
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

import dist_qap_model as D


def _reference_h(O, rows, logn):
    """d = NTT(coset(iNTT(A))) ∘ NTT(coset(iNTT(B))) − NTT(coset(iNTT(C'))) with the oracle's transforms (domain 2n)"""
    R, n = O.R_MOD, 1 << logn
    w2n = O.fr_omega(logn + 1)
    ev = []
    for x in rows:
        a = O.arr_to_ints(O.fr_ntt(O.ints_to_arr(x), True, domain_log=logn + 1))
        a = [v * pow(w2n, k, R) % R for k, v in enumerate(a)]
        ev.append(O.arr_to_ints(O.fr_ntt(O.ints_to_arr(a), False, domain_log=logn + 1)))
    return [(a * b - c) % R for b, a, c in zip(*ev)]


def _ntt_fns(O, logm, logn):
    inv = lambda v: O.arr_to_ints(O.fr_ntt(O.ints_to_arr(v), True, domain_log=logn + 1)) if logm else list(v)
    fwd = lambda v: O.arr_to_ints(O.fr_ntt(O.ints_to_arr(v), False, domain_log=logn + 1)) if logm else list(v)
    return inv, fwd


@pytest.mark.parametrize("logn,G", [(5, 2), (6, 4), (7, 8), (6, 8)])
def test_three_stage_decomposition_equals_construct_r1cs(O, logn, G):
    n, m, mb = 1 << logn, (1 << logn) // G, (1 << logn) // (G * G)
    rnd = random.Random(7 * logn + G)
    rows = [[rnd.randrange(O.R_MOD) for _ in range(n)] for _ in range(3)]     # [B | A | C'] evaluations on the domain
    want = _reference_h(O, rows, logn)
    wn, w2n = O.fr_omega(logn), O.fr_omega(logn + 1)
    inv, fwd = _ntt_fns(O, m.bit_length() - 1, logn)
    Y = [D.stage1([x[r::G] for x in rows], r, G, n, wn, inv) for r in range(G)]
    Z = []
    for b in range(G):      # exchange 1: rank b receives block b of every source
        recv = [[Y[j1][row][b * mb:(b + 1) * mb] for j1 in range(G)] for row in range(3)]
        Z.append(D.stage2(recv, b, G, n, wn, w2n))
    for r in range(G):      # exchange 2: rank r receives its Z_r block from every rank b, in block order
        z_rows = [sum((Z[b][row][r] for b in range(G)), []) for row in range(3)]
        assert D.stage3(z_rows, fwd) == want[r::G], (logn, G, r)


_WORKER = r'''
import importlib, os, random, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
import oracle as O
from test_dist_qap import _reference_h, _ntt_fns
import dist_qap_model as D
dist.init_process_group("gloo")
r, G = dist.get_rank(), dist.get_world_size()
logn = 6; n, m, mb = 1 << logn, (1 << logn) // G, (1 << logn) // (G * G)
rnd = random.Random(99)
rows = [[rnd.randrange(O.R_MOD) for _ in range(n)] for _ in range(3)]     # same on every rank (the witness is)
inv, fwd = _ntt_fns(O, m.bit_length() - 1, logn)
wn, w2n = O.fr_omega(logn), O.fr_omega(logn + 1)
def a2a(chunks):   # chunks[peer] = list of ints for that peer; returns what every peer sent to this rank (gloo has no all_to_all: gather + pick)
    mine = torch.from_numpy(O.ints_to_arr([v for c in chunks for v in c]).view("int64").copy())
    outs = [torch.empty_like(mine) for _ in range(G)]
    dist.all_gather(outs, mine)
    per = len(chunks[0])
    return [O.arr_to_ints(o.numpy().view("uint64").reshape(-1, 4))[r * per:(r + 1) * per] for o in outs]
Y = D.stage1([x[r::G] for x in rows], r, G, n, wn, inv)
recv = [a2a([Y[row][b * mb:(b + 1) * mb] for b in range(G)]) for row in range(3)]          # recv[row][j1]
Z = D.stage2(recv, r, G, n, wn, w2n)                                                          # Z[row][i1]
z_rows = [sum(a2a([Z[row][i1] for i1 in range(G)]), []) for row in range(3)]                  # blocks in source order
got = D.stage3(z_rows, fwd)
assert got == _reference_h(O, rows, logn)[r::G]
dist.barrier(); dist.destroy_process_group()
print("DIST_QAP_OK", r)
'''


def test_two_gloo_ranks_exchange_their_blocks():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-c", _WORKER % {"root": ROOT}]
    # torch.distributed.run has no -c: run the worker from a temp file
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(_WORKER % {"root": ROOT})
        path = f.name
    try:
        cmd = cmd[:-2] + [path]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="2"))
        assert out.returncode == 0 and out.stdout.count("DIST_QAP_OK") == 2, out.stdout[-2000:] + out.stderr[-3000:]
    finally:
        os.unlink(path)
