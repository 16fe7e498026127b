"""The rest of the ICICLE surface the wrapper crate binds, through the C ABI on the GPU against
Python-integer / oracle results: vec ops (div, accumulate, Σ, Π, scalar∘vector, batches in both layouts), projective
Montgomery conversion, NTT orderings (kNR/kRN/kRR/kNM/kMN) and columns_batch, batched MSM and msm_precompute_bases.
Semantics follow icicle/backend/cpu/src/field/cpu_vec_ops.cpp, icicle/include/icicle/ntt.h:32-43 and msm.h:21-53."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DIMS = {"g1": (2, 3), "g2": (4, 6)}


def rand_fr(rng, n):
    raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 61) - 1)
    return raw


def test_vec_div_accumulate_and_reductions(gpu, O):
    K, R = gpu, O.R_MOD
    rng = np.random.default_rng(5)
    n = 3000
    a, b = rand_fr(rng, n), rand_fr(rng, n)
    b[7] = 0                                           # inverse(0) = 0 in the reference ⇒ a/0 = 0
    ai, bi = O.arr_to_ints(a), O.arr_to_ints(b)
    want = [x * pow(y, R - 2, R) % R for x, y in zip(ai, bi)]
    assert np.array_equal(K.vec_op2("div", a, b), O.ints_to_arr(want))
    d = K.DeviceVec.from_host(a)
    K.accumulate_scalars(d, b)
    assert np.array_equal(d.to_host(a.shape), O.ints_to_arr([(x + y) % R for x, y in zip(ai, bi)]))
    d.free()
    for batch, cols in ((1, False), (3, False), (3, True)):
        size = n // batch
        v = a[:size * batch]
        vi = ai[:size * batch]
        rows = [[vi[(bb + i * batch) if cols else (bb * size + i)] for i in range(size)] for bb in range(batch)]
        s = K.reduce_scalars("sum", v, batch_size=batch, columns_batch=cols)
        assert np.array_equal(s, O.ints_to_arr([sum(r) % R for r in rows]))
        p = K.reduce_scalars("product", v, batch_size=batch, columns_batch=cols)
        prods = []
        for r in rows:
            acc = 1
            for x in r:
                acc = acc * x % R
            prods.append(acc)
        assert np.array_equal(p, O.ints_to_arr(prods))
    # a long vector (two-stage reduction, device resident)
    big = rand_fr(rng, 1 << 17)
    dv = K.DeviceVec.from_host(big)
    assert np.array_equal(K.reduce_scalars("sum", dv), O.ints_to_arr([sum(O.arr_to_ints(big)) % R]))
    dv.free()


@pytest.mark.parametrize("cols", [False, True])
def test_scalar_vector_ops(gpu, O, cols):
    K, R = gpu, O.R_MOD
    rng = np.random.default_rng(11)
    batch, size = 4, 257
    v, sc = rand_fr(rng, batch * size), rand_fr(rng, batch)
    vi, si = O.arr_to_ints(v), O.arr_to_ints(sc)
    for name, f in (("add", lambda s, x: (s + x) % R), ("sub", lambda s, x: (s - x) % R), ("mul", lambda s, x: s * x % R)):
        want = [f(si[(k % batch) if cols else (k // size)], x) for k, x in enumerate(vi)]
        assert np.array_equal(K.scalar_vec_op(name, sc, v, batch_size=batch, columns_batch=cols), O.ints_to_arr(want)), name
    # element-wise ops are layout-agnostic: batch 4 in columns layout equals the flat result
    assert np.array_equal(K.vec_op2("mul", v, v, batch_size=batch, columns_batch=cols), K.mul_scalars(v, v))


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_projective_convert_montgomery(gpu, O, grp):
    K = gpu
    gen = K.ec(grp, "generator")
    pts = np.stack([K.ec(grp, "mul_scalar", gen, k) for k in (1, 2, 12345)])
    m = K.projective_convert_montgomery(grp, pts, True)
    assert np.array_equal(m.reshape(-1, 4), O.fq_convert_montgomery(pts.reshape(-1, 4), True))
    assert np.array_equal(K.projective_convert_montgomery(grp, m, False), pts)


def _bitrev(x, logn):
    idx = np.array([int(format(i, f"0{logn}b")[::-1], 2) if logn else 0 for i in range(1 << logn)])
    return x[idx]


@pytest.mark.parametrize("logn", [3, 10, 12])
def test_ntt_orderings_and_columns_batch(gpu, O, logn):
    K = gpu
    K.release_domain()
    K.initialize_domain(K.get_root_of_unity(1 << 13))
    rng = np.random.default_rng(logn)
    n = 1 << logn
    x = rand_fr(rng, n)
    NN, NR, RN, RR, NM, MN = range(6)
    for inverse in (False, True):
        want = O.fr_ntt(x, inverse, domain_log=13)
        assert np.array_equal(K.ntt(x, inverse, ordering=NR), _bitrev(want, logn))
        assert np.array_equal(K.ntt(_bitrev(x, logn), inverse, ordering=RN), want)
        assert np.array_equal(K.ntt(_bitrev(x, logn), inverse, ordering=RR), _bitrev(want, logn))
        assert np.array_equal(K.ntt(x, inverse, ordering=NM), _bitrev(want, logn))       # radix-2 convention: M = R
        assert np.array_equal(K.ntt(_bitrev(x, logn), inverse, ordering=MN), want)
    # NR then RN round-trips without any reordering by the caller (the use the mixed orders exist for)
    assert np.array_equal(K.ntt(K.ntt(x, False, ordering=NM), True, ordering=MN), x)
    # columns_batch: element i of batch b at b + i·batch
    batch = 3
    xb = rand_fr(rng, batch * n)
    rows = O.fr_ntt(xb, False, batch=batch, domain_log=13).reshape(batch, n, 4)
    cols_in = np.ascontiguousarray(xb.reshape(batch, n, 4).transpose(1, 0, 2)).reshape(-1, 4)
    got = K.ntt(cols_in, False, batch_size=batch, columns_batch=True).reshape(n, batch, 4)
    assert np.array_equal(got.transpose(1, 0, 2), rows)
    d = K.DeviceVec.from_host(cols_in)                                                      # in place, on device, reversed output
    K.ntt(d, False, batch_size=batch, columns_batch=True, ordering=NR)
    got = d.to_host((n, batch, 4)).transpose(1, 0, 2)
    assert np.array_equal(got, np.stack([_bitrev(r, logn) for r in rows]))
    d.free()
    with pytest.raises(K.IcicleError):
        K.ntt(x, False, ordering=9)
    K.release_domain()


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_msm_batch_and_precompute(gpu, O, grp):
    K = gpu
    rng = np.random.default_rng(21 if grp == "g1" else 22)
    n, batch = 700, 3
    G = O.ec_to_affine(grp, O.ec_generator(grp))
    bases = O.fixed_base_mul(grp, G, rand_fr(rng, n))
    bases[5] = 0
    sc = rand_fr(rng, batch * n)
    want = [O.ec_to_affine(grp, O.msm(grp, sc[b * n:(b + 1) * n], bases)) for b in range(batch)]
    res = K.msm(grp, sc, bases, batch_size=batch, shared_points=True)
    for b in range(batch):
        assert np.array_equal(K.ec(grp, "to_affine", res[b]), want[b])
    # one base set per batch element
    bases2 = np.concatenate([bases, bases[::-1], bases])
    res = K.msm(grp, sc, bases2, batch_size=batch, shared_points=False)
    assert np.array_equal(K.ec(grp, "to_affine", res[0]), want[0])
    assert np.array_equal(K.ec(grp, "to_affine", res[1]), O.ec_to_affine(grp, O.msm(grp, sc[n:2 * n], bases[::-1].copy())))
    # precomputed bases: [f·i + j] = 2^(j·shift)·P_i ; an MSM over them gives the same result
    f = 3
    pre = K.msm_precompute_bases(grp, bases, f, c=0)
    assert pre.shape[0] == f * n and np.array_equal(pre[0::f], bases)
    c = max(4, min(16, int(np.ceil(np.log2(n))) - 4))
    shift = c * (((254 // c + 1) + f - 1) // f)
    for i in (0, 5, 17, n - 1):
        p = O.ec_from_affine(grp, bases[i])
        for j in range(1, f):
            p = O.ec_mul_scalar(grp, p, 1 << shift)
            assert np.array_equal(pre[f * i + j], O.ec_to_affine(grp, p)), (i, j)
    res = K.msm(grp, sc[:n], pre, size=n, precompute_factor=f)
    assert np.array_equal(K.ec(grp, "to_affine", res), want[0])
    # Montgomery-form bases in and out
    bm = O.fq_convert_montgomery(bases.reshape(-1, 4), True).reshape(bases.shape)
    prem = K.msm_precompute_bases(grp, bm, f, points_mont=True)
    assert np.array_equal(O.fq_convert_montgomery(prem.reshape(-1, 4), False).reshape(pre.shape), pre)


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_msm_bitsize_and_precompute_factor_semantics(gpu, O, grp):
    """MSMConfig.bitsize (msm.h:32-34: scalars below 2^bitsize → fewer windows) and precompute_factor > 1 read through
    every precomputed multiple (window w uses multiple w / nbms, bucket set w mod nbms — icicle/src/msm.cpp:45-72,
    cuda_msm.cuh:186-203): same group element as the plain MSM, for several (c, f, bitsize)."""
    K = gpu
    rng = np.random.default_rng(31 if grp == "g1" else 32)
    n = 3000
    G = O.ec_to_affine(grp, O.ec_generator(grp))
    bases = O.fixed_base_mul(grp, G, rand_fr(rng, n))
    bases[7] = 0
    sc = rand_fr(rng, n)
    sc[0] = 0
    sc[1, :] = np.frombuffer((O.R_MOD - 1).to_bytes(32, "little"), dtype=np.uint64)
    want = O.ec_to_affine(grp, O.msm(grp, sc, bases))
    for c, f in ((0, 2), (0, 4), (10, 3), (13, 7), (5, 64)):
        pre = K.msm_precompute_bases(grp, bases, f, c=c)
        got = K.ec(grp, "to_affine", K.msm(grp, sc, pre, size=n, precompute_factor=f, c=c))
        assert np.array_equal(got, want), (c, f)
    # short scalars: 64-bit and 100-bit, with the maximum value present
    for bits in (1, 64, 100, 253):
        s2 = sc.copy()
        if bits <= 64:
            s2[:, 1:] = 0
            s2[:, 0] &= np.uint64((1 << bits) - 1)
            s2[2, 0] = np.uint64((1 << bits) - 1)
        elif bits == 100:
            s2[:, 2:] = 0
            s2[:, 1] &= np.uint64((1 << 36) - 1)
            s2[2, 0], s2[2, 1] = np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64((1 << 36) - 1)
        else:
            s2[:, 3] &= np.uint64((1 << 61) - 1)
            s2[1] = 0
        w2 = O.ec_to_affine(grp, O.msm(grp, s2, bases))
        for c in (0, 9):
            got = K.ec(grp, "to_affine", K.msm(grp, s2, bases, bitsize=bits, c=c))
            assert np.array_equal(got, w2), (bits, c)
        if bits in (64, 100):
            f = 3
            pre = K.msm_precompute_bases(grp, bases, f, bitsize=bits)
            got = K.ec(grp, "to_affine", K.msm(grp, s2, pre, size=n, precompute_factor=f, bitsize=bits))
            assert np.array_equal(got, w2), (bits, "precompute")
    with pytest.raises(K.IcicleError):
        K.msm(grp, sc, bases, bitsize=300)
