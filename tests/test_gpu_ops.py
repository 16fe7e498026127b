"""Parity tests proper (need an MI355X): the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs and against the committed golden vectors.  Bit-exact everywhere (integer work).
Shapes follow the reference's own differential tests (wrappers/rust/icicle-core/src/{vec_ops,ntt,msm}/tests.rs)."""
import os

import numpy as np
import pytest

from conftest import load_golden, unhex, unhex_int

pytestmark = pytest.mark.gpu
DIMS = {"g1": (2, 3), "g2": (4, 6)}


def rand_fr(O, rng, n):
    raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 61) - 1)  # < 2^253 < r
    return raw


def test_runtime_memory_tracking(gpu):
    K = gpu
    d = K.DeviceVec(1024)
    assert K.lib().icicle_is_active_device_memory(K.ptr_of(d)) == 0
    assert K.lib().icicle_is_active_device_memory(K.ptr_of(d.slice(512, 64))) == 0   # interior pointer
    host = np.zeros(4, dtype=np.uint64)
    assert K.lib().icicle_is_active_device_memory(K.ptr_of(host)) != 0
    assert K.lib().icicle_is_host_memory(K.ptr_of(host)) == 0
    a = np.arange(128, dtype=np.uint64)
    d.copy_from_host(a)
    assert np.array_equal(d.to_host((128,)), a)
    st = K.IcicleStream()
    d2 = K.DeviceVec.from_host(a, st)
    assert np.array_equal(d2.to_host((128,), stream=st), a)
    st.destroy()
    d.free(); d2.free()
    assert K.lib().icicle_is_active_device_memory(K.ptr_of(d2)) != 0


@pytest.mark.parametrize("n", [1, 5, 255, 4096, 100003])
def test_vec_ops_all_residencies(gpu, O, n):
    K = gpu
    rng = np.random.default_rng(n)
    a, b = rand_fr(O, rng, n), rand_fr(O, rng, n)
    a[0] = 0
    want = {"mul": O.fr_vector_mul(a, b), "sub": O.fr_vector_sub(a, b), "add": O.fr_vector_add(a, b)}
    fn = {"mul": K.mul_scalars, "sub": K.sub_scalars, "add": K.add_scalars}
    for op in want:
        assert np.array_equal(fn[op](a, b), want[op]), op                      # host, host -> host
        da, db = K.DeviceVec.from_host(a), K.DeviceVec.from_host(b)
        out = K.DeviceVec(n * 32)
        fn[op](da, db, out)                                                     # device -> device
        assert np.array_equal(out.to_host((n, 4)), want[op]), op
        assert np.array_equal(fn[op](da, b), want[op]), op                     # mixed -> host
        st = K.IcicleStream()
        fn[op](da, db, da, stream=st, is_async=True)                            # in place, async
        st.synchronize(); st.destroy()
        assert np.array_equal(da.to_host((n, 4)), want[op]), op
        da.free(); db.free(); out.free()


def test_scalar_montgomery(gpu, O):
    K = gpu
    rng = np.random.default_rng(5)
    a = rand_fr(O, rng, 5000)
    assert np.array_equal(K.scalar_convert_montgomery(a, True), O.fr_convert_montgomery(a, True))
    assert np.array_equal(K.scalar_convert_montgomery(a, False), O.fr_convert_montgomery(a, False))
    d = K.DeviceVec.from_host(a)
    K.scalar_convert_montgomery(d, True)       # in place on device, like ScalarField::from_mont(&mut DeviceVec)
    K.scalar_convert_montgomery(d, False)
    assert np.array_equal(d.to_host(a.shape), a)
    d.free()


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_point_montgomery_golden(gpu, grp):
    K = gpu
    g = load_golden("curve.json")[grp]
    na = DIMS[grp][0]
    pts = unhex(g["mont_in"], 4, na, 4)
    assert np.array_equal(K.affine_convert_montgomery(grp, pts, True), unhex(g["to_mont"], 4, na, 4))
    assert np.array_equal(K.affine_convert_montgomery(grp, pts, False), unhex(g["from_mont"], 4, na, 4))


@pytest.fixture(scope="module")
def domain(gpu):
    K = gpu
    K.release_domain()
    K.initialize_domain(K.get_root_of_unity(1 << 20))
    yield K
    K.release_domain()


def test_ntt_golden(domain):
    K = domain
    for c in load_golden("ntt.json"):
        n = c["n"]
        x = unhex(c["x"], n, 4)
        assert np.array_equal(K.ntt(x, False), unhex(c["forward"], n, 4))
        assert np.array_equal(K.ntt(x, True), unhex(c["inverse"], n, 4))


@pytest.mark.parametrize("logn", [0, 1, 2, 3, 7, 9, 10, 11, 13, 16, 18, 19, 20])
def test_ntt_vs_oracle(domain, O, logn):
    """1-pass (≤2^9), 2-pass (2^10..2^18) and 3-pass (≥2^19) plans, batch 3, in place on device, async —
    the prover's configuration (icicle_helper.rs:13-32) — forward, inverse and round trip."""
    K = domain
    n = 1 << logn
    rng = np.random.default_rng(logn)
    batch = 3 if logn <= 18 else 1
    x = rand_fr(O, rng, batch * n)
    want_f = O.fr_ntt(x, False, batch=batch, domain_log=20)
    want_i = O.fr_ntt(x, True, batch=batch, domain_log=20)
    assert np.array_equal(K.ntt(x, False, batch_size=batch), want_f)          # host in/out
    st = K.IcicleStream()
    d = K.DeviceVec.from_host(x, st)
    K.ntt(d, True, batch_size=batch, stream=st, is_async=True)                 # in place on device
    assert np.array_equal(d.to_host(x.shape, stream=st), want_i)
    K.ntt(d, False, batch_size=batch, stream=st, is_async=True)                # round trip
    assert np.array_equal(d.to_host(x.shape, stream=st), x)
    st.destroy(); d.free()


@pytest.mark.parametrize("logn", [11, 12, 15, 16, 17, 20])
def test_ntt_extreme_inputs_on_the_lazy_field(domain, O, logn):
    """The radix-2^29 passes (csrc/fr29.h, ntt_pass29_kernel; every size from 2^11 on) do not reduce after additions: the value
    bounds the host plans for them (bounds29) must hold for the worst inputs.  A constant vector of r − 1 drives the all-sums path
    of every sub-transform to its bound (element 0 becomes n·(r − 1)), alternating 0 / r − 1 and r − 1 / 1 patterns the
    difference paths; all against the oracle, forward and inverse, batch 3."""
    K = domain
    n = 1 << logn
    rm1 = np.frombuffer(int(O.R_MOD - 1).to_bytes(32, "little"), dtype=np.uint64)
    one = np.frombuffer(int(1).to_bytes(32, "little"), dtype=np.uint64)
    rows = [np.tile(rm1, (n, 1)), np.tile(rm1, (n, 1)), np.tile(rm1, (n, 1))]
    rows[1][0::2] = 0
    rows[2][1::2] = one
    x = np.ascontiguousarray(np.concatenate(rows))
    for inverse in (False, True):
        assert np.array_equal(K.ntt(x, inverse, batch_size=3), O.fr_ntt(x, inverse, batch=3, domain_log=20)), (logn, inverse)


def test_ntt_8x32_kernels_behind_the_switch():
    """ICICLE_SNARK_NTT29=0 (and a domain whose Montgomery-261 twiddle table could not be allocated) sends every size through the
    8×32-bit pass kernels, which otherwise only see transforms below 2^11: the oracle comparison of test_ntt_vs_oracle and the golden
    vectors once more in a process with the switch set."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "test_ntt_vs_oracle or test_ntt_golden or test_ntt_coset"],
                         capture_output=True, text=True, env=dict(os.environ, ICICLE_SNARK_NTT29="0"), timeout=900)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_ntt_out_of_place_and_errors(domain, O):
    K = domain
    rng = np.random.default_rng(99)
    x = rand_fr(O, rng, 1 << 12)
    d_in, d_out = K.DeviceVec.from_host(x), K.DeviceVec(x.nbytes)
    K.ntt(d_in, False, out=d_out)
    assert np.array_equal(d_out.to_host(x.shape), O.fr_ntt(x, False, domain_log=20))
    assert np.array_equal(d_in.to_host(x.shape), x)
    with pytest.raises(K.IcicleError):
        K.ntt(x[:12], False)                       # not a power of two
    with pytest.raises(K.IcicleError):
        K.ntt(np.zeros((1 << 21, 4), dtype=np.uint64), False)   # larger than the domain
    d_in.free(); d_out.free()


def test_ntt_coset(domain, O):
    """size-2n NTT = NTT(n) of evens ∥ coset-NTT(n): the reference's own identity test (ntt/tests.rs:99-166),
    here in the form: coset NTT with g equals NTT of x_j·g^j."""
    K = domain
    rng = np.random.default_rng(4)
    n = 1 << 10
    x = rand_fr(O, rng, n)
    g = O.fr_omega(11)
    xs = O.arr_to_ints(x)
    shifted = O.ints_to_arr([v * pow(g, j, O.R_MOD) % O.R_MOD for j, v in enumerate(xs)])
    want = O.fr_ntt(shifted, False, domain_log=20)
    got = K.ntt(x, False, coset_gen=O.ints_to_arr([g])[0])
    assert np.array_equal(got, want)
    back = K.ntt(got, True, coset_gen=O.ints_to_arr([g])[0])
    assert np.array_equal(back, x)


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_msm_golden(gpu, O, grp):
    K = gpu
    na = DIMS[grp][0]
    for c in load_golden("msm.json")[grp]:
        n = c["n"]
        sc, bases = unhex(c["scalars"], n, 4), unhex(c["bases"], n, na, 4)
        want = unhex(c["result_affine"], na, 4)
        for cc in (0, 5, 16):
            res = K.msm(grp, sc, bases, c=cc)
            assert np.array_equal(K.ec(grp, "to_affine", res), want), (grp, n, c["kind"], cc)


def _bases(O, grp, rng, n, distinct=100):
    G = O.ec_to_affine(grp, O.ec_generator(grp))
    k = rand_fr(O, rng, min(n, distinct))
    pts = O.fixed_base_mul(grp, G, k)
    reps = (n + len(pts) - 1) // len(pts)
    return np.concatenate([pts] * reps)[:n].copy()    # repeats like rand_host_many (projective.h:246-256)


@pytest.mark.parametrize("grp,n", [("g1", 1), ("g1", 16), ("g1", 1000), ("g1", 1 << 14), ("g2", 1), ("g2", 1000), ("g2", 1 << 12)])
def test_msm_vs_oracle_device_resident(gpu, O, grp, n):
    """msm/tests.rs:24-95: scalars/points/result on device, async, two affine-zero points."""
    K = gpu
    rng = np.random.default_rng(n + (7 if grp == "g2" else 0))
    sc, bases = rand_fr(O, rng, n), _bases(O, grp, rng, n)
    if n > 2:
        bases[1] = 0
        bases[n - 1] = 0
    want = O.ec_to_affine(grp, O.msm(grp, sc, bases))
    st = K.IcicleStream()
    d_s, d_b = K.DeviceVec.from_host(sc, st), K.DeviceVec.from_host(bases, st)
    d_r = K.DeviceVec(32 * DIMS[grp][1], st)
    K.msm(grp, d_s, d_b, out=d_r, stream=st, is_async=True)
    res = d_r.to_host((DIMS[grp][1], 4), stream=st)
    assert np.array_equal(K.ec(grp, "to_affine", res), want)
    # Montgomery-form inputs (are_scalars_montgomery_form / are_points_montgomery_form)
    K.scalar_convert_montgomery(d_s, True)
    K.affine_convert_montgomery(grp, d_b, True)
    res2 = K.msm(grp, d_s, d_b, scalars_mont=True, points_mont=True)
    assert np.array_equal(K.ec(grp, "to_affine", res2), want)
    st.destroy(); d_s.free(); d_b.free(); d_r.free()


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_msm_skewed_scalars(gpu, O, grp):
    """msm/tests.rs:254-302: mostly 0/1 scalars (witness-like) → huge buckets, P+P doublings, large-bucket path."""
    K = gpu
    n = 20000 if grp == "g1" else 6000
    rng = np.random.default_rng(21)
    sc = rand_fr(O, rng, n)
    kind = rng.integers(0, 10, size=n)
    sc[kind < 4] = 0
    sc[(kind >= 4) & (kind < 7)] = np.array([1, 0, 0, 0], dtype=np.uint64)
    # bytes / small packed values: a few hundred entries in each of the lowest buckets — above the large-bucket threshold
    # (3 × average, at least 64), below one 1024-entry work item; the 0/1 bucket above spans several work items
    small = kind == 7
    sc[small] = 0
    sc[small, 0] = rng.integers(2, 10, size=int(small.sum()), dtype=np.uint64)
    sc[kind == 8] = O.ints_to_arr([O.R_MOD - 1])[0]
    bases = _bases(O, grp, rng, n, distinct=50)
    want = O.ec_to_affine(grp, O.msm(grp, sc, bases))
    assert np.array_equal(K.ec(grp, "to_affine", K.msm(grp, sc, bases)), want)
    assert np.array_equal(K.ec(grp, "to_affine", K.msm(grp, sc, bases, c=8)), want)


def test_msm_degenerate(gpu, O):
    K = gpu
    rng = np.random.default_rng(2)
    bases = _bases(O, "g1", rng, 8)
    zero = np.zeros((8, 4), dtype=np.uint64)
    res = K.msm("g1", zero, bases)
    assert not K.ec("g1", "to_affine", res).any() and np.array_equal(res, O.ec_zero("g1"))   # (0,1,0)
    # P − P
    sc = O.ints_to_arr([5, O.R_MOD - 5])
    b2 = np.stack([bases[0], bases[0]])
    assert not K.ec("g1", "to_affine", K.msm("g1", sc, b2)).any()


@pytest.mark.parametrize("c", [0, 5, 8, 16])
def test_msm_edge_scalars(gpu, O, c):
    """recoding limits: scalars around the negation threshold (r − 1)/2, around 2^253, r − 1, all-ones low parts"""
    K = gpu
    R = O.R_MOD
    half = (R - 1) // 2
    vals = [half, half + 1, half - 1, (1 << 253) - 1, 1 << 253, (1 << 253) + 1, R - 1, R - 2, 1, 0, (1 << 252) - 1,
            (half >> 230 << 230) - 1, (1 << 240) - 1, ((1 << 253) - 1) ^ (1 << 19)]
    vals = vals + [(R - v) % R for v in vals]
    rng = np.random.default_rng(4)
    bases = _bases(O, "g1", rng, len(vals))
    sc = O.ints_to_arr(vals)
    got = K.ec("g1", "to_affine", K.msm("g1", sc, bases, c=c))
    assert np.array_equal(got, O.ec_to_affine("g1", O.msm("g1", sc, bases)))


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_generator_mul(gpu, O, grp):
    K = gpu
    rng = np.random.default_rng(8)
    sc = rand_fr(O, rng, 300)
    sc[0] = 0
    sc[1] = np.array([1, 0, 0, 0], dtype=np.uint64)
    G = O.ec_to_affine(grp, O.ec_generator(grp))
    assert np.array_equal(K.generator_mul(grp, sc), O.fixed_base_mul(grp, G, sc))


def test_many_async_msms_in_flight_do_not_share_tail_slots(gpu, O):
    """VERDICT r1 robustness item: the pinned tail-slot ring holds 128 slots; more asynchronous bn254_msm calls than that
    in flight on one stream must neither corrupt earlier results nor fail — a slot is reused only once the event behind its
    last use has completed, and a call that finds no free slot finishes on the device instead."""
    K = gpu
    rng = np.random.default_rng(5)
    n, calls = 96, 300
    G = O.ec_to_affine("g1", O.ec_generator("g1"))
    bases = O.fixed_base_mul("g1", G, rand_fr(O, rng, n))
    st = K.IcicleStream()
    d_b = K.DeviceVec.from_host(bases, st)
    scs = [rand_fr(O, rng, n) for _ in range(4)]
    want = [O.ec_to_affine("g1", O.msm("g1", s, bases)) for s in scs]
    d_s = [K.DeviceVec.from_host(s, st) for s in scs]
    outs = [K.DeviceVec(96, st) for _ in range(calls)]
    for i in range(calls):
        K.msm("g1", d_s[i % 4], d_b, out=outs[i], stream=st, is_async=True, size=n)
    st.synchronize()
    for i in range(calls):
        got = K.ec("g1", "to_affine", outs[i].to_host((3, 4)))
        assert np.array_equal(got, want[i % 4]), i
    for d in outs + d_s + [d_b]:
        d.free()
    st.destroy()


@pytest.mark.parametrize("grp,n", [("g1", 40_000), ("g2", 33_000)])
def test_msm_automatic_fixed_base_tables(gpu, O, grp, n):
    """bn254_msm / bn254_g2_msm over the SAME device-resident base array: the first call runs the classic layout, the second
    builds a fixed-base table on the caller's stream, later ones use it (csrc/msm_plan.h: base_table_lookup).  Every result
    equals the oracle's, with fresh scalars each time, standard and Montgomery scalars, two streams; a write to the bases
    THROUGH the API (copy, in-place Montgomery conversion, free + reallocation at the same address) retires the table, and the
    results follow the new contents — icicle/src/msm.cpp:12-32 semantics are unchanged."""
    K = gpu
    rng = np.random.default_rng(2024 + n)
    bases = _bases(O, grp, rng, n)
    bases[3] = 0
    st, st2 = K.IcicleStream(), K.IcicleStream()
    d_b = K.DeviceVec.from_host(bases, st)
    st.synchronize()

    def check(points, points_mont=False, stream=None, rounds=1):
        for k in range(rounds):
            sc = rand_fr(O, rng, n)
            if k % 2:
                sc[: n // 2] = 0
                sc[5, 0] = 1
            want = O.ec_to_affine(grp, O.msm(grp, sc, points))
            d_s = K.DeviceVec.from_host(sc, stream)
            if k == 2:
                K.scalar_convert_montgomery(d_s, True, stream=stream)
            got = K.msm(grp, d_s, d_b, stream=stream, scalars_mont=(k == 2), points_mont=points_mont)
            assert np.array_equal(K.ec(grp, "to_affine", got), want), (k, points_mont)
            d_s.free()

    check(bases, rounds=4, stream=st)                      # classic, build, hit, hit (Montgomery scalars on the third)
    check(bases, rounds=2, stream=st2)                     # the table built on `st` serves another stream
    # the bases change behind the same pointer: copy from the host …
    bases2 = _bases(O, grp, rng, n)
    d_b.copy_from_host(bases2, st)
    st.synchronize()
    check(bases2, rounds=3, stream=st)
    # … converted in place to Montgomery form (the reference's cache does the opposite conversion on its points, src/cache.rs:228)
    K.affine_convert_montgomery(grp, d_b, True)
    check(bases2, points_mont=True, rounds=3)
    # … freed, and a new array of the same size (the allocation cache hands the block out again)
    d_b.free()
    bases3 = _bases(O, grp, rng, n)
    d_b = K.DeviceVec.from_host(bases3)
    check(bases3, rounds=3)
    st.destroy(); st2.destroy(); d_b.free()


@pytest.mark.parametrize("grp,n", [("g1", 40_000), ("g2", 33_000)])
def test_msm_tables_follow_writes_the_library_cannot_see(gpu, O, grp, n):
    """The bases are read at call time (icicle/src/msm.cpp:12-32) however the caller wrote them: between bn254_msm calls on the
    same pointer the array is overwritten by a RAW hipMemcpy of the HIP runtime (not through icicle_copy*, so the write tracking
    of the C ABI sees nothing) — whole array, a single point, two points swapped — and every result is the oracle's sum over the
    NEW contents (hash sum of the bases + guarded refresh of the table in stream order, csrc/msm_plan.h)."""
    import ctypes as C
    K = gpu
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipDeviceSynchronize.argtypes = []
    rng = np.random.default_rng(77 + n)
    bases = _bases(O, grp, rng, n)
    d_b = K.DeviceVec.from_host(bases)
    st = K.IcicleStream()

    def raw_write(arr, byte_off=0):
        a = np.ascontiguousarray(arr)
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemcpy(C.c_void_p(d_b.ptr + byte_off), a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0   # hipMemcpyHostToDevice

    def check(points, rounds, stream=None):
        for _ in range(rounds):
            sc = rand_fr(O, rng, n)
            want = O.ec_to_affine(grp, O.msm(grp, sc, points))
            d_s = K.DeviceVec.from_host(sc, stream)
            got = K.msm(grp, d_s, d_b, stream=stream)
            assert np.array_equal(K.ec(grp, "to_affine", got), want)
            d_s.free()

    check(bases, 3)                                   # classic, build, hit
    psize = bases[0].nbytes
    b2 = _bases(O, grp, rng, n)
    raw_write(b2)                                     # everything changes behind the library's back
    check(b2, 2)
    b3 = b2.copy()
    b3[n // 3] = _bases(O, grp, rng, 1)[0]            # one point
    raw_write(b3[n // 3], (n // 3) * psize)
    check(b3, 2, stream=st)
    b4 = b3.copy()
    b4[[5, n - 7]] = b4[[n - 7, 5]]                   # two points trade places (an order-blind checksum would miss this)
    raw_write(b4[5], 5 * psize)
    raw_write(b4[n - 7], (n - 7) * psize)
    check(b4, 2)
    b5 = b4.copy()
    b5[11] = 0                                        # a base becomes the identity
    raw_write(b5[11], 11 * psize)
    check(b5, 2, stream=st)
    st.destroy(); d_b.free()


@pytest.mark.parametrize("grp", ["g1", "g2"])
def test_table_mode_sort_with_runs_of_every_length(gpu, O, grp):
    """the LDS-staged digit sort behind the table mode (csrc/msm_sort.hip, round 5) copies its staged entries out run by run — 16 lanes per
    run, runs above 192 entries by the whole workgroup — in both of its passes.  Scalar vectors that make runs of every kind: all equal
    (ONE bucket per window holds everything: tile runs of 1024, row runs and bucket runs far above 192), the values 0 … 7 only (eight
    buckets), a few large values in a sea of ones, zeros at every second place, and dense ones beside them; 2^16 + 37 points (a ragged
    last tile), three calls per vector (classic layout, table build, table hit) against the oracle."""
    K = gpu
    n = (1 << 16) + 37
    rng = np.random.default_rng(99)
    bases = _bases(O, grp, rng, n, distinct=64)
    bases[11] = 0
    d_b = K.DeviceVec.from_host(bases)

    def ones():
        a = np.zeros((n, 4), dtype=np.uint64)
        a[:, 0] = 1
        return a
    same = np.tile(rand_fr(O, rng, 1), (n, 1))
    small = np.zeros((n, 4), dtype=np.uint64)
    small[:, 0] = rng.integers(0, 8, size=n, dtype=np.uint64)
    few = ones()
    few[rng.integers(0, n, size=50)] = rand_fr(O, rng, 50)
    holes = rand_fr(O, rng, n)
    holes[::2] = 0
    rmax = np.tile(np.frombuffer((O.R_MOD - 1).to_bytes(32, "little"), dtype=np.uint64), (n, 1))   # every digit at its negative extreme
    for name, sc in (("ones", ones()), ("same", same), ("small", small), ("few", few), ("holes", holes), ("r-1", rmax), ("dense", rand_fr(O, rng, n))):
        want = O.ec_to_affine(grp, O.msm(grp, sc, bases))
        d_s = K.DeviceVec.from_host(sc)
        for call in range(3):
            got = K.ec(grp, "to_affine", K.msm(grp, d_s, d_b))
            assert np.array_equal(got, want), (name, call)
        d_s.free()
    d_b.free()
