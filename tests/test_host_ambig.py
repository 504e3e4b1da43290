"""sa_load_ambig (the -a file, create_ambig_bases2 impl/pairwiseAligner.c:68-92) against the reference's own fixture and
known answer: tests/signalPairwiseAlignerTest.c:789-796 reads tests/test_position_code/test_positions_encoding.positions
and expects L -> "asdf".  The two fixture files are committed as data under tests/golden/position_code/.  No GPU."""
import ctypes as C
import os

import signalalign_amd as sa
from signalalign_amd import _capi

HERE = os.path.dirname(os.path.abspath(__file__))
POS = os.path.join(HERE, "golden", "position_code")


def _load(path):
    arr = (C.c_char_p * 256)()
    rc = _capi.lib().sa_load_ambig(path.encode(), arr)
    return rc, arr


def test_reference_positions_encoding_fixture():
    rc, arr = _load(os.path.join(POS, "test_positions_encoding.positions"))
    assert rc == 0
    assert arr[ord("L")] == b"asdf"            # the reference's known answer
    assert arr[ord("E")] == b"af" and arr[ord("C")] == b"EF"
    # the file REPLACES the built-in table (create_ambig_bases2 builds a fresh hash): X is not ambiguous any more
    assert arr[ord("X")] is None and arr[ord("A")] is None
    assert sum(1 for i in range(256) if arr[i] is not None) == 3


def test_reference_ambig_model_fixture_and_errors(tmp_path):
    rc, arr = _load(os.path.join(POS, "test_ambig.model"))
    assert rc == 0 and arr[ord("O")] == b"AD"
    rc, _ = _load(str(tmp_path / "missing.positions"))
    assert rc == sa.SA_EIO if hasattr(sa, "SA_EIO") else rc == -6
    # the built-in table (create_ambig_bases, impl/pairwiseAligner.c:32-65; known answer X -> ACGT, :780-786)
    d = sa.default_ambig()
    assert d[ord("X")] == b"ACGT" and d[ord("L")] == b"CEO" and d[ord("P")] == b"CE"
    # buffers of the reference: at most 300 lines are read
    p = tmp_path / "many.positions"
    p.write_text("".join("%s\tAC\n" % chr(33 + (i % 90)) for i in range(400)))
    rc, arr = _load(str(p))
    assert rc == 0 and arr[ord("!")] == b"AC"


def test_number_parser_rounds_like_strtod(tmp_path):
    """The loaders parse decimal text with a one-multiplication fast path (sa_io.c:sa_atod) and fall back to strtod for anything
    it cannot convert exactly: the values must equal Python's float() (correctly rounded, as glibc's strtod) bit for bit."""
    import numpy as np
    rng = np.random.default_rng(3)
    toks = ["0", "-0.0", "1", "102.77682312", "1.17174478293", "0.001", "5e-324", "1e22", "1e23", "1.7976931348623157e308",
            "123456789012345", "1234567890123456", "0.1234567890123456789", "9007199254740993", "4.35e-5", "6.02E+23",
            "000012.5000", ".5", "5.", "-3.25e-7", "+2.5", "0.000000000000000000000123", "1e-22", "1e-23", "0x1p3", "inf"]
    toks += ["%.*g" % (int(rng.integers(1, 18)), v) for v in rng.normal(0, 1, 400) * 10.0 ** rng.integers(-12, 12, 400)]
    toks += ["%.6f" % v for v in rng.uniform(0, 200, 400)]
    n_kmers = 4 ** 3
    while len(toks) < 5 * n_kmers:
        toks.append("1.5")
    toks = toks[:5 * n_kmers]
    path = tmp_path / "tiny.model"
    path.write_text("3\t4\tACGT\t3\n" + "\t".join(["0.1"] * 10) + "\n" + "\t".join(toks) + "\n")
    m = sa.Model.load(str(path))
    got = np.array(m.table5())
    exp = np.array([float(t) if not t.startswith("0x") else float.fromhex(t) for t in toks])
    assert got.tobytes() == exp.tobytes()


def test_result_pairs_survive_the_16_byte_record():
    """Pairs cross PCIe as 16-byte records (signalalign_amd/csrc/sa_internal.h: x and y in 28 bits, prob_e7 in 24, path in 16,
    kmer_id in 32) and sa_batch_pairs expands them: the extremes of every field and random values come back unchanged."""
    import numpy as np
    rng = np.random.default_rng(11)
    n = 5000
    a = np.zeros(n, dtype=_capi.PAIR_DTYPE)
    a["prob_e7"] = rng.integers(0, 10_000_001, n)
    a["x"] = rng.integers(0, 1 << 28, n)
    a["y"] = rng.integers(0, 1 << 28, n)
    a["path"] = rng.integers(0, 1 << 16, n)
    a["kmer_id"] = rng.integers(0, (1 << 31) - 1, n)
    a[0] = (10_000_000, (1 << 28) - 1, (1 << 28) - 1, 65535, (1 << 31) - 1)
    a[1] = (0, 0, 0, 0, 0)
    a[2] = (100_000, 3000, 5000, 728, 46655)          # hdCell worst case: 729 paths (tests/signalPairwiseAlignerTest.c:543-568)
    b = np.zeros(n, dtype=_capi.PAIR_DTYPE)
    rc = _capi.lib().sa_pair_roundtrip(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_int64(n))
    assert rc == 0 and a.tobytes() == b.tobytes()


def test_format_f6_prints_what_printf_prints():
    """sa_format_f6 (the TSV writers' "%f", signalalign_amd/csrc/sa_io.c) against the C library's formatter -- Python's % uses the
    same exact-decimal, ties-to-even conversion -- on magnitudes the writers see, on exact ties (binary fractions that end at the
    seventh decimal), on values that carry into the integer part, on signed zeros, denormals, huge values, inf and nan."""
    import ctypes as C
    import numpy as np
    L = sa.lib()
    L.sa_format_f6.argtypes = [C.c_char_p, C.c_double]
    L.sa_format_f6.restype = C.c_int
    buf = C.create_string_buffer(400)

    def f6(v):
        n = L.sa_format_f6(buf, v)
        assert buf.raw[n] == 0
        return buf.raw[:n].decode()

    rng = np.random.default_rng(5)
    vals = list(rng.uniform(-200, 200, 50000)) + list(rng.uniform(0, 1, 50000)) + list(rng.uniform(0, 1e-5, 20000))
    vals += list(10.0 ** rng.uniform(-12, 15, 50000)) + list(-(10.0 ** rng.uniform(-12, 15, 5000)))
    for j in range(1, 30):
        for k_ in range(64):
            vals += [k_ / 2.0 ** j, k_ / 2.0 ** j + 5e-7, -(k_ / 2.0 ** j)]
    vals += [0.0, -0.0, 0.5e-6, 1.5e-6, 2.5e-6, 0.9999995, 0.99999949999, 1e-300, -1e-300, 5e-324, 123456789.1234565, 9.0e15 - 1,
             2.0 ** 52 + 0.5, 0.1, 0.125, 0.0000005, 0.0000015, float("inf"), float("-inf"), 1e300, -1e22, 9.1e15, 999999.9999995]
    for v in vals:
        assert f6(float(v)) == "%f" % float(v), repr(v)
    assert f6(float("nan")) in ("nan", "-nan")
