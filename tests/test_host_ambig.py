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
