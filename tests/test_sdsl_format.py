"""Reader of the reference's own index files (sdsl-lite serialisation of csa_wt_int<>, reference
seal/cpp_modules/fm_index.cpp:186-199, seal/index.py:186-204) against files written by the oracle in the same layout.
PARITY UNPINNED below the SWIG boundary: sdsl is not in this image, both sides follow its published sources; what IS
pinned here is that the reader recovers exactly the text / suffix array / quirk table the file encodes, refuses what it
cannot vouch for, and that an index loaded this way answers like the oracle (-m gpu)."""
import ctypes
import os

import numpy as np
import pytest

from oracle.seal_oracle import OracleFMIndex, brute_sa, brute_text
from oracle.seal_oracle import lib as orc_lib
from seal_amd._lib import SealFMError, check, lib
from tests.helpers import make_docs


def _host_array(h, name, dt):
    n, e = ctypes.c_uint64(), ctypes.c_uint32()
    p = lib().fmi_host_array(h, name.encode(), ctypes.byref(n), ctypes.byref(e))
    assert p and e.value == np.dtype(dt).itemsize
    return np.frombuffer(ctypes.string_at(p, n.value * e.value), dtype=dt)


def _corpora():
    rng = np.random.default_rng(0)
    yield "mixed", make_docs(3, 80, 300, min_len=4, max_len=30)
    yield "tiny alphabet (quirk Q1 fires)", [rng.integers(4, 7, size=int(rng.integers(2, 9))).tolist() + [2] for _ in range(40)]
    yield "one document", [[5, 6, 7, 5, 6, 2]]
    for n in (63, 64, 65, 127, 128, 129):              # text lengths around the sample densities (SA/32, ISA/64)
        yield f"{n} symbols", [rng.integers(4, 40, size=n - 2).tolist() + [2]]


@pytest.mark.parametrize("name,docs", list(_corpora()), ids=[n for n, _ in _corpora()])
def test_reader_recovers_what_the_sdsl_file_encodes(name, docs, tmp_path):
    orc = OracleFMIndex()
    orc.initialize(docs)
    path = str(tmp_path / "ref.fmi")
    orc.save_sdsl(path)
    h = ctypes.c_void_p()
    check(lib().fmi_load(ctypes.byref(h), path.encode(), -1))          # host only: parse + LF walks + host builder
    try:
        text, _ = brute_text(docs)
        n = len(text)
        assert lib().fmi_size(h) == n == orc.size()
        assert _host_array(h, "text", np.uint16).tolist() == text
        assert _host_array(h, "sa", np.uint32).tolist() == brute_sa(text)
        # the quirk table comes from the FILE's bit layout, through sdsl's rank loop: rank(size()+1, c) - occ(c)
        q1 = _host_array(h, "q1", np.uint8)
        # (symbols that do not occur never reach a rank: backward search answers (1, 0) for them, fm_index.cpp:67-76)
        occ = [orc_lib().orc_rank(orc._h, n, c) for c in range(len(q1))]
        want = [orc_lib().orc_rank(orc._h, n + 1, c) - occ[c] if occ[c] else 0 for c in range(len(q1))]
        assert q1.tolist() == want
        if "quirk" in name:
            assert sum(want) > 0
    finally:
        lib().fmi_free(h)


def test_reader_refuses_files_it_cannot_vouch_for(tmp_path):
    orc = OracleFMIndex()
    orc.initialize(make_docs(1, 30, 50))
    path = str(tmp_path / "ref.fmi")
    orc.save_sdsl(path)
    raw = open(path, "rb").read()
    bad = {"truncated": raw[:len(raw) // 2], "trailing bytes": raw + b"\\0" * 8, "size field": (12345).to_bytes(8, "little") + raw[8:],
           "not an index": os.urandom(4096)}
    flipped = bytearray(raw)
    flipped[40] ^= 0x10                                  # one bit of the wavelet tree: the LF walks no longer close
    bad["one flipped tree bit"] = bytes(flipped)
    for what, blob in bad.items():
        p = str(tmp_path / "bad.fmi")
        open(p, "wb").write(blob)
        h = ctypes.c_void_p()
        rc = lib().fmi_load(ctypes.byref(h), p.encode(), -1)
        assert rc != 0, what
        assert b"sdsl" in lib().fmi_last_error() or b"short" in lib().fmi_last_error(), what


def test_reader_survives_corrupted_sample_and_alphabet_sections(tmp_path):
    """byte-level fuzz of the part of the file the reader DECODES with file-supplied sizes (SA / ISA samples, the
    alphabet's sd_vector -- low bits, unary high bits --, C): every mutated file is either refused or loads as an index of
    the original size; none may crash the process or index out of bounds (the alphabet decode checks every entry: strictly
    ascending symbols below the character range, no more entries than the low-bit vector holds)"""
    orc = OracleFMIndex()
    orc.initialize(make_docs(2, 40, 300, min_len=3, max_len=12))
    path = str(tmp_path / "ref.fmi")
    orc.save_sdsl(path)
    raw = open(path, "rb").read()
    rng = np.random.default_rng(0)
    tail = len(raw) * 2 // 5
    refused = loaded = 0
    for trial in range(400):
        blob = bytearray(raw)
        for _ in range(int(rng.integers(1, 4))):
            pos = len(raw) - 1 - int(rng.integers(0, tail))
            blob[pos] = int(rng.integers(0, 256)) if trial % 2 else blob[pos] ^ (1 << int(rng.integers(0, 8)))
        p = str(tmp_path / "fuzz.fmi")
        open(p, "wb").write(bytes(blob))
        h = ctypes.c_void_p()
        rc = lib().fmi_load(ctypes.byref(h), p.encode(), -1)
        if rc == 0:
            loaded += 1
            assert lib().fmi_size(h) == orc.size()
            lib().fmi_free(h)
        else:
            refused += 1
    assert refused > 200 and refused + loaded == 400


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_index_loaded_from_an_sdsl_file_answers_like_the_oracle(seed, tmp_path):
    """``FMIndex.load`` of the reference's pair of files (``.fmi`` = sdsl's, ``.oth`` = the pickle of index.py:190-192)"""
    import pickle
    from seal_amd import FMIndex
    docs = make_docs(seed, 120, 60 if seed else 9, min_len=3, max_len=20, title_sep=7)
    orc = OracleFMIndex()
    orc.initialize(docs)
    base = str(tmp_path / "index")
    orc.save_sdsl(base + ".fmi")
    with open(base + ".oth", "wb") as f:
        pickle.dump((orc.beginnings, orc.occurring, [f"d{i}" for i in range(len(docs))]), f)
    ix = FMIndex.load(base)
    assert ix.size() == orc.size() and ix.n_docs == orc.n_docs and ix.labels[3] == "d3"
    assert (ix.occurring_distinct, ix.occurring_counts) == (orc.occurring_distinct, orc.occurring_counts)
    rng = np.random.default_rng(seed)
    for t in ix.occurring_distinct:                      # every first step, quirk rows included
        assert ix.get_range([t]) == orc.get_range([t]), t
    for _ in range(200):
        d = docs[int(rng.integers(len(docs)))]
        a = int(rng.integers(0, len(d)))
        seq = d[a:a + int(rng.integers(1, 5))]
        lo, hi = orc.get_range(seq)
        assert ix.get_range(seq) == (lo, hi)
        assert ix.get_distinct_count(lo, hi) == orc.get_distinct_count(lo, hi)
        for r in range(lo, min(hi, lo + 4, orc.size())):
            assert ix.locate(r) == orc.locate(r) and ix.get_doc_index_from_row(r) == orc.get_doc_index_from_row(r)
    assert [ix.get_doc(i) for i in range(len(docs))] == docs
    ix.save(str(tmp_path / "native"))                   # and back out in the engine's own container
    again = FMIndex.load(str(tmp_path / "native"))
    assert again.get_range(docs[0][:2]) == orc.get_range(docs[0][:2])


@pytest.mark.gpu
def test_published_nq_index_reproduces_the_readme_numbers():
    """hook for the day a real SEAL index is on the box: reference README.md:105-113 -- keys ' eating soup' with corpus
    frequency 10 and ' fork' with 9390 on SEAL_NQ.  Needs SEAL_NQ_INDEX (path prefix of the .fmi/.oth pair) and a BART
    tokenizer directory in SEAL_BART_TOKENIZER; skipped otherwise (neither can be fetched offline)."""
    prefix, tok_dir = os.environ.get("SEAL_NQ_INDEX"), os.environ.get("SEAL_BART_TOKENIZER")
    if not prefix or not tok_dir or not os.path.exists(prefix + ".fmi"):
        pytest.skip("SEAL_NQ_INDEX / SEAL_BART_TOKENIZER not provided")
    from transformers import AutoTokenizer
    from seal_amd import FMIndex
    tok = AutoTokenizer.from_pretrained(tok_dir)
    ix = FMIndex.load(prefix)
    for text, freq in ((" eating soup", 10), (" fork", 9390)):
        ids = tok(text, add_special_tokens=False)["input_ids"]
        assert ix.get_count(ids) == freq, text
