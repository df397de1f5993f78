"""CPU-side tests of libsealfm.so: the ABI surface, the host builder and the
on-disk format.  No query kernels run here (there is no GPU in this container)."""
import ctypes
import os
import random
import re

import numpy as np
import pytest

from oracle.seal_oracle import SHIFT, CppFMIndex, brute_bwt, brute_sa, brute_text, lib as orc_lib
from seal_amd._lib import SIGNATURES, SealFMError, check, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_p64 = ctypes.POINTER(ctypes.c_uint64)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sealfm.h")).read()
    declared = set(re.findall(r"\b(fmi_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(SIGNATURES), (declared ^ set(SIGNATURES))
    L = lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.fmi_abi_version() == 1
    from seal_amd._lib import NN_SIGNATURES
    nn = set(re.findall(r"\b(sealnn_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", "sealnn.h")).read()))
    assert nn == set(NN_SIGNATURES) and all(hasattr(L, n) for n in nn)


def test_the_library_carries_the_digest_of_its_sources_and_another_one_is_refused(tmp_path, monkeypatch):
    """the digest of the sources, headers and flags lives INSIDE libsealfm.so (``fmi_source_digest()``): read from the file without loading it
    (``_build.built_digest``) and from the mapped image at load; a binary built from other sources -- here: the same file with one digit of its
    digest changed, dropped in place of the real one, no sidecar to forge -- is refused before anything is called in it"""
    import shutil
    from seal_amd import _build, _lib
    want = _build.source_digest()
    assert len(want) == 64 and _build.built_digest() == want and not _build.stale()
    assert lib().fmi_source_digest().decode() == want
    forged = tmp_path / "libsealfm.so"
    blob = open(_build.LIB, "rb").read()
    at = blob.find(_build.MARKER) + len(_build.MARKER)
    assert at > len(_build.MARKER) and blob.count(_build.MARKER) == 1
    other = b"0" if blob[at:at + 1] != b"0" else b"1"
    forged.write_bytes(blob[:at] + other + blob[at + 1:])
    monkeypatch.setattr(_build, "LIB", str(forged))
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.delenv("SEALFM_ALLOW_STALE_LIB", raising=False)
    _build._digest_cache.clear()
    try:
        assert _build.stale() and _build.built_digest() != want
        with pytest.raises(ImportError, match="built from other sources"):
            _lib.lib()
    finally:
        _build._digest_cache.clear()


def _host_index(data):
    h = ctypes.c_void_p()
    check(lib().fmi_create(ctypes.byref(h)))
    a = np.ascontiguousarray(np.asarray(data, dtype=np.uint64))
    check(lib().fmi_build(h, a.ctypes.data_as(_p64), len(a), -1))   # host only, no upload
    return h


def _arr(h, name):
    n = ctypes.c_uint64()
    e = ctypes.c_uint32()
    p = lib().fmi_host_array(h, name.encode(), ctypes.byref(n), ctypes.byref(e))
    assert p, name
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[e.value]
    buf = (ctypes.c_uint8 * (n.value * e.value)).from_address(p)
    return np.frombuffer(buf, dtype=dt).copy()


def _wm_child_positions(wm, sbase, nblk, nsb, sb_shift, k, p):
    """Python restatement of the documented 128-byte block layout: where position p
    of level k goes on level k+1 for each of the sixteen digits."""
    blk, within = divmod(p, 128)
    base = (k * nblk + blk) * 16                      # u64 words
    dwords = []
    for w in wm[base:base + 16]:
        dwords += [int(w) & 0xFFFFFFFF, int(w) >> 32]
    row = sbase[(k * nsb + (blk >> sb_shift)) * 16:][:16]
    out = [int(row[d]) + dwords[d] for d in range(16)]
    planes = [sum(dwords[16 + 4 * j + x] << (32 * x) for x in range(4)) for j in range(4)]
    for i in range(within):
        out[sum(((planes[j] >> i) & 1) << j for j in range(4))] += 1
    return out


def _rand_data(rng, n, vocab):
    return [rng.randrange(1, vocab) for _ in range(n)]


@pytest.mark.parametrize("seed,n,vocab,force_sb", [(0, 50, 4, None), (1, 400, 30, None), (2, 3000, 700, None), (3, 5000, 60000, None),
                                                   (4, 2000, 2, None), (2, 3000, 700, 1), (3, 5000, 60000, 0)])
def test_host_builder_matches_brute_force(seed, n, vocab, force_sb, monkeypatch):
    # force_sb: superblocks of 2^force_sb blocks, the layout texts beyond 2^32 symbols get
    if force_sb is not None:
        monkeypatch.setenv("SEALFM_FORCE_SB", str(force_sb))
    sb_shift = 40 if force_sb is None else force_sb
    rng = random.Random(seed)
    data = _rand_data(rng, n, vocab)
    if seed == 1:   # long repeats: stress the doubling rounds
        data = (data[:40] * 10)[:n]
    h = _host_index(data)
    try:
        text = data + [0]
        sa = brute_sa(text)
        bwt = brute_bwt(text, sa)
        N = len(text)
        assert lib().fmi_size(h) == N
        assert _arr(h, "sa").tolist() == sa
        assert _arr(h, "bwt").tolist() == bwt
        assert _arr(h, "text").tolist() == text
        L = lib().fmi_levels(h)
        assert L == max(1, max(text).bit_length())
        C = _arr(h, "C")
        max_sym = max(text)
        assert len(C) == max_sym + 2
        for c in set(text):
            assert C[c] == sum(1 for x in text if x < c)
        assert lib().fmi_sigma(h) == len(set(text))
        wm, dbase, sbase, leaf = _arr(h, "wm"), _arr(h, "dbase"), _arr(h, "sbase"), _arr(h, "leaf")
        D = (L + 3) // 4
        assert len(dbase) == 16 * D
        nblk = len(wm) // (16 * D)
        assert nblk == N // 128 + 2
        nsb = len(sbase) // (16 * D)
        assert nsb == (nblk >> sb_shift) + 1
        if nsb == 1:
            assert sbase.tolist() == dbase.tolist()
        # rank_c(i) through the hex wavelet matrix == naive count
        for _ in range(300):
            c = rng.choice(text)
            i = rng.randrange(0, N + 1)
            p = i
            for k in range(D):
                d = (c >> (4 * (D - 1 - k))) & 15
                p = _wm_child_positions(wm, sbase, nblk, nsb, sb_shift, k, p)[d]
            assert p - int(leaf[c]) == bwt[:i].count(c)
        # quirk table == what the faithful sdsl-layout oracle computes for rank(size()+1, c)
        orc = CppFMIndex()
        orc.initialize(data)
        q1 = _arr(h, "q1")
        for c in set(text):
            occ = text.count(c)
            assert int(orc_lib().orc_rank(orc._h, N + 1, c)) - occ == int(q1[c]), c
    finally:
        lib().fmi_free(h)


def test_zipf_corpus_q1_against_oracle():
    rng = np.random.default_rng(5)
    data = (np.minimum(rng.zipf(1.3, size=20000), 50000) + SHIFT).tolist()
    h = _host_index(data)
    try:
        orc = CppFMIndex()
        orc.initialize(data)
        N = len(data) + 1
        q1 = _arr(h, "q1")
        C = _arr(h, "C")
        fired = 0
        for c in set(data) | {0}:
            occ = int(C[c + 1] - C[c])
            d = int(orc_lib().orc_rank(orc._h, N + 1, c)) - occ
            assert d == int(q1[c])
            fired += d
        assert fired <= 8   # rare
    finally:
        lib().fmi_free(h)


@pytest.mark.parametrize("force_sb", [None, 1])
def test_save_load_round_trip(tmp_path, force_sb, monkeypatch):
    rng = random.Random(9)
    data = _rand_data(rng, 1000, 300)
    if force_sb is not None:          # the superblocked layout; the file carries sb_shift, the loader must not need the env
        monkeypatch.setenv("SEALFM_FORCE_SB", str(force_sb))
    h = _host_index(data)
    monkeypatch.delenv("SEALFM_FORCE_SB", raising=False)
    path = str(tmp_path / "x.fmi").encode()
    check(lib().fmi_save(h, path))
    h2 = ctypes.c_void_p()
    check(lib().fmi_load(ctypes.byref(h2), path, -1))
    try:
        for name in ("sa", "bwt", "text", "C", "leaf", "q1", "dbase", "sbase", "wm"):
            assert np.array_equal(_arr(h, name), _arr(h2, name)), name
        assert lib().fmi_size(h2) == 1001 and lib().fmi_levels(h2) == lib().fmi_levels(h)
        assert len(_arr(h2, "sbase")) == (16 * ((1001 // 128 + 2 >> force_sb) + 1) if force_sb is not None else 16) * ((lib().fmi_levels(h) + 3) // 4)
    finally:
        lib().fmi_free(h)
        lib().fmi_free(h2)


def test_build_from_file_little_endian(tmp_path):
    data = [11, 12, 13, 11, 12, 300]
    p = tmp_path / "d.bin"
    np.asarray(data, dtype="<i4").tofile(p)   # FORMAT '<l', width 4 (reference index.py:18,65)
    h = ctypes.c_void_p()
    check(lib().fmi_create(ctypes.byref(h)))
    check(lib().fmi_build_from_file(h, str(p).encode(), 4, -1))
    assert _arr(h, "text").tolist() == data + [0]
    lib().fmi_free(h)


def test_errors_are_loud_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = _host_index([5, 6, 7])
    out = np.zeros(2, dtype=np.uint64)
    with pytest.raises(SealFMError) as e:
        check(lib().fmi_backward_search_step(h, 5, 0, 3, out.ctypes.data_as(_p64)))
    assert e.value.code == 3 and "no CPU query path" in str(e.value)
    with pytest.raises(SealFMError):
        check(lib().fmi_to_device(h, 0))
    # zero symbols are rejected like sdsl would (text must not contain the sentinel)
    bad = np.asarray([4, 0, 4], dtype=np.uint64)
    with pytest.raises(SealFMError):
        check(lib().fmi_build(h, bad.ctypes.data_as(_p64), 3, -1))
    lib().fmi_free(h)


def test_lightning_loader_accepts_both_layouts_and_rejects_a_wrong_checkpoint(tmp_path):
    """reference seal/utils.py:31-39 loads a PLAIN HF state dict strictly; lightning wrappers prefix keys with
    "model." under "state_dict".  Both must load to the same weights; a checkpoint of another geometry must raise
    instead of loading silently with missing keys."""
    import pytest
    import torch
    from seal_amd.utils import load_state_dict_from_lightning_checkpoint
    from tests.helpers import tiny_bart
    src = tiny_bart(120, seed=5)
    plain, wrapped = str(tmp_path / "plain.pt"), str(tmp_path / "wrapped.pt")
    torch.save(src.state_dict(), plain)
    torch.save({"state_dict": {"model." + k: v for k, v in src.state_dict().items()}}, wrapped)
    for path in (plain, wrapped):
        dst = tiny_bart(120, seed=9)
        load_state_dict_from_lightning_checkpoint(dst, path)
        for k, v in src.state_dict().items():
            assert torch.equal(dst.state_dict()[k], v), k
    other = tiny_bart(120, seed=5, layers=1)
    bad = str(tmp_path / "bad.pt")
    torch.save(other.state_dict(), bad)
    with pytest.raises(RuntimeError, match="does not match"):
        load_state_dict_from_lightning_checkpoint(tiny_bart(120, seed=9), bad)


def test_beam_step_struct_matches_the_header_field_for_field(tmp_path):
    """``_lib.FmiBeamStep`` (ctypes) against ``fmi_beam_step_t`` as a C compiler lays it out from include/sealfm.h"""
    import ctypes
    import os
    import subprocess
    from seal_amd._lib import FmiBeamStep
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = [f[0] for f in FmiBeamStep._fields_]
    src = tmp_path / "offs.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s/include/sealfm.h"\nint main(void) {\n%s\n  printf("%%zu\\n", sizeof(fmi_beam_step_t));\n  return 0;\n}\n'
                   % (root, "\n".join('  printf("%%zu\\n", offsetof(fmi_beam_step_t, %s));' % n for n in names)))
    exe = tmp_path / "offs"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got[:-1] == [getattr(FmiBeamStep, n).offset for n in names]
    assert got[-1] == ctypes.sizeof(FmiBeamStep)
