"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the SEAL FM-index path.

PARITY UNPINNED for the C++ layer (see oracle/fm_oracle.c header): sdsl-lite + SWIG are absent and the
reference tree holds no golden vectors, so ``CppFMIndex`` is pinned by brute force only.  The Python
layer above it is pinned: tests/golden/ref_index_and_mask.json was produced by the reference's own
``seal/index.py`` running on ``CppFMIndex`` (tests/golden/make_reference_golden.py) and
tests/test_reference_golden.py checks ``OracleFMIndex`` against it.

Two layers, each a restatement of a reference layer:

* ``CppFMIndex``   -- the SWIG class ``seal.cpp_modules.fm_index.FMIndex``
  (reference seal/cpp_modules/fm_index.{hpp,cpp,i}); arithmetic in
  oracle/fm_oracle.c, bound with ctypes.
* ``OracleFMIndex`` -- ``seal.index.FMIndex`` (reference seal/index.py:20-204):
  SHIFT re-basing, per-document reversal, ``beginnings``, the ``d > 0`` filters.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module.  Nothing under seal_amd/ does.
"""
import bisect
import ctypes
import os
import subprocess
from typing import Iterable, List, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfm_oracle.so")

SHIFT = 10  # reference seal/index.py:16

_u64 = ctypes.c_uint64
_p64 = ctypes.POINTER(ctypes.c_uint64)


def build_oracle_lib(force: bool = False) -> str:
    """Compile oracle/fm_oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "fm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libfm_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle_lib()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_build.restype = ctypes.c_void_p
        L.orc_build.argtypes = [_p64, _u64]
        L.orc_build_from_bwt.restype = ctypes.c_void_p
        L.orc_build_from_bwt.argtypes = [ctypes.POINTER(ctypes.c_uint32), _u64, _p64, _p64]
        L.orc_free.argtypes = [ctypes.c_void_p]
        L.orc_set_threads.argtypes = [ctypes.c_int]
        for name in ("orc_size", "orc_sigma"):
            getattr(L, name).restype = _u64
            getattr(L, name).argtypes = [ctypes.c_void_p]
        L.orc_max_level.restype = ctypes.c_uint32
        L.orc_max_level.argtypes = [ctypes.c_void_p]
        L.orc_backward_search_step.argtypes = [ctypes.c_void_p, _u64, _u64, _u64, _p64]
        L.orc_backward_search_multi.argtypes = [ctypes.c_void_p, _p64, _u64, _p64]
        L.orc_distinct_count.restype = _u64
        L.orc_distinct_count.argtypes = [ctypes.c_void_p, _u64, _u64, _p64]
        L.orc_distinct.restype = _u64
        L.orc_distinct.argtypes = [ctypes.c_void_p, _u64, _u64, _p64]
        L.orc_distinct_count_multi.argtypes = [ctypes.c_void_p, _u64, _p64, _p64,
                                               ctypes.POINTER(_p64), _p64]
        L.orc_free_buf.argtypes = [_p64]
        L.orc_locate.restype = _u64
        L.orc_locate.argtypes = [ctypes.c_void_p, _u64]
        L.orc_extract_text.restype = _u64
        L.orc_extract_text.argtypes = [ctypes.c_void_p, _u64, _u64, _p64]
        for name in ("orc_bwt", "orc_sa", "orc_isa"):
            getattr(L, name).restype = _u64
            getattr(L, name).argtypes = [ctypes.c_void_p, _u64]
        L.orc_rank.restype = _u64
        L.orc_rank.argtypes = [ctypes.c_void_p, _u64, _u64]
        L.orc_get_range_batch.argtypes = [ctypes.c_void_p, _u64, _p64, _p64, _p64, _p64, ctypes.c_int]
        L.orc_locate_bin_batch.argtypes = [ctypes.c_void_p, _u64, _p64, _p64, _u64, _p64, _p64, ctypes.c_int]
        L.orc_distinct_count_sizes.argtypes = [ctypes.c_void_p, _u64, _p64, _p64, _p64, ctypes.c_int]
        L.orc_extract_batch.argtypes = [ctypes.c_void_p, _u64, _p64, _p64, _p64, ctypes.c_int]
        L.orc_save_sdsl.restype = ctypes.c_int
        L.orc_save_sdsl.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_distinct_bitmaps.argtypes = [ctypes.c_void_p, _u64, _p64, _p64, _u64, _u64, ctypes.POINTER(ctypes.c_uint32), _p64, _p64,
                                           ctypes.c_int]
        L.orc_extract_batch_tokens.argtypes = [ctypes.c_void_p, _u64, _p64, _p64, _p64, _p64, ctypes.c_int]
        _lib = L
    return _lib


def _arr(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_p64)


class CppFMIndex:
    """Restates the SWIG-exported C++ class (reference fm_index.hpp:20-43)."""

    def __init__(self):
        self._h = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_free(self._h)
            self._h = None

    # ref cpp:33
    def initialize(self, data) -> None:
        a = _arr(list(data))
        if self._h:
            lib().orc_free(self._h)
        self._h = lib().orc_build(_ptr(a), len(a))

    # ref cpp:43 -- raw little-endian ints of `width` bytes
    def initialize_from_file(self, path: str, width: int) -> None:
        dt = {1: "<u1", 2: "<u2", 4: "<u4", 8: "<u8"}[width]
        self.initialize(np.fromfile(path, dtype=dt).astype(np.uint64))

    def initialize_from_bwt(self, bwt_u32: np.ndarray, sa_samples: np.ndarray, isa_samples: np.ndarray) -> None:
        """bench-only: wrap a BWT + samples built elsewhere (see orc_build_from_bwt)."""
        b = np.ascontiguousarray(bwt_u32, dtype=np.uint32)
        s, i = _arr(sa_samples), _arr(isa_samples)
        self._h = lib().orc_build_from_bwt(b.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(b), _ptr(s), _ptr(i))

    def save_sdsl(self, path: str) -> None:
        """the file the reference's ``FMIndex::save`` would write (sdsl ``store_to_file`` of the csa_wt_int<>, ref cpp:186-189),
        as far as the layout is recalled (fm_oracle.c orc_save_sdsl: parity unpinned)"""
        if lib().orc_save_sdsl(self._h, path.encode()) != 0:
            raise IOError(f"cannot write {path}")

    def size(self) -> int:  # ref cpp:50
        return int(lib().orc_size(self._h))

    def backward_search_step(self, symbol: int, low: int, high: int) -> Tuple[int, int]:  # ref cpp:67
        out = np.zeros(2, dtype=np.uint64)
        lib().orc_backward_search_step(self._h, symbol, low & (2**64 - 1), high & (2**64 - 1), _ptr(out))
        return int(out[0]), int(out[1])

    def backward_search_multi(self, query) -> Tuple[int, int]:  # ref cpp:55
        q = _arr(list(query))
        out = np.zeros(2, dtype=np.uint64)
        lib().orc_backward_search_multi(self._h, _ptr(q), len(q), _ptr(out))
        return int(out[0]), int(out[1])

    def _scratch(self, low, high):
        w = max(0, high - low)
        return np.zeros(2 * min(w, int(lib().orc_sigma(self._h))) + 2, dtype=np.uint64)

    def distinct(self, low: int, high: int) -> Tuple[int, ...]:  # ref cpp:78
        out = self._scratch(low, high)
        k = lib().orc_distinct(self._h, low, high, _ptr(out))
        return tuple(int(x) for x in out[:k])

    def distinct_count(self, low: int, high: int) -> Tuple[int, ...]:  # ref cpp:91
        out = self._scratch(low, high)
        k = lib().orc_distinct_count(self._h, low, high, _ptr(out))
        return tuple(int(x) for x in out[:k])

    def distinct_count_multi(self, lows, highs):  # ref cpp:111
        lo, hi = _arr(list(lows)), _arr(list(highs))
        m = len(lo)
        bufs = (_p64 * m)()
        sizes = np.zeros(m, dtype=np.uint64)
        lib().orc_distinct_count_multi(self._h, m, _ptr(lo), _ptr(hi), bufs, _ptr(sizes))
        ret = []
        for i in range(m):
            k = int(sizes[i])
            ret.append(tuple(int(bufs[i][j]) for j in range(k)))
            lib().orc_free_buf(bufs[i])
        return tuple(ret)

    def locate(self, row: int) -> int:  # ref cpp:163
        return int(lib().orc_locate(self._h, row))

    def extract_text(self, begin: int, end: int) -> Tuple[int, ...]:  # ref cpp:169
        out = np.zeros(max(end - begin, 1), dtype=np.uint64)
        k = lib().orc_extract_text(self._h, begin, end, _ptr(out))
        return tuple(int(x) for x in out[:k])

    # --- batched drivers (cpu baseline) ---
    def get_range_batch(self, seqs: List[List[int]], threads: int = 1):
        """index.py:102-111 per sequence (tokens are shifted here)."""
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(s) for s in seqs])
        toks = _arr([t + SHIFT for s in seqs for t in s]) if int(offs[-1]) else np.zeros(1, dtype=np.uint64)
        lo = np.zeros(len(seqs), dtype=np.uint64)
        hi = np.zeros(len(seqs), dtype=np.uint64)
        lib().orc_get_range_batch(self._h, len(seqs), _ptr(offs), _ptr(toks), _ptr(lo), _ptr(hi), threads)
        return lo, hi

    def locate_bin_batch(self, rows, beginnings, threads: int = 1):
        r, b = _arr(rows), _arr(beginnings)
        pos = np.zeros(len(r), dtype=np.uint64)
        doc = np.zeros(len(r), dtype=np.uint64)
        lib().orc_locate_bin_batch(self._h, len(r), _ptr(r), _ptr(b), len(b), _ptr(pos), _ptr(doc), threads)
        return pos, doc

    def extract_batch(self, begins, ends, threads: int = 1):
        b, e = _arr(begins), _arr(ends)
        n = np.zeros(len(b), dtype=np.uint64)
        lib().orc_extract_batch(self._h, len(b), _ptr(b), _ptr(e), _ptr(n), threads)
        return n

    def distinct_bitmaps(self, lows, highs, vocab: int, threads: int = 1):
        """get_distinct_count_multi per interval as bitmaps over raw token ids: (bits uint32 [m, ceil(vocab/32)],
        number of distinct tokens, sum of their counts)"""
        lo, hi = _arr(lows), _arr(highs)
        words = (int(vocab) + 31) // 32
        bits = np.zeros((len(lo), words), dtype=np.uint32)
        k = np.zeros(len(lo), dtype=np.uint64)
        cs = np.zeros(len(lo), dtype=np.uint64)
        lib().orc_distinct_bitmaps(self._h, len(lo), _ptr(lo), _ptr(hi), SHIFT, words,
                                   bits.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), _ptr(k), _ptr(cs), threads)
        return bits, k, cs

    def extract_batch_tokens(self, begins, ends, threads: int = 1):
        """get_doc for many documents: (flat raw token ids int64, offsets)"""
        b, e = _arr(begins), _arr(ends)
        offs = np.zeros(len(b) + 1, dtype=np.uint64)
        np.cumsum(e - b, out=offs[1:])
        out = np.zeros(max(int(offs[-1]), 1), dtype=np.uint64)
        lib().orc_extract_batch_tokens(self._h, len(b), _ptr(b), _ptr(e), _ptr(offs), _ptr(out), threads)
        return out[:int(offs[-1])].astype(np.int64) - SHIFT, offs.astype(np.int64)

    def distinct_count_sizes(self, lows, highs, threads: int = 1):
        lo, hi = _arr(lows), _arr(highs)
        k = np.zeros(len(lo), dtype=np.uint64)
        lib().orc_distinct_count_sizes(self._h, len(lo), _ptr(lo), _ptr(hi), _ptr(k), threads)
        return k


class OracleFMIndex(CppFMIndex):
    """Restates ``seal.index.FMIndex`` (reference seal/index.py:20-204)."""

    def __init__(self):
        super().__init__()
        self.beginnings: List[int] = [0]
        self.occurring = set()
        self.occurring_distinct: List[int] = []
        self.occurring_counts: List[int] = []
        self.labels = None

    # ref index.py:39-66 (both branches feed the same symbols; the file branch
    # only differs in how sdsl is handed the data)
    def initialize(self, sequences: Iterable[List[int]], in_memory: bool = True) -> None:
        data: List[int] = []
        occurring = set()
        for seq in sequences:
            seq = list(seq)
            self.beginnings.append(self.beginnings[-1] + len(seq))
            occurring |= set(seq)
            data.extend(x + SHIFT for x in reversed(seq))
        self.occurring = list(occurring)
        CppFMIndex.initialize(self, data)
        self.occurring_distinct, self.occurring_counts = self.get_distinct_count(0, len(self))

    def get_doc(self, doc_index: int) -> List[int]:  # ref index.py:68-75
        doc = self.extract_text(self.beginnings[doc_index], self.beginnings[doc_index + 1])
        return [x - SHIFT for x in doc]

    def get_doc_index(self, token_index: int) -> int:  # ref index.py:77-82
        return bisect.bisect_right(self.beginnings, token_index) - 1

    def get_doc_length(self, doc_index: int) -> int:  # ref index.py:84-88
        return self.beginnings[doc_index + 1] - self.beginnings[doc_index]

    def get_token_index_from_row(self, row: int) -> int:  # ref index.py:90-94
        return self.locate(row)

    def get_doc_index_from_row(self, row: int) -> int:  # ref index.py:96-100
        return self.get_doc_index(self.locate(row))

    def get_range(self, sequence: List[int]) -> Tuple[int, int]:  # ref index.py:102-111
        start_row, end_row = 0, self.size()
        for token in sequence:
            start_row, end_row = self.backward_search_step(token + SHIFT, start_row, end_row)
        return start_row, end_row + 1

    def get_count(self, sequence: List[int]) -> int:  # ref index.py:113-118
        start, end = self.get_range(sequence)
        return end - start

    def get_doc_indices(self, sequence: List[int]):  # ref index.py:120-126
        start, end = self.get_range(sequence)
        for row in range(start, end):
            yield self.get_doc_index_from_row(row)

    def get_continuations(self, sequence: List[int]) -> List[int]:  # ref index.py:128-134
        start, end = self.get_range(sequence)
        return self.get_distinct(start, end)

    def get_distinct(self, low: int, high: int) -> List[int]:  # ref index.py:136-142
        return [c - SHIFT for c in self.distinct(low, high) if c > 0]

    @staticmethod
    def _unzip(data):
        distinct, counts = [], []
        for d, c in zip(data[0::2], data[1::2]):
            if d > 0:
                distinct.append(d - SHIFT)
                counts.append(c)
        return distinct, counts

    def get_distinct_count(self, low: int, high: int):  # ref index.py:144-156
        return self._unzip(self.distinct_count(low, high))

    def get_distinct_count_multi(self, lows, highs):  # ref index.py:158-171
        return [self._unzip(d) for d in self.distinct_count_multi(lows, highs)]

    def __len__(self) -> int:  # ref index.py:173-177
        return self.beginnings[-1]

    @property
    def n_docs(self) -> int:  # ref index.py:179-184
        return len(self.beginnings) - 1


# --------------------------------------------------------------------------
# Brute force (ground truth for the oracle itself): naive suffix sort + naive
# substring counting.  Pure python/numpy; small inputs only.
# --------------------------------------------------------------------------
def brute_text(sequences: Iterable[List[int]]) -> Tuple[List[int], List[int]]:
    """reversed, +SHIFT, concatenated, 0 sentinel (index.py:46-62 + sdsl's sentinel)."""
    text: List[int] = []
    beginnings = [0]
    for seq in sequences:
        beginnings.append(beginnings[-1] + len(seq))
        text.extend(x + SHIFT for x in reversed(list(seq)))
    text.append(0)
    return text, beginnings


def brute_sa(text: List[int]) -> List[int]:
    return sorted(range(len(text)), key=lambda i: text[i:])


def brute_bwt(text: List[int], sa: List[int]) -> List[int]:
    return [text[i - 1] for i in sa]  # i == 0 -> text[-1], the sentinel


def brute_range(text: List[int], sa: List[int], shifted_pattern: List[int]) -> Tuple[int, int]:
    """half-open row range of suffixes starting with reversed(pattern)."""
    pat = list(reversed(shifted_pattern))
    m = len(pat)
    rows = [r for r, i in enumerate(sa) if text[i:i + m] == pat]
    if not rows:
        return (0, 0)
    assert rows == list(range(rows[0], rows[-1] + 1))
    return rows[0], rows[-1] + 1
