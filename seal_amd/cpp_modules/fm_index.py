"""Drop-in for the SWIG module ``seal.cpp_modules.fm_index``
(reference seal/cpp_modules/fm_index.i:7-21, fm_index.hpp:20-45).

Same class name, same ten methods, same value conventions (tuples of python
ints; inclusive interval for ``backward_search_step``, half-open for
``distinct*``; ``2**64-1`` from ``locate`` past the end) -- but every query is a
HIP kernel launch on the MI355X that holds the index (libsealfm.so)."""
import ctypes
from typing import Sequence, Tuple

import numpy as np

from .._lib import _p64, check, lib

_MASK = (1 << 64) - 1


def _arr(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_p64)


def default_device() -> int:
    """GPU the index is uploaded to: LOCAL_RANK-th visible device when launched
    one-process-per-GPU, else torch's current device, else 0."""
    import os
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except Exception:
        pass
    return int(os.environ.get("LOCAL_RANK", "0"))


class FMIndex:
    def __init__(self):
        h = ctypes.c_void_p()
        check(lib().fmi_create(ctypes.byref(h)))
        self._h = h
        self._device = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().fmi_free(h)
            except Exception:
                pass
            self._h = None

    # -- construction ------------------------------------------------------
    def initialize(self, data: Sequence[int]) -> None:  # reference fm_index.cpp:33
        a = _arr(data)
        check(lib().fmi_build(self._h, _ptr(a), len(a), default_device()))

    def initialize_from_file(self, path: str, width: int) -> None:  # reference fm_index.cpp:43
        check(lib().fmi_build_from_file(self._h, path.encode(), int(width), default_device()))

    def save(self, path: str) -> None:  # reference fm_index.cpp:186
        check(lib().fmi_save(self._h, path.encode()))

    # -- queries -----------------------------------------------------------
    def size(self) -> int:  # reference fm_index.cpp:50
        return int(lib().fmi_size(self._h))

    def backward_search_step(self, symbol: int, low: int, high: int) -> Tuple[int, int]:  # cpp:67
        out = np.zeros(2, dtype=np.uint64)
        check(lib().fmi_backward_search_step(self._h, symbol & _MASK, low & _MASK, high & _MASK, _ptr(out)))
        return int(out[0]), int(out[1])

    def backward_search_multi(self, query: Sequence[int]) -> Tuple[int, int]:  # cpp:55
        q = _arr(list(query)) if len(query) else np.zeros(1, dtype=np.uint64)
        out = np.zeros(2, dtype=np.uint64)
        check(lib().fmi_backward_search_multi(self._h, _ptr(q), len(query), _ptr(out)))
        return int(out[0]), int(out[1])

    def _distinct_csr(self, lows, highs, with_counts: bool):
        lo, hi = _arr(lows), _arr(highs)
        n = len(lo)
        offs = np.zeros(n + 1, dtype=np.uint64)
        sigma = int(lib().fmi_sigma(self._h))
        widths = np.where(hi > lo, hi - lo, 0)
        cap = int(np.minimum(widths, sigma).sum()) + 1
        syms = np.zeros(cap, dtype=np.uint64)
        cnts = np.zeros(cap, dtype=np.uint64) if with_counts else None
        check(lib().fmi_distinct_count_multi(self._h, n, _ptr(lo), _ptr(hi), _ptr(offs), _ptr(syms),
                                             _ptr(cnts) if with_counts else None, cap))
        return offs, syms, cnts

    def distinct(self, low: int, high: int) -> Tuple[int, ...]:  # cpp:78
        offs, syms, _ = self._distinct_csr([low], [high], False)
        return tuple(int(x) for x in syms[:int(offs[1])])

    def distinct_count(self, low: int, high: int) -> Tuple[int, ...]:  # cpp:91, flat (c0, n0, c1, n1, ...)
        return self.distinct_count_multi([low], [high])[0]

    def distinct_count_multi(self, lows: Sequence[int], highs: Sequence[int]):  # cpp:111
        offs, syms, cnts = self._distinct_csr(list(lows), list(highs), True)
        ret = []
        for i in range(len(offs) - 1):
            a, b = int(offs[i]), int(offs[i + 1])
            flat = np.empty(2 * (b - a), dtype=np.uint64)
            flat[0::2] = syms[a:b]
            flat[1::2] = cnts[a:b]
            ret.append(tuple(int(x) for x in flat))
        return tuple(ret)

    def locate(self, row: int) -> int:  # cpp:163
        r = _arr([row & _MASK])
        out = np.zeros(1, dtype=np.uint64)
        check(lib().fmi_locate(self._h, 1, _ptr(r), _ptr(out), None))
        return int(out[0])

    def extract_text(self, begin: int, end: int) -> Tuple[int, ...]:  # cpp:169
        if end - begin <= 0:
            return ()
        out = np.zeros(end - begin, dtype=np.uint64)
        check(lib().fmi_extract_text(self._h, begin, end, _ptr(out)))
        return tuple(int(x) for x in out)


def load_FMIndex(path: str) -> FMIndex:  # reference fm_index.cpp:191
    ix = FMIndex.__new__(FMIndex)
    h = ctypes.c_void_p()
    check(lib().fmi_load(ctypes.byref(h), path.encode(), default_device()))
    ix._h = h
    ix._device = None
    return ix
