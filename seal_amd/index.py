"""``seal.index.FMIndex`` on the MI355X engine.

Keeps the reference's Python surface (reference seal/index.py:20-204): same
method names, argument meaning and returns, ``SHIFT`` re-basing, per-document
reversal, ``beginnings`` / ``occurring`` / ``occurring_distinct`` /
``occurring_counts`` / ``labels`` attributes and the ``.oth`` pickle sidecar --
on top of ``seal_amd.cpp_modules.fm_index.FMIndex`` (libsealfm.so, HIP kernels).

Additions (not in the reference) are the ``*_batch`` / ``dev_*`` methods the
GPU decode and retrieval paths use to avoid one launch per Python call.
"""
import bisect
import ctypes
import os
import pickle
from typing import Iterable, Iterator, List, Optional, Sequence, Set, Tuple

import numpy as np

from ._lib import _p64, check, lib


def _h2d(arr, dev):
    from .keys import _h2d as f          # staged through pinned memory: does not block the host (seal_amd/keys.py)
    return f(arr, dev)
from .cpp_modules.fm_index import FMIndex as _FMIndex
from .cpp_modules.fm_index import _arr, _ptr, default_device, load_FMIndex

SHIFT = 10        # reference seal/index.py:16
FORMAT = "<l"     # reference seal/index.py:18


class FMIndex(_FMIndex):
    """FM-index over token-id documents, resident in HBM."""

    beginnings: List[int]
    occurring: Set[int]
    occurring_distinct: List[int]
    occurring_counts: List[int]
    labels: Optional[List[str]]

    def __init__(self):
        super().__init__()
        self.beginnings = [0]
        self.occurring = set()
        self.occurring_distinct = []
        self.occurring_counts = []
        self.labels = None
        self._trace = None      # optional list; bench.py records the index operations of a batch here

    # -- construction (reference index.py:39-66) ---------------------------
    def initialize(self, sequences: Iterable[List[int]], in_memory: bool = False) -> None:
        """Build the index from an iterable of token-id lists.

        ``in_memory`` is accepted for signature compatibility; the reference's
        two branches differ only in how sdsl is handed the symbols (temp file
        of little-endian int32 vs. vector) and produce the same index.
        """
        occurring = set()
        chunks = []
        for seq in sequences:
            seq = np.asarray(list(seq), dtype=np.int64)
            self.beginnings.append(self.beginnings[-1] + len(seq))
            occurring.update(seq.tolist())
            chunks.append(seq[::-1] + SHIFT)
        self.occurring = list(occurring)
        data = np.concatenate(chunks).astype(np.uint64) if chunks else np.zeros(0, dtype=np.uint64)
        _FMIndex.initialize(self, data)
        self._after_build()

    def initialize_from_device(self, data, beginnings, occurring=None, keep_host: bool = False) -> None:
        """Build on the GPU (``fmi_build_device``) from symbols that are ALREADY in
        index order: per-document reversed and +SHIFT, concatenated (what
        reference index.py:50-53 produces), as a uint32/int32 torch tensor on the
        target GPU.  ``beginnings`` = cumulative document lengths (len n_docs+1).
        For corpora the host builder cannot reach (NQ scale)."""
        import torch
        assert data.is_cuda and data.dtype in (torch.int32, torch.uint32) and data.is_contiguous()
        self.beginnings = [int(x) for x in beginnings]
        if occurring is None:
            present = torch.zeros(int(data.max()) + 1, dtype=torch.bool, device=data.device)
            for a in range(0, data.numel(), 1 << 28):      # chunked: torch.unique refuses > 2^31 elements
                present[data[a:a + (1 << 28)].long()] = True
            occurring = (torch.nonzero(present).flatten() - SHIFT).tolist()
        self.occurring = list(occurring)
        torch.cuda.synchronize(data.device)
        check(lib().fmi_build_device(self._h, data.data_ptr(), data.numel(), data.device.index or 0, int(keep_host)))
        self._after_build()

    def initialize_from_device_text(self, text, beginnings, occurring=None, slice_rows: int = 0) -> None:
        """``initialize_from_device`` for texts whose construction workspace there does not fit the GPU (BASELINE configs[4],
        1.4e10 symbols): the suffix array is sorted in slices (``fmi_build_device_sliced``).  ``text``: the symbols in index order
        -- per-document reversed, +SHIFT -- INCLUDING the final 0 sentinel, as an int16 / uint16 (two's-complement view of the
        16-bit symbol) or int32 tensor on the target GPU; it becomes the index's resident text (nothing is copied) and is kept
        alive by this object.  ``slice_rows``: suffixes per slice (0: 2^30)."""
        import torch
        assert text.is_cuda and text.is_contiguous() and text.element_size() in (2, 4) and text.dim() == 1
        self.beginnings = [int(x) for x in beginnings]
        wide = text.element_size() == 4
        if occurring is None:
            present = torch.zeros(1 << 17 if wide else 1 << 16, dtype=torch.bool, device=text.device)
            for a in range(0, text.numel(), 1 << 28):      # chunked: torch refuses index tensors beyond 2^31 elements
                chunk = text[a:a + (1 << 28)].long()
                present[chunk if wide else chunk & 0xFFFF] = True
            present[0] = False                              # the sentinel
            occurring = (torch.nonzero(present).flatten() - SHIFT).tolist()
        self.occurring = list(occurring)
        self._text_tensor = text                            # the library reads it for as long as the index lives
        torch.cuda.synchronize(text.device)
        check(lib().fmi_build_device_sliced(self._h, text.data_ptr(), text.numel(), text.element_size(), text.device.index or 0, int(slice_rows)))
        self._after_build()

    def initialize_rank_only_from_bwt(self, bwt, max_symbol: int) -> None:
        """Rank/select-only index from a BWT already on the GPU (int16/uint16 or int32 tensor with exactly
        one 0): backward search, ranges, counts and continuations work; ``locate``/``get_doc`` raise.
        For the bandwidth stress tier whose suffix array does not fit one GPU (SURVEY.md 8d tier X)."""
        import torch
        assert bwt.is_cuda and bwt.is_contiguous()
        sym_bytes = bwt.element_size()
        torch.cuda.synchronize(bwt.device)
        check(lib().fmi_build_from_bwt_device(self._h, bwt.data_ptr(), bwt.numel(), sym_bytes, int(max_symbol), bwt.device.index or 0))
        self.beginnings = [0, bwt.numel() - 1]
        self.occurring_distinct, self.occurring_counts = self.get_distinct_count(0, len(self))
        self.occurring = list(self.occurring_distinct)
        self.__dict__.pop("_first_bits_cache", None)

    def _after_build(self) -> None:
        self._push_beginnings()
        self.occurring_distinct, self.occurring_counts = self.get_distinct_count(0, len(self))
        # per-corpus caches hung on the index (the first decode step's allowed-token bitmap, beam_search.py) die with the corpus
        self.__dict__.pop("_first_bits_cache", None)

    def _push_beginnings(self) -> None:
        b = np.asarray(self.beginnings, dtype=np.uint64)
        self._beginnings_np = b.astype(np.int64)      # numpy twin of the (possibly 21M-element) python list
        check(lib().fmi_set_doc_beginnings(self._h, _ptr(b), len(b)))

    # -- reference API ------------------------------------------------------
    def get_doc(self, doc_index: int) -> List[int]:  # index.py:68-75
        doc = self.extract_text(self.beginnings[doc_index], self.beginnings[doc_index + 1])
        return [x - SHIFT for x in doc]

    def get_doc_index(self, token_index: int) -> int:  # index.py:77-82
        return bisect.bisect_right(self.beginnings, token_index) - 1

    def get_doc_length(self, doc_index: int) -> int:  # index.py:84-88
        return self.beginnings[doc_index + 1] - self.beginnings[doc_index]

    def get_token_index_from_row(self, row: int) -> int:  # index.py:90-94
        return self.locate(row)

    def get_doc_index_from_row(self, row: int) -> int:  # index.py:96-100
        return self.get_doc_index(self.locate(row))

    def get_range(self, sequence: List[int]) -> Tuple[int, int]:  # index.py:102-111
        # one launch for the whole prefix instead of one SWIG call per token;
        # identical arithmetic: start at (0, size()), one backward step per token
        lo, hi = self.get_range_batch([sequence])
        return int(lo[0]), int(hi[0])

    def get_count(self, sequence: List[int]) -> int:  # index.py:113-118
        start, end = self.get_range(sequence)
        return end - start

    def get_doc_indices(self, sequence: List[int]) -> Iterator[int]:  # index.py:120-126
        start, end = self.get_range(sequence)
        if end > start:
            _, docs = self.locate_batch(np.arange(start, end, dtype=np.uint64))
            for d in docs:
                yield int(d)

    def get_continuations(self, sequence: List[int]) -> List[int]:  # index.py:128-134
        start, end = self.get_range(sequence)
        return self.get_distinct(start, end)

    def get_distinct(self, low: int, high: int) -> List[int]:  # index.py:136-142
        return [c - SHIFT for c in self.distinct(low, high) if c > 0]

    def get_distinct_count(self, low: int, high: int) -> Tuple[List[int], List[int]]:  # index.py:144-156
        return self.get_distinct_count_multi([low], [high])[0]

    def get_distinct_count_multi(self, lows: List[int], highs: List[int]):  # index.py:158-171
        offs, syms, cnts = self._distinct_csr(list(lows), list(highs), True)
        ret = []
        for i in range(len(offs) - 1):
            a, b = int(offs[i]), int(offs[i + 1])
            s, c = syms[a:b], cnts[a:b]
            keep = s > 0                      # drops the sentinel (index.py:153,167)
            ret.append(((s[keep].astype(np.int64) - SHIFT).tolist(), c[keep].astype(np.int64).tolist()))
        return ret

    def __len__(self) -> int:  # index.py:173-177
        return self.beginnings[-1]

    @property
    def n_docs(self) -> int:  # index.py:179-184
        return len(self.beginnings) - 1

    def save(self, path: str) -> None:  # index.py:186-192
        with open(path + ".oth", "wb") as f:
            pickle.dump((self.beginnings, self.occurring, self.labels), f)
        return super().save(path + ".fmi")

    @classmethod
    def load(cls, path: str) -> "FMIndex":  # index.py:194-204
        index = load_FMIndex(path + ".fmi")
        index.__class__ = cls
        with open(path + ".oth", "rb") as f:
            index.beginnings, index.occurring, index.labels = pickle.load(f)
        index._after_build()
        return index

    # -- batched extras (GPU-friendly forms of the calls above) -------------
    def get_range_batch(self, sequences: Sequence[Sequence[int]]):
        """``get_range`` for many sequences in one launch -> (lo[], hi[]) uint64.  Runs on the current torch stream with
        torch-managed buffers (no hipMalloc/hipFree on the way: those synchronise the whole device, i.e. every other
        pipeline's stream)."""
        n = len(sequences)
        offs = np.zeros(n + 1, dtype=np.int64)
        if n:
            offs[1:] = np.cumsum([len(s) for s in sequences])
        total = int(offs[-1])
        toks = np.zeros(max(total, 1), dtype=np.int64)
        if total:
            toks[:total] = np.fromiter((t for s in sequences for t in s), dtype=np.int64, count=total)
        return self.get_range_csr(offs, toks)

    def get_range_csr(self, offsets: np.ndarray, tokens: np.ndarray):
        """``get_range_batch`` for sequences that are already arrays: sequence i = ``tokens[offsets[i]:offsets[i + 1]]`` (int64)"""
        import torch
        n = len(offsets) - 1
        if n <= 0:
            return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64)
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        toks = np.ascontiguousarray(tokens, dtype=np.int64)
        if toks.size == 0:
            toks = np.zeros(1, dtype=np.int64)
        dev = torch.device("cuda", lib().fmi_device(self._h))
        st = torch.cuda.current_stream(dev)
        d_off = _h2d(offs, dev)
        d_tok = _h2d(toks, dev)
        out = torch.empty(2, n, dtype=torch.int64, device=dev)
        check(lib().fmi_dev_get_range(self._h, st.cuda_stream, n, d_off.data_ptr(), d_tok.data_ptr(), SHIFT, out[0].data_ptr(), out[1].data_ptr()))
        res = out.cpu().numpy().view(np.uint64)
        lo, hi = res[0], res[1]
        if getattr(self, "_trace", None) is not None:      # bench.py: the operation AND what the GPU answered
            self._trace.append(("ranges", [toks[offs[i]:offs[i + 1]].tolist() for i in range(n)], lo.copy(), hi.copy()))
        return lo, hi

    def get_count_batch(self, sequences: Sequence[Sequence[int]]) -> np.ndarray:
        lo, hi = self.get_range_batch(sequences)
        return (hi - lo).astype(np.int64)

    def locate_batch(self, rows) -> Tuple[np.ndarray, np.ndarray]:
        """(positions, doc indices) for many rows: ``locate`` + ``get_doc_index``."""
        r = _arr(rows)
        pos = np.zeros(len(r), dtype=np.uint64)
        doc = np.zeros(len(r), dtype=np.uint64)
        check(lib().fmi_locate(self._h, len(r), _ptr(r), _ptr(pos), _ptr(doc)))
        return pos, doc

    def _side_stream(self, dev):
        """the stream of the retrieval-side launches.  The index itself uses a non-default stream of its own, so that
        those launches (and the host waiting on them) do not queue behind the decoder's work on torch's current
        stream; a view (``view()``: one per concurrent pipeline) runs everything on its pipeline's current stream."""
        import torch
        if self.__dict__.get("_is_view"):
            return torch.cuda.current_stream(dev)
        # a HIGH-PRIORITY stream.  Not for the priority itself: HIP maps streams onto a few
        # hardware queues per priority level, round robin, and at the default priority this stream has shared its hardware queue with
        # the searcher's rescoring stream (rocprofv3: stream 10 and stream 1 both on queue 2) -- the index kernels then ran in the
        # rescoring's queue order, behind its event waits.  Streams of another priority level get queues of their own
        # (tools/stream_overlap_probe.py, profiles/r5_stream_overlap_probe.txt: a side chain beside graph replays ends after 4.6 ms on a
        # high-priority stream, after the 35 ms of replays on a default-priority one that landed on the replays' queue).
        prio = -1
        st = self.__dict__.get("_svc_stream")
        if st is None or self.__dict__.get("_svc_stream_priority") != prio:
            st = self.__dict__["_svc_stream"] = torch.cuda.Stream(device=dev, priority=prio)
            self.__dict__["_svc_stream_priority"] = prio
        return st

    def set_trace(self, trace) -> None:
        """bench.py: record every index operation (with the GPU's answer) of this index and its views into ``trace``"""
        self._trace = trace
        for v in self.__dict__.get("_views", []):
            v._trace = trace

    def view(self) -> "FMIndex":
        """A second handle on the same resident index (``fmi_view_create``): shares the device arrays and the python-side
        tables, owns the mutable per-pipeline state (constraint workspace and incremental ranges, aggregation
        buffers).  The searcher gives each of its concurrent query-batch pipelines one."""
        h = ctypes.c_void_p()
        check(lib().fmi_view_create(self._h, ctypes.byref(h)))
        v = FMIndex.__new__(FMIndex)
        v.__dict__.update({k: val for k, val in self.__dict__.items() if k not in ("_h", "_svc_stream", "_agg_buffers", "_agg_debug", "_views")})
        v._h = h
        v._device = None
        v.__dict__["_is_view"] = True
        v.__dict__["_parent"] = self          # keeps the owner of the device arrays alive
        self.__dict__.setdefault("_views", []).append(v)
        return v

    def locate_ranges(self, lows, highs, max_per_range: int):
        """``locate`` + ``get_doc_index`` for the first ``max_per_range`` rows of many
        half-open row ranges (``islice(range(*get_range(ngram)), max_hits)``, reference
        keys.py:320-324) in one launch.  Returns (pos, doc, offsets) as int64 numpy."""
        import torch
        lo = np.asarray(lows, dtype=np.int64)
        hi = np.asarray(highs, dtype=np.int64)
        width = np.minimum(np.maximum(hi - lo, 0), int(max_per_range))
        offs = np.zeros(len(lo) + 1, dtype=np.int64)
        np.cumsum(width, out=offs[1:])
        total = int(offs[-1])
        if total == 0:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), offs
        dev = torch.device("cuda", lib().fmi_device(self._h))
        st = self._side_stream(dev)
        with torch.cuda.stream(st):
            d_lo = _h2d(lo, dev)
            d_hi = _h2d(hi, dev)
            d_off = _h2d(offs, dev)
            out = torch.empty(2, total, dtype=torch.int64, device=dev)
            check(lib().fmi_dev_locate_ranges(self._h, st.cuda_stream, len(lo), d_lo.data_ptr(), d_hi.data_ptr(),
                                              int(max_per_range), d_off.data_ptr(), total, out[0].data_ptr(), out[1].data_ptr()))
            res = out.cpu().numpy()
        if getattr(self, "_trace", None) is not None:
            self._trace.append(("locate", lo.copy(), hi.copy(), int(max_per_range), res[0].copy(), res[1].copy()))
        return res[0], res[1], offs

    def get_docs_batch(self, doc_indices, as_arrays: bool = False):
        """``get_doc`` for many documents in one launch (reference index.py:68-75); lists of python
        ints, or int64 numpy views of one flat buffer with ``as_arrays``."""
        import torch
        docs = np.asarray(list(doc_indices), dtype=np.int64)
        if len(docs) == 0:
            return []
        b = self.__dict__.get("_beginnings_np")
        if b is None or len(b) != len(self.beginnings):
            b = self._beginnings_np = np.asarray(self.beginnings, dtype=np.int64)
        lens = b[docs + 1] - b[docs]
        offs = np.zeros(len(docs) + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        dev = torch.device("cuda", lib().fmi_device(self._h))
        st = self._side_stream(dev)
        with torch.cuda.stream(st):
            d_docs = _h2d(docs, dev)
            d_off = _h2d(offs, dev)
            out = torch.empty(max(int(offs[-1]), 1), dtype=torch.int64, device=dev)
            check(lib().fmi_dev_get_docs(self._h, st.cuda_stream, len(docs), d_docs.data_ptr(), d_off.data_ptr(), SHIFT,
                                         out.data_ptr()))
            flat = out.cpu().numpy()
        if getattr(self, "_trace", None) is not None:
            self._trace.append(("docs", docs.copy(), flat[:int(offs[-1])].copy(), offs.copy()))
        if as_arrays == "flat":
            return flat[:int(offs[-1])], offs
        if as_arrays:
            return [flat[offs[i]:offs[i + 1]] for i in range(len(docs))]
        return [flat[offs[i]:offs[i + 1]].tolist() for i in range(len(docs))]

    # -- device-pointer forms (torch tensors on the index's GPU) -------------
    @property
    def handle(self):
        return self._h

    def device_bytes(self) -> int:
        return int(lib().fmi_device_bytes(self._h))
