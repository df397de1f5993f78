"""seal_amd -- MI355X-native FM-index constrained decoding, drop-in for the hot
path of facebookresearch/SEAL (``seal.FMIndex``, ``seal.fm_index_generate``,
``seal.IndexBasedLogitsProcessor``, ``seal.SEALSearcher``; reference
seal/__init__.py:7-9)."""
from .index import FMIndex  # noqa: F401
from .beam_search import IndexBasedLogitsProcessor, fm_index_generate  # noqa: F401

__all__ = ["FMIndex", "fm_index_generate", "IndexBasedLogitsProcessor"]
