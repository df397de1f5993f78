"""hipBLASLt algorithm picks for the GEMMs of the captured decode step.

The reference runs BART through HF ``generate()`` (seal/beam_search.py:231-238) and takes whatever algorithm the BLAS library's
heuristic returns.  On gfx950 the heuristic's fp32 pick for the step's skinny shapes (M = batch x beams rows: 600 / 300 for the
searcher's defaults, N = 1024 / 3072 / 4096 / 50 265) is up to 27 % slower than the best algorithm the library holds
(tools/gemm_tune_probe2.py: the GEMMs of one 600-row step 2.996 -> 2.672 ms), and the captured step replays the same shapes for
the life of the process -- so the pick is made ONCE, through PyTorch's TunableOp (the GEMM arithmetic itself is untouched: fp32
in, fp32 accumulate; a different algorithm only re-orders the sums, within the 1e-4 score tolerance of north_star).

``SEAL_TUNED_GEMMS``:
    ``file`` (default)  read the picks shipped in ``tuned_gemm_gfx950.csv`` (made on an MI355X with this image's libraries; the
                        file carries the library versions and TunableOp ignores it when they differ); shapes that are not in it
                        run the library default.  Nothing is tuned at run time.
    ``tune:<path>``     additionally time the candidates for every NEW step shape during the eager warm-up steps that precede a
                        graph capture (a few seconds per shape, once), append the picks to ``<path>`` and use them.
    ``0``               leave the library default everywhere.
A process whose user enabled TunableOp themselves (``PYTORCH_TUNABLEOP_ENABLED=1``) is left as they configured it.
"""
import contextlib
import logging
import os

import torch

logger = logging.getLogger(__name__)
SHIPPED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gemm_gfx950.csv")
_mode = None


def setup() -> str:
    """once per process, before the first capture; returns ``off`` / ``user`` / ``file`` / ``tune``"""
    global _mode
    if _mode is not None:
        return _mode
    spec = os.environ.get("SEAL_TUNED_GEMMS", "file")
    if spec == "0" or not torch.cuda.is_available():
        _mode = "off"
        return _mode
    import torch.cuda.tunable as tn
    if tn.is_enabled():
        _mode = "user"
        return _mode
    tn.enable(True)
    tn.tuning_enable(False)
    tn.record_untuned_enable(False)
    _mode = "file"
    if os.path.exists(SHIPPED) and not tn.read_file(SHIPPED):
        logger.warning("tuned GEMM picks in %s were made with other library versions and are ignored (library defaults run)", SHIPPED)
    if spec.startswith("tune:"):
        out = spec[len("tune:"):]
        if not out:
            raise ValueError("SEAL_TUNED_GEMMS=tune:<path> needs the file the picks are appended to")
        tn.set_filename(out)
        if os.path.exists(out):
            tn.read_file(out)
        tn.set_max_tuning_duration(100)
        tn.set_max_tuning_iterations(30)
        _mode = "tune"
    elif spec != "file":
        raise ValueError(f"SEAL_TUNED_GEMMS={spec!r}: expected file, tune:<path> or 0")
    return _mode


@contextlib.contextmanager
def tuning():
    """around the eager warm-up steps of a capture: new shapes are timed here in ``tune`` mode, nowhere else"""
    if setup() != "tune":
        yield
        return
    import torch.cuda.tunable as tn
    before = len(tn.get_results())
    tn.tuning_enable(True)
    try:
        yield
    finally:
        tn.tuning_enable(False)
        if len(tn.get_results()) != before:
            _write(tn, tn.get_filename())


def _write(tn, path: str) -> None:
    """the picks so far in TunableOp's own file format, written now rather than left to the library's at-exit writer (which a
    process that ends abruptly never reaches)"""
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        for name, value in tn.get_validators():
            f.write(f"Validator,{name},{value}\n")
        for op, params, solution, ms in tn.get_results():
            f.write(f"{op},{params},{solution},{ms}\n")
    os.replace(tmp, path)
