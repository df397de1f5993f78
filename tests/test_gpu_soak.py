"""Soak of the searcher's DEFAULT path: ``SEALSearcher.batch_search`` with the decodes of batch i+1
enqueued ahead of batch i's rescoring / aggregation (two streams + the index's retrieval stream) must always return -- as the
reference's ``batch_search`` (seal/retrieval.py:649-718, with its ``imap`` pipeline :762-775) always does -- and must return what
the one-batch-at-a-time path returns.

Round 3's bench stalled on the GPU in exactly this call (BENCH_r03.json).  The cause, found in round 4 (profiles/r4_hang_*): the
hipBLASLt algorithm picks that ``seal_amd/tuned_gemm.py`` replayed inside the captured decode step (PyTorch TunableOp); with the
library's own picks 150 repetitions x 20 overlapped batches at NQ size completed.  The geometry here is the bench's -- BART-large,
beam 15, batch 20, both decodes as one loop -- on an index past the 3 GiB inside which random gathers still hit the TLBs
(DESIGN.md 5.1), 36 overlapped batches per call, several calls."""
import faulthandler
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DOCS = 3_200_000            # x ~137 tokens: 4.4e8 symbols, ~4.2 GiB resident (matrix + suffix array + text)
BATCHES = 36
CALLS = 3
WALL_S = 180                # a stalled GPU ends the test process with every thread's stack instead of hanging the suite


def test_overlapped_batches_complete_and_equal_the_sequential_path():
    import torch
    import bench
    from seal_amd import FMIndex
    from seal_amd.bart_decoder import BartStepDecoder
    from seal_amd.distributed import pack_topk
    from seal_amd.retrieval import SEALSearcher
    from transformers import BartConfig, BartForConditionalGeneration
    dev = torch.device("cuda:0")
    # (pytest captures fd 2: the stacks of a stall go to a file that survives the process)
    log_dir = "gpurun_out" if os.path.isdir("gpurun_out") else "/tmp"
    stall_log = open(os.path.join(log_dir, "test_gpu_soak_stall.txt"), "w")
    faulthandler.dump_traceback_later(WALL_S, exit=True, file=stall_log)
    try:
        data, beg, title_len, ids_by_rank = bench.synth_corpus(DOCS, dev, seed=0, phrases=3_000_000)
        queries, bias = bench.synth_queries(BATCHES * 20, data, beg, title_len, ids_by_rank, dev, seed=1)
        index = FMIndex()
        index.initialize_from_device(data, beg.tolist())
        index.labels = None
        del data
        assert index.device_bytes() >= 4 * 2**30
        torch.manual_seed(0)
        cfg = BartConfig()
        cfg.forced_bos_token_id = None
        with torch.device(dev):
            model = BartForConditionalGeneration(cfg).eval()
        with torch.no_grad():
            for tok in (cfg.pad_token_id, cfg.bos_token_id, bench.VOCAB - 1):
                model.final_logits_bias[0, tok] = float("-inf")
        model._seal_step_decoder = BartStepDecoder(model)

        def search(overlap):
            s = SEALSearcher(index, None, model, add_query_to_keys=True, detokenize=False, beam=15, batch_size=20, overlap=overlap)
            assert s._overlapped() == overlap
            s.logit_bias = bias
            t = time.perf_counter()
            res = s.batch_search(queries, k=100)
            torch.cuda.synchronize()
            return pack_topk(res, 100), time.perf_counter() - t

        want, t_seq = search(False)                          # one batch after the other (captures the graphs)
        assert float((want[:, :, 0] >= 0).float().mean()) > 0.9, "the synthetic queries must retrieve documents"
        for call in range(CALLS):
            got, t_ov = search(True)
            assert torch.equal(got[:, :, 0], want[:, :, 0]), f"call {call}: other documents than the sequential path"
            assert torch.equal(got[:, :, 1], want[:, :, 1]), f"call {call}: other scores than the sequential path"
        print(f"soak: {CALLS} x {BATCHES} overlapped batches, {BATCHES * 20 / t_ov:.0f} queries/s overlapped vs {BATCHES * 20 / t_seq:.0f} sequential",
              file=sys.stderr)
    finally:
        faulthandler.cancel_dump_traceback_later()
        stall_log.close()
        if os.path.getsize(stall_log.name) == 0:
            os.remove(stall_log.name)
