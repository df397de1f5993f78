"""BASELINE.json configs[3] and configs[4] at their REAL sizes, inside the driver's `pytest -m gpu` run: an index of more than 2^32
symbols cannot be faked at a test size -- the 40-bit suffix array (sa_lo + sa_hi), the superblocked counters of the wavelet matrix
(FMI_SB_SHIFT), the 64-bit two-pass sort of the builder and, at configs[4], the sliced suffix-array construction only exist there
(reference seal/cpp_modules/fm_index.cpp:163-167 needs > 32-bit positions at these sizes too).

Each test runs `bench.py` for ONE timed batch on the tier's workload in a process of its own (the index takes 51 / 142 GiB of HBM:
nothing of it may outlive the test) and reads its line: the complete default search (both decodes, rescoring, device aggregation),
then bench.py's parity leg on the same index -- the suffix-array audit (2^20 adjacent suffix pairs compared symbol by symbol on the
device, sum(SA) = n(n-1)/2, 2^20 LF steps through fmi_dev_bs_step), the CPU oracle's replay of every recorded FM-index operation of a
batch (allowed-token bitmaps as APPLIED, ranges / counts, every 16th / 64th located row and scored document), the aggregation against
the host float64 routines and against oracle/keys_oracle.py, and the float check against HF's forward.  A mismatch anywhere makes
bench.py exit non-zero.  Skipped (with the reason) when the GPU or the host is too small for the tier."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host_ram_gib():
    try:
        with open("/proc/meminfo") as f:
            return int(next(l for l in f if l.startswith("MemAvailable")).split()[1]) / 2**20
    except Exception:
        return 0.0


def _run_tier(tag, argv, need_hbm_gib, need_host_gib, timeout_s):
    import torch
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info(0)
    if free / 2**30 < need_hbm_gib:
        pytest.skip(f"{tag}: {free / 2**30:.0f} GiB of HBM free, the tier needs {need_hbm_gib}")
    if _host_ram_gib() < need_host_gib:
        pytest.skip(f"{tag}: {_host_ram_gib():.0f} GiB of host memory available, the oracle index of this tier needs {need_host_gib}")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--latency-batches", "0",
           "--keys-oracle-queries", "1"] + argv
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):                      # evidence for profiles/ when the builder runs this through gpurun
        with open(os.path.join(out_dir, f"large_tier_{tag}.log"), "w") as f:
            f.write(r.stderr[-20000:])
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"large_tier_{tag}.json"), "w") as f:
            json.dump(line, f)
    return line


def _check_parity(line, min_symbols, tol_key="tol"):
    par = line["parity_check"]
    kinds = par["by_kind"]
    assert par["mismatches"] == 0, kinds
    # every leg ran and compared something
    for k in ("allowed_token_sets", "distinct_symbol_counts", "ranges_and_counts", "located_positions_and_doc_ids", "extracted_document_tokens",
              "aggregated_documents_scores_and_keys", "aggregated_documents_vs_keys_oracle", "suffix_array_audit", "beam_scores", "rescore_scores"):
        assert k in kinds and kinds[k].get("values", 1) > 0 and kinds[k]["mismatches"] == 0, (k, kinds.get(k))
    assert "source" in kinds["allowed_token_sets"]            # the APPLIED bitmaps, not a recomputation
    audit = kinds["suffix_array_audit"]
    assert audit["suffix_order_violations"] == 0 and audit["bwt_lf_violations"] == 0 and audit["sum_of_sa_is_n_choose_2"] and audit["text_equals_input_corpus"]
    assert audit["adjacent_suffix_pairs"] >= 1 << 20
    # the index really is beyond 32-bit positions
    import re
    n = int(re.search(r"(\d+) symbols", line["config"]["workload"]).group(1))
    assert n >= min_symbols > (1 << 32)
    assert line["value"] > 0 and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0
    return n


def test_configs3_kilt_size_index_complete_search_and_oracle_replay():
    """configs[3]: 36 M passages, 4.93 G symbols -- 40-bit suffix array, superblocked counters, the 64-bit prefix-doubling builder;
    BART-large fp32, beam 15, batch 20; oracle replay on every 16th located row; scores within 1e-4 of HF's fp32 forward"""
    line = _run_tier("kilt", ["--docs", "36000000", "--cpu-locate-sample", "16"], need_hbm_gib=150, need_host_gib=120, timeout_s=900)
    n = _check_parity(line, min_symbols=4_800_000_000)
    assert line["config"]["workload"].startswith("configs[3]")
    kinds = line["parity_check"]["by_kind"]
    assert kinds["beam_scores"]["tol"] == 1e-4 and kinds["beam_scores"]["max_abs_err"] <= 1e-4
    assert kinds["rescore_scores"]["max_abs_err"] <= 1e-4
    assert kinds["located_positions_and_doc_ids"]["values"] >= 100_000
    print(f"configs[3]: n = {n}, {line['value']} queries/s, {line['parity_check']['values_compared']} values, 0 mismatches")


def test_configs4_stress_index_complete_search_and_oracle_replay():
    """configs[4]: 100 M passages, 1.37e10 symbols -- the suffix array sorted in slices (fmi_build_device_sliced), 141.5 GiB resident;
    BART-large bf16, beam 30; oracle replay on every 64th located row; the recorded bf16 scores no further from HF's FP32 forward than
    1.5 x HF's own bf16 forward is"""
    line = _run_tier("stress", ["--workload", "stress"], need_hbm_gib=260, need_host_gib=400, timeout_s=1500)
    n = _check_parity(line, min_symbols=13_000_000_000)
    assert line["config"]["workload"].startswith("configs[4]")
    kinds = line["parity_check"]["by_kind"]
    for part in ("body", "title"):
        v = kinds["beam_scores"][part]["vs_fp32"]
        assert not v["violation"] and v["max_abs_err_vs_hf_fp32"] <= v["tol_vs_fp32"]
    print(f"configs[4]: n = {n}, {line['value']} queries/s, {line['parity_check']['values_compared']} values, 0 mismatches")
