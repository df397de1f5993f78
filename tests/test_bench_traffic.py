"""bench.py's ``roofline.traffic``: HBM bytes per constraint call from a rocprofv3 ``--pmc FETCH_SIZE`` pass (tools/prof_bench.sh ->
tools/summarize_pmc.py -> profiles/r*_pmc_fetch_size*.json).  A line may only cite counters taken over the kernel sources it runs and on
its own workload; per call = every launch of the constraint calls summed, divided by the calls, x 1024 x 2 (gfx950 FETCH_SIZE counts
128-byte requests at 64 bytes: MI355X guide)."""
import json
import os


def _root(tmp_path, sources=b"kernels"):
    csrc = tmp_path / "seal_amd" / "csrc"
    csrc.mkdir(parents=True)
    for f in ("fmi_kernels.hip", "fmi_device.h", "fmi_internal.h"):
        (csrc / f).write_bytes(sources + f.encode())
    (tmp_path / "profiles").mkdir()
    return str(tmp_path)


def _pmc(root, name, sha, workload, kernels):
    d = {k: {"FETCH_SIZE": {"launches": n, "sum": float(s), "avg": float(s) / n}} for k, (n, s) in kernels.items()}
    d.update({"_kernel_source_sha256": sha, "_workload": workload, "_commit": name})
    with open(os.path.join(root, "profiles", name), "w") as f:
        json.dump(d, f)


def test_traffic_is_per_call_over_every_launch_of_a_call(tmp_path):
    import bench
    root = _root(tmp_path)
    sha = bench.kernel_source_sha256(root)
    # 12 row-first calls (two launches each) + 1 table call (two launches) per batch, 6 batches
    _pmc(root, "r4_pmc_fetch_size.json", sha, "nq-21", {"k_constrain_rows": (72, 72 * 300.0), "void k_constrain<false, 8>": (72, 72 * 7000.0),
                                                         "void k_constrain_table<false>": (6, 6 * 100000.0), "k_table_bits": (6, 6 * 4000.0),
                                                         "void k_expand_dense<false>": (1, 500.0)})
    t, src = bench.cite_traffic("nq-21", root)
    want_kib = (72 * 300.0 + 72 * 7000.0 + 6 * 100000.0 + 6 * 4000.0) / 78
    assert t == round(want_kib * 1024 * 2, 1) and src["file"] == os.path.join("profiles", "r4_pmc_fetch_size.json") and src["workload"] == "nq-21"


def test_counters_of_other_kernels_or_workloads_are_refused(tmp_path):
    import bench
    root = _root(tmp_path)
    sha = bench.kernel_source_sha256(root)
    k = {"k_constrain_rows": (10, 3000.0), "void k_constrain<true, 8>": (10, 70000.0)}
    _pmc(root, "r9_pmc_fetch_size.json", "0" * 64, "nq-21", k)                 # newer, but another kernel generation
    _pmc(root, "r4_pmc_fetch_size_kilt.json", sha, "nq-36", k)                 # these kernels, another workload
    t, src = bench.cite_traffic("nq-21", root)
    assert t is None and "refused" in src and sha[:16] in src["refused"]
    t, src = bench.cite_traffic("nq-36", root)
    assert t == round(7300.0 * 1024 * 2, 1) and src["file"].endswith("r4_pmc_fetch_size_kilt.json")
    # an edit of the kernel sources invalidates every file
    with open(os.path.join(root, "seal_amd", "csrc", "fmi_kernels.hip"), "ab") as f:
        f.write(b"// edit")
    assert bench.cite_traffic("nq-36", root)[0] is None


def test_by_call_adds_the_chains_to_the_call_they_precede():
    """``bench.merge_call_logs``: three passes over one batch -> one record per constraint call.  The chains that k_beam_advance ran for a
    call a model step ahead are charged to that call: what they add to the launch they ride in (fused duration - bookkeeping duration, never
    below zero), with the chains as a launch of their own as the upper bound; a pass that launched something else voids the attribution."""
    import bench
    def rec(cur_len, rows, form, us=-1.0, blocks=0):
        return {"cur_len": cur_len, "rows": rows, "form": form, "us": us, "blocks": blocks}
    timed = [rec(2, 600, "table", 60.0), rec(3, 600, "advance", 8.0), rec(3, 600, "advance+chains", 15.0), rec(3, 600, "chained", 50.0),
             rec(4, 600, "advance", 9.0), rec(4, 600, "advance+chains", 14.0), rec(4, 600, "chained", 20.0), rec(5, 300, "advance", 8.5)]
    counted = [rec(2, 600, "table", blocks=1000000), rec(3, 600, "advance"), rec(3, 600, "advance+chains", blocks=8000), rec(3, 600, "chained", blocks=500000),
               rec(4, 600, "advance"), rec(4, 600, "advance+chains", blocks=4000), rec(4, 600, "chained", blocks=100000), rec(5, 300, "advance")]
    fused = [rec(2, 600, "table", 61.0), rec(3, 600, "advance+chains", 18.0), rec(3, 600, "chained", 50.0), rec(4, 600, "advance+chains", 8.0),
             rec(4, 600, "chained", 20.0), rec(5, 300, "advance", 8.4)]
    calls, other = bench.merge_call_logs(timed, counted, fused)
    assert [(c["cur_len"], c["form"]) for c in calls] == [(2, "table"), (3, "chained"), (4, "chained")]
    assert calls[0]["us"] == 60.0 and calls[0]["MB"] == 128.0 and "chains_us" not in calls[0]
    assert calls[1]["chains_us"] == 10.0 and calls[1]["chains_alone_us"] == 15.0 and calls[1]["us"] == 60.0 and calls[1]["us_upper_bound"] == 65.0
    assert abs(calls[1]["MB"] - (500000 + 8000) * 128 / 1e6) < 1e-3 and abs(calls[1]["frac"] - calls[1]["MB"] / 60.0 / 8000 * 1e3) < 1e-3
    assert calls[2]["chains_us"] == 0.0 and calls[2]["us"] == 20.0                       # (fused < bookkeeping on that step: nothing is subtracted)
    assert [o["cur_len"] for o in other] == [3, 4, 5] and other[0]["fused_with_chains_us"] == 18.0
    # without the fused pass the chains are charged as their own launch
    calls2, _ = bench.merge_call_logs(timed, counted, None)
    assert calls2[1]["chains_us"] == 15.0 and calls2[1]["us"] == 65.0
    # the counting pass launched another sequence: no attribution at all
    assert bench.merge_call_logs(timed, counted[:-1], fused) == ([], [])


def test_aggregate_roofline_prices_the_stages_from_what_they_processed():
    import bench

    class Ix:
        def size(self):
            return 2_879_038_742
    t = {"stage_ms": [0.2332, 0.4552, 0.1637, 0.3962, 0.0666, 0.2348, 1.1196, 0.0175, 0.3592, 0.0933, 0.0936], "counts": [5775748, 5699723, 30000, 2000, 4250343], "calls": 1}
    r = bench.aggregate_roofline(t, Ix())
    loc = r["stages"][0]
    assert loc["stage"] == "k_agg_locate" and abs(loc["algorithmic_MB"] - 5775748 * 16 / 1e6) < 0.01       # SA 4 + 12 B written (round 6: the document comes later, in position order)
    assert abs(loc["achieved"] - 5775748 * 16 / 233.2e-6 / 1e9) < 1.0 and r["frac"] == loc["frac"] and r["kernel"] == "k_agg_locate"
    assert r["stages"][2]["stage"].startswith("documents+coverage") and abs(r["stages"][2]["algorithmic_MB"] - 5775748 * 51 / 1e6) < 0.01
    assert "algorithmic_MB" not in r["stages"][1] and r["located_rows"] == 5775748 and abs(r["total_us"] - sum(t["stage_ms"]) * 1e3) < 0.1
    assert bench.aggregate_roofline({"stage_ms": [0] * 11, "counts": [0] * 5, "calls": 0}, Ix()) is None
