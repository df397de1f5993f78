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
