"""What does one host pay when N ranks of `bench.py --gpus N` start together?    The GPU work of a rank is its own GPU's;
what the ranks SHARE is the host: every rank holds the reference API's `beginnings` python list of the 21 M passages (+ its numpy twin,
+ the copy the library keeps), builds its query lists, and runs one python thread.  This dry run does exactly that host work, without a
GPU, in N processes at once, and prints wall time and peak RSS per rank and in total.

  python tools/host_budget_dryrun.py [--ranks 8] [--docs 21015324]"""
import argparse, multiprocessing as mp, os, resource, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def rank_work(rank, docs, q):
    import numpy as np
    t0 = time.perf_counter()
    rng = np.random.default_rng(0)
    lens = np.clip(np.round(rng.normal(137.0, 25.0, docs)), 40, 256).astype(np.int64)            # bench.synth_corpus: document lengths
    beg = np.zeros(docs + 1, dtype=np.int64)
    np.cumsum(lens, out=beg[1:])
    t1 = time.perf_counter()
    as_list = beg.tolist()                                                                          # beg.tolist() handed to the index
    beginnings = [int(x) for x in as_list]                                                          # FMIndex.initialize_from_device
    twin = np.asarray(beginnings, dtype=np.uint64)                                                  # FMIndex._push_beginnings (+ int64 twin)
    twin64 = twin.astype(np.int64)
    lib_copy = twin.copy()                                                                          # fmi_set_doc_beginnings keeps its own
    t2 = time.perf_counter()
    qrng = np.random.default_rng(1 + rank)                                                          # bench.synth_queries: 26 batches of 20
    queries = [[0] + (np.minimum(qrng.zipf(1.3, size=int(qrng.integers(8, 25))), 50000) + 3).tolist() + [2] for _ in range(26 * 20)]
    t3 = time.perf_counter()
    rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20
    q.put((rank, t1 - t0, t2 - t1, t3 - t2, rss, len(beginnings), len(queries), int(twin64[-1]), int(lib_copy[-1])))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--docs", type=int, default=21015324)
    a = ap.parse_args()
    q = mp.Queue()
    t = time.perf_counter()
    ps = [mp.Process(target=rank_work, args=(r, a.docs, q)) for r in range(a.ranks)]
    for p in ps:
        p.start()
    rows = sorted(q.get() for _ in ps)
    for p in ps:
        p.join()
    wall = time.perf_counter() - t
    print(f"{a.ranks} ranks at once on {os.cpu_count()} host cores, {a.docs} passages each:")
    for r, c, b, qs, rss, nb, nq, last, _ in rows:
        print(f"  rank {r}: lengths+offsets {c:.1f}s, beginnings list + numpy twins {b:.1f}s, {nq} queries {qs:.2f}s, peak RSS {rss:.2f} GiB")
    print(f"wall {wall:.1f}s; peak RSS summed over ranks {sum(x[4] for x in rows):.1f} GiB, max {max(x[4] for x in rows):.2f} GiB")
