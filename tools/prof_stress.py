import cProfile, pstats, sys, os
sys.argv = ["bench.py", "--workload", "stress", "--stress-symbols", "2e9", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
pr = cProfile.Profile()
import seal_amd.retrieval as R
real = R._process_batch
cnt = [0]
def wrapped(*a, **kw):
    cnt[0] += 1
    if cnt[0] >= 3:
        pr.enable()
    try:
        return real(*a, **kw)
    finally:
        pr.disable()
R._process_batch = wrapped
try:
    bench.main()
finally:
    pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(35)
