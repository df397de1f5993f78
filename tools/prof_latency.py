"""where the host's time goes in ONE un-pipelined batch of bench.py's workload (the p50 latency of the line): cProfile over ten single batches after a
short bench run, sorted by own time and by cumulative time.   python tools/prof_latency.py 2> profiles/<tag>_latency_host_profile.txt"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
from seal_amd import retrieval
orig = retrieval.SEALSearcher.batch_search
calls = []
def wrapped(self, *a, **kw):
    calls.append((self, a, kw))
    return orig(self, *a, **kw)
retrieval.SEALSearcher.batch_search = wrapped
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--latency-batches", "3"]
bench.main()
retrieval.SEALSearcher.batch_search = orig
singles = [c for c in calls if len(c[1][0]) <= 20][-3:]
torch.cuda.synchronize()
ts = []
for rep in range(4):
    for s, a, kw in singles:
        t = time.perf_counter(); orig(s, *a, **kw); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print("single batches ms:", " ".join("%.1f" % x for x in ts), file=sys.stderr)
pr = cProfile.Profile(); pr.enable()
for rep in range(4):
    for s, a, kw in singles:
        orig(s, *a, **kw); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stderr)
st.sort_stats("tottime").print_stats(30)
st.sort_stats("cumulative").print_stats(45)
