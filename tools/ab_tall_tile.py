"""the tall tile against the 4-wave tiles it replaced, same box: bench.py's timed call with split_gemm.HAND_CONFIGS as of the commit before (every product of a
decode step in 64 x 64 .. 128 x 128 tiles, fc1 as one slab: `before`) or as it is (`after`); one arm per process, the caller alternates:
for a in before after before after; do python tools/ab_tall_tile.py $a 2> /dev/null; done > profiles/r6_ab_tall_tile.txt"""
import json, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[2] == "run":
    import bench
    from seal_amd import split_gemm
    if sys.argv[1] == "before":
        split_gemm.HAND_CONFIGS.clear()
        split_gemm.HAND_CONFIGS.update({
            (1024, 12288): ((320, 2 | (2 << 8) | (2 << 12) | (4 << 16)), (640, 2 | (2 << 8) | (1 << 12) | (4 << 16))),
            (1024, 3072): ((320, 2 | (2 << 8) | (2 << 12) | (4 << 16)), (640, 2 | (2 << 8) | (1 << 12) | (4 << 16))),
            (3072, 3072): ((320, 2 | (2 << 8) | (2 << 12) | (2 << 16)), (640, 4 | (3 << 8) | (1 << 12) | (2 << 16))),
            (4096, 3072): ((320, 2 | (2 << 8) | (2 << 12) | (1 << 16)), (640, (4 + 128) | (3 << 8) | (1 << 12) | (1 << 16))),
            (50265, 3072): ((640, (1 + 128) | (2 << 8) | (1 << 12) | (1 << 16)),),
        })
    sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--latency-batches", "6"]
    bench.main()
else:
    arm = sys.argv[1]
    out = subprocess.run([sys.executable, os.path.abspath(__file__), arm, "run"], stdout=subprocess.PIPE, check=True).stdout.decode()
    d = json.loads(out.strip().splitlines()[-1])
    print(f"{'4-wave tiles' if arm == 'before' else 'tall tile':13s}: {d['value']:.1f} queries/s, {d['ms_per_step']:.2f} ms per batch, un-pipelined p50 {d['p50_batch_latency_ms']:.1f} ms, "
          f"one batch serialised {d['extra']['phase_ms_one_batch']}", flush=True)
