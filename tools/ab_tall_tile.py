"""the steps of round 6's GEMM work on ONE box: bench.py's timed call with split_gemm.HAND_CONFIGS as of the commit before the tall tile (every product of a decode
step in 64 x 64 .. 128 x 128 tiles, fc1 as one slab: `before`), with the tall tile over three-block planes (`after`: split_gemm.PAIRS off), and as it is (`pairs`:
hi / lo pair planes, three products per K step); one arm per process, the caller alternates:
for a in before after pairs before after pairs; do python tools/ab_tall_tile.py $a 2> /dev/null; done > profiles/r6_ab_gemm_steps.txt"""
import json, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[2] == "run":
    import bench
    from seal_amd import split_gemm
    split_gemm.PAIRS = sys.argv[1] == "pairs"
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
    print(f"{ {'before': '4-wave tiles', 'after': 'tall tile', 'pairs': 'tall tile + pair planes'}[arm]:24s}: {d['value']:.1f} queries/s, {d['ms_per_step']:.2f} ms per batch, un-pipelined p50 {d['p50_batch_latency_ms']:.1f} ms, "
          f"one batch serialised {d['extra']['phase_ms_one_batch']}", flush=True)
