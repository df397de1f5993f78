#!/usr/bin/env python
"""Do short kernels on a side stream run WHILE a stream of back-to-back small kernels (graph replays, as a decode) occupies another stream?
The searcher's aggregation (~110 dependent launches, 3.5 ms of GPU time) was seen to finish only after the decode beside it had ended
(SEAL_OVERLAP_TIMING=2).  This probe times a chain of 100 tiny dependent kernels on a side stream (normal / high priority) while the main
work runs (a) on the default (NULL) stream, (b) on a torch stream; main work = replays of a graph of 600-row GEMMs + elementwise kernels.
usage: python tools/stream_overlap_probe.py   (needs a GPU)"""
import time
import torch


def main():
    dev = torch.device("cuda", 0)
    a = torch.randn(600, 1024, device=dev, dtype=torch.float16)
    w = torch.randn(1024, 1024, device=dev, dtype=torch.float16)
    small = torch.zeros(4096, device=dev)

    def body():
        x = a
        for _ in range(40):
            x = torch.relu(x @ w) * 0.01
        return x
    cap = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(cap):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        body()
    torch.cuda.synchronize()

    big = torch.zeros(6_000_000, device=dev)
    keys = torch.randint(0, 1 << 40, (300_000,), device=dev)
    HEAVY = [False]

    def chain(st, n=100):
        with torch.cuda.stream(st):
            if HEAVY[0]:                    # as an aggregation: wide grids (23 000 workgroups), sorts, fills: ~3 ms alone
                for _ in range(max(1, n // 10)):
                    big.add_(1.0)
                    torch.sort(keys)
                    big.zero_()
                    big.mul_(2.0)
            else:
                for _ in range(n):
                    small.add_(1.0)

    import sys
    HEAVY[0] = len(sys.argv) > 1 and sys.argv[1] == "heavy"
    for main_name in ("the default (NULL) stream", "a torch stream"):
        main_st = torch.cuda.default_stream(dev) if main_name.startswith("the default") else torch.cuda.Stream(device=dev)
        for prio in (0, -1):
            side = torch.cuda.Stream(device=dev, priority=prio)
            chain(side, 5)
            torch.cuda.synchronize()
            # the chain alone
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side); chain(side); e1.record(side)
            torch.cuda.synchronize()
            alone = e0.elapsed_time(e1)
            # the main work alone
            m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(main_st):
                m0.record(main_st)
                for _ in range(60):
                    g.replay()
                m1.record(main_st)
            torch.cuda.synchronize()
            main_alone = m0.elapsed_time(m1)
            # both: the chain is enqueued a little after the main work has started
            with torch.cuda.stream(main_st):
                m0.record(main_st)
                for _ in range(60):
                    g.replay()
                m1.record(main_st)
            time.sleep(0.002)
            e0.record(side); chain(side); e1.record(side)
            torch.cuda.synchronize()
            print("main work on %-26s side stream priority %2d: chain alone %6.2f ms, main alone %6.2f ms; together: chain %6.2f ms (ends %6.2f ms "
                  "after the main work began), main %6.2f ms" % (main_name + ",", prio, alone, main_alone, e0.elapsed_time(e1), m0.elapsed_time(e1), m0.elapsed_time(m1)))


if __name__ == "__main__":
    main()
