#!/usr/bin/env python3
"""Do two hipGraph replays on two streams run side by side?  (round 4, visit R)

Each graph is a chain of N one-workgroup spin kernels (`torch.cuda._sleep`), so two of them fit on the chip a hundred times over:
side by side they take the time of one, serialized the time of two.  Variants: plain chains, chains forked over two capture
streams (the shape of the forked block step), eager launches for comparison, and replays issued from two host threads.
    python tools/graph_concurrency.py [out.json] [--one-graph]
"""
import json
import os
import sys
import threading
import time

import torch

dev = torch.device("cuda")
N, CYC = 200, 100_000     # 200 kernels of ~40-50 us


def chain(n):
    for _ in range(n):
        torch.cuda._sleep(CYC)


def capture(stream, forked):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        chain(2)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            if not forked:
                chain(N)
            else:
                for _ in range(N // 20):
                    side.wait_stream(stream)
                    with torch.cuda.stream(side):
                        chain(10)
                    chain(10)
                    stream.wait_stream(side)
    return g


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return round(best * 1e3, 3)


def main():
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    out = {"kernels_per_graph": N}

    def eager_one():
        with torch.cuda.stream(s1):
            chain(N)

    def eager_two():
        for _ in range(N // 10):
            with torch.cuda.stream(s1):
                chain(10)
            with torch.cuda.stream(s2):
                chain(10)
    out["eager_one_ms"] = timed(eager_one)
    out["eager_two_streams_ms"] = timed(eager_two)
    for forked in (False, True):
        g1, g2 = capture(s1, forked), capture(s2, forked)
        tag = "forked" if forked else "plain"

        def one():
            with torch.cuda.stream(s1):
                g1.replay()

        def two():
            with torch.cuda.stream(s1):
                g1.replay()
            with torch.cuda.stream(s2):
                g2.replay()

        def two_threads():
            def run(s, g):
                with torch.cuda.stream(s):
                    g.replay()
            th = [threading.Thread(target=run, args=a) for a in ((s1, g1), (s2, g2))]
            [t.start() for t in th]
            [t.join() for t in th]

        def four():      # two replays per stream, interleaved like the block loop
            for _ in range(2):
                with torch.cuda.stream(s1):
                    g1.replay()
                with torch.cuda.stream(s2):
                    g2.replay()
        out[f"{tag}_one_ms"] = timed(one)
        out[f"{tag}_two_streams_ms"] = timed(two)
        out[f"{tag}_two_streams_two_threads_ms"] = timed(two_threads)
        out[f"{tag}_two_streams_two_replays_each_ms"] = timed(four)
    if "--one-graph" not in sys.argv:
        print(json.dumps(out, indent=1))
        return
    # both "groups" (two forked chains) captured into ONE graph: four branches under one launch.  On ROCm 7.2 this capture
    # segfaults inside hipStreamEndCapture (a side stream forked off a stream that is itself forked off the origin), which
    # is why it is behind a flag and why the codec keeps one capture per chain group.
    g = torch.cuda.CUDAGraph()
    sa, sb, t2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s1):
            t2.wait_stream(s1)
            for main, side in ((s1, sa), (t2, sb)):
                with torch.cuda.stream(main):
                    for _ in range(N // 20):
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            chain(10)
                        chain(10)
                        main.wait_stream(side)
            s1.wait_stream(t2)

    def both():
        with torch.cuda.stream(s1):
            g.replay()
    out["two_forked_groups_in_one_graph_ms"] = timed(both)
    print(json.dumps(out, indent=1))
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if args:
        json.dump(out, open(args[0], "w"), indent=1)


if __name__ == "__main__":
    main()
