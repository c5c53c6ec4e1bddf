#!/usr/bin/env python3
"""Dev aid (GPU box): what a kernel node costs in a replayed HIP graph -- a chain of K dependent tiny
kernels on one stream, and the same with a fork / join to a second stream every 4 kernels."""
import time, torch
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev); y = torch.zeros(64, device=dev)


def bench(fn, K, tag):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 200 * 1e6
    print(f"{tag}: {us:7.1f} us per replay, {us / K:5.2f} us per kernel node ({K} nodes)")


for K in (8, 32, 128):
    bench(lambda: [x.add_(1.0) for _ in range(K)], K, f"chain of {K}")
side = torch.cuda.Stream()


def forked(K):
    cur = torch.cuda.current_stream()
    for i in range(K // 4):
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            y.add_(1.0); y.add_(1.0)
        x.add_(1.0); x.add_(1.0)
        cur.wait_stream(side)


bench(lambda: forked(32), 32, "32 nodes, fork / join every 4")
