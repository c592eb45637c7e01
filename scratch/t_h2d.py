#!/usr/bin/env python3
"""experiment: what does the host -> device path deliver?  one big copy, chunked copies on 1 / 2 / 4 streams"""
import time, torch
N = 2 << 30
host = torch.empty(N, dtype=torch.uint8, pin_memory=True); host.fill_(3)
dev = torch.empty(N, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
def run(nstreams, chunk):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k = 0
    for o in range(0, N, chunk):
        with torch.cuda.stream(streams[k % nstreams]):
            dev[o:o + chunk].copy_(host[o:o + chunk], non_blocking=True)
        k += 1
    torch.cuda.synchronize()
    return N / (time.perf_counter() - t0) / 1e9
for ns, ch in ((1, N), (1, 256 << 20), (1, 64 << 20), (2, 256 << 20), (2, 64 << 20), (4, 64 << 20), (4, 16 << 20), (8, 16 << 20)):
    r = [run(ns, ch) for _ in range(4)]
    print("streams %d chunk %4d MiB: %s GB/s" % (ns, ch >> 20, ["%.1f" % x for x in r]))
