#!/usr/bin/env python3
"""What does a plain streaming kernel reach for the ingest pass's 3:1 read:write mix (6.2 GB in, 2.07 GB out)?
torch.addcmul on three 2.07 GB uint8 tensors (one vectorised elementwise kernel) against the ingest pass itself."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
px, B = 1920 * 1080, 1024
n = px * B
a = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda:0"); b = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda:0")
c = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda:0"); out = torch.empty_like(a)
def t(fn, reps=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, fn, gb in [("addcmul u8 (3 in, 1 out)", lambda: torch.addcmul(a, b, c, out=out), 4 * n / 1e9),
                     ("add u8 (2 in, 1 out)", lambda: torch.add(a, b, out=out), 3 * n / 1e9),
                     ("copy u8 (1 in, 1 out)", lambda: out.copy_(a), 2 * n / 1e9),
                     ("add1 u8 (1 in, 1 out)", lambda: torch.add(a, 1, out=out), 2 * n / 1e9)]:
    for dt, k in ((torch.uint8, 1), (torch.int32, 4)):
        if dt is torch.int32:
            A, Bt, C, O = (x.view(torch.int32) for x in (a, b, c, out))
            f2 = {"addcmul u8 (3 in, 1 out)": lambda: torch.addcmul(A, Bt, C, out=O), "add u8 (2 in, 1 out)": lambda: torch.add(A, Bt, out=O), "copy u8 (1 in, 1 out)": lambda: O.copy_(A), "add1 u8 (1 in, 1 out)": lambda: torch.add(A, 1, out=O)}[name]
            ms = t(f2); print("%-28s as int32: %.3f ms  %.0f GB/s" % (name, ms, gb / ms * 1e3))
        else:
            ms = t(fn); print("%-28s as uint8: %.3f ms  %.0f GB/s" % (name, ms, gb / ms * 1e3))
