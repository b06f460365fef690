#!/usr/bin/env python3
"""What the runtime's own copies cost on this box (sizes of the C2 host path): H2D of 1.58 GB from pageable / page-locked
memory, D2H of 160 MB into fresh pageable / touched pageable / page-locked memory.  (torch is the copy engine here.)"""
import time
import numpy as np
import torch

def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best

n = 1_580_000_000
a = torch.from_numpy(np.ones(n, dtype=np.uint8))
d = torch.empty(n, dtype=torch.uint8, device="cuda")
x = t(lambda: d.copy_(a)); print(f"H2D pageable     {n/1e9:.2f} GB: {x*1e3:6.1f} ms = {n/x/1e9:5.1f} GB/s")
ap = a.pin_memory()
x = t(lambda: d.copy_(ap, non_blocking=True)); print(f"H2D page-locked  {n/1e9:.2f} GB: {x*1e3:6.1f} ms = {n/x/1e9:5.1f} GB/s")
m = 160_000_000
dd = d[:m]
x = t(lambda: dd.cpu()); print(f"D2H fresh pageable   {m/1e6:.0f} MB: {x*1e3:6.1f} ms = {m/x/1e9:5.1f} GB/s")
h = torch.from_numpy(np.ones(m, dtype=np.uint8))
x = t(lambda: h.copy_(dd)); print(f"D2H touched pageable {m/1e6:.0f} MB: {x*1e3:6.1f} ms = {m/x/1e9:5.1f} GB/s")
hp = h.pin_memory()
x = t(lambda: hp.copy_(dd, non_blocking=True)); print(f"D2H page-locked      {m/1e6:.0f} MB: {x*1e3:6.1f} ms = {m/x/1e9:5.1f} GB/s")
import threading
def par_copy(dst, src, T=8):
    k = (len(src) + T - 1) // T
    th = [threading.Thread(target=lambda i=i: np.copyto(dst[i*k:(i+1)*k], src[i*k:(i+1)*k])) for i in range(T)]
    [q.start() for q in th]; [q.join() for q in th]
hpn = hp.numpy()
for T in (1, 4, 8, 16):
    def f():
        fresh = np.empty(m, dtype=np.uint8); par_copy(fresh, hpn, T)
    x = t(f); print(f"host copy page-locked -> fresh pageable, {T:2d} threads, {m/1e6:.0f} MB: {x*1e3:6.1f} ms = {m/x/1e9:5.1f} GB/s")
an = a.numpy(); apn = ap.numpy()
for T in (1, 4, 8, 16):
    x = t(lambda: par_copy(apn, an, T)); print(f"host copy pageable -> page-locked, {T:2d} threads, {n/1e9:.2f} GB: {x*1e3:6.1f} ms = {n/x/1e9:5.1f} GB/s")
