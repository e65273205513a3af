#!/usr/bin/env python3
"""Control for tools/step_spikes.py: the same wall-clock statistics for a plain torch loop (one small kernel +
synchronize per iteration, nothing of this repository loaded)."""
import time, numpy as np, torch
x = torch.zeros(1 << 20, device="cuda")
for _ in range(1000):
    x.add_(1); torch.cuda.synchronize()
K = 60000
marks = np.zeros(K + 1)
marks[0] = time.perf_counter()
for i in range(K):
    x.add_(1); torch.cuda.synchronize()
    marks[i + 1] = time.perf_counter()
d = np.diff(marks) * 1e3
t = marks[1:] - marks[0]
slow = [(round(float(t[i]) * 1e3, 1), round(float(d[i]), 3)) for i in range(K) if d[i] > 1.0]
print("median %.4f ms, mean %.4f ms, total %.2f s, iterations over 1 ms (at ms, took ms):" % (float(np.median(d)), float(d.mean()), float(t[-1])), slow[:40])
