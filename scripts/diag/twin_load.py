"""Diagnostic twin (GPU): runtime-level activity with no kernels of this library, beside a victim that replays msda ops
(op_replay_race.py).  modes: streams | h2d | malloc | sync | procs"""
import os, subprocess, sys, time
import numpy as np
mode, secs = sys.argv[1], float(sys.argv[2])
t0 = time.time()
if mode == "procs":      # short-lived processes: context + queue creation / destruction
    while time.time() - t0 < secs:
        subprocess.run([sys.executable, "-c", "import torch; torch.zeros(1 << 20, device='cuda').sum().item()"])
    sys.exit(0)
import torch
d = torch.device("cuda:0")
x = torch.randn(1 << 22, device=d)
while time.time() - t0 < secs:
    if mode == "streams":        # new HIP streams (hardware queues) with a little work on each
        ss = [torch.cuda.Stream() for _ in range(4)]
        for s in ss:
            with torch.cuda.stream(s):
                (x * 2).sum()
        torch.cuda.synchronize()
        del ss
    elif mode == "h2d":          # pageable host -> device copies + blocking sync
        for _ in range(20):
            torch.tensor(np.random.rand(6, 4, 4)).to(d)
            (x * 2).sum().item()
    elif mode == "malloc":       # driver-level allocation churn (page-table updates)
        ys = [torch.empty(64 << 20, dtype=torch.uint8, device=d) for _ in range(8)]
        ys[0].zero_()
        del ys
        torch.cuda.empty_cache()
    elif mode == "sync":         # short bursts of work separated by idle gaps (queue goes empty and wakes again)
        (x * 2).sum().item()
        time.sleep(0.002)
