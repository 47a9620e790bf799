import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
for k in (18, 20, 22):
    n = 1 << k
    params = h2.ParamsKZG.setup(k, 0x1234567)
    for pre in (False, True):
        if pre: params.precompute()
        polys = [torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda") for _ in range(16)]
        for p in polys: p[:, 3] &= (1 << 59) - 1
        for M in (1, 4, 16):
            params.commit_many(polys[:M]); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3): params.commit_many(polys[:M])
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
            print(f"k={k} pre={pre} M={M}: {dt*1e3:.2f} ms/batch {dt*1e3/M:.3f} ms/commit", flush=True)
    params.release()
