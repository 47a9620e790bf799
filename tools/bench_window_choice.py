"""time one commitment with window tables for every table width c: calibration of msm_cost() in capi.hip.
    python tools/bench_window_choice.py [k ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
for k in [int(x) for x in sys.argv[1:]] or [18, 20, 22, 23, 24]:
    n = 1 << k
    sc = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda"); sc[:, 3] &= (1 << 59) - 1
    row = []
    for c in [0] + list(range(14, 23)):
        p = h2.ParamsKZG.setup(k, 0x5343524f4c4c0001)
        try:
            p.precompute(c=c, lagrange=False)
        except Exception as e:
            row.append(f"c={c}: {type(e).__name__}"); p.release(); continue
        p.commit(sc); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): p.commit(sc)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        row.append(f"c={'auto' if c == 0 else c}: {dt*1e3:.2f}")
        p.release()
    print(f"k={k}: " + "  ".join(row), flush=True)
