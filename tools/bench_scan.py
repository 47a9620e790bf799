"""throughput of the multiplicative scans on device-resident Fr vectors: python tools/bench_scan.py [k ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
for k in [int(x) for x in sys.argv[1:]] or [20, 24, 26]:
    n = 1 << k
    a = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda"); a[:, 3] &= (1 << 59) - 1
    b = torch.empty_like(a)
    z = h2.fr(0x1234567890abcdef)
    for name, fn, bytes_per in (("batch_invert", lambda: h2.batch_invert(a), 64), ("prefix_product", lambda: h2.prefix_product(a, dst=b), 96),
                                ("kate_division", lambda: h2.kate_division(a, z, dst=b), 96), ("axpy", lambda: h2.fr_vec_axpy(b, a, a, z), 96)):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print(f"k={k} {name}: {dt*1e3:.3f} ms, {n/dt/1e9:.2f} G elem/s, {n*bytes_per/dt/1e12:.2f} TB/s", flush=True)
