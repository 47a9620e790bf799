"""H2D / D2H rates the host-pointer entry points see: pageable numpy buffers (reused, pages already touched), several sizes.
mi355_ntt_fr_host = H2D + transform + D2H; mi355_eval_polynomial_host = H2D + a streaming kernel; the device-resident transform for the difference."""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
for k in (18, 20, 22, 24, 26):
    n = 1 << k
    host = np.zeros((n, 4), dtype=np.uint64); host[:, 0] = np.arange(n, dtype=np.uint64)
    dev = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    w = h2.fr(pow(h2.FR_ROOT_OF_UNITY, 1 << (28 - k), h2.R_MOD)); pt = h2.fr(12345)
    def t(fn, reps):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    reps = 10 if k <= 22 else 3
    ntt_h = t(lambda: h2.best_fft(host, w, k), reps)
    ntt_d = t(lambda: h2.best_fft(dev, w, k), reps)
    ev_h = t(lambda: h2.eval_polynomial(host, pt), reps)
    ev_d = t(lambda: h2.eval_polynomial(dev, pt), reps)
    mb = n * 32 / 1e6
    h2d = ev_h - ev_d; d2h = ntt_h - ntt_d - h2d
    print(f"k={k} ({mb:.0f} MB): ntt_host {ntt_h:.2f} ms, ntt_dev {ntt_d:.2f}, eval_host {ev_h:.2f}, eval_dev {ev_d:.2f}  => H2D {h2d:.2f} ms = {mb / h2d:.1f} GB/s, D2H {d2h:.2f} ms = {mb / max(d2h, 1e-6):.1f} GB/s", flush=True)
    pin = torch.zeros((n, 4), dtype=torch.int64).pin_memory()
    tp_h2d = t(lambda: dev.copy_(pin, non_blocking=True), reps); tp_d2h = t(lambda: pin.copy_(dev, non_blocking=True), reps)
    print(f"      pinned reference: H2D {mb / tp_h2d:.1f} GB/s, D2H {mb / tp_d2h:.1f} GB/s", flush=True)
    del host, dev, pin
