"""time ParamsKZG.downsize (g_to_lagrange = G1 inverse DFT) on the device: python tools/bench_g1fft.py [k ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
for k in [int(x) for x in sys.argv[1:]] or [14, 16, 18, 20]:
    p = h2.ParamsKZG.setup(k + 1, 0x5343524f4c4c0001)
    torch.cuda.synchronize(); t = time.perf_counter()
    p.downsize(k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    ref = h2.ParamsKZG.setup(k, 0x5343524f4c4c0001)
    ok = bool((p.read_g(lagrange=True) == ref.read_g(lagrange=True)).all())
    smul = (1 << (k - 1)) * (k - 1) + (1 << k)
    print(f"downsize to k={k}: {dt*1e3:.1f} ms, {smul/dt/1e6:.2f} M scalar-muls/s, matches closed form: {ok}", flush=True)
    p.release(); ref.release()
