"""a synthetic params file of the real size (k = 26: 8 589 934 852 bytes, the size pinned by [REF params-sha256sum]) written to local disk, then
Prover::load_params through mi355_srs_load_params_file (pinned double-buffered reads, on-device validation of every point) and a commitment
with the loaded basis checked in the field.  python tools/bench_params_file.py [k] [dir]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import cref
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
k = int(sys.argv[1]) if len(sys.argv) > 1 else 26
d = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
tau = 0x5343524F4C4C0001
path = os.path.join(d, f"params{k}_synthetic")
p = h2.ParamsKZG.setup(k, tau)
t = time.perf_counter(); p.write(path); tw = time.perf_counter() - t
size = os.path.getsize(path)
p.release(); del p; torch.cuda.empty_cache()
print(f"wrote {path}: {size} bytes (expected {h2.params_file_size(k)}) in {tw:.1f} s", flush=True)
for validate in (False, True):
    t = time.perf_counter(); q = h2.params_from_file(path, validate=validate); tl = time.perf_counter() - t
    print(f"load_params (validate={validate}): {tl:.2f} s = {size / tl / 1e9:.2f} GB/s", flush=True)
    if validate:
        n = 1 << k
        sc = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda"); sc[:, 3] &= (1 << 59) - 1
        got = q.commit(sc)
        want = cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), cref.eval_polynomial(sc.cpu().numpy().view(np.uint64), h2.fr(tau))))
        print("commit with the loaded basis == p(tau) G:", bool((np.asarray(got)[:8] == want).all()), flush=True)
    q.release()
os.remove(path)
