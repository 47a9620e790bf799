"""one uniform-scalar MSM of size 2^k (window tables on) repeated a few times: the workload for a rocprofv3 kernel trace read by rocpd_timeline.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
k = int(sys.argv[1]); n = 1 << k
p = h2.ParamsKZG.setup(k, 0x5343524f4c4c0001); p.precompute()
sc = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda"); sc[:, 3] &= (1 << 59) - 1
for _ in range(4):
    p.commit(sc)
torch.cuda.synchronize()
