import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
zk = ge.load_package()
exe = zk.replay.exe_path()
d = tempfile.mkdtemp()
e = dict(os.environ); e["MI355_TRACE"] = "1"
r = subprocess.run([exe, "--protocol", os.path.join(ROOT, "tests/golden/protocol_layer2.json"), "--out", d, "--proofs", "2"], capture_output=True, text=True, env=e)
lines = r.stderr.splitlines()
print(len(lines))
# the last proof's calls: print the last 120 trace lines
for l in lines[-130:]: print(l[:200])
