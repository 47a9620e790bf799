"""full-size replays from the reference's own protocol fixtures: layer 2 (k = 25) and layer 4 (k = 26); the verifier on the bytes"""
import os, sys, time, json, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
zk = ge.load_package()
from oracle import plonk
cases = [(2, None, []), (4, None, []), (3, None, []), (1, None, []), (5, None, []), (6, None, []), (0, None, [])]
if len(sys.argv) > 1:
    cases = [c for c in cases if str(c[0]) in sys.argv[1:]]
for layer, k, args in cases:
    t0 = time.time()
    fx = os.path.join(ROOT, "tests", "golden", f"protocol_layer{layer}.json")
    rec = zk.replay.run(layer, k, args=args, protocol_file=fx if os.path.exists(fx) else None)
    if not rec.get("ok"):
        print("LAYER", layer, "REPLAY FAILED", rec.get("error")); continue
    try:
        pr = plonk.Protocol(json.load(open(rec["protocol_path"])))
        inst = plonk.mont_to_ints(__import__("numpy").frombuffer(rec["instances"], dtype="uint64").reshape(-1, 4))
        tv = time.time()
        ver = plonk.verify(pr, rec["vk"], inst, rec["proof"], 0x5343524F4C4C0001 + (rec["layer"] if rec["layer"] >= 0 else 0))
        slim = {k_: v for k_, v in rec.items() if k_ not in ("proof", "vk", "instances", "replay")}
        print("LAYER", layer, "verify", ver["ok"], ver.get("error"), "verify_s", round(time.time() - tv, 2), "wall", round(time.time() - t0, 1), json.dumps(slim), flush=True)
    except Exception:
        print("LAYER", layer, "CHECK FAILED"); traceback.print_exc()
