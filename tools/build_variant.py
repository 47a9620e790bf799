#!/usr/bin/env python3
"""build_variant.py NAME -DFLAG ... -- builds scroll-prover_amd/libmi355zk_NAME.so with extra compile-time switches for A/B runs
(MI355ZK_LIB=<path> selects it in the Python binding).  E.g.:  python tools/build_variant.py chain -DZK_NTT_CHAIN=true
Objects go to scroll-prover_amd/build_NAME/ (per-unit, content-hashed like the shipped build: scroll-prover_amd/build.py)."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_mi355_build", os.path.join(ROOT, "scroll-prover_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "scroll-prover_amd", f"libmi355zk_{name}.so")
print(b.build(lib=out, extra_flags=flags, objdir=os.path.join(ROOT, "scroll-prover_amd", f"build_{name}"), verbose=True))
