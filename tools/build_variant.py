#!/usr/bin/env python3
"""build_variant.py NAME -DFLAG ... -- builds scroll-prover_amd/libmi355zk_NAME.so with extra compile-time switches for A/B runs
(MI355ZK_LIB=<path> selects it in the Python binding).  E.g.:  python tools/build_variant.py chain -DZK_NTT_CHAIN=true -DZK_MADD_CHAIN_DEFAULT=true"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "scroll-prover_amd", f"libmi355zk_{name}.so")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DNDEBUG", "-Wno-unused-result", *flags, "-o", out,
       os.path.join(ROOT, "scroll-prover_amd", "csrc", "capi.hip")]
print(" ".join(cmd)); subprocess.check_call(cmd); print(out)
