#!/usr/bin/env python3
"""isa_block_census.py UNIT KERNEL_SUBSTRING [OUT.md] -- per-basic-block VALU census of one kernel, from `hipcc -S` of a translation unit
(csrc/UNIT.hip).  A whole-kernel census counts cold paths (doubling, bucket start, tails) together with the loop body; this one shows which blocks hold
the multiplies and which hold the moves.   e.g.  python tools/isa_block_census.py lib_msm 'k_msm_accumulateILi4' profiles/r04_accumulate_isa_blocks.md"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit, pat = sys.argv[1], sys.argv[2]
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
asm = os.path.join(tempfile.gettempdir(), unit + ".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DNDEBUG", "-S", "--cuda-device-only", "-o", asm,
                       os.path.join(ROOT, "scroll-prover_amd", "csrc", unit + ".hip")], stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith("\t.section\t.rodata"))
vgpr = next((l.split()[-1] for l in lines[start:start + 20000] if ".amdhsa_next_free_vgpr" in l), "?")
blocks, cur = [], None
for l in lines[start:end]:
    s = l.strip(); m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m or cur is None:
        cur = {"name": m.group(1) if m else "entry", "ops": collections.Counter(), "loop": "Loop" in s}; blocks.append(cur)
        if m: continue
    if not s or s.startswith(";") or s.startswith("."): continue
    cur["ops"][s.split()[0]] += 1
tot = collections.Counter()
for b in blocks: tot.update(b["ops"])
valu = lambda o: sum(n for k, n in o.items() if k.startswith("v_"))
mad = lambda o: o["v_mad_u64_u32"] + o["v_mad_i64_i32"]
mov = lambda o: o["v_mov_b32_e32"] + o["v_mov_b64_e32"]
print(f"kernel `{lines[start].split(':')[0]}`: {vgpr} VGPRs; whole kernel: {valu(tot)} VALU, {mad(tot)} v_mad_[ui]64, {mov(tot)} v_mov, {tot['v_and_b32_e32']} v_and, {tot['v_lshrrev_b64']} v_lshrrev_b64, {tot['v_mul_lo_u32']} v_mul_lo, {tot['s_nop']} s_nop\n", file=out)
print("| block | in loop | VALU | v_mad_[ui]64 | v_mov | v_and | v_lshrrev_b64 | v_mul_lo | v_add / v_sub | s_nop |", file=out)
print("|---|---|---|---|---|---|---|---|---|---|", file=out)
for b in blocks:
    o = b["ops"]
    if valu(o) >= 30:
        print(f"| {b['name']} | {'yes' if b['loop'] else ''} | {valu(o)} | {mad(o)} | {mov(o)} | {o['v_and_b32_e32']} | {o['v_lshrrev_b64']} | {o['v_mul_lo_u32']} | {o['v_add_u32_e32'] + o['v_sub_u32_e32']} | {o['s_nop']} |", file=out)
