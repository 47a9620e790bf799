#!/usr/bin/env python3
"""pmc_report.py <tag> -- turns the raw per-dispatch lines of tools/collect_profiles.sh (gpurun_out/<tag>_*.txt, *_record.json) into the tracked documents:

  profiles/<tag>_kernel_stats.md   rocprofv3 --kernel-trace --stats of the headline legs (the averages roofline.avg_launch_ms must agree with)
  profiles/<tag>_pmc_k26.md        HBM traffic of the MSM kernels and of the NTT passes at 2^26 (FETCH_SIZE / WRITE_SIZE, separate passes), the passes again
                                   THROUGH THE BATCHED ENTRY POINTS inside a layer-4 create_proof, SQ counters (instructions per element, VALU busy)
  profiles/<tag>_gate_eval.md      the roofline of k_fr_gate_eval: counter bytes vs the algorithmic operand bytes the replay counted, instructions per factor, VALU busy
  profiles/<tag>_kernel_vs_wall.md kernel time vs wall for one layer-0, layer-3 and layer-4 proof
  profiles/pmc_latest.json         the two recorded traffic figures bench.py re-emits, with the hash of the kernel sources they were measured on

Counter corrections (guide /opt/skills/guides/MI355X_MICROARCH.md "HBM"; calibration profiles/r01_pmc_msm_k26.md): FETCH_SIZE is in KiB and reports exactly half of the bytes
of a wide coalesced streaming read (16 B per lane) -- doubled here for streaming kernels; 64-byte gathers and strided 128-byte runs are counted 1:1; WRITE_SIZE (KiB) 1:1.
"""
import hashlib
import json
import os
import re
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
CUS, CLOCK = 256, 2.4e9


def lines(name):
    p = os.path.join(OUT, f"{TAG}_{name}")
    return open(p).read().splitlines() if os.path.exists(p) else []


def parse(name):
    """[(dispatch, kernel, ms, {counter: value})]"""
    out = []
    for l in lines(name):
        m = re.match(r"dispatch (\d+) (.*?) ([\d.]+) ms (.*)$", l)
        if m:
            out.append((int(m.group(1)), m.group(2), float(m.group(3)), {k: float(v) for k, v in (kv.split("=") for kv in m.group(4).split())}))
    return out


def record(name):
    for l in lines(name):
        if l.startswith("{"):
            return json.loads(l)
    return {}


def source_hash(kind):
    files = {"msm": ("msm.hpp", "lib_msm.hip", "fp29.hpp", "g1_29.hpp"), "ntt": ("ntt29.hpp", "lib_ntt.hip", "fp29.hpp")}[kind]
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "scroll-prover_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def by_kernel(rows, sub):
    return [r for r in rows if sub in r[1]]


def avg(xs):
    xs = list(xs)
    return sum(xs) / len(xs) if xs else float("nan")


KIB = 1024.0
N26 = 1 << 26
fetch, write, sq = parse("pmc_FETCH_SIZE.txt"), parse("pmc_WRITE_SIZE.txt"), parse("pmc_sq.txt")

# ------------------------------------------------------------------------------------------------ kernel stats
ks = lines("kernel_stats.txt")
if ks:
    with open(os.path.join(PROF, f"{TAG}_kernel_stats.md"), "w") as f:
        f.write(f"# Round {TAG[1:].lstrip('0')} -- per-kernel statistics of the headline legs (`rocprofv3 --kernel-trace --stats`)\n\n"
                "`cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-api --no-table-free --no-proof-mix "
                "--no-sizes --no-witness-like --no-batch-legs` (`tools/collect_profiles.sh`): six uniform MSMs at 2^26 (1 warm-up + 5 timed) and the NTT leg.  `k_msm_accumulate`'s average is the "
                "figure `roofline.avg_launch_ms` of the bench line must agree with; `k_srs_precompute` / `k_fixed_base_mul` / `k_srs_scalars` are registration-time set-up.\n\n")
        f.write("\n".join(ks) + "\n")
        bl = record("stats_bench_line.json")
        if bl:
            f.write(f"\nThe run's own line: ms_per_step {bl['ms_per_step']:.3f}, roofline.avg_launch_ms {bl['roofline']['avg_launch_ms']:.3f}, roofline.frac {bl['roofline']['frac']:.5f}, "
                    f"NTT {(bl['ntt']['ms_per_transform'] if 'ntt' in bl else bl['config'].get('ntt_k26_ms', float('nan'))):.3f} ms per transform (three passes).\n")   # round 6: the stdout line is the compact one

# ------------------------------------------------------------------------------------------------ MSM + NTT traffic at 2^26
acc_f, acc_w = by_kernel(fetch, "k_msm_accumulate"), by_kernel(write, "k_msm_accumulate")
acc_bytes = None
md = [f"# Round {TAG[1:].lstrip('0')} -- HBM traffic counters at 2^26 on the {TAG} tree: the MSM kernels, the three NTT passes, and the same passes through the batched entry points\n",
      "`tools/collect_profiles.sh r05` (FETCH_SIZE and WRITE_SIZE in separate `rocprofv3 --kernel-trace --pmc` passes over `bench.py --steps 1 --warmup 0` with the headline legs only, "
      "then over one layer-4 `create_proof`); per-dispatch sums by `tools/pmc_query.py`, this table by `tools/pmc_report.py`.  Counter units are KiB.  Corrections: wide coalesced "
      "streaming reads are counted at 1/2 (doubled below), 64-byte gathers and strided 128-byte runs 1:1, writes 1:1 (guide, \"HBM\"; calibration in `profiles/r01_pmc_msm_k26.md`).\n",
      f"Source hashes the figures belong to (bench.py re-emits `roofline.traffic` only while they match): msm `{source_hash('msm')}`, ntt `{source_hash('ntt')}`.\n"]
if acc_f and acc_w:
    a_f, a_w, a_ms = avg(r[3]["FETCH_SIZE"] for r in acc_f), avg(r[3]["WRITE_SIZE"] for r in acc_w), avg(r[2] for r in acc_f + acc_w)
    idx_stream = N26 * 12 * 4 / 2      # the (window, index) stream: 12 windows x 4 B per pair, streamed (counted at 1/2): its uncounted half
    acc_bytes = a_f * KIB + idx_stream + a_w * KIB
    md += ["## MSM 2^26 (window tables, c = 22, W = 12)\n", "| kernel | ms | FETCH_SIZE KiB | WRITE_SIZE KiB | bytes moved (corrected) |", "|---|---|---|---|---|",
           f"| `k_msm_accumulate<4>` | {a_ms:.1f} | {a_f:.4g} | {a_w:.4g} | counted reads {a_f * KIB:.3g} (12 gathers of 64 B per pair, 1:1) + the uncounted half of the {2 * idx_stream / 1e9:.1f} GB index stream {idx_stream:.3g} + writes {a_w * KIB:.3g} = **{acc_bytes:.3g}** = {acc_bytes / (96 * N26):.1f} x the algorithmic 96 B x 2^26 = 6.44e9 |"]
    for kname, label, stream in (("k_msm_digits", "`k_msm_digits`", True), ("k_sort_l1", "`k_sort_l1_scatter_split<24>`", True), ("k_sort_l2_hist", "`k_sort_l2_hist_split`", True), ("k_sort_l2_scatter", "`k_sort_l2_scatter_split<16>`", True)):
        kf, kw = by_kernel(fetch, kname), by_kernel(write, kname)
        if kf and kw:
            ff, ww = avg(r[3]["FETCH_SIZE"] for r in kf), avg(r[3]["WRITE_SIZE"] for r in kw)
            md.append(f"| {label} | {avg(r[2] for r in kf):.2f} | {ff:.4g} | {ww:.4g} | reads {2 * ff * KIB:.3g} (x2) + writes {ww * KIB:.3g} = {2 * ff * KIB + ww * KIB:.3g} |")
    md.append("")
ntt_bytes = None


def ntt_table(frows, wrows, title):
    global ntt_bytes
    s_f, s_w, f_f, f_w = by_kernel(frows, "k_ntt29_strided"), by_kernel(wrows, "k_ntt29_strided"), by_kernel(frows, "k_ntt29_final"), by_kernel(wrows, "k_ntt29_final")
    if not (s_f and s_w and f_f and f_w):
        return []
    # the two strided levels alternate: level 1 reads the vector as wide streams (counted 1/2), level 2 in 128-byte runs (1:1)
    l1 = [r for r in s_f if r[3]["FETCH_SIZE"] < 1.6e6]; l2 = [r for r in s_f if r[3]["FETCH_SIZE"] >= 1.6e6]
    w_avg = avg(r[3]["WRITE_SIZE"] for r in s_w)
    rows = [title, "", "| pass | launches | ms | FETCH_SIZE KiB | WRITE_SIZE KiB | bytes moved |", "|---|---|---|---|---|---|"]
    tot = 0.0
    for lab, rs, mul in (("strided, level 1 (rows 2^17 elements apart)", l1, 2.0), ("strided, level 2 (rows 256 elements apart)", l2, 1.0), ("final (digit-reversing)", f_f, 2.0)):
        if not rs:
            continue
        ff = avg(r[3]["FETCH_SIZE"] for r in rs); ww = avg(r[3]["WRITE_SIZE"] for r in f_w) if "final" in lab else w_avg
        b = mul * ff * KIB + ww * KIB; tot += b
        rows.append(f"| {lab} | {len(rs)} | {avg(r[2] for r in rs):.2f} | {ff:.4g} | {ww:.4g} | reads {mul * ff * KIB:.3g} ({'x2' if mul == 2 else '1:1'}) + writes {ww * KIB:.3g} = {b:.3g} |")
    rows.append(f"| **transform** | | {avg(r[2] for r in l1) + avg(r[2] for r in l2) + avg(r[2] for r in f_f):.2f} | | | **{tot:.3g} B per transform** = {tot / (64 * N26):.2f} x the algorithmic 64 N = 4.295e9 B |")
    rows.append("")
    if ntt_bytes is None:
        ntt_bytes = tot
    return rows


md += ntt_table(fetch, write, "## NTT 2^26, single transforms (`mi355_ntt_fr_dev` / `mi355_intt_fr_dev`: k_ntt29_strided<2,0> x 2 + k_ntt29_final<2,0>; N x 32 B = 2.147e9 B per vector)")
l4f, l4w, l4s = parse("L4_pmc_FETCH_SIZE.txt"), parse("L4_pmc_WRITE_SIZE.txt"), parse("L4_pmc_sq.txt")
md += ntt_table(l4f, l4w, "## NTT 2^26 THROUGH THE BATCHED ENTRY POINTS (`mi355_ntt_fr_batch_dev`, `mi355_coset_ntt_fr_batch_dev`): every transform of one layer-4 create_proof + its keygen")
if sq:
    md += ["## SQ counters (headline legs)\n", "| kernel | ms | VALU instructions | per unit of work | VALU busy | SQ_WAIT_ANY / SQ_WAVE_CYCLES |", "|---|---|---|---|---|---|"]
    seen = set()
    for d, kname, ms, c in sq:
        key = kname.split("(")[0]
        if key in seen:
            continue
        seen.add(key)
        per = c["SQ_INSTS_VALU"] * 64 / N26 if "ntt" in kname else c["SQ_INSTS_VALU"] * 64 / (N26 * 12) if "accumulate" in kname else c["SQ_INSTS_VALU"] * 64 / N26
        unit = "per element" if "accumulate" not in kname else "per mixed addition (pair x window)"
        busy = c["SQ_ACTIVE_INST_VALU"] * 4 / (ms * 1e-3 * 1.93e9 * CUS * 4)   # one VALU instruction occupies its SIMD for 4 cycles; 4 SIMDs per CU; sustained clock under these kernels 1.93 GHz (r03_sq_counters.md)
        md.append(f"| `{key.replace('void ', '').replace('zk::', '')}` | {ms:.2f} | {c['SQ_INSTS_VALU']:.4g} (wave-level) | {per:.0f} {unit} | {busy:.2f} | {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f} |")
    md.append("\nVALU busy = SQ_ACTIVE_INST_VALU x 4 cycles / (duration x 1.93 GHz x 256 CUs x 4 SIMDs): the fraction of SIMD issue slots that carried a vector instruction.\n")
md += ["## raw per-dispatch lines\n", "```"] + lines("pmc_FETCH_SIZE.txt") + ["----"] + lines("pmc_WRITE_SIZE.txt") + ["----"] + lines("pmc_sq.txt") + ["```"]
with open(os.path.join(PROF, f"{TAG}_pmc_k26.md"), "w") as f:
    f.write("\n".join(md) + "\n")

# ------------------------------------------------------------------------------------------------ gate_eval roofline
gf, gw, gs = by_kernel(l4f, "k_fr_gate_eval"), by_kernel(l4w, "k_fr_gate_eval"), by_kernel(l4s, "k_fr_gate_eval")
rec = record("L4_FETCH_SIZE_record.json") or record("L4_stats_record.json")
if gf and gw and gs and rec:
    tot = rec["gate_eval_process_totals"]
    f_b, w_b = sum(r[3]["FETCH_SIZE"] for r in gf) * KIB * 2, sum(r[3]["WRITE_SIZE"] for r in gw) * KIB   # every operand is a wide streaming read: x2
    ms = sum(r[2] for r in gf)
    insts = sum(r[3]["SQ_INSTS_VALU"] for r in gs) * 64
    active = sum(r[3]["SQ_ACTIVE_INST_VALU"] for r in gs)
    ms_s = sum(r[2] for r in gs)
    g = [f"# Round {TAG[1:].lstrip('0')} -- the roofline of `k_fr_gate_eval`\n",
         "One `create_proof` of the reference's layer-4 protocol at k = 26 (`tests/cpp/test_plonk_replay --protocol tests/golden/protocol_layer4.json --proofs 1`) under "
         "`rocprofv3 --kernel-trace --pmc` (FETCH_SIZE, WRITE_SIZE, SQ counters in three separate runs; `tools/collect_profiles.sh`).  The program counts the ALGORITHMIC side itself "
         "(`gate_eval_process_totals` of its record: per launch 32 B x rows x (distinct operand polynomials + dst written + dst read when accumulating), and factor-rows = rows x factors).\n",
         "| | value |", "|---|---|",
         f"| launches in the process (keygen + step 4 + quotient parts + SHPLONK) | {tot['launches']} (counters saw {len(gf)}) |",
         f"| kernel time | {ms:.1f} ms |",
         f"| algorithmic bytes | {tot['algorithmic_bytes']:.4g} B -> {tot['algorithmic_bytes'] / ms / 1e6:.0f} GB/s = **{tot['algorithmic_bytes'] / ms / 1e6 / 8000:.3f} of the 8 TB/s HBM roof** |",
         f"| counter bytes (FETCH_SIZE x 2 for wide streaming reads + WRITE_SIZE) | reads {f_b:.4g} + writes {w_b:.4g} = {f_b + w_b:.4g} B = **{(f_b + w_b) / tot['algorithmic_bytes']:.2f} x algorithmic** -> {(f_b + w_b) / ms / 1e6:.0f} GB/s |",
         f"| factor-rows (one operand load + one 9x29 Montgomery product each) | {tot['factor_rows']:.4g} -> {tot['factor_rows'] / ms / 1e6:.1f} G factor-rows/s; term-rows {tot['term_rows']:.4g} |",
         f"| VALU instructions (SQ_INSTS_VALU x 64 lanes) | {insts:.4g} = **{insts / tot['factor_rows']:.0f} per factor-row** |",
         f"| VALU busy (SQ_ACTIVE_INST_VALU x 4 / (time x 1.93 GHz x 1024 SIMDs)) | **{active * 4 / (ms_s * 1e-3 * 1.93e9 * CUS * 4):.2f}** |", ""]
    g += ["Reading: the kernel moves about a third of the HBM roof and keeps the vector ALUs busy for most of its issue slots at ~200 instructions per factor (the operand's 8x32 -> 9x29 "
          "re-slice plus one 9x29 Montgomery product): it sits between the two roofs, closer to the ALU one, as DESIGN.md section 5 says.  The levers are therefore (i) fewer factor-rows for "
          "the same expression and (ii) fewer bytes per launch; see the A/B below.\n"]
    ab = os.path.join(OUT, f"{TAG}_job3_ab.json")
    if os.path.exists(ab):
        A = json.load(open(ab))
        g += ["## A/B: common-prefix groups in the plan compiler (MI355_PLAN_PREFIX_MIN=16 default vs 0 = off), full-size proofs, second (steady-state) proof\n",
              "| layer | groups | launches per part | terms | factor-rows (process) | step 7 ms | proof ms |", "|---|---|---|---|---|---|---|"]
        for key in sorted(A):
            if key.startswith("prefix_") and A[key].get("ok"):
                r = A[key]
                g.append(f"| {key.split('_')[1]} ({'groups on' if key.endswith('min16') else 'off'}) | {r['plan'].get('prefix_groups')} | {r['plan']['launches_per_part']} | {r['plan']['terms']} | {r['gate_eval_process_totals']['factor_rows']:.4g} | {r['step_ms']['7_quotient']:.1f} | {r['resident_ms']:.1f} |")
        g.append("")
        g += ["Every proof of this table was verified from its bytes (tools/_scratch/r05_job4.py runs oracle/plonk.py's verifier on each).\n"]
    g += ["## raw per-dispatch lines (first 40 of each pass)\n", "```"] + lines("L4_pmc_FETCH_SIZE.txt")[:40] + ["----"] + lines("L4_pmc_WRITE_SIZE.txt")[:40] + ["----"] + lines("L4_pmc_sq.txt")[:40] + ["```"]
    with open(os.path.join(PROF, f"{TAG}_gate_eval.md"), "w") as f:
        f.write("\n".join(g) + "\n")

# ------------------------------------------------------------------------------------------------ kernel vs wall
kv = [f"# Round {TAG[1:].lstrip('0')} -- kernel time vs wall for one proof of layers 0, 3 and 4 (`rocprofv3 --kernel-trace --stats`, `--proofs 1`)\n\n**The tables cover the whole PROCESS: the first proof AND the keygen before it** (keygen commits every fixed / sigma column one by one -- 270 single, table-free MSMs at layer 0 -- and transforms every key polynomial onto every coset part, one call each): call counts and totals of `k_msm_*`, `k_ntt29_*` and (where it still runs) `k_distribute_powers` are NOT those of a proof.  One proof's own phase totals, keygen excluded, are in `profiles/{TAG}_phase_profile.json` (the replay's `--phase-profile`): at layer 0 a proof issues 34 MSM passes, not 311.\n"]
for L in (0, 3, 4):
    st, r = lines(f"L{L}_kernel_stats.txt"), record(f"L{L}_stats_record.json")
    if st and r:
        kv += [f"## layer {L} (k = {r['k']}): proof {r['resident_ms']:.0f} ms wall under the profiler; steps {json.dumps(r['step_ms'])}\n"] + st[:16] + [""]
with open(os.path.join(PROF, f"{TAG}_kernel_vs_wall.md"), "w") as f:
    f.write("\n".join(kv) + "\n")

# ------------------------------------------------------------------------------------------------ one proof's own phase totals (keygen excluded)
ph = {}
for L in (0, 3, 4):
    r = record(f"L{L}_phase_record.json")
    if r.get("phase_profile"):
        ph[f"layer{L}"] = {"k": r["k"], "resident_ms": r["resident_ms"], "step_ms": r["step_ms"], "msm": r["msm"], "intt": r["intt"], "coset_ntt": r["coset_ntt"], "phase_profile": r["phase_profile"],
                          "reduction_tail_share_of_proof": round(r["phase_profile"]["msm_reduce"]["ms"] / r["phase_profile"]["proof_ms"], 4)}
if ph:
    json.dump(ph, open(os.path.join(PROF, f"{TAG}_phase_profile.json"), "w"), indent=1)

# ------------------------------------------------------------------------------------------------ the recorded figures bench.py re-emits
if acc_bytes and ntt_bytes:
    json.dump({"msm_accumulate_k26_hbm_bytes_per_launch": float(f"{acc_bytes:.4g}"), "ntt_k26_hbm_bytes_per_transform": float(f"{ntt_bytes:.4g}"),
               "source": f"profiles/{TAG}_pmc_k26.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, {TAG} tree; recorded, not measured inside the bench run)",
               "source_sha16": {"msm": source_hash("msm"), "ntt": source_hash("ntt")}}, open(os.path.join(PROF, "pmc_latest.json"), "w"))
print("wrote", [f for f in os.listdir(PROF) if f.startswith(TAG + "_") or f == "pmc_latest.json"])
