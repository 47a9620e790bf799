// test_plonk_replay.cpp -- a COMPILED caller that proves one layer of scroll-prover's proof stack on the MI355X from the layer's PlonkProtocol: set-up (synthetic SRS with a
// known tau, a circuit instance that satisfies the protocol, keygen with the proving key resident under an HBM plan), then mi355zk::plonk::create_proof
// (include/mi355zk_plonk.hpp) through the C-ABI, and the proof / verifying key / instances written out in the reference's byte layouts.  The reference reaches
// create_proof once per layer [REF integration/src/prove.rs:36-43,67,95-97] and verifies what comes out [REF integration/src/prove.rs:50-53,75-80]; here the verifier is
// oracle/plonk.py (TEST INFRASTRUCTURE, run by tests/ and bench.py on the files this program writes -- nothing in this program links or calls the oracle).
//
//   --protocol FILE   a PlonkProtocol JSON: the reference's own (tests/golden/protocol_layer2.json = [REF release-v0.13.1/chunk.protocol], protocol_layer4.json) or one
//                     written by scroll-prover_amd/protocols.py (any layer at any k)
//   --out DIR         proof.bin, vk.bin, instances.bin, result.json; with --dump-inputs also the circuit instance (for the CPU restatement of the prover)
//   --builder-only    no GPU: build the circuit instance, dump it, exit 0 (the CPU-only tests drive oracle/plonk.py with it)
//   --host-api        additionally replay the MSM / NTT / evaluation counts through the host-pointer entry points (what a shim without DevicePoly pays)
// Prints one JSON line; exit code 0 = a proof was written, 2 = no GPU (mi355_init failed).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "mi355zk_plonk.hpp"

using namespace mi355zk::plonk;
using mi355zk::halo2::G1;
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
static void write_file(const std::string &path, const void *p, size_t bytes) { std::ofstream f(path, std::ios::binary); if (!f) throw std::invalid_argument("cannot write " + path); f.write(static_cast<const char *>(p), (std::streamsize)bytes); }

int main(int argc, char **argv) {
  std::string protocol_path, out_dir, tables = "auto", pk_mode = "auto";
  int devices = 1, threads = (int)std::thread::hardware_concurrency(), proofs = 2, upload_threads = 1, early_intt = -1;
  bool host_api = false, phase_profile = false, builder_only = false, dump_inputs = false, pinned_witness = false, corrupt = false, sparse_uploads = false, packed_m = true; uint64_t seed = 1, blind_seed = 0; bool zero_blinding = false; double fill = 0.9, assign_density = 1.0; TranscriptKind transcript = TranscriptKind::Blake2b; bool transcript_auto = true;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto next = [&]() -> long { return i + 1 < argc ? std::atol(argv[++i]) : 0; };
    auto nexts = [&]() -> std::string { return i + 1 < argc ? std::string(argv[++i]) : std::string(); };
    if (a == "--protocol") protocol_path = nexts(); else if (a == "--out") out_dir = nexts(); else if (a == "--devices") devices = (int)next();
    else if (a == "--threads") threads = (int)next(); else if (a == "--host-api") host_api = true; else if (a == "--builder-only") builder_only = true; else if (a == "--dump-inputs") dump_inputs = true;
    else if (a == "--no-tables") tables = "off"; else if (a == "--tables") tables = nexts(); else if (a == "--pk-cosets") pk_mode = nexts(); else if (a == "--proofs") proofs = (int)next();
    else if (a == "--upload-threads") upload_threads = (int)next(); else if (a == "--early-intt") early_intt = (int)next(); else if (a == "--pinned-witness") pinned_witness = true;
    else if (a == "--seed") seed = (uint64_t)next(); else if (a == "--blind-seed") blind_seed = (uint64_t)next(); else if (a == "--zero-blinding") zero_blinding = true; else if (a == "--phase-profile") phase_profile = true; else if (a == "--fill") fill = std::atof(nexts().c_str()); else if (a == "--corrupt-witness") corrupt = true;
    else if (a == "--sparse-uploads") sparse_uploads = true; else if (a == "--packed-multiplicities") packed_m = true; else if (a == "--no-packed-multiplicities") packed_m = false; else if (a == "--assign-density") assign_density = std::atof(nexts().c_str());
    else if (a == "--transcript") { const std::string tn = nexts(); if (tn == "auto") continue; transcript_auto = false; try { transcript = transcript_kind_from_name(tn); } catch (const std::exception &e) { std::printf("%s\n", e.what()); return 1; } }
    else if (a == "--transcript-selftest") {   // host only: a fixed byte stream through the Blake2b transcript (tests compare with hashlib)
      Transcript T; T.common_scalar(fr_u64(5)); const Fr c1 = fr_to_canonical(T.squeeze_challenge());
      G1 g{}; { const Fr one = fr_one(); (void)one; mi355zk::halo2::G1Affine gen{}; zk::fe_t x = zk::Fq::one(), y = zk::Fq::add(zk::Fq::one(), zk::Fq::one()); std::memcpy(gen.data(), &x, 32); std::memcpy(gen.data() + 4, &y, 32); std::memcpy(g.data(), gen.data(), 64); std::memcpy(g.data() + 8, &x, 32); }
      T.write_point(g); T.write_scalar(fr_u64(0xDEADBEEFull)); const Fr c2 = fr_to_canonical(T.squeeze_challenge());
      std::vector<uint8_t> vkb(40, 7); const Fr c3 = fr_to_canonical(vk_transcript_repr(vkb));
      auto hex = [](const Fr &c) { char b[65]; std::snprintf(b, sizeof b, "%016llx%016llx%016llx%016llx", (unsigned long long)c[3], (unsigned long long)c[2], (unsigned long long)c[1], (unsigned long long)c[0]); return std::string(b); };
      std::string ph; for (uint8_t b : T.proof) { char t[3]; std::snprintf(t, sizeof t, "%02x", b); ph += t; }
      // the same stream through the Poseidon transcript, plus squeezes on every buffer length 0 .. 9 (short chunk, exact multiple of RATE, empty) and the first constants
      Transcript S(TranscriptKind::Poseidon); S.common_scalar(fr_u64(5)); const Fr p1 = fr_to_canonical(S.squeeze_challenge());
      S.write_point(g); S.write_scalar(fr_u64(0xDEADBEEFull)); const Fr p2 = fr_to_canonical(S.squeeze_challenge());
      std::string lens; Transcript L(TranscriptKind::Poseidon);
      for (int len = 0; len < 10; len++) { for (int i = 0; i < len; i++) L.common_scalar(fr_u64(1000 * len + i)); lens += (len ? ", \"" : "\"") + hex(fr_to_canonical(L.squeeze_challenge())) + "\""; }
      Transcript E(TranscriptKind::Evm); E.common_scalar(fr_u64(5)); const Fr e1 = fr_to_canonical(E.squeeze_challenge()); const Fr e1b = fr_to_canonical(E.squeeze_challenge());   // the second: hash | 0x01
      E.write_point(g); E.write_scalar(fr_u64(0xDEADBEEFull)); const Fr e2 = fr_to_canonical(E.squeeze_challenge());
      std::string eh; for (uint8_t b : E.proof) { char t[3]; std::snprintf(t, sizeof t, "%02x", b); eh += t; }
      std::vector<uint8_t> kin(200); for (size_t i = 0; i < kin.size(); i++) kin[i] = (uint8_t)(i * 7 + 1);
      std::string kh; for (uint8_t b : keccak256(kin)) { char t[3]; std::snprintf(t, sizeof t, "%02x", b); kh += t; }
      const PoseidonSpec &PS = PoseidonSpec::get();
      std::printf("{\"c1\": \"%s\", \"c2\": \"%s\", \"vk_repr\": \"%s\", \"proof\": \"%s\", \"poseidon\": {\"c1\": \"%s\", \"c2\": \"%s\", \"by_length\": [%s], \"rc0\": \"%s\", \"rc_last\": \"%s\", \"mds00\": \"%s\", \"mds44\": \"%s\", \"proof_equal\": %s}, \"evm\": {\"c1\": \"%s\", \"c1b\": \"%s\", \"c2\": \"%s\", \"proof\": \"%s\", \"keccak_200\": \"%s\"}}\n",
                  hex(c1).c_str(), hex(c2).c_str(), hex(c3).c_str(), ph.c_str(), hex(p1).c_str(), hex(p2).c_str(), lens.c_str(), hex(fr_to_canonical(PS.rc.front())).c_str(), hex(fr_to_canonical(PS.rc.back())).c_str(),
                  hex(fr_to_canonical(PS.mds[0][0])).c_str(), hex(fr_to_canonical(PS.mds[4][4])).c_str(), S.proof == T.proof ? "true" : "false", hex(e1).c_str(), hex(e1b).c_str(), hex(e2).c_str(), eh.c_str(), kh.c_str());
      return 0;
    }
    else { std::printf("usage: %s --protocol FILE --out DIR [--builder-only] [--dump-inputs] [--devices D] [--threads T] [--proofs N] [--upload-threads U] [--early-intt 0|1] [--pinned-witness]\n"
                       "          [--tables auto|on|lagrange|off] [--pk-cosets auto|resident|on-the-fly] [--host-api] [--seed S] [--fill F] [--corrupt-witness] [--sparse-uploads] [--no-packed-multiplicities] [--transcript auto|blake2b|poseidon|evm] [--assign-density D] [--blind-seed S] [--zero-blinding] [--phase-profile] | --transcript-selftest\n", argv[0]); return 1; }
  }
  if (protocol_path.empty() || out_dir.empty()) { std::printf("--protocol and --out are required\n"); return 1; }
  if (threads < 1) threads = 1; if (threads > 16) threads = 16;
  const char *tq = std::getenv("MI355_REPLAY_THREADS"); if (tq) threads = std::max(1, std::atoi(tq));
  if (proofs < 1) proofs = 1;
  Protocol P;
  try { P.load(protocol_path); } catch (const std::exception &e) { std::printf("cannot load the protocol: %s\n", e.what()); return 1; }
  const uint32_t k = P.k, Q = P.Q; const uint64_t n = P.n;
  if (transcript_auto) transcript = reference_transcript(P);   // what the reference proves this layer with
  const Fr tau = fr_u64(0x5343524F4C4C0001ull + (uint64_t)(P.layer < 0 ? 0 : P.layer));
  if (builder_only) {
    try {
      CircuitOptions co; co.seed = seed; co.threads = threads; co.fill = fill; co.assign_density = assign_density; co.blind_seed = blind_seed; co.zero_blinding = zero_blinding;
      auto C = build_circuit(P, co);
      dump_circuit(*C, out_dir, tau, protocol_path);
      std::printf("{\"builder_only\": true, \"layer\": %d, \"k\": %u, \"copy_pairs\": %zu, \"gates_active\": %llu}\n", P.layer, k, C->pairs.size(), (unsigned long long)C->gates_active);
    } catch (const std::exception &e) { std::printf("FAILED with exception: %s\n", e.what()); return 1; }
    return 0;
  }
  {
    std::vector<int> ids(devices); for (int d = 0; d < devices; d++) ids[d] = d;
    if (std::getenv("MI355_ALLOW_DUP_DEVICES")) for (auto &d : ids) d = 0;
    const int rc = devices == 1 ? mi355_init(0) : mi355_init_multi(ids.data(), devices);
    if (rc != MI355_OK) { std::printf("mi355_init failed (%d): %s\n", rc, mi355_last_error()); return 2; }
  }
  int rc_main = 0;
  try {
    const mi355zk::halo2::EvaluationDomain dom(Q + 1, k);
    // ---- ParamsKZG (set-up time): synthetic SRS on the device, both bases registered
    uint64_t hg = 0, hl = 0;
    {
      DevicePoly g(2 * n, 0), gl(2 * n, 0);   // 64 bytes per point
      check(mi355_srs_setup_dev(g.p, gl.p, k, tau.data(), dom.omega.data()));
      check(mi355_srs_register_dev(g.p, n, 1, &hg)); check(mi355_srs_register_dev(gl.p, n, 1, &hl));
      check(mi355_synchronize());
    }
    check(mi355_buf_trim());
    // ---- HBM plan: what must live in HBM for this layer's prover, and which optional residents fit on top
    uint64_t hbm_free = 0, hbm_total = 0; check(mi355_mem_info(0, &hbm_free, &hbm_total, nullptr, nullptr, nullptr));
    const double GiB = 1024.0 * 1024 * 1024;
    const PkSizes sz = pk_sizes(P);
    uint32_t NW = 1; for (auto w : P.num_witness) NW += w;   // instance + witness polynomials
    const double working = sz.working_bytes;                  // a proof's own blocks at their peak (mi355zk_plonk.hpp pk_sizes)
    const double table_one = (double)n * 64 * (k >= 24 ? 12 : 15);
    const double usable = 0.94 * (double)hbm_free;
    bool resident = true; int n_tables = 0;
    if (pk_mode == "on-the-fly") resident = false;
    else if (pk_mode == "auto" && sz.base_bytes + sz.coset_bytes + working > usable) resident = false;
    // ---- the circuit instance (host side: what keygen and create_proof are handed) and keygen
    const auto t_build = Clock::now();
    CircuitOptions co; co.seed = seed; co.threads = threads; co.pinned = pinned_witness; co.fill = fill; co.assign_density = assign_density; co.blind_seed = blind_seed; co.zero_blinding = zero_blinding;
    auto C = build_circuit(P, co);
    const double build_ms = ms_since(t_build);
    if (corrupt) C->advice[0][3] = fr_add(C->advice[0][3], fr_one());   // one cell off: the first gate no longer holds, the verifier must reject what comes out
    if (dump_inputs) dump_circuit(*C, out_dir, tau, protocol_path);
    const auto t_keygen = Clock::now();
    auto pk = keygen(P, *C, hl, resident, devices);
    const double keygen_ms = ms_since(t_keygen);
    for (auto &c : C->pre) { Column().swap(c); }   // keygen has uploaded them; the prover reads the witness only
    std::vector<Fr>().swap(C->omega_pow);
    check(mi355_buf_trim());
    uint64_t free_after_pk = 0; check(mi355_mem_info(0, &free_after_pk, nullptr, nullptr, nullptr, nullptr));
    const double room = 0.97 * (double)free_after_pk - working - (resident ? 0.0 : sz.lean_tmp_bytes);
    if (tables == "on") n_tables = 2; else if (tables == "lagrange") n_tables = 1; else if (tables == "off") n_tables = 0;
    else { const double margin = 0.08 * (double)hbm_total; n_tables = room >= 2 * table_one + margin ? 2 : room >= table_one + margin ? 1 : 0; }
    if (devices > 1 && tables == "auto") n_tables = 2;
    if (n_tables >= 1 && mi355_srs_precompute(hl, 0, 0) != MI355_OK) { std::printf("window tables for g_lagrange did not fit (%s): table-free schedule\n", mi355_last_error()); n_tables = 0; }
    if (n_tables >= 2 && mi355_srs_precompute(hg, 0, 0) != MI355_OK) { std::printf("window tables for g did not fit (%s): Lagrange basis only\n", mi355_last_error()); n_tables = 1; }
    ProofOptions opt; opt.devices = devices; opt.threads = threads; opt.upload_threads = upload_threads; opt.early_intt = early_intt; opt.sparse_uploads = sparse_uploads; opt.packed_multiplicities = packed_m; opt.transcript = transcript;
    // ---- the proofs: a prover process runs proof after proof; the first one grows the workspace arena and the buffer pool, the last one is reported
    ProofResult R; double first_ms = 0;
    for (int it = 0; it < proofs; it++) { R = create_proof(hg, hl, *pk, *C, opt); if (it == 0) first_ms = R.total_ms; }
    // --phase-profile: one MORE proof with the library's phase events on (a few microseconds per phase: kept out of the timed proofs), so that the commitment tail, the
    // transform passes and the evaluations of ONE proof -- keygen excluded -- can be read next to its wall time (how much of a proof is the MSM reduction tail)
    std::string phases = "null";
    if (phase_profile) {
      check(mi355_profile_reset()); check(mi355_profile_enable(1));
      const ProofResult RP = create_proof(hg, hl, *pk, *C, opt);
      check(mi355_synchronize()); check(mi355_profile_enable(0));
      char buf[160]; phases = "{\"proof_ms\": "; std::snprintf(buf, sizeof buf, "%.2f", RP.total_ms); phases += buf;
      for (const char *nm : {"msm_total", "msm_digits", "msm_sort", "msm_accumulate", "msm_reduce", "ntt_total", "ntt_pass", "eval_poly"}) {
        double ms = 0; uint64_t cnt = 0; check(mi355_profile_get(nm, &ms, &cnt));
        std::snprintf(buf, sizeof buf, ", \"%s\": {\"ms\": %.2f, \"launches\": %llu}", nm, ms, (unsigned long long)cnt); phases += buf;
      }
      phases += "}";
    }
    uint64_t live = 0, pooled = 0, ws = 0, fr_end = 0; check(mi355_mem_info(0, &fr_end, nullptr, &live, &pooled, &ws));
    write_file(out_dir + "/proof.bin", R.proof.data(), R.proof.size());
    write_file(out_dir + "/vk.bin", pk->vk.data(), pk->vk.size());
    write_file(out_dir + "/instances.bin", C->instances.data(), C->instances.size() * 32);
    // ---- the same MSM / NTT / evaluation counts through the host-pointer entry points (what a shim without DevicePoly pays)
    double host_ms = -1, host_fft_ms = -1, host_fft_batched_ms = -1;
    uint32_t n_commit_lag = 0, n_commit_coef = Q + 3; for (auto w : P.num_witness) n_commit_lag += w; n_commit_lag -= 1;
    if (host_api) {
      std::vector<Fr> hp(C->advice[0].begin(), C->advice[0].end()); std::vector<Fr> hext(Q * n); G1 out; std::vector<std::vector<Fr>> cols(std::min<uint32_t>(8, NW), hp);
      const uint32_t n_intt = NW - 1, n_ntt = R.coset_ntt, n_ev = R.evals + Q;
      const auto t1 = Clock::now();
      for (uint32_t i = 0; i < n_commit_lag; i++) check(mi355_msm_g1_host(hl, 0, hp.data(), n, out.data()));
      for (uint32_t i = 0; i < n_commit_coef; i++) check(mi355_msm_g1_host(hg, 0, hp.data(), n, out.data()));
      const auto t2 = Clock::now();
      for (uint32_t i = 0; i < n_intt; i++) check(mi355_intt_fr_host(hp.data(), k, dom.omega_inv.data(), dom.ifft_divisor.data()));
      for (uint32_t i = 0; i < n_ntt; i++) check(mi355_ntt_fr_host(hp.data(), k, dom.omega.data()));
      host_fft_ms = ms_since(t2);
      check(mi355_extended_to_coeff_host(hext.data(), dom.extended_k, dom.g_coset.data(), dom.g_coset_inv.data(), dom.extended_omega_inv.data(), dom.extended_ifft_divisor.data()));
      Fr e; for (uint32_t i = 0; i < n_ev; i++) check(mi355_eval_polynomial_host(hp.data(), n, tau.data(), e.data()));
      host_ms = ms_since(t1);
      const uint32_t B = (uint32_t)cols.size();
      std::vector<void *> ptrs(B); for (uint32_t i = 0; i < B; i++) ptrs[i] = cols[i].data();
      const auto t3 = Clock::now();
      for (uint32_t done = 0; done < n_intt; done += B) check(mi355_ntt_fr_batch_host(ptrs.data(), std::min(B, n_intt - done), k, dom.omega_inv.data(), dom.ifft_divisor.data()));
      for (uint32_t done = 0; done < n_ntt; done += B) check(mi355_ntt_fr_batch_host(ptrs.data(), std::min(B, n_ntt - done), k, dom.omega.data(), nullptr));
      host_fft_batched_ms = ms_since(t3);
    }
    char line[6144];
    std::snprintf(line, sizeof line,
      "{\"replay\": \"mi355zk::plonk::create_proof (include/mi355zk_plonk.hpp): the layer's PlonkProtocol compiled and proven through the C-ABI, polynomials and proving key resident\", \"layer\": %d, \"k\": %u, \"devices\": %d, "
      "\"protocol\": {\"num_preprocessed\": %u, \"num_witness\": [%u, %u, %u], \"quotient_pieces\": %u, \"evaluations\": %zu, \"queries\": %zu, \"permutation_chunks\": %zu, \"lookups\": %zu, \"gates\": %zu, \"last_rotation\": %d}, "
      "\"circuit\": {\"copy_pairs\": %zu, \"gates_active\": %llu, \"lookup_rows\": %llu, \"build_ms\": %.1f, \"keygen_ms\": %.1f}, "
      "\"plan\": {\"constraints\": %u, \"launches_per_part\": %u, \"terms\": %u, \"temporaries\": %u, \"prefix_groups\": %u, \"common_polynomials\": %zu}, "
      "\"gate_eval_process_totals\": {\"launches\": %llu, \"algorithmic_bytes\": %llu, \"factor_rows\": %llu, \"term_rows\": %llu, \"distinct_operand_rows\": %llu, \"hot_repeat_rows\": %llu}, "
      "\"window_tables\": %s, \"window_table_bases\": %d, \"pk_cosets\": \"%s\", \"upload_threads\": %d, \"pinned_witness\": %s, \"sparse_uploads\": {\"on\": %s, \"columns\": %llu, \"packed_columns\": %llu, \"witness_link_gib\": %.2f, \"assign_density\": %.2f}, "
      "\"msm\": %u, \"intt\": %u, \"coset_ntt\": %u, \"gate_launches\": %u, \"evals\": %u, \"rotation_sets\": %u, \"proof_bytes\": %zu, \"resident_ms\": %.3f, \"first_proof_ms\": %.3f, \"proofs\": %d, "
      "\"host_api_ms\": %.3f, \"host_api_fft_ms\": %.3f, \"host_api_fft_batched_ms\": %.3f, "
      "\"step_ms\": {\"1_instance\": %.2f, \"2_3_advice_lookup_commits\": %.2f, \"4_products\": %.2f, \"5_random\": %.2f, \"6_to_coeff\": %.2f, \"7_quotient\": %.2f, \"8_commit_h\": %.2f, \"9_evals\": %.2f, \"10_shplonk\": %.2f}, "
      "\"hbm\": {\"total_gib\": %.1f, \"peak_used_gib\": %.1f, \"proving_key_gib\": %.1f, \"live_buffers_gib\": %.1f, \"pooled_gib\": %.1f, \"workspace_gib\": %.1f, \"planned\": {\"pk_base_gib\": %.1f, \"pk_cosets_gib\": %.1f, \"working_set_gib\": %.1f, \"one_table_gib\": %.1f, \"usable_gib\": %.1f}}, "
      "\"transcript\": \"%s\", \"phase_profile\": %s, \"ok\": true}",
      P.layer, k, devices, P.num_pre, P.num_witness[0], P.num_witness[1], P.num_witness[2], Q, P.evaluations.size(), P.queries.size(), P.perm.size(), P.lookups.size(), P.gates.size(), P.last_rot,
      C->pairs.size(), (unsigned long long)C->gates_active, (unsigned long long)C->lookup_rows, build_ms, keygen_ms,
      R.plan_constraints, R.plan_launches, R.plan_terms, R.plan_tmps, R.plan_prefix_groups, pk->commons.defs.size(),
      (unsigned long long)gate_stats().launches.load(), (unsigned long long)gate_stats().bytes.load(), (unsigned long long)gate_stats().factor_rows.load(), (unsigned long long)gate_stats().term_rows.load(), (unsigned long long)gate_stats().distinct_rows.load(), (unsigned long long)gate_stats().hot_repeat_rows.load(),
      n_tables ? "true" : "false", n_tables, resident ? "resident" : "on-the-fly", upload_threads, pinned_witness ? "true" : "false", sparse_uploads ? "true" : "false", (unsigned long long)R.sparse_columns, (unsigned long long)R.packed_columns, R.witness_link_bytes / GiB, assign_density,
      R.msm, R.intt, R.coset_ntt, R.gate_launches, R.evals, R.rotation_sets, R.proof.size(), R.total_ms, first_ms, proofs,
      host_ms, host_fft_ms, host_fft_batched_ms,
      R.step_ms[1], R.step_ms[2], R.step_ms[4], R.step_ms[5], R.step_ms[6], R.step_ms[7], R.step_ms[8], R.step_ms[9], R.step_ms[10],
      hbm_total / GiB, (hbm_total - fr_end) / GiB, pk->bytes / GiB, live / GiB, pooled / GiB, ws / GiB, sz.base_bytes / GiB, sz.coset_bytes / GiB, working / GiB, table_one / GiB, usable / GiB, transcript_name(transcript), phases.c_str());
    std::printf("%s\n", line);
    write_file(out_dir + "/result.json", line, std::strlen(line));
    pk.reset();
    check(mi355_srs_release(hg)); check(mi355_srs_release(hl));
  } catch (const std::exception &e) { std::printf("FAILED with exception: %s\n", e.what()); rc_main = 1; }
  (void)mi355_shutdown();
  std::fflush(stdout);
  return rc_main;
}
