// test_prover_process.cpp -- ONE prover process that holds SEVERAL layers, as the reference's provers do: a ChunkProver owns the SRS of the degrees {20, 24, 25} and the proving
// keys of layers 0, 1, 2 and runs their three create_proof calls back to back inside gen_halo2_chunk_proof [REF integration/src/prove.rs:30-43], a BatchProver the degrees {21, 26}
// and layers 3, 4 (then 5, 6 for the bundle) [REF integration/src/prove.rs:11-21,67,95-97], [REF bin/src/trace_prover.rs:35-36].  Everything resident at once does not fit
// 288 GiB: mi355zk::plonk::plan_residency (include/mi355zk_plonk.hpp) decides which keys keep their coset parts and which bases get window tables, and this
// program runs exactly that plan -- set-up for every layer first (SRS per degree, circuit instance, keygen), then `--proofs` rounds of the layers' proofs in order.
//
//   test_prover_process --protocol L0.json --protocol L1.json --protocol L2.json --out DIR [--proofs N] [--threads T]
// writes DIR/<i>/{proof,vk,instances}.bin per protocol (verified by oracle/plonk.py in tests / bench.py) and prints one JSON line.  Exit code 2 = no GPU.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <sys/stat.h>

#include "mi355zk_plonk.hpp"

using namespace mi355zk::plonk;
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
static void write_file(const std::string &path, const void *p, size_t bytes) { std::ofstream f(path, std::ios::binary); if (!f) throw std::invalid_argument("cannot write " + path); f.write(static_cast<const char *>(p), (std::streamsize)bytes); }

int main(int argc, char **argv) {
  std::vector<std::string> paths; std::string out_dir; int threads = (int)std::thread::hardware_concurrency(), proofs = 2; uint64_t seed = 1; bool trim_between = false;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--protocol" && i + 1 < argc) paths.push_back(argv[++i]); else if (a == "--out" && i + 1 < argc) out_dir = argv[++i];
    else if (a == "--proofs" && i + 1 < argc) proofs = std::atoi(argv[++i]); else if (a == "--threads" && i + 1 < argc) threads = std::atoi(argv[++i]);
    else if (a == "--seed" && i + 1 < argc) seed = (uint64_t)std::atol(argv[++i]); else if (a == "--trim") trim_between = true;
    else { std::printf("usage: %s --protocol FILE [--protocol FILE ...] --out DIR [--proofs N] [--threads T] [--seed S] [--trim]\n", argv[0]); return 1; }
  }
  if (paths.empty() || out_dir.empty()) { std::printf("--protocol (one or more) and --out are required\n"); return 1; }
  threads = std::max(1, std::min(threads, 16)); proofs = std::max(1, proofs);
  const size_t NLY = paths.size();
  std::vector<std::unique_ptr<Protocol>> P(NLY);
  try { for (size_t i = 0; i < NLY; i++) { P[i] = std::make_unique<Protocol>(); P[i]->load(paths[i]); } } catch (const std::exception &e) { std::printf("cannot load a protocol: %s\n", e.what()); return 1; }
  if (mi355_init(0) != MI355_OK) { std::printf("mi355_init failed: %s\n", mi355_last_error()); return 2; }
  int rc_main = 0;
  try {
    const double GiB = 1024.0 * 1024 * 1024;
    uint64_t hbm_free = 0, hbm_total = 0; check(mi355_mem_info(0, &hbm_free, &hbm_total, nullptr, nullptr, nullptr));
    std::vector<const Protocol *> ptrs; for (auto &p : P) ptrs.push_back(p.get());
    const ResidencyPlan plan = plan_residency(ptrs, (double)hbm_total / GiB);
    // ---- set-up: one SRS per DEGREE (layers of one degree share it, as params_map does), then every layer's circuit instance and proving key
    struct Srs { uint64_t g = 0, gl = 0; Fr tau; };
    std::map<uint32_t, Srs> srs;
    for (auto &p : P) if (!srs.count(p->k)) {
      Srs s; s.tau = fr_u64(0x5343524F4C4C0001ull + 1000 + p->k);
      const mi355zk::halo2::EvaluationDomain dom(2, p->k);
      DevicePoly g(2 * p->n, 0), gl(2 * p->n, 0);
      check(mi355_srs_setup_dev(g.p, gl.p, p->k, s.tau.data(), dom.omega.data()));
      check(mi355_srs_register_dev(g.p, p->n, 1, &s.g)); check(mi355_srs_register_dev(gl.p, p->n, 1, &s.gl));
      check(mi355_synchronize());
      srs[p->k] = s;
    }
    check(mi355_buf_trim());
    std::vector<std::unique_ptr<Circuit>> C(NLY); std::vector<std::unique_ptr<ProvingKey>> pk(NLY);
    const auto t_setup = Clock::now();
    for (size_t i = 0; i < NLY; i++) {
      CircuitOptions co; co.seed = seed + i; co.threads = threads;
      C[i] = build_circuit(*P[i], co);
      pk[i] = keygen(*P[i], *C[i], srs[P[i]->k].gl, plan.layers[i].cosets_resident, 1);
      for (auto &c : C[i]->pre) Column().swap(c);
      std::vector<Fr>().swap(C[i]->omega_pow);
      check(mi355_buf_trim());
    }
    std::string tables_note; std::set<uint64_t> tabled;
    for (size_t i = 0; i < NLY; i++) {   // window tables where the plan placed them; a table that does not fit after all is dropped (table-free schedule), not an error
      Srs &s = srs[P[i]->k];
      if (plan.layers[i].table_lagrange && tabled.insert(s.gl).second && mi355_srs_precompute(s.gl, 0, 0) != MI355_OK) tables_note += "k" + std::to_string(P[i]->k) + " lagrange table dropped; ";
      if (plan.layers[i].table_coeff && tabled.insert(s.g).second && mi355_srs_precompute(s.g, 0, 0) != MI355_OK) tables_note += "k" + std::to_string(P[i]->k) + " coefficient table dropped; ";
    }
    const double setup_ms = ms_since(t_setup);
    // ---- the proofs: every round runs the layers in order, as gen_halo2_chunk_proof / gen_batch_proof do; the last round is reported
    ProofOptions opt; opt.threads = threads; opt.packed_multiplicities = true;
    std::vector<ProofResult> R(NLY); std::vector<double> round_ms;
    for (int it = 0; it < proofs; it++) {
      const auto t0 = Clock::now();
      for (size_t i = 0; i < NLY; i++) {
        // the previous layer's blocks have another size: the library carves this layer's blocks out of the same slabs (lib_core.hip "Slabs"), nothing goes back to HIP in between.
        // --trim forces mi355_buf_trim at every layer boundary (the behaviour a caller had to choose before the slabs existed; kept for A/B)
        if (NLY > 1 && trim_between) check(mi355_buf_trim());
        opt.transcript = reference_transcript(*P[i]);   // Poseidon for the layers the next one verifies in-circuit
        R[i] = create_proof(srs[P[i]->k].g, srs[P[i]->k].gl, *pk[i], *C[i], opt);
      }
      round_ms.push_back(ms_since(t0));
    }
    uint64_t fr_end = 0, live = 0, pooled = 0, ws = 0; check(mi355_mem_info(0, &fr_end, nullptr, &live, &pooled, &ws));
    std::string layers_json;
    for (size_t i = 0; i < NLY; i++) {
      const std::string d = out_dir + "/" + std::to_string(i); mkdir(d.c_str(), 0755);
      write_file(d + "/proof.bin", R[i].proof.data(), R[i].proof.size()); write_file(d + "/vk.bin", pk[i]->vk.data(), pk[i]->vk.size());
      write_file(d + "/instances.bin", C[i]->instances.data(), C[i]->instances.size() * 32);
      const Fr tc = fr_to_canonical(srs[P[i]->k].tau); char hex[65];
      std::snprintf(hex, sizeof hex, "%016llx%016llx%016llx%016llx", (unsigned long long)tc[3], (unsigned long long)tc[2], (unsigned long long)tc[1], (unsigned long long)tc[0]);
      char b[1024];
      std::snprintf(b, sizeof b, "%s{\"index\": %zu, \"layer\": %d, \"k\": %u, \"tau\": \"%s\", \"transcript\": \"%s\", \"ms\": %.3f, \"msm\": %u, \"coset_ntt\": %u, \"proof_bytes\": %zu, \"cosets_resident\": %s, \"table_lagrange\": %s, \"table_coeff\": %s, \"proving_key_gib\": %.1f}",
                    i ? ", " : "", i, P[i]->layer, P[i]->k, hex, transcript_name(reference_transcript(*P[i])), R[i].total_ms, R[i].msm, R[i].coset_ntt, R[i].proof.size(), plan.layers[i].cosets_resident ? "true" : "false",
                    plan.layers[i].table_lagrange ? "true" : "false", plan.layers[i].table_coeff ? "true" : "false", pk[i]->bytes / GiB);
      layers_json += b;
    }
    std::printf("{\"prover_process\": \"several layers resident in one process (plan_residency), proofs back to back\", \"layers\": [%s], \"round_ms\": %.3f, \"first_round_ms\": %.3f, \"rounds\": %d, \"setup_ms\": %.1f, "
                "\"plan\": {\"srs_gib\": %.1f, \"keys_gib\": %.1f, \"tables_gib\": %.1f, \"working_gib\": %.1f, \"total_gib\": %.1f, \"budget_gib\": %.1f, \"fits\": %s, \"note\": \"%s\"}, "
                "\"hbm\": {\"total_gib\": %.1f, \"peak_used_gib\": %.1f, \"live_buffers_gib\": %.1f, \"pooled_gib\": %.1f, \"workspace_gib\": %.1f}, \"ok\": true}\n",
                layers_json.c_str(), round_ms.back(), round_ms.front(), proofs, setup_ms, plan.srs_gib, plan.keys_gib, plan.tables_gib, plan.working_gib, plan.total_gib, plan.budget_gib, plan.fits ? "true" : "false", tables_note.c_str(),
                hbm_total / GiB, (hbm_total - fr_end) / GiB, live / GiB, pooled / GiB, ws / GiB);
    pk.clear();
    for (auto &kv : srs) { check(mi355_srs_release(kv.second.g)); check(mi355_srs_release(kv.second.gl)); }
  } catch (const std::exception &e) { std::printf("FAILED with exception: %s\n", e.what()); rc_main = 1; }
  (void)mi355_shutdown();
  std::fflush(stdout);
  return rc_main;
}
