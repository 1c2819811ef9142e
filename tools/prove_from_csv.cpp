// The production path of the reference's backend -- Snapshot::new + generate_proof_of_inclusion
// [REF backend/src/apis/round.rs:132-174] -- as ONE compiled program over the library, no interpreter anywhere:
//   SRS file (halo2 RawBytes) + CSV of (username, balances..) + user index  ->  {"proof": "0x..", "public_inputs": [..]}
// Steps: parse the CSV, hash the usernames (Keccak), build the Merkle sum tree on the device (sg_mst_build_dev), read /
// downsize the parameters, lay out MstInclusionCircuit<LEVELS, N_CURRENCIES, 8> with the reference's floor plan
// (include/summa_circuit.hpp), commit to its fixed and permutation columns (the verifying key), build the proving key's forms,
// synthesize the user's witness on the device (sg_mst_inclusion_witness_dev), create_proof (Keccak transcript, SHPLONK), and --
// as create_proof_checked does [REF utils.rs:181-193] -- verify the proof right away (sp_verify_proof, the file's g2 / s_g2).
//   build: hipcc -O2 -std=c++17 -Iinclude tools/prove_from_csv.cpp -o tools/prove_from_csv -Lcircuits_halo2_amd -lsumma_gpu
//   usage: prove_from_csv <srs file> <csv> <user index> <k> <out.json> [vk digest as 0x.. (default: this build's digest)] [reps]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "summa_circuit.hpp"
#include "summa_prover.h"

using namespace summa::prover;
using namespace summa::circuit;
using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); }

static std::string hex32(const uint8_t be[32]) {
  static const char* d = "0123456789abcdef";
  std::string s = "0x";
  for (int i = 0; i < 32; i++) {
    s += d[be[i] >> 4];
    s += d[be[i] & 15];
  }
  return s;
}

int main(int argc, char** argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s <srs file> <csv> <user index> <k> <out.json> [vk digest 0x..] [reps]\n", argv[0]);
    return 2;
  }
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  try {
    const std::string srs_path = argv[1], csv_path = argv[2], out_path = argv[5];
    const size_t user = (size_t)std::atoll(argv[3]);
    const uint32_t k = (uint32_t)std::atoi(argv[4]);
    const int reps = argc > 7 ? std::atoi(argv[7]) : 3;
    ck(sg_init(0), "sg_init");
    auto t_all = clk::now();

    // ---- CSV -> entries (utils/csv_parser.rs: one balance per header column after the username)
    std::ifstream csv(csv_path);
    if (!csv) throw std::runtime_error("cannot open the csv");
    std::string line;
    std::getline(csv, line);
    const char delim = line.find(';') != std::string::npos ? ';' : ',';
    auto split = [&](const std::string& l) {
      std::vector<std::string> out;
      std::stringstream ss(l);
      std::string cell;
      while (std::getline(ss, cell, delim)) {
        while (!cell.empty() && (cell.back() == '\r' || cell.back() == '\n')) cell.pop_back();
        out.push_back(cell);
      }
      return out;
    };
    const uint32_t nc = (uint32_t)split(line).size() - 1;
    if (nc == 0 || nc > 64) throw std::runtime_error("csv: no balance columns");
    std::vector<Fr> users, balances;
    while (std::getline(csv, line)) {
      if (line.empty()) continue;
      const auto cells = split(line);
      if (cells.size() != nc + 1) throw std::runtime_error("csv: wrong number of columns in a row");
      users.push_back(fr_from_username(cells[0]));
      for (uint32_t c = 0; c < nc; c++) balances.push_back(fr_from_decimal(cells[1 + c]));
    }
    if (users.empty() || user >= users.size()) throw std::runtime_error("user index out of bounds");
    uint32_t depth = 0;
    while (((size_t)1 << depth) < users.size()) depth++;
    const size_t size = (size_t)1 << depth;
    users.resize(size, Fr::zero());             // zero entries: username 0, balances 0 (entry.rs:30-38)
    balances.resize(size * nc, Fr::zero());

    // ---- Merkle sum tree on the device, kept there
    DevCol d_users(size), d_bals(size * nc), d_h(2 * size - 1), d_b((2 * size - 1) * nc);
    d_users.upload(users.data(), 0, size);
    d_bals.upload(balances.data(), 0, size * nc);
    auto t = clk::now();
    ck(sg_mst_build_dev(d_users.p, d_bals.p, depth, nc, d_h.p, d_b.p, nullptr), "mst build");
    hk(hipDeviceSynchronize(), "sync");
    const double tree_ms = ms_since(t);

    // ---- parameters: ParamsKZG::read, downsized when the file is larger (utils.rs:52-66)
    std::ifstream sf(srs_path, std::ios::binary);
    if (!sf) throw std::runtime_error("couldn't load params");
    uint32_t k_file = 0;
    sf.read(reinterpret_cast<char*>(&k_file), 4);
    if (!sf || k_file > 28) throw std::runtime_error("Failed to read params");
    if (k_file < k) throw std::runtime_error("k is too large for the given params");
    const size_t n_file = (size_t)1 << k_file, n = (size_t)1 << k;
    std::vector<uint8_t> g(64 * n_file), gl(64 * n_file);
    sf.read(reinterpret_cast<char*>(g.data()), (std::streamsize)g.size());
    sf.read(reinterpret_cast<char*>(gl.data()), (std::streamsize)gl.size());
    if (!sf) throw std::runtime_error("Failed to read params");
    uint8_t g2[128], s_g2[128];   // the verifier's side of the parameters: the container's last 256 bytes
    sf.read(reinterpret_cast<char*>(g2), 128);
    sf.read(reinterpret_cast<char*>(s_g2), 128);
    if (!sf) throw std::runtime_error("Failed to read params");
    if (k_file > k) {
      g.resize(64 * n);
      gl.resize(64 * n);
      ck(sg_g1_to_lagrange(g.data(), k, gl.data()), "downsize");
    }
    uint64_t srs = 0;
    ck(sg_srs_upload(k, g.data(), gl.data(), &srs), "srs upload");
    ck(sg_srs_precompute(srs, 0, 0), "precompute");
    ck(sg_srs_precompute(srs, 1, 0), "precompute");
    ck(sg_srs_precompute(srs, 2, 0), "precompute");

    // ---- key generation: the empty circuit's floor plan -> fixed / permutation columns -> commitments (vk) -> pk forms
    t = clk::now();
    FloorPlan fp(k, depth, nc, 8);
    uint8_t omega_b[32];
    ck(sg_domain_constant(k, 0, omega_b), "domain constant");
    Fr omega;
    std::memcpy(omega.l, omega_b, 32);
    const auto sigma_host = fp.sigma(omega);
    std::vector<DevCol> fixed, sigma;
    std::vector<void*> col_ptrs;
    for (auto& c : fp.fixed) {
      fixed.emplace_back(n);
      fixed.back().upload(c.data(), 0, n);
      col_ptrs.push_back(fixed.back().p);
    }
    for (auto& c : sigma_host) {
      sigma.emplace_back(n);
      sigma.back().upload(c.data(), 0, n);
      col_ptrs.push_back(sigma.back().p);
    }
    std::vector<std::array<uint8_t, 64>> comms(col_ptrs.size());
    ck(sg_commit_batch_dev(srs, 1, col_ptrs.data(), col_ptrs.size(), n, nullptr, comms[0].data()), "vk commitments");
    ProvingKey pk;
    if (argc > 6 && std::string(argv[6]).size() == 66) {   // halo2's own digest, where known (the contract's vk_digest)
      for (int i = 0; i < 32; i++) pk.vk_digest_be[i] = (uint8_t)std::stoul(std::string(argv[6]).substr(2 + 2 * i, 2), nullptr, 16);
    } else {
      const auto d = verifying_key_digest(k, nc, comms);
      std::memcpy(pk.vk_digest_be, d.data(), 32);
    }
    pk.gates = gate_graph(nc);
    pk.gate_challenge_exps = gate_challenge_exponents(nc);
    pk.lookup_input = lookup_input_graph();
    pk.build(k, srs, std::move(fixed), std::move(sigma));
    const double keygen_ms = ms_since(t);

    // ---- the user's witness on the device, the public inputs
    DevCol d_prog((fp.program.size() + 7) / 8), d_idx(1), advice_all(3 * n);
    hk(hipMemcpy(d_prog.p, fp.program.data(), 4 * fp.program.size(), hipMemcpyHostToDevice), "H2D");
    const uint32_t idx = (uint32_t)user;
    hk(hipMemcpy(d_idx.p, &idx, 4, hipMemcpyHostToDevice), "H2D");
    std::vector<Fr> instances(2 + nc);
    hk(hipMemcpy(instances[0].l, d_h.at(user), 32, hipMemcpyDeviceToHost), "D2H");          // leaf hash
    hk(hipMemcpy(instances[1].l, d_h.at(2 * size - 2), 32, hipMemcpyDeviceToHost), "D2H");  // root hash
    hk(hipMemcpy(instances[2].l, d_b.at((2 * size - 2) * nc), 32 * nc, hipMemcpyDeviceToHost), "D2H");

    std::vector<uint8_t> proof;
    double best = 1e30, witness_ms = 0, verify_ms = 0;
    for (int r = 0; r < reps + 1; r++) {
      hk(hipDeviceSynchronize(), "sync");
      auto t0 = clk::now();
      ck(sg_mst_inclusion_witness_dev(d_prog.p, fp.n_items, fp.n_absorbs, d_users.p, d_h.p, d_b.p, depth, nc, d_idx.p, 1, advice_all.p, n,
                                      nullptr), "witness");
      hk(hipDeviceSynchronize(), "sync");
      witness_ms = ms_since(t0);
      std::vector<DevCol> advice;
      for (int c = 0; c < 3; c++) advice.push_back(DevCol::borrow(advice_all.at((size_t)c * n), n));
      t0 = clk::now();
      proof = create_proof(pk, advice, instances);
      if (r) best = std::min(best, ms_since(t0));
    }

    // ---- create_proof_checked: the proof is verified before it is handed out
    int accepted = 0;
    {
      const auto t0 = clk::now();
      if (sp_verify_proof(k, nc, pk.vk_digest_be, comms[0].data(), comms[summa::prover::NUM_FIXED].data(), g2, s_g2, proof.data(), proof.size(),
                          reinterpret_cast<const uint8_t*>(instances.data()), (uint32_t)instances.size(), SP_TRANSCRIPT_EVM, &accepted) != SG_OK)
        throw std::runtime_error(std::string("verify_proof: ") + sp_verify_last_error());
      verify_ms = ms_since(t0);
      if (!accepted) throw std::runtime_error("the proof just made does not verify");
    }
    // ---- calldata JSON (what gen_proof_solidity_calldata hands back) + the verifying key for whoever checks it
    std::ofstream out(out_path);
    out << "{\"proof\": \"0x";
    static const char* hx = "0123456789abcdef";
    for (uint8_t b : proof) out << hx[b >> 4] << hx[b & 15];
    out << "\", \"public_inputs\": [";
    for (size_t i = 0; i < instances.size(); i++) {
      uint8_t be[32];
      instances[i].to_be_bytes(be);
      out << (i ? ", " : "") << "\"" << hex32(be) << "\"";
    }
    out << "], \"vk_digest\": \"" << hex32(pk.vk_digest_be) << "\", \"commitments\": [";
    for (size_t i = 0; i < comms.size(); i++) {
      uint8_t x[32], y[32];
      fq_mont_to_be(comms[i].data(), x);
      fq_mont_to_be(comms[i].data() + 32, y);
      out << (i ? ", " : "") << "[\"" << hex32(x) << "\", \"" << hex32(y) << "\"]";
    }
    out << "]}\n";
    std::printf("{\"k\": %u, \"levels\": %u, \"n_currencies\": %u, \"users\": %zu, \"rows_used\": %u, \"tree_ms\": %.3f, \"keygen_ms\": %.3f, "
                "\"witness_ms\": %.3f, \"create_proof_ms\": %.3f, \"verify_ms\": %.3f, \"verified\": true, \"proof_bytes\": %zu, \"total_ms\": %.1f}\n",
                k, depth, nc, users.size(), fp.rows_used, tree_ms, keygen_ms, witness_ms, best, verify_ms, proof.size(), ms_since(t_all));
    sg_srs_free(srs);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "prove_from_csv: %s\n", e.what());
    return 1;
  }
  sg_shutdown();
  return 0;
}
