// C++ driver of include/summa_prover.hpp: reads a bundle written by circuits_halo2_amd.prover.export_bundle (SRS,
// proving-key columns, GraphEvaluator programs, an assignment), builds the proving key's device forms, runs
// create_proof `reps` times, writes the last proof to <out> and prints one JSON line with the best wall time.
//   build: hipcc -O2 -std=c++17 -Iinclude tools/create_proof_main.cpp -o tools/create_proof_cpp -Lcircuits_halo2_amd -lsumma_gpu
//   usage: create_proof_cpp <bundle> <proof out> [reps = 5]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "summa_prover.hpp"

using namespace summa::prover;

struct Reader {
  std::ifstream f;
  explicit Reader(const char* path) : f(path, std::ios::binary) {
    if (!f) throw std::runtime_error("cannot open bundle");
  }
  void read(void* dst, size_t bytes) {
    f.read(static_cast<char*>(dst), (std::streamsize)bytes);
    if ((size_t)f.gcount() != bytes) throw std::runtime_error("short bundle");
  }
  uint32_t u32() {
    uint32_t v;
    read(&v, 4);
    return v;
  }
  Graph graph() {
    Graph g;
    g.constants.resize(32 * (size_t)u32());
    read(g.constants.data(), g.constants.size());
    g.rotations.resize(u32());
    read(g.rotations.data(), 4 * g.rotations.size());
    g.calculations.resize(u32());
    for (auto& c : g.calculations) {
      uint32_t w[9];
      read(w, sizeof w);
      c.op = w[0];
      c.a = sg_value_source{w[1], w[2], w[3]};
      c.b = sg_value_source{w[4], w[5], w[6]};
      c.parts_offset = w[7];
      c.parts_len = w[8];
    }
    g.parts.resize(u32());
    for (auto& p : g.parts) {
      uint32_t w[3];
      read(w, sizeof w);
      p = sg_value_source{w[0], w[1], w[2]};
    }
    return g;
  }
  DevCol column(size_t n) {
    std::vector<uint8_t> h(32 * n);
    read(h.data(), h.size());
    DevCol c(n);
    c.upload(h.data(), 0, n);
    return c;
  }
};

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <bundle> <proof out> [reps]\n", argv[0]);
    return 2;
  }
  const int reps = argc > 3 ? std::atoi(argv[3]) : 5;
  // the prover issues independent chains on side streams; with HIP's default of 4 hardware queues they end up sharing
  // a queue with the main stream (measured: the phase-1 transforms serialised with the commitments) -- ask for more
  // before the runtime initialises
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  try {
    ck(sg_init(0), "sg_init");
    if (const char* sp = std::getenv("SG_PARAMS")) {   // e.g. SG_PARAMS=msm.log_seg=4,msm.quad=0 (tuning sweeps)
      std::string all(sp);
      size_t pos = 0;
      while (pos < all.size()) {
        const size_t end = all.find(',', pos) == std::string::npos ? all.size() : all.find(',', pos);
        const std::string kv = all.substr(pos, end - pos);
        const size_t eq = kv.find('=');
        if (eq != std::string::npos) ck(sg_set_param(kv.substr(0, eq).c_str(), std::atoll(kv.substr(eq + 1).c_str())), "sg_set_param");
        pos = end + 1;
      }
    }
    const char* acc_log = std::getenv("SG_ACC_LOG");   // profiling: the library's msm_accumulate launch log, written at the end
    if (acc_log) ck(sg_set_param("msm.acc_log", 1), "sg_set_param");
    Reader rd(argv[1]);
    char magic[8];
    rd.read(magic, 8);
    if (std::memcmp(magic, "SGPB2\0\0\0", 8)) throw std::runtime_error("not a prover bundle");
    const uint32_t k = rd.u32(), n_inst = rd.u32();
    const size_t n = (size_t)1 << k;
    std::vector<uint8_t> g(64 * n), gl(64 * n);
    rd.read(g.data(), g.size());
    rd.read(gl.data(), gl.size());
    uint64_t srs;
    ck(sg_srs_upload(k, g.data(), gl.data(), &srs), "srs upload");
    ck(sg_srs_precompute(srs, 0, 0), "precompute");
    ck(sg_srs_precompute(srs, 1, 0), "precompute");
    ck(sg_srs_precompute(srs, 2, 0), "precompute");
    std::vector<DevCol> fixed, sigma, advice;
    for (uint32_t i = 0; i < NUM_FIXED; i++) fixed.push_back(rd.column(n));
    for (uint32_t i = 0; i < NUM_SIGMA; i++) sigma.push_back(rd.column(n));
    for (uint32_t i = 0; i < NUM_ADVICE; i++) advice.push_back(rd.column(n));
    std::vector<Fr> instances(n_inst);
    rd.read(instances.data(), 32 * (size_t)n_inst);   // Montgomery
    ProvingKey pk;
    rd.read(pk.vk_digest_be, 32);
    pk.gates = rd.graph();
    pk.lookup_input = rd.graph();
    pk.gate_challenge_exps.resize(rd.u32());
    for (auto& group : pk.gate_challenge_exps) {
      group.resize(rd.u32());
      for (auto& e : group) e = rd.u32();
    }
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    pk.build(k, srs, std::move(fixed), std::move(sigma));
    const double keygen_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
    std::vector<uint8_t> proof;
    double best = 1e30;
    for (int r = 0; r < reps + 1; r++) {   // the first run warms work spaces, plans and the program cache
      hk(hipDeviceSynchronize(), "sync");
      t0 = clk::now();
      proof = create_proof(pk, advice, instances);
      const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
      if (r) best = std::min(best, ms);
      if (std::getenv("SG_PROVER_VERBOSE")) std::fprintf(stderr, "run %d: %.3f ms\n", r, ms);
    }
    Timings tm;
    create_proof(pk, advice, instances, &tm);
    std::ofstream out(argv[2], std::ios::binary);
    out.write(reinterpret_cast<const char*>(proof.data()), (std::streamsize)proof.size());
    std::printf("{\"k\": %u, \"driver\": \"c++\", \"create_proof_ms\": %.3f, \"keygen_transforms_ms\": %.3f, \"proof_bytes\": %zu", k,
                best, keygen_ms, proof.size());
    for (auto& kv : tm.ms) std::printf(", \"%s\": %.3f", kv.first.c_str(), kv.second);
    std::printf("}\n");
    if (acc_log) {
      size_t held = 0;
      ck(sg_msm_launch_log(nullptr, 0, &held), "launch log");
      std::vector<uint32_t> words(8 * std::max<size_t>(held, 1));
      ck(sg_msm_launch_log(words.data(), held, &held), "launch log");
      if (FILE* f = std::fopen(acc_log, "w")) {
        std::fprintf(f, "{\"region\": \"whole process\", \"launches\": [");
        for (size_t i = 0; i < held; i++) {
          const uint32_t* w = &words[8 * i];
          std::fprintf(f, "%s{\"entries\": %llu, \"n\": %u, \"M\": %u, \"threads\": %u, \"fixed\": %u, \"jobs_in_flight\": %u, \"task_len\": %u}",
                       i ? ", " : "", (unsigned long long)(w[0] | ((uint64_t)w[1] << 32)), w[2], w[3], w[4], w[5], w[6], w[7]);
        }
        std::fprintf(f, "]}\n");
        std::fclose(f);
      }
    }
    sg_srs_free(srs);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "create_proof_cpp: %s\n", e.what());
    return 1;
  }
  sg_shutdown();
  return 0;
}
