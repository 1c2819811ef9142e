// Host-only dump of include/summa_circuit.hpp for the byte-for-byte comparison with the Python twin
// (tests/test_host_logic.py): circuit_dump <k> <levels> <n_currencies> <n_bytes> <out file>
//   build: g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/circuit_dump.cpp -o tools/circuit_dump -L/opt/rocm/lib -lamdhip64
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "summa_circuit.hpp"

using namespace summa::circuit;

static void put_u32(std::ofstream& f, uint32_t v) { f.write(reinterpret_cast<const char*>(&v), 4); }
static void put_graph(std::ofstream& f, const summa::prover::Graph& g) {   // the layout of prover.export_bundle's graph_bytes
  put_u32(f, (uint32_t)(g.constants.size() / 32));
  f.write(reinterpret_cast<const char*>(g.constants.data()), (std::streamsize)g.constants.size());
  put_u32(f, (uint32_t)g.rotations.size());
  f.write(reinterpret_cast<const char*>(g.rotations.data()), (std::streamsize)(4 * g.rotations.size()));
  put_u32(f, (uint32_t)g.calculations.size());
  for (auto& c : g.calculations)
    for (uint32_t w : {c.op, c.a.kind, c.a.index, c.a.rotation, c.b.kind, c.b.index, c.b.rotation, c.parts_offset, c.parts_len}) put_u32(f, w);
  put_u32(f, (uint32_t)g.parts.size());
  for (auto& p : g.parts)
    for (uint32_t w : {p.kind, p.index, p.rotation}) put_u32(f, w);
}

int main(int argc, char** argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s <k> <levels> <n_currencies> <n_bytes> <out>\n", argv[0]);
    return 2;
  }
  try {
    const uint32_t k = (uint32_t)std::atoi(argv[1]), levels = (uint32_t)std::atoi(argv[2]), nc = (uint32_t)std::atoi(argv[3]),
                   nb = (uint32_t)std::atoi(argv[4]);
    FloorPlan fp(k, levels, nc, nb);
    // omega of the 2^k domain: ROOT_OF_UNITY^(2^(28 - k))
    static const uint64_t ROOT_C[4] = {0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL, 0x98865ea93dd31f74ULL, 0x03ddb9f5166d18b7ULL};
    summa::prover::Fr omega = summa::prover::Fr::from_canonical_limbs(ROOT_C);
    for (uint32_t i = k; i < 28; i++) omega = omega * omega;
    const auto sigma = fp.sigma(omega);
    std::ofstream f(argv[5], std::ios::binary);
    put_u32(f, fp.n_items);
    put_u32(f, fp.n_absorbs);
    put_u32(f, fp.rows_used);
    put_u32(f, (uint32_t)fp.instance_symbols.size());
    for (uint32_t s : fp.instance_symbols) put_u32(f, s);
    f.write(reinterpret_cast<const char*>(fp.program.data()), (std::streamsize)(4 * fp.program.size()));
    for (auto& col : fp.fixed) f.write(reinterpret_cast<const char*>(col.data()), (std::streamsize)(32 * col.size()));
    for (auto& col : sigma) f.write(reinterpret_cast<const char*>(col.data()), (std::streamsize)(32 * col.size()));
    put_graph(f, gate_graph(nc));
    put_graph(f, lookup_input_graph());
    {   // the gate program's challenge list
      const auto groups = gate_challenge_exponents(nc);
      put_u32(f, (uint32_t)groups.size());
      for (auto& g : groups) {
        put_u32(f, (uint32_t)g.size());
        for (uint32_t e : g) put_u32(f, e);
      }
    }
    // the verifying-key digest of 17 stand-in commitments (Montgomery bytes = 1, 2, ..), compared with the Python formula
    std::vector<std::array<uint8_t, 64>> comms(NUM_FIXED + NUM_PERM);
    for (size_t i = 0; i < comms.size(); i++) {
      comms[i].fill(0);
      comms[i][0] = (uint8_t)(2 * i + 1);
      comms[i][32] = (uint8_t)(2 * i + 2);
    }
    const auto digest = verifying_key_digest(k, nc, comms);
    f.write(reinterpret_cast<const char*>(digest.data()), 32);
    const auto u = fr_from_username("dxGaEAii");
    const auto b = fr_from_decimal("11888");
    f.write(reinterpret_cast<const char*>(u.l), 32);
    f.write(reinterpret_cast<const char*>(b.l), 32);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "circuit_dump: %s\n", e.what());
    return 1;
  }
  return 0;
}
