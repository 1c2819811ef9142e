// CPU check of bn254_curve29.cuh (driven by tests/checks/limb_curve29_check.py).
// input: lines "madd <16 hex words> <neg>" | "reset" | "swap" (acc <-> other) | "addother" | "double" | "dump"
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
#include "../../circuits_halo2_amd/csrc/bn254_curve29.cuh"
using namespace sg;
int main() {
  xyzz29 acc = xyzz29_identity(), other = xyzz29_identity();
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream ss(line);
    std::string op; ss >> op;
    if (op == "madd") {
      uint32_t w[16]; for (int i = 0; i < 16; i++) ss >> std::hex >> w[i];
      int neg; ss >> neg;
      affine29 q = affine29_from_words(w);
      if (neg) affine29_negate(q);
      xyzz29_madd(acc, q);
    } else if (op == "reset") acc = xyzz29_identity();
    else if (op == "swap") { xyzz29 t = acc; acc = other; other = t; }
    else if (op == "addother") xyzz29_add(acc, other);
    else if (op == "double") acc = xyzz29_double(acc);
    else if (op == "dump") {
      uint32_t w[32]; xyzz29_to_words(acc, w);
      for (int i = 0; i < 32; i++) printf("%08x%c", w[i], i == 31 ? '\n' : ' ');
      // invariants
      auto top = [](const f29& a) { return a.l[8]; };
      fprintf(stderr, "top limbs %x %x %x %x\n", top(acc.x), top(acc.y), top(acc.zz), top(acc.zzz));
    }
  }
}
