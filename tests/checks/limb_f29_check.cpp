// CPU check of circuits_halo2_amd/csrc/bn254_f29.cuh against big integers (driven by
// tests/checks/limb_f29_check.py): reads "op field a b" lines, a/b as 9 hex limbs each, prints the result limbs.
#include <cstdio>
#include <cstring>
#include <string>
#include <iostream>
#include <sstream>
#include "../../circuits_halo2_amd/csrc/bn254_f29.cuh"
using namespace sg;
template <class P> void run(const std::string& op, const f29& a, const f29& b) {
  f29 r = f29_zero();
  if (op == "mul") r = f29_mul<P>(a, b);
  else if (op == "sqr") r = f29_sqr<P>(a);
  else if (op == "mul2") r = f29_mul2<P>(a, b, b, a);
  else if (op == "muladd") r = f29_mul_add<P>(a, b, b);        // a b 2^-261 + b
  else if (op == "dot3") { const f29 x[3] = {a, b, a}, y[3] = {b, b, a}; r = f29_dot<P, 3>(x, y); }   // (ab + bb + aa) 2^-261
  else if (op == "dot5") { const f29 x[5] = {a, b, a, a, b}, y[5] = {b, b, a, b, a}; r = f29_dot<P, 5>(x, y); }   // (3ab + bb + aa) 2^-261
  else if (op == "x3nc") r = f29_carry(f29_sub_nc<P, 1>(f29_sub_nc<P, 0>(a, b), f29_add_nc(b, b)));   // a + 2p - b + 4p - 2b, one carry step
  else if (op == "tnc") r = f29_carry_top(f29_sub_nc<P, 2>(a, b));                                    // a + 8p - b, top limb carried only
  else if (op == "add") r = f29_add(a, b);
  else if (op == "sub2") r = f29_sub<P, 0>(a, b);
  else if (op == "sub8") r = f29_sub<P, 2>(a, b);
  else if (op == "sub64") r = f29_sub<P, 5>(a, b);
  else if (op == "canon") r = f29_canonical<P>(a);
  else if (op == "step") r = f29_mont_step<P>(a);
  else if (op == "redsmall") r = f29_reduce_small<P>(a);
  else if (op == "inv") r = f29_canonical<P>(f29_inv_safegcd<P>(a));
  else if (op == "iszero") { r.l[0] = f29_is_zero_mod_p<P>(a); }
  else if (op == "norm") r = f29_normalize(a);
  else if (op == "from0" || op == "from5") {
    uint32_t w[8]; for (int i = 0; i < 8; i++) w[i] = a.l[i];
    r = op == "from0" ? f29_from_words<0>(w) : f29_from_words<5>(w);
  } else if (op == "towords") { uint32_t w[8]; f29_to_words(a, w); for (int i = 0; i < 8; i++) r.l[i] = w[i]; r.l[8] = 0; }
  for (int i = 0; i < 9; i++) printf("%x%c", r.l[i], i == 8 ? '\n' : ' ');
}
int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream ss(line);
    std::string op, fld; ss >> op >> fld;
    f29 a, b;
    for (int i = 0; i < 9; i++) ss >> std::hex >> a.l[i];
    for (int i = 0; i < 9; i++) ss >> std::hex >> b.l[i];
    if (fld == "q") run<Fq29>(op, a, b); else run<Fr29>(op, a, b);
  }
}
