// Host-side product code (no HIP) under AddressSanitizer + UBSan: host_field.hpp (window combine, affine output) and
// host_keccak.hpp (generator derivation hashes).  Built and run by tests/test_host_sanitize.py; GPU sanitizers are not
// available on the pool, so this is the sanitizer coverage the product gets.
#include "../spartan-bn254_amd/csrc/host_field.hpp"
#include "../spartan-bn254_amd/csrc/host_keccak.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using namespace sbn_host;

static Pt gen() { Pt g; g.X = one(); g.Y = dbl(one()); g.ZZ = one(); g.ZZZ = one(); return g; }
static Pt mul_small(const Pt& p, unsigned k) { Pt acc = inf(); for (int b = 31; b >= 0; b--) { acc = pdbl(acc); if ((k >> b) & 1u) acc = padd(acc, p); } return acc; }

int main() {
  int bad = 0;
  // field identities
  Fq a = to_mont(Fq{{123456789, 987654321, 5, 7}}), b = to_mont(Fq{{42, 0, 0, 1}});
  if (!eq(mul(a, inv(a)), one())) { printf("inv\n"); bad++; }
  { Fq x = a; for (int i = 0; i < 100; i++) { x = add(sqr(x), b); if (!eq(inv(x), inv_fermat(x)) || !eq(mul(x, inv(x)), one())) { printf("inv vs fermat\n"); bad++; break; } }
    Fq pm1 = sub(zero(), one()); if (!eq(inv(pm1), inv_fermat(pm1)) || !eq(inv(one()), one()) || !is_zero(inv(zero()))) { printf("inv edge\n"); bad++; } }
  if (!eq(sub(add(a, b), b), a)) { printf("addsub\n"); bad++; }
  if (!eq(from_mont(to_mont(Fq{{9, 8, 7, 6}})), Fq{{9, 8, 7, 6}})) { printf("mont\n"); bad++; }
  // scalar field: the binary-Euclid inverse against Fermat's, and a * 1/a = 1, on edge values and a pseudo-random walk
  {
    using namespace sbn_host::fr;
    El x = {{0x9E3779B97F4A7C15ull, 0xBF58476D1CE4E5B9ull, 0x94D049BB133111EBull, 0x0123456789abcdefull}};
    El pm1 = {{P[0] - 1, P[1], P[2], P[3]}}, two = from_u64(2), big = {{0, 0, 0, 1ull << 60}};
    std::vector<El> xs = {from_u64(1), two, pm1, big, from_u64(0xffffffffffffffffull)};
    for (int i = 0; i < 200; i++) { x = mmul(x, x); x = add(x, from_u64(i + 1)); xs.push_back(x); }
    for (const El& e : xs) {
      const El i1 = inv(e), i2 = inv_fermat(e), pr = mmul(to_m(e), i1);
      if (memcmp(i1.v, i2.v, 32) || !(pr.v[0] == 1 && (pr.v[1] | pr.v[2] | pr.v[3]) == 0) || geq(i1.v)) { printf("fr inv\n"); bad++; break; }
    }
    if (!is_zero(inv(from_u64(0)))) { printf("fr inv 0\n"); bad++; }
  }
  // group law: 2G+3G == 5G, P + (-P) == inf, window combine == scalar multiple
  Pt G = gen();
  uint8_t x1[64], x2[64]; int i1 = 0, i2 = 0;
  to_affine_bytes(padd(mul_small(G, 2), mul_small(G, 3)), x1, &i1); to_affine_bytes(mul_small(G, 5), x2, &i2);
  if (memcmp(x1, x2, 64) || i1 || i2) { printf("2G+3G\n"); bad++; }
  Pt nG = G; nG.Y = sub(zero(), G.Y);
  to_affine_bytes(padd(G, nG), x1, &i1); if (!i1) { printf("cancel\n"); bad++; }
  // sum_w 2^(c w) S_w with S_w = (w+1) G, c = 5, W = 4  ==  (1 + 2*32 + 3*1024 + 4*32768) G
  std::vector<Pt> S; for (unsigned w = 0; w < 4; w++) S.push_back(mul_small(G, w + 1));
  to_affine_bytes(combine_windows(S.data(), 4, 5), x1, &i1); to_affine_bytes(mul_small(G, 1 + 2 * 32 + 3 * 1024 + 4 * 32768), x2, &i2);
  if (memcmp(x1, x2, 64)) { printf("combine\n"); bad++; }
  // keccak: SHA3-256("abc") first bytes 3a985da7..., SHAKE256 incremental squeeze == one-shot
  uint8_t h[32]; sha3_256((const uint8_t*)"abc", 3, h);
  if (h[0] != 0x3a || h[1] != 0x98 || h[31] != 0x32) { printf("sha3\n"); bad++; }
  Keccak k1(136, 0x1f), k2(136, 0x1f); uint8_t o1[500], o2[500];
  k1.absorb((const uint8_t*)"gens", 4); k2.absorb((const uint8_t*)"ge", 2); k2.absorb((const uint8_t*)"ns", 2);
  k1.squeeze(o1, 500); for (int i = 0; i < 500; i += 7) k2.squeeze(o2 + i, (i + 7 <= 500) ? 7 : 500 - i);
  if (memcmp(o1, o2, 500)) { printf("shake\n"); bad++; }
  printf(bad ? "HOST SANITIZE FAIL\n" : "HOST SANITIZE OK\n");
  return bad;
}
