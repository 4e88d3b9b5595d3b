// host_keccak.hpp — SHA3-256 and SHAKE256 (FIPS 202) for the setup-time generator derivation
// (reference src/commitments.rs:31-62 uses sha3::Shake256, src/group.rs:110-131 uses sha3::Sha3_256).
// Host-only, product code; the permutation is written lane-wise (x + 5y indexing) with rho/pi tables
// computed at start-up from their defining recurrences rather than tabulated.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace sbn_host {

class Keccak {
 public:
  // rate in bytes, domain-separation suffix (0x06 SHA3, 0x1f SHAKE)
  Keccak(size_t rate, uint8_t suffix) : rate_(rate), suffix_(suffix), pos_(0), squeezing_(false) {
    memset(st_, 0, sizeof st_);
    init_tables();
  }
  void absorb(const uint8_t* in, size_t len) {
    for (size_t i = 0; i < len; i++) {
      st_[pos_ >> 3] ^= (uint64_t)in[i] << (8 * (pos_ & 7));
      if (++pos_ == rate_) { permute(); pos_ = 0; }
    }
  }
  void squeeze(uint8_t* out, size_t len) {
    if (!squeezing_) {
      st_[pos_ >> 3] ^= (uint64_t)suffix_ << (8 * (pos_ & 7));
      st_[(rate_ - 1) >> 3] ^= (uint64_t)0x80 << (8 * ((rate_ - 1) & 7));
      permute(); pos_ = 0; squeezing_ = true;
    }
    for (size_t i = 0; i < len; i++) {
      if (pos_ == rate_) { permute(); pos_ = 0; }
      out[i] = (uint8_t)(st_[pos_ >> 3] >> (8 * (pos_ & 7)));
      pos_++;
    }
  }

 private:
  uint64_t st_[25];
  size_t rate_; uint8_t suffix_; size_t pos_; bool squeezing_;
  int rho_[25]; int pi_[25]; uint64_t rc_[24];

  static uint64_t rol(uint64_t x, int n) { n &= 63; return n ? (x << n) | (x >> (64 - n)) : x; }
  void init_tables() {
    // rho offsets and pi permutation from the (x,y) -> (y, 2x+3y) walk
    for (int i = 0; i < 25; i++) { rho_[i] = 0; pi_[i] = i; }
    int x = 1, y = 0;
    for (int t = 0; t < 24; t++) {
      rho_[x + 5 * y] = ((t + 1) * (t + 2) / 2) % 64;
      int nx = y, ny = (2 * x + 3 * y) % 5; x = nx; y = ny;
    }
    for (int xx = 0; xx < 5; xx++) for (int yy = 0; yy < 5; yy++) pi_[yy + 5 * ((2 * xx + 3 * yy) % 5)] = xx + 5 * yy;  // dest <- src
    // round constants from the degree-8 LFSR
    uint8_t lfsr = 1;
    for (int r = 0; r < 24; r++) {
      uint64_t c = 0;
      for (int j = 0; j < 7; j++) {
        if (lfsr & 1) c |= (uint64_t)1 << ((1 << j) - 1);
        lfsr = (uint8_t)((lfsr << 1) ^ ((lfsr & 0x80) ? 0x71 : 0));
      }
      rc_[r] = c;
    }
  }
  void permute() {
    for (int r = 0; r < 24; r++) {
      uint64_t C[5], D[5], B[25];
      for (int x = 0; x < 5; x++) C[x] = st_[x] ^ st_[x + 5] ^ st_[x + 10] ^ st_[x + 15] ^ st_[x + 20];
      for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rol(C[(x + 1) % 5], 1);
      for (int i = 0; i < 25; i++) st_[i] ^= D[i % 5];
      for (int i = 0; i < 25; i++) B[i] = rol(st_[pi_[i]], rho_[pi_[i]]);
      for (int y = 0; y < 5; y++) for (int x = 0; x < 5; x++) st_[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
      st_[0] ^= rc_[r];
    }
  }
};

static inline void sha3_256(const uint8_t* in, size_t len, uint8_t out[32]) { Keccak k(136, 0x06); k.absorb(in, len); k.squeeze(out, 32); }

}  // namespace sbn_host
