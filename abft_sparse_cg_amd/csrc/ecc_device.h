// ecc_device.h -- the reference's per-element software ECC, device side (gfx950).
//
// Codeword layout (reference CSR/ecc.h:7-11, COO/ecc.h:11-16), as 32-bit words:
//   CSR  96 bit: w0,w1 = IEEE double value (lo,hi), w2 = column
//   COO 128 bit: w0 = column, w1 = row, w2,w3 = value (lo,hi)
// The ECC byte is bits 24-31 of the column word: bit 32-p holds Hamming check
// bit p (p = 1..7), bit 24 the overall parity (sec8/secded); SED uses bit 31.
//
// The 7 x {3,4} masks are not typed in: they are generated at compile time
// from the construction rule the reference documents (COO/ecc.h:136-170) and
// end up as instruction immediates.  tests/ check them against the reference's
// constants through the golden fixtures.
#pragma once
#include <stdint.h>

#define ABFT_COLMASK 0x00FFFFFFu
#ifndef ABFT_CFG_BITOP3
#define ABFT_CFG_BITOP3 1  // check-bit folds with v_bitop3_b32 (A/B: -DABFT_CFG_BITOP3=0)
#endif

enum { FMT_CSR = 0, FMT_COO = 1 };
enum { MODE_NONE = 0, MODE_CONSTRAINTS = 1, MODE_SED = 2, MODE_SEC7 = 3, MODE_SEC8 = 4, MODE_SECDED = 5 };

template <int FMT> struct EccLayout {
  static constexpr int NW = FMT == FMT_CSR ? 3 : 4;     // words per element
  static constexpr int EW = FMT == FMT_CSR ? 2 : 0;     // word that carries the ECC byte
  static constexpr int NBITS = 32 * NW;
};

constexpr bool ecc_is_pow2(uint32_t x) { return x && !(x & (x - 1)); }

// mask of check bit p (1..7) over word w: walk all bits in word order handing
// Hamming positions 3,5,6,7,9,... to every bit except the ECC byte.
constexpr uint32_t ecc_mask(int fmt, int p, int w) {
  const int nw = fmt == FMT_CSR ? 3 : 4, ew = fmt == FMT_CSR ? 2 : 0;
  uint32_t x = 3, out = 0;
  for (int ww = 0; ww < nw; ww++) {
    for (int b = 0; b < 32; b++) {
      if (ecc_is_pow2(x)) x++;
      uint32_t in = 0;
      if (ww == ew && b >= 24) {
        in = (32 - b == p);
      } else {
        in = (x >> (p - 1)) & 1u;
        x++;
      }
      if (ww == w && in) out |= 1u << b;
    }
  }
  return out;
}

__device__ __forceinline__ uint32_t popc(uint32_t x) { return (uint32_t)__builtin_popcount(x); }

// XOR-fold of the words under the masks of check bit P; its popcount's low bit
// is that check bit's parity (reference CSR/ecc.h:59-78).
template <int FMT, int P>
__device__ __forceinline__ uint32_t ecc_fold(const uint32_t *w) {
  constexpr uint32_t m0 = ecc_mask(FMT, P, 0), m1 = ecc_mask(FMT, P, 1), m2 = ecc_mask(FMT, P, 2);
#if ABFT_CFG_BITOP3 && defined(__HIP_DEVICE_COMPILE__)
  // gfx950's three-input boolean op: acc ^ (word & mask) is ONE instruction (truth table 0x6c over
  // {word, acc, mask}), so a check bit costs and + 2 x bitop3 + popcount instead of 3 and + 2 xor + popcount
  uint32_t acc = w[0] & m0;
  acc = __builtin_amdgcn_bitop3_b32(w[1], acc, m1, 0x6c);
  acc = __builtin_amdgcn_bitop3_b32(w[2], acc, m2, 0x6c);
  if (FMT == FMT_COO) {
    constexpr uint32_t m3 = ecc_mask(FMT_COO, P, 3);
    acc = __builtin_amdgcn_bitop3_b32(w[3], acc, m3, 0x6c);
  }
  return acc;
#else
  uint32_t acc = (w[0] & m0) ^ (w[1] & m1) ^ (w[2] & m2);
  if (FMT == FMT_COO) {
    constexpr uint32_t m3 = ecc_mask(FMT_COO, P, 3);
    acc ^= w[3] & m3;
  }
  return acc;
#endif
}

// Low bit set iff any of the 7 Hamming checks fails -- the hot-path test; the
// individual bits are only assembled on the cold path.
template <int FMT>
__device__ __forceinline__ uint32_t ecc_any_check(const uint32_t *w) {
  uint32_t a = popc(ecc_fold<FMT, 1>(w)) | popc(ecc_fold<FMT, 2>(w)) | popc(ecc_fold<FMT, 3>(w));
  uint32_t b = popc(ecc_fold<FMT, 4>(w)) | popc(ecc_fold<FMT, 5>(w)) | popc(ecc_fold<FMT, 6>(w));
  return (a | b | popc(ecc_fold<FMT, 7>(w))) & 1u;
}

// Hamming position of the failing bit: bit p-1 = parity of check p.
template <int FMT>
__device__ __forceinline__ uint32_t ecc_hamming(const uint32_t *w) {
  return ((popc(ecc_fold<FMT, 1>(w)) & 1u) << 0) | ((popc(ecc_fold<FMT, 2>(w)) & 1u) << 1) |
         ((popc(ecc_fold<FMT, 3>(w)) & 1u) << 2) | ((popc(ecc_fold<FMT, 4>(w)) & 1u) << 3) |
         ((popc(ecc_fold<FMT, 5>(w)) & 1u) << 4) | ((popc(ecc_fold<FMT, 6>(w)) & 1u) << 5) |
         ((popc(ecc_fold<FMT, 7>(w)) & 1u) << 6);
}

// The reference's "syndrome" word: check bit p at bit 32-p (CSR/ecc.h:51-81).
template <int FMT>
__device__ __forceinline__ uint32_t ecc_syndrome_word(const uint32_t *w) {
  uint32_t h = ecc_hamming<FMT>(w);
  return __builtin_bitreverse32(h);  // bit p-1 -> bit 32-p
}

template <int FMT>
__device__ __forceinline__ uint32_t ecc_parity(const uint32_t *w) {
  uint32_t acc = w[0] ^ w[1] ^ w[2];
  if (FMT == FMT_COO) acc ^= w[3];
  return popc(acc) & 1u;  // reference CSR/ecc.h:89-93, COO/ecc.h:109-113
}

// Hamming position -> bit index inside the element (reference CSR/ecc.h:97-113,
// COO/ecc.h:117-134): powers of two are check bits (bit 32-p of the ECC word),
// anything else is data bit h - floor(log2 h) - 2; COO data bits above 23 sit
// behind the ECC byte.
template <int FMT>
__device__ __forceinline__ uint32_t ecc_position_to_bit(uint32_t h) {
  uint32_t lg = 31u - (uint32_t)__builtin_clz(h);
  if (ecc_is_pow2(h)) return (31u - lg) + (FMT == FMT_CSR ? 64u : 0u);
  uint32_t d = h - lg - 2u;
  if (FMT == FMT_COO && d >= 24u) d += 8u;
  return d;
}

// generate_ecc_bits for every mode (reference CSR/CPUContext.cpp:209-212,
// 247-250, 291-295, 347-351; COO/CPUContext.cpp:196-199, 234-237, 277-281,
// 330-334).  `w` holds the clean element; the ECC word is updated in place.
template <int FMT>
__device__ __forceinline__ void ecc_encode(int mode, uint32_t *w) {
  constexpr int EW = EccLayout<FMT>::EW;
  if (mode == MODE_SED) {
    w[EW] |= ecc_parity<FMT>(w) << 31;
  } else if (mode >= MODE_SEC7) {
    w[EW] |= ecc_syndrome_word<FMT>(w);
    if (mode != MODE_SEC7) w[EW] |= ecc_parity<FMT>(w) << 24;
  }
}
