// swt_dedup.hip -- word-level dedup inside one encode call (see swt_dedup.h), the kernels and the two host halves.
#include <cstdlib>

#include "swt_dedup.h"

namespace swt {

constexpr uint8_t kClsWs = SWT_CLS_BERT_WS, kClsPunct = SWT_CLS_BERT_PUNCT, kClsPySpace = SWT_CLS_PY_SPACE;
constexpr int kClsLds = 1024;  // code points whose class is served from LDS (64 lanes x 16 B)
#ifndef SWT_DTILE
#define SWT_DTILE 1024
#endif
#ifndef SWT_DCAP
#define SWT_DCAP 1536
#endif
constexpr int kDTile = SWT_DTILE;  // (build parameters: tools/gpu_dd_sweep.sh)
constexpr int kDCap = SWT_DCAP;  // staged bytes per chunk: a 1-KiB tile's span plus the end of its last sentence fits; with 2048 the LDS
                             // (8.1 KB) left 4.9 waves per SIMD and the waves waited 57 % of their cycles -- 1536 (6.2 KB, 6.4
                             // waves per SIMD) made wordref 16 % faster on the 141 MB FastWP batch
constexpr int kDBlocks = kDCap / 64;
#ifndef SWT_WORDREF_WAVES
#define SWT_WORDREF_WAVES 7  // waves per SIMD the register allocation of wordref_kernel aims at (5,600 bytes of LDS allow 7)
#endif
constexpr unsigned long long kDOffMask = (1ull << 40) - 1ull;
constexpr uint32_t kRefSlot = 0x80000000u;

typedef uint64_t u64u __attribute__((aligned(1)));  // unaligned 8-byte access (one global_load / ds_read on gfx950)

struct DedupTab {
  unsigned long long *slot;      // epoch:8 | tag:8 | byte length:8 | representative offset:40
  unsigned long long *rec;       // per slot: the inserter leaves the byte length; the unique-word encode replaces it by
                                 // token count:32 | place of the tokens in its scratch:32
  uint64_t n_bytes;              // size of the text (wide compares stay inside it)
  uint32_t diag;                 // diagnostics: bit 0 tallies CAS successes / failures behind `overflow`
  uint32_t bits;
  uint32_t epoch;
  // The table is kept SMALL (a few MB: its hot lines stay in the 4-MiB L2 of every XCD) and is not sized for the worst
  // case.  A word that finds neither itself nor a free slot within kDdMaxProbes steps -- and any word of 255+ bytes --
  // is simply not deduplicated: it gets a record slot of its own behind the table, numbered by its position in the
  // text, and is encoded like a new word.  Correct whatever the load; only repeated words that overflow cost extra.
  uint32_t ovf_shift;            // position >> ovf_shift is unique per tabled word (1: two bytes at least; 0: one)
  // New words are NOT numbered with a global counter (one hot address serialises every returning atomic of the chip:
  // that alone cost 170 us of a 230 us kernel).  The tile that inserted a word lists it; a scan over the tiles'
  // (count, bytes) numbers the words afterwards (bpe_ureg_kernel).
  unsigned long long *newlist;   // slot:32 | position:32 (the inserter leaves the byte length in rec[slot]); a tile's entries start at [span_base >> 1]
  unsigned long long *tile_new;  // per tile: new words:32 | their bytes:32 (a dedup call holds at most 2^30 bytes)
  unsigned int *overflow;
};

constexpr uint32_t kDdMaxProbes = 24;
__device__ __forceinline__ uint32_t dd_own_slot(const DedupTab &D, uint64_t gpos) { return (1u << D.bits) + (uint32_t)(gpos >> D.ovf_shift); }

// the same function as the wide form in dd_find_or_insert_lds, byte by byte (words in global memory)
__device__ __forceinline__ unsigned long long dd_pack8(const uint8_t *p, uint32_t n) {
  unsigned long long w = 0;
  for (uint32_t i = 0; i < n && i < 8; i++) w |= (unsigned long long)p[i] << (8 * i);
  return w;
}
__device__ __forceinline__ unsigned long long dd_hash(const uint8_t *p, uint32_t n) {
  const unsigned long long w0 = dd_pack8(p, n), w1 = n > 8 ? dd_pack8(p + 8, n - 8) : 0ull;
  unsigned long long h = (w0 ^ 0x9E3779B97F4A7C15ull) * 0xff51afd7ed558ccdull;
  h ^= h >> 32;
  h = (h ^ w1 ^ ((unsigned long long)n << 56)) * 0xc4ceb9fe1a85ec53ull;
  for (uint32_t i = 16; i < n; i += 8) {
    h ^= h >> 29;
    h = (h ^ dd_pack8(p + i, n - i)) * 0x9E3779B97F4A7C15ull;
  }
  h ^= h >> 29; h *= 0x94d049bb133111ebull; h ^= h >> 32;
  return h;
}

__device__ uint32_t dd_find_or_insert(const DedupTab &D, const uint8_t *__restrict__ text, const uint8_t *mine, uint32_t len,
                                      uint64_t gpos, bool &is_new) {
  is_new = false;
  if (len >= 255u) { is_new = true; return dd_own_slot(D, gpos); }
  const unsigned long long h = dd_hash(mine, len);
  const uint32_t mask = (1u << D.bits) - 1u;
  const uint32_t lf = len;
  uint32_t probes = 0;
  const unsigned long long head = ((unsigned long long)D.epoch << 56) | (((h >> 40) & 0xFFull) << 48) | ((unsigned long long)lf << 40);
  uint32_t idx = (uint32_t)h & mask;
  for (;;) {
    unsigned long long v = __hip_atomic_load(&D.slot[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(v >> 56) != D.epoch) {  // free in this call (never used, or left over from an earlier call)
      const unsigned long long prev = atomicCAS(&D.slot[idx], v, head | gpos);
      if (prev == v) {
        is_new = true;
        return idx;
      }
      v = prev;
      if ((uint32_t)(v >> 56) != D.epoch) continue;  // changed to another stale value?  look again
    }
    if ((v & ~kDOffMask) == head) {
      const uint8_t *rep = text + (v & kDOffMask);
      bool same = true;
      for (uint32_t i = 0; i < len; i++)
        if (rep[i] != mine[i]) { same = false; break; }
      if (same) return idx;
    }
    if (++probes >= kDdMaxProbes) { is_new = true; return dd_own_slot(D, gpos); }
    idx = (idx + 1) & mask;
  }
}

// The common case: the word sits in LDS.  Its first 16 bytes are taken with two unaligned 8-byte reads, hashed as two
// words, and compared with the representative's bytes by two unaligned global loads (one L2 trip), not byte by byte.
__device__ __forceinline__ uint32_t dd_find_or_insert_lds(const DedupTab &D, const uint8_t *__restrict__ text, const uint8_t *mine,
                                                          uint32_t len, uint64_t gpos, bool &is_new) {
  is_new = false;
  if (len >= 255u) { is_new = true; return dd_own_slot(D, gpos); }
  unsigned long long w0 = *reinterpret_cast<const u64u *>(mine), w1 = *reinterpret_cast<const u64u *>(mine + 8);
  if (len < 8) { w0 &= (1ull << (8 * len)) - 1ull; w1 = 0; }
  else if (len < 16) w1 &= (1ull << (8 * (len - 8))) - 1ull;
  unsigned long long h = (w0 ^ 0x9E3779B97F4A7C15ull) * 0xff51afd7ed558ccdull;
  h ^= h >> 32;
  h = (h ^ w1 ^ ((unsigned long long)len << 56)) * 0xc4ceb9fe1a85ec53ull;
  for (uint32_t i = 16; i < len; i += 8) {
    unsigned long long wk = *reinterpret_cast<const u64u *>(mine + i);  // txt[] has 16 bytes of slack behind the chunk
    if (len - i < 8) wk &= (1ull << (8 * (len - i))) - 1ull;
    h ^= h >> 29;
    h = (h ^ wk) * 0x9E3779B97F4A7C15ull;
  }
  h ^= h >> 29; h *= 0x94d049bb133111ebull; h ^= h >> 32;
  const uint32_t mask = (1u << D.bits) - 1u;
  const uint32_t lf = len;
  uint32_t probes = 0;
  const unsigned long long head = ((unsigned long long)D.epoch << 56) | (((h >> 40) & 0xFFull) << 48) | ((unsigned long long)lf << 40);
  uint32_t idx = (uint32_t)h & mask;
  if (!(D.diag & 2u)) {
    // The common case first, through the caches: a word of running text has usually been tabled long ago, and a slot of this
    // call is written once and never changes, so a plain (L2-cached) load can only show it complete or not at all -- a hit
    // found here is exact, anything else (the slot looks free, or stale from an earlier call, or holds another word twice in a
    // row) is settled by the coherent path below, from the home slot.  The device-scope loads of that path fetch a 64-byte
    // line from the fabric per lookup: 1.4 GB of the FastWP call's 2.8 GB (profiles/r03_wp_encode_FETCH_SIZE_per_kernel.csv).
    uint32_t j = idx;
    for (int p = 0; p < 2; p++) {
      const unsigned long long v = D.slot[j];
      if ((v & ~kDOffMask) == head) {
        const uint64_t ro = v & kDOffMask;
        if (ro + 16 <= D.n_bytes) {
          const uint8_t *rep = text + ro;
          unsigned long long r0 = *reinterpret_cast<const u64u *>(rep), r1 = *reinterpret_cast<const u64u *>(rep + 8);
          if (len < 8) { r0 &= (1ull << (8 * len)) - 1ull; r1 = 0; }
          else if (len < 16) r1 &= (1ull << (8 * (len - 8))) - 1ull;
          bool same = r0 == w0 && r1 == w1;
          for (uint32_t i = 16; i < len && same; i++) same = rep[i] == mine[i];
          if (same) return j;
        } else {
          break;
        }
      } else if ((uint32_t)(v >> 56) != D.epoch) {
        break;
      }
      j = (j + 1) & mask;
    }
  }
  for (;;) {
    unsigned long long v = __hip_atomic_load(&D.slot[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(v >> 56) != D.epoch) {  // free in this call (never used, or left over from an earlier call)
      const unsigned long long prev = atomicCAS(&D.slot[idx], v, head | gpos);
      if (D.diag & 1u) atomicAdd(&D.overflow[prev == v ? 1 : 2], 1u);
      if (prev == v) {
        is_new = true;  // listed by the caller
        return idx;
      }
      v = prev;
      if ((uint32_t)(v >> 56) != D.epoch) continue;
    }
    if ((v & ~kDOffMask) == head) {
      const uint64_t ro = v & kDOffMask;
      const uint8_t *rep = text + ro;
      bool same;
      if (ro + 16 <= D.n_bytes) {
        unsigned long long r0 = *reinterpret_cast<const u64u *>(rep), r1 = *reinterpret_cast<const u64u *>(rep + 8);
        if (len < 8) { r0 &= (1ull << (8 * len)) - 1ull; r1 = 0; }
        else if (len < 16) r1 &= (1ull << (8 * (len - 8))) - 1ull;
        same = r0 == w0 && r1 == w1;
        for (uint32_t i = 16; i < len && same; i++) same = rep[i] == mine[i];
      } else {
        same = true;
        for (uint32_t i = 0; i < len; i++)
          if (rep[i] != mine[i]) { same = false; break; }
      }
      if (same) return idx;
    }
    if (++probes >= kDdMaxProbes) { is_new = true; return dd_own_slot(D, gpos); }
    idx = (idx + 1) & mask;
  }
}

// ---- the split, sixteen bytes per lane (FastWP chunks) -------------------------------------------------------------------------
// wordref_kernel<kDedupWp> issues 254 M vector and 199 M scalar instructions per 141 MB call -- 81 % of the vector pipe's time
// (profiles/r03_sq_counters.txt) -- and two thirds of them are the split loop: one byte per lane, 64 bytes per trip, five
// ballots and ~70 scalar mask operations per trip.  A chunk between str.isspace characters needs none of the pre-tokenizer's
// classes, so the lane form takes SIXTEEN bytes per lane (one ds_read_b128), finds the ASCII whitespace of its four dwords with
// carry-free byte arithmetic, looks only its (rare) non-ASCII lead bytes up in the class table, and does the mask algebra of the
// byte-lane loop on its own 16 bits -- the whitespace smear and the "byte before me" across the lane boundary from the lane
// below (one shuffle) -- a whole KiB per trip.  It leaves EXACTLY what the byte-lane loop leaves (endm / wst / nwb per 64-byte
// block, the list of word starts, the cut): the rest of the kernel does not know the difference, and every dedup test compares
// the two forms' outputs through the oracle.  SWT_DD_OLD_SPLIT=1 runs the byte-lane loop instead (comparison).
__device__ __forceinline__ uint32_t dd_nibble(uint32_t flags_bit7) {  // bit 7 of each of the 4 bytes -> 4 bits, byte 0 lowest
  return ((((flags_bit7 >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
}

struct WordrefLds {
  __attribute__((aligned(16))) uint8_t txt[kDCap + 16];
  uint16_t wl[kDCap];
  unsigned long long sbits[kDBlocks + 1];
  unsigned long long endm[kDBlocks + 1];
  uint32_t cls2[kClsLds / 16];  // classes of U+0000..U+03FF, two bits each: bit 0 = ends a word (the mode's whitespace), bit 1 = punctuation (BPE words)
  unsigned long long wst[kDBlocks + 1];  // word starts per 64-byte block, and how many came before the block
  uint32_t nwb[kDBlocks + 1];
  uint64_t giant_end;
  uint32_t giant_new, giant_word;
};

// Mode kDedupBpe: words end at BertPreTokenizer whitespace, every punctuation code point is a word of its own, and a
// one-symbol word is recorded as its own token.  Mode kDedupWp: words are the chunks between str.isspace characters and
// every one of them goes through the table.
// Lanes16: the split takes sixteen bytes per lane (an instance of its own, so that neither form pays the other's registers)
template <int Mode, bool Lanes16 = false>
__global__ __launch_bounds__(64, SWT_WORDREF_WAVES) void wordref_kernel(const uint8_t *__restrict__ text, uint64_t n_bytes,
                                                     const uint64_t *__restrict__ sent_off, const uint64_t *__restrict__ plan,
                                                     const uint8_t *__restrict__ cls_tab, DedupTab D, uint32_t *__restrict__ wref,
                                                     uint32_t *__restrict__ sent_word, uint32_t *__restrict__ tile_words,
                                                     uint32_t dbg) {
  constexpr uint8_t kWsBit = Mode == kDedupWp ? kClsPySpace : kClsWs;
  constexpr bool kPunctSplits = Mode == kDedupBpe;
  __shared__ WordrefLds L;
  const int lane = threadIdx.x;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const uint64_t t = blockIdx.x;
  const uint64_t s_lo = plan[t], s_hi = plan[t + 1];
  if (s_lo == s_hi) {
    if (lane == 0) { D.tile_new[t] = 0ull; tile_words[t] = 0u; }
    return;
  }
  {
    // sixteen code points per lane, two bits each (the 1-KiB byte table cost a wave per SIMD: 6,368 -> 5,600 bytes of LDS)
    const uint4 v = reinterpret_cast<const uint4 *>(cls_tab)[lane];
    auto pk = [](uint32_t d) {
      uint32_t ws = d & (0x01010101u * kWsBit), pn = kPunctSplits ? (d & (0x01010101u * kClsPunct)) : 0u;
      ws = ws / kWsBit;                    // bit 0 of each byte
      pn = (pn / kClsPunct) << 1;          // bit 1 of each byte
      uint32_t x = ws | pn;                // two bits in each byte
      x = (x | (x >> 6)) & 0x000F000Fu;
      return (x | (x >> 12)) & 0xFFu;
    };
    L.cls2[lane] = pk(v.x) | (pk(v.y) << 8) | (pk(v.z) << 16) | (pk(v.w) << 24);
  }
  const uint64_t span_base = sent_off[s_lo], span_end = sent_off[s_hi];
  uint64_t s_next = s_lo;
  uint64_t cb = span_base;
  // room for one entry per tabled word of the span: such a word has two bytes (BPE), or one byte (WP: "a" next to a sentence end)
  unsigned long long *const my_list = D.newlist + (Mode == kDedupWp ? span_base : (span_base >> 1));
  uint32_t *const my_rec = wref + span_base;  // the tile's word records, dense, in text order (at most one per byte)
  uint32_t n_new = 0, words_done = 0;  // wave-uniform
  unsigned long long my_bytes = 0;     // per lane, summed at the end
  for (;;) {
    const uint64_t abase = cb & ~15ull;
    const uint32_t off0 = (uint32_t)(cb - abase);
    const uint64_t avail = span_end - abase;
    const bool last = avail <= (uint64_t)kDCap;
    const uint32_t staged = last ? (uint32_t)avail : (uint32_t)kDCap;
    const uint32_t nblk = (staged + 63) >> 6;
    for (uint32_t c = lane * 16; c < staged; c += 64 * 16) {
      const uint64_t g = abase + c;
      if (g + 16 <= n_bytes && ((reinterpret_cast<uintptr_t>(text + g) & 15) == 0)) {
        *reinterpret_cast<uint4 *>(&L.txt[c]) = *reinterpret_cast<const uint4 *>(text + g);
      } else {
        for (int i = 0; i < 16; i++) L.txt[c + i] = (g + i < n_bytes) ? text[g + i] : (uint8_t)' ';
      }
    }
    if (lane <= kDBlocks) L.sbits[lane] = 0ull;
    __syncthreads();
    for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
      const uint64_t o = sent_off[s];
      if (o >= abase + staged) break;
      if (o >= cb) atomicOr(&L.sbits[(o - abase) >> 6], 1ull << ((o - abase) & 63));
    }
    __syncthreads();
    uint32_t nw = 0;
    bool prev_wb = true;
    int cut = -1;
    if (Lanes16) {
      // carried from trip to trip: the last lane's raw whitespace and continuation bits (the lane below lane 0)
      uint32_t carry_wb = 0x8000u, carry_ct = 0u;  // chunk start: "the byte before belongs to whitespace"
      const uint32_t n_groups = nblk * 4;          // 16-byte groups that lie in the blocks the kernel looks at
      for (uint32_t g0 = 0; g0 < n_groups; g0 += 64) {
        const uint32_t g = g0 + (uint32_t)lane, p0 = g * 16;
        const uint4 x = *reinterpret_cast<const uint4 *>(&L.txt[p0 < (uint32_t)kDCap ? p0 : 0]);
        uint32_t sp_all = 0, pn_all = 0, ct_all = 0, hi_all = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t w = q == 0 ? x.x : (q == 1 ? x.y : (q == 2 ? x.z : x.w)), x7 = w & 0x7F7F7F7Fu, hi = w & 0x80808080u;
          // a byte below 0x80 lies in [lo, hi] iff bit 7 of (b + 0x80 - lo) is set and bit 7 of (b + 0x7F - hi) is not: no carry
          // leaves a byte.  The ASCII ranges are checked against the class table on the host (dedup_front).
#define SWT_IN(lo_, hi_) ((x7 + (0x80u - (lo_)) * 0x01010101u) & ~(x7 + (0x7Fu - (hi_)) * 0x01010101u))
          uint32_t in_ws, in_pn = 0;
          if (Mode == kDedupWp) {
            in_ws = SWT_IN(0x09u, 0x0Du) | SWT_IN(0x1Cu, 0x20u);  // str.isspace
          } else {
            in_ws = SWT_IN(0x09u, 0x0Du) | SWT_IN(0x20u, 0x20u);  // the pre-tokenizer's whitespace ...
            in_pn = SWT_IN(0x21u, 0x2Fu) | SWT_IN(0x3Au, 0x40u) | SWT_IN(0x5Bu, 0x60u) | SWT_IN(0x7Bu, 0x7Eu);  // ... and punctuation
          }
#undef SWT_IN
          sp_all |= dd_nibble(in_ws & ~hi & 0x80808080u) << (4 * q);
          if (kPunctSplits) pn_all |= dd_nibble(in_pn & ~hi & 0x80808080u) << (4 * q);
          ct_all |= dd_nibble(hi & ~(w << 1)) << (4 * q);  // 10xxxxxx: a continuation byte
          hi_all |= dd_nibble(hi & (w << 1)) << (4 * q);   // 11xxxxxx: a lead byte whose class the table knows
        }
        // my 16 bytes' place in [off0, staged)
        const uint32_t lo = off0 > p0 ? (off0 - p0 < 16u ? off0 - p0 : 16u) : 0u;
        const uint32_t hi_n = staged > p0 ? (staged - p0 < 16u ? staged - p0 : 16u) : 0u;
        const uint32_t inr16 = (((1u << hi_n) - 1u) & ~((1u << lo) - 1u)) & 0xFFFFu;
        uint32_t sp16 = sp_all & inr16, pn16 = pn_all & inr16;
        const uint32_t ct16 = ct_all & inr16;
        // the non-ASCII lead bytes: decode and look up, as the byte-lane loop does for every byte
        for (uint32_t m = hi_all & inr16; m; m &= m - 1u) {
          const uint32_t j = (uint32_t)__builtin_ctz(m), p = p0 + j;
          const uint8_t b = L.txt[p];
          int len = utf8_len(b);
          if (p + len > staged) len = (int)(staged - p);
          uint32_t cp = b;
          if (len > 1) {
            cp = b & (0xFF >> (len + 1));
            for (int i = 1; i < len; i++) cp = (cp << 6) | (L.txt[p + i] & 0x3F);
          }
          uint32_t c2;  // bit 0: whitespace of the mode, bit 1: punctuation
          if (cp < (uint32_t)kClsLds) {
            c2 = (L.cls2[cp >> 4] >> ((cp & 15u) << 1)) & 3u;
          } else {
            const uint8_t c = cp < kNumCodePoints ? cls_tab[cp] : (uint8_t)0;
            c2 = ((c & kWsBit) ? 1u : 0u) | ((kPunctSplits && (c & kClsPunct)) ? 2u : 0u);
          }
          if (c2 & 1u) sp16 |= 1u << j;
          if (c2 & 2u) pn16 |= 1u << j;
        }
        const uint32_t wsm16 = (sp16 | ~inr16) & 0xFFFFu;  // bytes outside the chunk behave as whitespace
        const uint32_t lead16 = ~ct16 & 0xFFFFu;
        // whitespace smear over continuation bytes and "the byte before me", with the lane below as the low half of a window
        const uint32_t raw16 = wsm16 | pn16;  // bytes of whitespace / punctuation characters, before the smear
        uint32_t below_wb = __shfl_up(raw16, 1), below_ct = __shfl_up(ct16, 1);
        if (lane == 0) { below_wb = carry_wb; below_ct = carry_ct; }
        const uint32_t c32 = below_ct | (ct16 << 16);
        uint32_t wb32 = below_wb | (raw16 << 16);
        wb32 |= (wb32 << 1) & c32;
        wb32 |= (wb32 << 1) & c32;
        wb32 |= (wb32 << 1) & c32;
        const uint32_t ss16 = reinterpret_cast<const uint16_t *>(L.sbits)[g < (uint32_t)(kDBlocks + 1) * 4u ? g : 0];
        const uint32_t first16 = g == 0 ? (1u << off0) : 0u;
        const uint32_t before16 = ((wb32 >> 15) | ss16 | first16) & 0xFFFFu;
        const uint32_t sym16 = lead16 & ~wsm16 & inr16;
        const uint32_t ws16 = sym16 & (pn16 | before16);
        // chunk cut candidates: word boundaries strictly inside, with room for a whole UTF-8 char behind them
        const uint32_t c_lo = off0 + 1u > p0 ? (off0 + 1u - p0 < 16u ? off0 + 1u - p0 : 16u) : 0u;
        const uint32_t c_hi = staged >= p0 + 4u ? (staged - 3u - p0 < 16u ? staged - 3u - p0 : 16u) : 0u;  // p + 4 <= staged
        const uint32_t cutr16 = (c_hi > c_lo ? (((1u << c_hi) - 1u) & ~((1u << c_lo) - 1u)) : 0u) & 0xFFFFu;
        const uint32_t cut16 = lead16 & (wsm16 | pn16 | ss16) & cutr16 & inr16;
        const unsigned long long CUTL = __ballot(cut16 != 0u);
        if (CUTL) {
          const int src = 63 - __builtin_clzll(CUTL);
          const uint32_t top = __shfl(cut16, src);
          cut = (int)((g0 + (uint32_t)src) * 16u + 31u - (uint32_t)__builtin_clz(top));
        }
        // words: ranks by a wave scan of the lanes' counts
        const uint32_t n_mine = (uint32_t)__popc(ws16);
        uint32_t incl = n_mine;
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t y = __shfl_up(incl, d);
          if (lane >= d) incl += y;
        }
        uint32_t at = nw + incl - n_mine;
        if (g < n_groups) {
          reinterpret_cast<uint16_t *>(L.endm)[g] = (uint16_t)(wsm16 | ws16);
          reinterpret_cast<uint16_t *>(L.wst)[g] = (uint16_t)ws16;
          if ((g & 3u) == 0u) L.nwb[g >> 2] = at;
          for (uint32_t m = ws16; m; m &= m - 1u) L.wl[at++] = (uint16_t)(p0 + (uint32_t)__builtin_ctz(m));
        }
        nw += __shfl(incl, 63);
        carry_wb = __shfl(raw16, 63);
        carry_ct = __shfl(ct16, 63);
      }
    } else
    for (uint32_t blk = 0; blk < nblk; blk++) {
      const uint32_t p = blk * 64 + lane;
      const bool inr = p >= off0 && p < staged;
      const uint8_t b = inr ? L.txt[p] : (uint8_t)' ';
      const bool lead = !utf8_is_cont(b);
      uint32_t cp = b;
      if (b >= 0xC0) {
        int len = utf8_len(b);
        if (p + len > staged) len = (int)(staged - p);
        if (len > 1) {
          cp = b & (0xFF >> (len + 1));
          for (int i = 1; i < len; i++) cp = (cp << 6) | (L.txt[p + i] & 0x3F);
        }
      }
      uint32_t c2 = 1u;  // bit 0: whitespace of the mode (bytes outside the chunk count as such), bit 1: punctuation
      if (inr && lead) {
        if (cp < (uint32_t)kClsLds) {
          c2 = (L.cls2[cp >> 4] >> ((cp & 15u) << 1)) & 3u;
        } else {
          const uint8_t c = cp < kNumCodePoints ? cls_tab[cp] : (uint8_t)0;
          c2 = ((c & kWsBit) ? 1u : 0u) | ((kPunctSplits && (c & kClsPunct)) ? 2u : 0u);
        }
      }
      const unsigned long long INR = __ballot(inr);
      const unsigned long long LEAD = __ballot(lead);
      const unsigned long long WSm = __ballot(lead && (c2 & 1u));
      const unsigned long long PNm = kPunctSplits ? __ballot(lead && (c2 & 2u)) : 0ull;
      const unsigned long long CONT = ~LEAD;
      unsigned long long WB = WSm | PNm | ((prev_wb && (CONT & 1ull)) ? 1ull : 0ull);
      WB |= (WB << 1) & CONT;
      WB |= (WB << 1) & CONT;
      WB |= (WB << 1) & CONT;
      const unsigned long long SS = L.sbits[blk];
      const unsigned long long first_bit = blk == 0 ? (1ull << off0) : 0ull;
      const unsigned long long before = (WB << 1) | (prev_wb ? 1ull : 0ull) | SS | first_bit;
      const unsigned long long SYM = LEAD & ~WSm & INR;
      const unsigned long long WSTART = SYM & (PNm | before);
      const unsigned long long CUT = LEAD & (WSm | PNm | SS) & __ballot(inr && p > off0 && p + 4 <= staged);
      if (CUT) cut = (int)(blk * 64 + 63 - __builtin_clzll(CUT));
      if (lane == 0) {
        L.endm[blk] = WSm | WSTART | ~INR;
        L.wst[blk] = WSTART;
        L.nwb[blk] = nw;
      }
      if ((WSTART >> lane) & 1ull) L.wl[nw + __popcll(WSTART & lt)] = (uint16_t)p;
      nw += __popcll(WSTART);
      prev_wb = (WB >> 63) & 1ull;
    }
    __syncthreads();
    uint32_t ce = staged;
    if (!last) {
      if (cut < 0) {
        // a single word longer than the chunk (or a lone separator in front of one): one lane, global memory
        if (lane == 0) {
          uint64_t s = s_next;
          while (s < s_hi && sent_off[s] <= cb) s++;
          const uint64_t send = sent_off[s];
          uint64_t e = cb;
          bool first = true, has_word = true;
          uint32_t nchar = 0, cp0 = 0;
          while (e < send) {
            const uint8_t b = text[e];
            int len = utf8_len(b);
            if (e + len > send) len = (int)(send - e);
            uint32_t cp = b;
            if (b >= 0x80 && len > 1) {
              cp = b & (0xFF >> (len + 1));
              for (int i = 1; i < len; i++) cp = (cp << 6) | (text[e + i] & 0x3F);
            }
            const uint8_t c = utf8_is_cont(b) ? kWsBit : (cp < kNumCodePoints ? cls_tab[cp] : (uint8_t)0);
            if (c & kWsBit) { if (first) { e += len; has_word = false; } break; }
            if (kPunctSplits && (c & kClsPunct)) { if (first) { e += len; nchar = 1; cp0 = cp; } break; }
            if (first) cp0 = cp;
            nchar++;
            e += len;
            first = false;
          }
          L.giant_new = 0;
          L.giant_word = 0;
          if (has_word && e > cb) {
            L.giant_word = 1;
            if (Mode == kDedupBpe && nchar == 1) {
              my_rec[words_done] = cp0;
            } else {
              bool fresh;
              const uint64_t wl = e - cb;
              const uint32_t idx = dd_find_or_insert(D, text, text + cb, (uint32_t)wl, cb, fresh);
              my_rec[words_done] = kRefSlot | idx;
              if (fresh) {
                D.rec[idx] = wl;
                my_list[n_new] = ((unsigned long long)idx << 32) | cb;
                my_bytes += wl;
                L.giant_new = 1;
              }
            }
          }
          L.giant_end = e;
        }
        __syncthreads();
        const uint32_t wd_before = words_done;
        cb = L.giant_end;
        n_new += L.giant_new;
        words_done += L.giant_word;
        uint32_t gone = 0;
        for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
          if (sent_off[s] >= cb) break;
          sent_word[s] = wd_before;
          gone++;
        }
        for (int d = 32; d >= 1; d >>= 1) gone += __shfl_xor(gone, d);
        s_next += gone;
        __syncthreads();
        if (cb >= span_end) break;
        continue;
      }
      ce = (uint32_t)cut;
    }
    // one lane per word
    const uint32_t wd0 = words_done;
    if (!(dbg & 1))
    for (uint32_t k0 = 0; k0 < nw; k0 += 64) {
      const uint32_t k = k0 + lane;
      const uint32_t s = k < nw ? L.wl[k] : 0xFFFFu;
      const bool mine_w = k < nw && s < ce;
      bool is_new = false;
      uint32_t idx = 0, wlen = 0;
      if (mine_w) {
        uint32_t w = s >> 6;
        unsigned long long m = (s & 63) == 63 ? 0ull : (L.endm[w] & ~((2ull << (s & 63)) - 1ull));
        while (!m && w + 1 < nblk) m = L.endm[++w];
        uint32_t e = m ? w * 64 + (uint32_t)__builtin_ctzll(m) : ce;
        if (e > ce) e = ce;
        wlen = e - s;
        const uint8_t b0 = L.txt[s];
        const uint32_t l0 = (uint32_t)utf8_len(b0);
        uint32_t r;
        if (Mode == kDedupBpe && wlen <= l0) {  // a single symbol: it is its own token
          uint32_t cp = b0;
          if (b0 >= 0xC0 && wlen > 1) {
            cp = b0 & (0xFF >> (wlen + 1));
            for (uint32_t i = 1; i < wlen; i++) cp = (cp << 6) | (L.txt[s + i] & 0x3F);
          }
          r = cp;
        } else {
          idx = dd_find_or_insert_lds(D, text, &L.txt[s], wlen, abase + s, is_new);
          r = kRefSlot | idx;
        }
        my_rec[wd0 + k] = r;  // the words before the cut are a prefix of the list
      }
      const unsigned long long NEWm = __ballot(is_new);
      if (is_new) {
        D.rec[idx] = wlen;
        my_list[n_new + __popcll(NEWm & lt)] = ((unsigned long long)idx << 32) | (abase + s);
        my_bytes += wlen;
      }
      n_new += (uint32_t)__popcll(NEWm);
      words_done += (uint32_t)__popcll(__ballot(mine_w));
    }
    // sentences that start in what this chunk consumed: how many of the tile's words come before them
    cb = abase + ce;
    uint32_t gone = 0;
    for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
      const uint64_t o = sent_off[s];
      if (!last && o >= cb) break;
      const uint32_t rel = (uint32_t)(o - abase);
      sent_word[s] = rel >= staged ? words_done : wd0 + L.nwb[rel >> 6] + (uint32_t)__popcll(L.wst[rel >> 6] & ((1ull << (rel & 63)) - 1ull));
      gone++;
    }
    if (last) break;
    for (int d = 32; d >= 1; d >>= 1) gone += __shfl_xor(gone, d);
    s_next += gone;
    __syncthreads();
  }
  for (int d = 32; d >= 1; d >>= 1) my_bytes += __shfl_xor(my_bytes, d);
  if (lane == 0) {
    D.tile_new[t] = ((unsigned long long)n_new << 32) | my_bytes;
    tile_words[t] = words_done;
  }
}

// Numbers the new words of every tile (scan of tile_new) and copies them into the unique-word text.
template <int Mode>
__global__ __launch_bounds__(64) void ureg_kernel(const uint8_t *__restrict__ text, const uint64_t *__restrict__ sent_off,
                                                      const uint64_t *__restrict__ plan, DedupTab D,
                                                      const unsigned long long *__restrict__ new_local,
                                                      const unsigned long long *__restrict__ new_blk_base,
                                                      const unsigned long long *__restrict__ d_total, uint32_t *__restrict__ uslot,
                                                      uint64_t *__restrict__ uoff, uint8_t *__restrict__ utext,
                                                      uint64_t *__restrict__ plan2, uint64_t n_tiles2, uint32_t tile2_min) {
  const int lane = threadIdx.x;
  const uint64_t t = blockIdx.x;
  // most tiles of running text table nothing new: they leave before anything else is loaded or divided (a tile without
  // sentences has no new words either; workgroup 0 also writes the ends of the plan)
  const uint32_t n_new = (uint32_t)(D.tile_new[t] >> 32);
  if (!n_new && t != 0) return;
  const uint64_t s_lo = plan[t], s_hi = plan[t + 1];
  // The plan of the launch that encodes the unique words (plan2[tt] = first unique word that starts at or after byte
  // tt * tile2) is written here too: that launch has a fixed n_tiles2 workgroups, so the tile size follows from the
  // number of unique bytes, which only the device knows.  A boundary inside (start, end] of a word belongs to the word
  // after it; boundary 0 and the boundaries behind the text are written by workgroup 0.
  const uint64_t n_uniq = *d_total >> 32, ubytes = *d_total & 0xFFFFFFFFull;
  uint64_t tile2 = n_tiles2 ? (ubytes + n_tiles2 - 1) / n_tiles2 : 0;
  if (tile2 < tile2_min) tile2 = tile2_min;
  if (t == 0) {
    if (lane == 0) uoff[n_uniq] = ubytes;  // the end of the last unique word
    if (plan2)
      for (uint64_t tt = lane; tt <= n_tiles2; tt += 64)
        if (tt == 0 || tt * tile2 > ubytes) plan2[tt] = tt == 0 ? 0ull : n_uniq;
  }
  if (s_lo == s_hi || !n_new) return;
  const unsigned long long base = new_blk_base[t >> 10] + new_local[t];
  const uint64_t u0 = base >> 32;
  uint64_t b0 = base & 0xFFFFFFFFull;
  const unsigned long long *my_list = D.newlist + (Mode == kDedupWp ? sent_off[s_lo] : (sent_off[s_lo] >> 1));
  for (uint32_t k0 = 0; k0 < n_new; k0 += 64) {
    const uint32_t k = k0 + lane;
    const unsigned long long e = k < n_new ? my_list[k] : 0ull;
    const uint32_t idx = (uint32_t)(e >> 32);
    const uint64_t pos = e & 0xFFFFFFFFull;
    const uint32_t len = k < n_new ? (uint32_t)D.rec[idx] : 0u;
    uint32_t x = len;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (k < n_new) {
      const uint64_t bo = b0 + (x - len);
      uoff[u0 + k] = bo;
      uslot[u0 + k] = idx;
      for (uint32_t i = 0; i < len; i++) utext[bo + i] = text[pos + i];
      if (plan2)
        for (uint64_t T = (bo / tile2 + 1) * tile2; T <= bo + len; T += tile2) plan2[T / tile2] = u0 + k + 1;
    }
    b0 += __shfl(x, 63);
  }
}

// a word record: a code point (BPE: the one-symbol word is its own token), or kRefSlot | table slot until the counting
// pass has run, kRefSlot | unique index afterwards
__device__ __forceinline__ uint32_t ref_count_dense(uint32_t v, const unsigned long long *__restrict__ drec, uint64_t &src) {
  src = 0;
  if (!(v & kRefSlot)) return 1;
  const unsigned long long r = drec[v & ~kRefSlot];
  src = r & 0xFFFFFFFFull;
  return (uint32_t)(r >> 32);
}

// tokens per tile (a tile's word records are the first tile_words[t] entries behind wref[span_base]); the records change
// from table slots to unique-word indices here, so that refwrite_kernel gathers from the dense drec[] only
__global__ __launch_bounds__(64) void refcount_kernel(const uint64_t *__restrict__ sent_off, const uint64_t *__restrict__ plan,
                                                      uint32_t *__restrict__ wref, const uint32_t *__restrict__ tile_words,
                                                      const unsigned long long *__restrict__ rec, uint32_t *__restrict__ tile_tok) {
  const uint64_t t = blockIdx.x;
  const uint64_t s_lo = plan[t], s_hi = plan[t + 1];
  uint32_t total = 0;
  if (s_lo != s_hi) {
    uint32_t *my_rec = wref + sent_off[s_lo];
    const uint32_t n_w = tile_words[t];
    for (uint32_t k = threadIdx.x; k < n_w; k += 64) {
      const uint32_t v = my_rec[k];
      if (v & kRefSlot) {
        const unsigned long long r = rec[v & ~kRefSlot];  // count:32 | unique index:32
        total += (uint32_t)(r >> 32);
        my_rec[k] = kRefSlot | (uint32_t)r;
      } else {
        total += 1;
      }
    }
    for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d);
  }
  if (threadIdx.x == 0) tile_tok[t] = total;
}

// final tokens + sentence offsets
__global__ __launch_bounds__(64) void refwrite_kernel(const uint64_t *__restrict__ sent_off, const uint64_t *__restrict__ plan,
                                                          uint64_t n_tiles, uint64_t n_sent, const uint32_t *__restrict__ wref,
                                                          const uint32_t *__restrict__ tile_words, const uint32_t *__restrict__ sent_word,
                                                          const unsigned long long *__restrict__ rec,
                                                          const uint32_t *__restrict__ u_ids, const uint32_t *__restrict__ tile_base,
                                                          const unsigned long long *__restrict__ blk_base,
                                                          const uint64_t *__restrict__ n_tokens, uint32_t *__restrict__ out_ids,
                                                          uint64_t *__restrict__ out_off) {
  __shared__ uint32_t pre[kDCap];
  const int lane = threadIdx.x;
  const uint64_t t = blockIdx.x;
  const uint64_t s_lo = plan[t], s_hi = plan[t + 1];
  if (t == n_tiles - 1 && lane == 0) out_off[n_sent] = *n_tokens;
  if (s_lo == s_hi) return;
  const uint64_t base = blk_base[t >> 10] + tile_base[t];
  const uint32_t *my_rec = wref + sent_off[s_lo];
  const uint32_t n_w = tile_words[t];
  uint64_t s_next = s_lo;
  uint32_t run = 0;
  for (uint32_t k0 = 0;; k0 += kDCap) {
    const uint32_t k1 = n_w - k0 > (uint32_t)kDCap ? k0 + kDCap : n_w;
    for (uint32_t j0 = k0; j0 < k1; j0 += 64) {
      const uint32_t j = j0 + lane;
      uint64_t src = 0;
      uint32_t v = 0, n = 0;
      if (j < k1) {
        v = my_rec[j];
        n = ref_count_dense(v, rec, src);
      }
      uint32_t x = n;
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= d) x += y;
      }
      const uint32_t ex = run + x - n;
      if (j < k1) pre[j - k0] = ex;
      if (n == 1 && !(v & kRefSlot)) {
        out_ids[base + ex] = v;
      } else if (n) {
        // the first tokens of the run together (most words have one to three), then the rest
        const uint32_t t0 = u_ids[src], t1 = n > 1 ? u_ids[src + 1] : 0u, t2 = n > 2 ? u_ids[src + 2] : 0u;
        out_ids[base + ex] = t0;
        if (n > 1) out_ids[base + ex + 1] = t1;
        if (n > 2) out_ids[base + ex + 2] = t2;
        for (uint32_t i = 3; i < n; i++) out_ids[base + ex + i] = u_ids[src + i];
      }
      run += __shfl(x, 63);
    }
    __syncthreads();
    // sentences whose first word lies in [k0, k1) -- and, on the last chunk, those behind the last word
    const bool lastc = k1 == n_w;
    uint32_t mine = 0;
    for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
      const uint32_t w = sent_word[s];
      if (w > k1 || (w == k1 && !lastc)) break;
      out_off[s] = base + (w < k1 ? pre[w - k0] : run);
      mine++;
    }
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    s_next += mine;
    __syncthreads();
    if (lastc) break;
  }
}


// ---- FastWP back half: a sentence with a failed chunk (the reference never returns on it) yields NO tokens and a status,
// so the counts are summed per sentence.  One kernel, two uses: Write = false leaves the tile's token total, Write = true
// copies the tokens and writes sentence offsets and statuses.  `rec` is the table's rec[] for the counting launch -- which
// also rewrites every record from its table slot to the unique index found there -- and the dense drec[] for the writing one.
template <bool Write>
__device__ void wp_refs_serial(const uint64_t *__restrict__ sent_off, uint64_t s_lo, uint64_t s_hi, uint32_t *__restrict__ my_rec,
                               uint32_t n_w, const uint32_t *__restrict__ sent_word, const unsigned long long *__restrict__ rec,
                               const uint32_t *__restrict__ u_ids, uint64_t base, uint32_t *__restrict__ tile_tok_out,
                               uint32_t *__restrict__ out_ids, uint64_t *__restrict__ out_off, uint8_t *__restrict__ status) {
  // a tile with more words than the LDS holds (a giant sentence): one lane, sentence by sentence
  uint32_t run = 0;
  for (uint64_t s = s_lo; s < s_hi; s++) {
    const uint32_t w0 = sent_word[s], w1 = s + 1 < s_hi ? sent_word[s + 1] : n_w;
    bool bad = false;
    uint32_t tot = 0;
    for (uint32_t w = w0; w < w1; w++) {
      const unsigned long long r = rec[my_rec[w] & ~kRefSlot];
      if (!Write) my_rec[w] = kRefSlot | (uint32_t)r;
      const uint32_t c = (uint32_t)(r >> 32);
      bad |= c == kRecFailed;
      tot += c;
    }
    if (Write) {
      out_off[s] = base + run;
      status[s] = bad ? (uint8_t)SWT_WP_NONTERMINATING : (uint8_t)SWT_WP_OK;
      if (!bad)
        for (uint32_t w = w0; w < w1; w++) {
          const unsigned long long r = rec[my_rec[w] & ~kRefSlot];
          const uint32_t c = (uint32_t)(r >> 32);
          const uint64_t src = r & 0xFFFFFFFFull;
          for (uint32_t i = 0; i < c; i++) out_ids[base + run + i] = u_ids[src + i];
          run += c;
        }
    } else if (!bad) {
      run += tot;
    }
  }
  if (!Write) *tile_tok_out = run;
}

template <bool Write>
__global__ __launch_bounds__(64) void wp_refs_kernel(const uint64_t *__restrict__ sent_off, const uint64_t *__restrict__ plan,
                                                     uint64_t n_tiles, uint64_t n_sent, uint32_t *__restrict__ wref,
                                                     const uint32_t *__restrict__ tile_words, const uint32_t *__restrict__ sent_word,
                                                     const unsigned long long *__restrict__ rec, const uint32_t *__restrict__ u_ids,
                                                     const uint32_t *__restrict__ tile_base, const unsigned long long *__restrict__ blk_base,
                                                     const uint64_t *__restrict__ n_tokens, uint32_t *__restrict__ tile_tok,
                                                     uint32_t *__restrict__ out_ids, uint64_t *__restrict__ out_off,
                                                     uint8_t *__restrict__ status) {
  __shared__ uint32_t cnt[kDCap];  // per word: token count (0 for the words of a failed sentence), then its exclusive prefix
  const int lane = threadIdx.x;
  const uint64_t t = blockIdx.x;
  const uint64_t s_lo = plan[t], s_hi = plan[t + 1];
  if (Write && t == n_tiles - 1 && lane == 0) out_off[n_sent] = *n_tokens;
  if (s_lo == s_hi) {
    if (!Write && lane == 0) tile_tok[t] = 0;
    return;
  }
  const uint64_t base = Write ? blk_base[t >> 10] + tile_base[t] : 0ull;
  uint32_t *my_rec = wref + sent_off[s_lo];
  const uint32_t n_w = tile_words[t];
  if (n_w > (uint32_t)kDCap) {
    if (lane == 0) wp_refs_serial<Write>(sent_off, s_lo, s_hi, my_rec, n_w, sent_word, rec, u_ids, base, tile_tok + t, out_ids, out_off, status);
    return;
  }
  bool any_bad = false;
  uint32_t my_sum = 0;
  // the records of a lane's first four words stay in registers for the copy below (a 1-KiB tile holds ~190 words): the write
  // launch then gathers each word's record once, and the four gathers are in flight together
  constexpr int kKeep = Write ? 4 : 0;  // (the counting launch has no second use for them: measured slower with)
  unsigned long long keep[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
  for (int u = 0; u < kKeep; u++) {
    const uint32_t k = (uint32_t)u * 64 + lane;
    keep[u] = k < n_w ? rec[my_rec[k] & ~kRefSlot] : 0ull;
  }
#pragma unroll
  for (int u = 0; u < kKeep; u++) {
    const uint32_t k = (uint32_t)u * 64 + lane;
    if (k < n_w) {
      const unsigned long long r = keep[u];
      const uint32_t c = (uint32_t)(r >> 32);
      cnt[k] = c;
      any_bad |= c == kRecFailed;
      my_sum += c;
    }
  }
  for (uint32_t k = kKeep * 64 + lane; k < n_w; k += 64) {
    const unsigned long long r = rec[my_rec[k] & ~kRefSlot];
    if (!Write) my_rec[k] = kRefSlot | (uint32_t)r;
    const uint32_t c = (uint32_t)(r >> 32);
    cnt[k] = c;
    any_bad |= c == kRecFailed;
    my_sum += c;
  }
  any_bad = __any(any_bad);
  __syncthreads();
  if (any_bad) {
    // rare: some chunk of this tile cannot be encoded -- find its sentence(s), one lane per sentence
    for (uint64_t s = s_lo + lane; s < s_hi; s += 64) {
      const uint32_t w0 = sent_word[s], w1 = s + 1 < s_hi ? sent_word[s + 1] : n_w;
      bool bad = false;
      for (uint32_t w = w0; w < w1; w++) bad |= cnt[w] == kRecFailed;
      if (bad)
        for (uint32_t w = w0; w < w1; w++) cnt[w] = 0u;
      if (Write) status[s] = bad ? (uint8_t)SWT_WP_NONTERMINATING : (uint8_t)SWT_WP_OK;
    }
    __syncthreads();
  } else {
    if (Write)
      for (uint64_t s = s_lo + lane; s < s_hi; s += 64) status[s] = (uint8_t)SWT_WP_OK;
    if (!Write) {  // the common case of the counting launch: the tile's total is the plain sum
      for (int d = 32; d >= 1; d >>= 1) my_sum += __shfl_xor(my_sum, d);
      if (lane == 0) tile_tok[t] = my_sum;
      return;
    }
  }
  uint32_t run = 0;
  for (uint32_t k0 = 0; k0 < n_w; k0 += 64) {
    const uint32_t k = k0 + lane;
    const uint32_t n = k < n_w ? cnt[k] : 0u;
    uint32_t x = n;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    const uint32_t ex = run + x - n;
    if (k < n_w) cnt[k] = ex;
    if (Write && n) {
      uint64_t src;
      if (k0 < (uint32_t)kKeep * 64) {
        unsigned long long r = keep[0];
#pragma unroll
        for (int u = 1; u < 4; u++) r = k0 == (uint32_t)u * 64 ? keep[u] : r;
        src = r & 0xFFFFFFFFull;
      } else {
        src = rec[my_rec[k] & ~kRefSlot] & 0xFFFFFFFFull;
      }
      // the first tokens of the run together (most words have one to three), then the rest
      const uint32_t t0 = u_ids[src], t1 = n > 1 ? u_ids[src + 1] : 0u, t2 = n > 2 ? u_ids[src + 2] : 0u;
      out_ids[base + ex] = t0;
      if (n > 1) out_ids[base + ex + 1] = t1;
      if (n > 2) out_ids[base + ex + 2] = t2;
      for (uint32_t i = 3; i < n; i++) out_ids[base + ex + i] = u_ids[src + i];
    }
    run += __shfl(x, 63);
  }
  __syncthreads();
  if (!Write) {
    if (lane == 0) tile_tok[t] = run;
    return;
  }
  for (uint64_t s = s_lo + lane; s < s_hi; s += 64) {
    const uint32_t w0 = sent_word[s];
    out_off[s] = base + (w0 < n_w ? cnt[w0] : run);
  }
}


void DedupEngine::release() {
  for (DevBuf *b : {&slot, &rec, &drec, &uslot, &utext, &uoff, &misc, &newlist, &tile_new, &new_local, &new_blk, &tile_words}) b->release();
  seen.release();
  bits = epoch = 0;
}

// The unique-word pass costs about as much per BYTE of unique words as the direct path costs per byte of text, and the front
// and back halves come on top: measured, the dedup stops paying once the unique words make up about a third of the text
// (S85k-open: 0.50 of the bytes, 560 us against 500 us direct; S85k-lex: 0.10, 196 us against 505 us).
bool DedupEngine::pays(uint64_t n_bytes) {
  if (!seen.p) return true;
  const volatile unsigned long long *h = seen.as<unsigned long long>();
  const unsigned long long total = h[0], of = h[1];
  if (!of || of * 2 < n_bytes || n_bytes * 2 < of) return true;  // nothing seen yet, or a batch of another size: look
  const unsigned long long ubytes = total & 0xFFFFFFFFull;
  if (ubytes * pay_ratio <= of) return true;
  if (++skipped >= kDedupRetry) { skipped = 0; return true; }
  return false;
}

void DedupEngine::note(uint64_t n_bytes, hipStream_t st) {
  if (!seen.p) {
    if (seen.reserve(16)) return;
    seen.as<unsigned long long>()[0] = 0;
    seen.as<unsigned long long>()[1] = 0;
  }
  unsigned long long *h = seen.as<unsigned long long>();
  h[1] = n_bytes;
  (void)hipMemcpyAsync(h, misc.p, 8, hipMemcpyDeviceToHost, st);
  skipped = 0;
}

int dedup_front(DedupEngine &E, TileWorkspace &ws, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off, uint64_t n_sent,
                const uint8_t *d_cls, DedupMode mode, hipStream_t st, uint64_t *d_plan2, uint64_t n_tiles2, uint32_t tile2_min) {
  int rc;
  if (n_bytes > kDedupMaxBytes) return 1;
  const uint64_t n_tiles = tile_count(n_bytes, kDTile);
  if ((rc = ws.reserve(n_bytes, n_sent, n_tiles))) return rc;
  // table: one slot per 32 bytes of text (S85k: one distinct word per 107 bytes), 2^16 .. 2^24 slots: small on purpose
  // (see DedupTab); never cleared (epoch).  rec[] = the table's slots, then the own slots of the words that overflowed.
  uint32_t bits = 16;
  while ((1ull << bits) < n_bytes / 32 && bits < 24) bits++;
  static const int env_bits = getenv("SWT_DD_BITS") ? atoi(getenv("SWT_DD_BITS")) : 0;  // measurement knob
  if (env_bits >= 4 && env_bits <= 24 && !E.opt_table_bits) bits = (uint32_t)env_bits;
  if (E.opt_table_bits) {  // SWT_OPT_DEDUP_TABLE_BITS (tests: a table so small that words overflow it)
    bits = E.opt_table_bits;
    if (bits != E.bits) E.bits = 0;
  }
  if (bits > E.bits) {
    E.slot.release();
    if ((rc = E.slot.reserve(((size_t)1 << bits) * 8))) return rc;
    SWT_HIP(hipMemsetAsync(E.slot.p, 0, ((size_t)1 << bits) * 8, st));
    E.bits = bits;
    E.epoch = 0;
  }
  const uint32_t ovf_shift = mode == kDedupWp ? 0u : 1u;
  if ((rc = E.rec.reserve((((size_t)1 << E.bits) + (n_bytes >> ovf_shift) + 2) * 8))) return rc;
  if (++E.epoch >= 256) {
    SWT_HIP(hipMemsetAsync(E.slot.p, 0, ((size_t)1 << E.bits) * 8, st));
    E.epoch = 1;
  }
  // a tabled word has at least two bytes (BPE: a one-symbol word is not tabled) or one (WP)
  const uint64_t max_uniq = (mode == kDedupWp ? n_bytes : n_bytes / 2) + 2;
  const uint64_t nb_new = (n_tiles + 1023) / 1024;
  if ((rc = E.utext.reserve(n_bytes + 64)) || (rc = E.uoff.reserve((max_uniq + 2) * 8)) || (rc = E.misc.reserve(64)) ||
      (rc = E.uslot.reserve((max_uniq + 2) * 4)) || (rc = E.drec.reserve((max_uniq + 2) * 8)) || (rc = E.newlist.reserve((max_uniq + 2) * 8)) ||
      (rc = E.tile_new.reserve((n_tiles + 1) * 8)) || (rc = E.new_local.reserve((n_tiles + 1) * 8)) ||
      (rc = E.tile_words.reserve((n_tiles + 1) * 4)))
    return rc;
  if (E.new_blk.cap < (2 * nb_new + 2) * 8) {
    if ((rc = E.new_blk.reserve((2 * nb_new + 2) * 8))) return rc;
    SWT_HIP(hipMemsetAsync(E.new_blk.p, 0, E.new_blk.cap, st));  // the scan's ticket starts at zero (and leaves it so)
  }
  unsigned long long *d_misc = E.misc.as<unsigned long long>();  // [0] unique words:32 | their bytes:32, [1..2] diagnostics
  DedupTab D;
  D.slot = E.slot.as<unsigned long long>();
  D.rec = E.rec.as<unsigned long long>();
  D.n_bytes = n_bytes;
  static const bool coherent_only = getenv("SWT_DD_COHERENT") != nullptr;  // comparison runs: no cached first look
  D.diag = ((ablation_knob(2) & 4) ? 1u : 0u) | (coherent_only ? 2u : 0u);
  D.bits = E.bits;
  D.epoch = E.epoch;
  D.ovf_shift = ovf_shift;
  D.newlist = E.newlist.as<unsigned long long>();
  D.tile_new = E.tile_new.as<unsigned long long>();
  D.overflow = reinterpret_cast<unsigned int *>(d_misc + 1);
  if (D.diag & 1u) SWT_HIP(hipMemsetAsync(d_misc, 0, 32, st));
  uint32_t *wref = ws.scratch.as<uint32_t>();
  uint64_t *plan1 = ws.plan.as<uint64_t>();
  unsigned long long *new_local = E.new_local.as<unsigned long long>(), *new_blk = E.new_blk.as<unsigned long long>();
  launch_plan(d_sent_off, n_sent, n_tiles, kDTile, plan1, st);
  // the lane-per-16-bytes split knows str.isspace in ASCII as two ranges: if the class table ever says otherwise, or on request
  // (SWT_DD_OLD_SPLIT=1), the byte-lane loop runs
  static const bool ascii_ok = [] {
    const uint8_t *cls = host_class_table();
    for (uint32_t c = 0; c < 128; c++) {
      if (((cls[c] & kClsPySpace) != 0) != ((c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x20))) return false;
      if (((cls[c] & kClsWs) != 0) != ((c >= 0x09 && c <= 0x0D) || c == 0x20)) return false;
      if (((cls[c] & kClsPunct) != 0) != ((c >= 0x21 && c <= 0x2F) || (c >= 0x3A && c <= 0x40) || (c >= 0x5B && c <= 0x60) || (c >= 0x7B && c <= 0x7E))) return false;
    }
    return true;
  }();
  const char *old_split = getenv("SWT_DD_OLD_SPLIT");
  const uint32_t split_flag = (!ascii_ok || (old_split && *old_split && *old_split != '0')) ? 0x80000000u : 0u;
  prof_begin(st, 3);
  // (Two launches -- a first one over 1/8 .. 1/128 of the tiles to table the frequent words, so that every compute unit of the
  // second finds them through its caches -- were measured: 0.843 - 0.862 ms against 0.839 ms per FastWP call.  One launch.)
  {
    const dim3 grid((unsigned)n_tiles);
    if (mode == kDedupWp && !split_flag)
      hipLaunchKernelGGL((wordref_kernel<kDedupWp, true>), grid, dim3(64), 0, st, d_text, n_bytes, d_sent_off, plan1, d_cls, D,
                         wref, ws.sent_local.as<uint32_t>(), E.tile_words.as<uint32_t>(), (uint32_t)ablation_knob(2));
    else if (mode == kDedupWp)
      hipLaunchKernelGGL(wordref_kernel<kDedupWp>, grid, dim3(64), 0, st, d_text, n_bytes, d_sent_off, plan1, d_cls, D, wref,
                         ws.sent_local.as<uint32_t>(), E.tile_words.as<uint32_t>(), (uint32_t)ablation_knob(2));
    else if (!split_flag)
      hipLaunchKernelGGL((wordref_kernel<kDedupBpe, true>), grid, dim3(64), 0, st, d_text, n_bytes, d_sent_off, plan1, d_cls, D,
                         wref, ws.sent_local.as<uint32_t>(), E.tile_words.as<uint32_t>(), (uint32_t)ablation_knob(2));
    else
      hipLaunchKernelGGL(wordref_kernel<kDedupBpe>, grid, dim3(64), 0, st, d_text, n_bytes, d_sent_off, plan1, d_cls, D, wref,
                         ws.sent_local.as<uint32_t>(), E.tile_words.as<uint32_t>(), (uint32_t)ablation_knob(2));
  }
  prof_end(st, 3);
  launch_scan_u64(n_tiles, D.tile_new, new_local, new_blk, reinterpret_cast<uint64_t *>(d_misc), st);
  if (mode == kDedupWp)
    hipLaunchKernelGGL(ureg_kernel<kDedupWp>, dim3((unsigned)n_tiles), dim3(64), 0, st, d_text, d_sent_off, plan1, D, new_local,
                       new_blk + 1 + nb_new, d_misc, E.uslot.as<uint32_t>(), E.uoff.as<uint64_t>(), E.utext.as<uint8_t>(), d_plan2,
                       n_tiles2, tile2_min);
  else
    hipLaunchKernelGGL(ureg_kernel<kDedupBpe>, dim3((unsigned)n_tiles), dim3(64), 0, st, d_text, d_sent_off, plan1, D, new_local,
                       new_blk + 1 + nb_new, d_misc, E.uslot.as<uint32_t>(), E.uoff.as<uint64_t>(), E.utext.as<uint8_t>(), d_plan2,
                       n_tiles2, tile2_min);
  SWT_HIP(hipGetLastError());
  if (D.diag & 1u) {
    unsigned long long h[3] = {0, 0, 0};
    SWT_HIP(hipMemcpyAsync(h, d_misc, 24, hipMemcpyDeviceToHost, st));
    SWT_HIP(hipStreamSynchronize(st));
    const unsigned int *c = reinterpret_cast<const unsigned int *>(h + 1);
    fprintf(stderr, "[swt] dedup: %llu unique words, %llu bytes; CAS %u inserted, %u lost to another lane\n", h[0] >> 32,
            h[0] & 0xFFFFFFFFull, c[1], c[2]);
  }
  return SWT_OK;
}

int dedup_back(DedupEngine &E, TileWorkspace &ws, const uint64_t *d_sent_off, uint64_t n_sent, uint64_t n_bytes,
               const uint32_t *d_unique_tokens, DedupMode mode, uint8_t *d_status, uint32_t *d_out_ids, uint64_t *d_out_off,
               uint64_t *d_n_tokens, hipStream_t st) {
  const uint64_t n_tiles = tile_count(n_bytes, kDTile);
  const uint64_t nb = (n_tiles + 1023) / 1024;
  uint32_t *wref = ws.scratch.as<uint32_t>();
  const uint64_t *plan1 = ws.plan.as<uint64_t>();
  const unsigned long long *rec = E.rec.as<unsigned long long>(), *drec = E.drec.as<unsigned long long>();
  const unsigned long long *blk_base = ws.blk.as<unsigned long long>() + 1 + nb;
  if (mode == kDedupWp) {
    hipLaunchKernelGGL(wp_refs_kernel<false>, dim3((unsigned)n_tiles), dim3(64), 0, st, d_sent_off, plan1, n_tiles, n_sent, wref,
                       E.tile_words.as<uint32_t>(), ws.sent_local.as<uint32_t>(), rec, d_unique_tokens, ws.tile_base.as<uint32_t>(),
                       blk_base, d_n_tokens, ws.tile_tok.as<uint32_t>(), d_out_ids, d_out_off, d_status);
    launch_scan_only(n_tiles, ws, d_n_tokens, st);
    hipLaunchKernelGGL(wp_refs_kernel<true>, dim3((unsigned)n_tiles), dim3(64), 0, st, d_sent_off, plan1, n_tiles, n_sent, wref,
                       E.tile_words.as<uint32_t>(), ws.sent_local.as<uint32_t>(), drec, d_unique_tokens, ws.tile_base.as<uint32_t>(),
                       blk_base, d_n_tokens, ws.tile_tok.as<uint32_t>(), d_out_ids, d_out_off, d_status);
  } else {
    hipLaunchKernelGGL(refcount_kernel, dim3((unsigned)n_tiles), dim3(64), 0, st, d_sent_off, plan1, wref, E.tile_words.as<uint32_t>(), rec,
                       ws.tile_tok.as<uint32_t>());
    launch_scan_only(n_tiles, ws, d_n_tokens, st);
    hipLaunchKernelGGL(refwrite_kernel, dim3((unsigned)n_tiles), dim3(64), 0, st, d_sent_off, plan1, n_tiles, n_sent, wref,
                       E.tile_words.as<uint32_t>(), ws.sent_local.as<uint32_t>(), drec, d_unique_tokens, ws.tile_base.as<uint32_t>(),
                       blk_base, d_n_tokens, d_out_ids, d_out_off);
  }
  SWT_HIP(hipGetLastError());
  return SWT_OK;
}

}  // namespace swt
