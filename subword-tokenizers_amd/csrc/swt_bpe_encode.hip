// swt_bpe_encode.hip -- FastBPE batched encode on gfx950.
//
// Replaces, for a whole batch of sentences at once:
//   SubwordTokenizer.preprocessing   /root/reference/source/utils.py:26-29   (split only; lower() is the caller's)
//   FastBPE.tokenize                 /root/reference/source/bpe.py:245-249
//   FastBPE.encode_word / _pairs     /root/reference/source/bpe.py:202-243
//   the rank dict                    /root/reference/source/bpe.py:200,257
//
// One 64-lane wavefront per tile (workgroup = one wave, so every barrier is a wave-local fence):
//   tile   = the sentences whose first byte lies in one window of the text (whole sentences, no data-path atomics between
//            workgroups, a tile's tokens are contiguous in the output)
//   chunk  = the part of the tile's span staged in LDS at a time; longer spans are cut at word boundaries
// Two kernels share that skeleton and the two-choice rank table (slot_value):
//   bpe_lane_kernel   (the default, round 3; described where it stands, below): the split stays byte-parallel and looks every
//                     adjacent pair up on the way, then ONE LANE OWNS ONE WORD through the merge rounds (live-slot mask, a lane takes
//                     the next word when its own is finished, four lanes a word once few words are left)
//   bpe_encode_kernel (rounds 1-2, SWT_BPE_KERNEL=bytes: the comparison form): one lane per BYTE POSITION in every phase --
//     B/C  64 bytes per step: decode the code point at each UTF-8 lead byte, class from an LDS copy of the table,
//          then everything structural (word starts, next symbol of the word, head of the word) comes from 64-bit
//          ballot masks and scalar bit arithmetic
//     C2/D the table value of every adjacent pair, then merge rounds over the list of symbols that still belong to an
//          unfinished word: per-word minimum by LDS atomicMin, the winners merge in place (symbols are linked by a
//          next-pointer, nothing is shifted), and only symbols next to a merge look the table up again
//     E/F  order-preserving ballot compaction to the tile's output run, per-sentence offsets
// A word longer than a chunk falls to a one-lane global-memory path (correct, slow, pathological inputs only).
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <utility>

#include "swt_dedup.h"
#include "swt_tile.h"
#include "swt_words.h"

namespace swt {

struct alignas(16) BpeSlot {
  uint64_t key;     // left << 32 | right, kEmptyKey when free
  uint32_t rank;    // table value: rank, or rank << 16 | (merged - SWT_SYM_BASE) in a packed table
  uint32_t merged;  // symbol id of left+right
};

constexpr uint8_t kClsWs = 1, kClsPunct = 2;
constexpr uint32_t kNoRank = 0xFFFFFFFFu;
#ifndef SWT_CLS_LDS
#define SWT_CLS_LDS 1024
#endif
constexpr int kClsLds = SWT_CLS_LDS;    // code points whose class is served from LDS (16 B a lane)

#ifndef SWT_BPE_TILE
#define SWT_BPE_TILE 192
#endif
#ifndef SWT_BPE_CAP
#define SWT_BPE_CAP 256
#endif
constexpr int kBpeTile = SWT_BPE_TILE;   // bytes of sentence starts per tile.  Measured on S85k-open with 256-byte chunks (tools/gpu_enc_sweep.sh):
                                         // 96: 489 us, 128: 481, 160: 467, 192: 453, 224: 456, 256: 460; 512-byte chunks: 558 (fewer resident waves)
constexpr int kBpeCap = SWT_BPE_CAP;    // staged bytes per chunk
#ifndef SWT_LANE_TILE
#define SWT_LANE_TILE 384
#endif
// (the tail of the merge rounds, lane_tail: four lanes a word once 16 words are left in a chunk.  Measured on S85k-open, ms per call:
// no tail 0.187; 2 lanes a word from 32 words 0.184; 4 from 16: 0.176; 8 from 8: 0.180; 4 from 16 and then 8 from 8: 0.1775)
#ifndef SWT_LANE_CAP
#define SWT_LANE_CAP 512
#endif
constexpr int kLaneTile = SWT_LANE_TILE;  // the word-lane kernel (bpe_lane_kernel): bytes of sentence starts per tile ...
constexpr int kLaneCap = SWT_LANE_CAP;    // ... and staged bytes per chunk (a batch of 64 word lanes wants ~350 bytes of text)
constexpr uint64_t kDirectBytes = 1024, kDirectSents = 64;  // up to here one workgroup and one launch do the whole call
constexpr uint32_t kNoPos = 0xFFFFu;

// The rank table is a two-choice cuckoo table (built once on the host, swt_bpe_table_create): a pair lives in slot h1 or in
// slot h2, so a lookup is two independent 16-byte loads and two compares -- no probe loop, and a wave never goes round
// again because one of its lanes met a collision.  The hashes are 24-bit multiplies (full-rate v_mul_u32_u24; symbol ids
// are below 2^24 for any table under five million merges, larger ids only hash worse) with one xor-shift between them.
struct BpeHash { uint32_t a, b, c; };
constexpr BpeHash kHash1{0x9E3779u, 0x85EBCBu, 0xC2B2AFu}, kHash2{0x27D4EBu, 0x165667u, 0x9E3779u};
__host__ __device__ __forceinline__ uint32_t bpe_hash(uint32_t l, uint32_t r, uint32_t sh, BpeHash k) {
  uint32_t x = (l & 0xFFFFFFu) * k.a + (r & 0xFFFFFFu) * k.b;
  x ^= x >> 16;
  return ((x & 0xFFFFFFu) * k.c) >> sh;   // sh = 32 - log2(slots)
}

__device__ __forceinline__ bool slot_lookup(const BpeSlot *__restrict__ slots, uint32_t sh, uint32_t l, uint32_t r,
                                            uint32_t &rank, uint32_t &merged) {
  const uint4 r1 = *reinterpret_cast<const uint4 *>(&slots[bpe_hash(l, r, sh, kHash1)]);
  const uint4 r2 = *reinterpret_cast<const uint4 *>(&slots[bpe_hash(l, r, sh, kHash2)]);
  const bool h1 = r1.x == r && r1.y == l, h2 = r2.x == r && r2.y == l;
  rank = h1 ? r1.z : r2.z;
  merged = h1 ? r1.w : r2.w;
  return h1 || h2;
}

__device__ __forceinline__ uint32_t slot_value(const BpeSlot *__restrict__ slots, uint32_t sh, uint32_t l, uint32_t r) {
  const uint4 r1 = *reinterpret_cast<const uint4 *>(&slots[bpe_hash(l, r, sh, kHash1)]);
  const uint4 r2 = *reinterpret_cast<const uint4 *>(&slots[bpe_hash(l, r, sh, kHash2)]);
  // both loads are issued before either is looked at; a pair lives in at most one slot and kNoRank is all ones, so the two
  // selects combine with AND (written as a select of a select the compiler makes the second load wait for the first compare)
  const uint32_t a = (r1.x == r && r1.y == l) ? r1.z : kNoRank;
  const uint32_t b = (r2.x == r && r2.y == l) ? r2.z : kNoRank;
  return a & b;
}

// FastBPE.encode_word on a symbol array (bpe.py:210-238), serial form for the one-lane fallback: repeat { lowest-rank
// adjacent pair; replace all of its occurrences left to right, non-overlapping }.  Returns the new length.
__device__ uint32_t merge_word(uint32_t *s, uint32_t n, const BpeSlot *__restrict__ slots, uint32_t sh) {
  while (n >= 2) {
    uint32_t best = 0xFFFFFFFFu, bm = 0, bl = 0, br = 0;
    uint32_t a = s[0];
    for (uint32_t i = 0; i + 1 < n; i++) {
      const uint32_t b = s[i + 1];
      uint32_t rk, mg;
      if (slot_lookup(slots, sh, a, b, rk, mg) && rk < best) { best = rk; bm = mg; bl = a; br = b; }
      a = b;
    }
    if (best == 0xFFFFFFFFu) break;
    uint32_t j = 0, i = 0;
    while (i < n) {
      const uint32_t x = s[i];
      if (i + 1 < n && x == bl && s[i + 1] == br) { s[j++] = bm; i += 2; }
      else { s[j++] = x; i++; }
    }
    n = j;
  }
  return n;
}

struct GiantResult { uint64_t end; uint32_t ntok; };

// One lane, global memory only: the word (or single separator) starting at byte `pos`, bounded by
// `send` (end of its sentence).  Tokens go to `out` (which doubles as the symbol workspace).
__device__ GiantResult giant_word(const uint8_t *__restrict__ text, uint64_t pos, uint64_t send,
                                  const uint8_t *__restrict__ cls_tab, const BpeSlot *__restrict__ slots, uint32_t sh,
                                  uint32_t *out) {
  GiantResult r{pos, 0};
  uint32_t n = 0;
  bool first = true;
  while (r.end < send) {
    const uint8_t b = text[r.end];
    int len = utf8_len(b);
    if (r.end + len > send) len = (int)(send - r.end);
    uint32_t cp = b;
    if (b >= 0x80 && len > 1) {
      cp = b & (0xFF >> (len + 1));
      for (int i = 1; i < len; i++) cp = (cp << 6) | (text[r.end + i] & 0x3F);
    }
    const uint8_t c = utf8_is_cont(b) ? kClsWs : ((cls_tab && cp < kNumCodePoints) ? cls_tab[cp] : (uint8_t)0);
    if (c & kClsWs) { if (first) r.end += len; break; }
    if (c & kClsPunct) { if (first) { out[n++] = cp; r.end += len; } break; }
    out[n++] = cp;
    r.end += len;
    first = false;
  }
  n = merge_word(out, n, slots, sh);
  for (uint32_t i = 1; i < n; i++) out[i] |= SWT_BPE_CONT;
  r.ntok = n;
  return r;
}

template <int Cap>
struct BpeLds {
  static constexpr int Blocks = Cap / 64;
  __attribute__((aligned(16))) uint8_t txt[Cap + 16];
  uint32_t sym[Cap];           // per byte position: symbol id (token id at the end) or kInvalidTok
  uint16_t nxt[Cap];           // per byte position: next live symbol of the same word, kNoPos for the last one
  uint32_t act[Cap];           // symbols of unfinished words: position | head position << 16
  uint32_t aval[Cap];          // per list entry: table value of (this symbol, next symbol); kNoRank when none
  uint32_t wm[2][Cap / 2];     // per word (indexed by head >> 1): minimum table value, this round / next round
  unsigned long long sbits[Blocks + 1];  // sentence-start bit per byte
  unsigned long long mark[Blocks + 1];   // "my predecessor is a candidate of a twin pair (a,a)"
  unsigned long long tk[Blocks + 1];     // taken: this symbol merges with its next one this round
  unsigned long long dead[Blocks + 1];   // consumed by the symbol before it this round
  unsigned long long vmask[Blocks + 1];  // phase E
  uint32_t blkpre[Blocks + 1];
  __attribute__((aligned(16))) uint8_t cls_lo[kClsLds];  // classes of U+0000..U+03FF
  GiantResult giant;
};

__device__ __forceinline__ bool bit_at(const unsigned long long *m, uint32_t p) { return (m[p >> 6] >> (p & 63)) & 1ull; }

// Packed = true: the table value of a pair is rank << 16 | (merged - SWT_SYM_BASE), so a merge round learns the
// merged symbol without touching memory (tables below 65,534 merges); false: the value is the rank and the merged
// symbol is read from merged_of_rank[].
// Mode: 0 = tiles of running text (plan, sent_local / tile_tok for the scan + gather), 1 = the unique-word pass of the dedup
// path (every "sentence" is a unique word: rec / drec / uslot), 2 = one workgroup writing the caller's arrays (DirectOut).  A
// template parameter, not a run-time test: each form keeps only its own arguments in scalar registers (one kernel for all
// three spilled 44 of them).
template <bool Packed, int Cap, int Mode>
__global__ __launch_bounds__(64) void bpe_encode_kernel(
    const uint8_t *__restrict__ text, uint64_t n_bytes, const uint64_t *__restrict__ sent_off,
    const uint64_t *__restrict__ plan, const uint8_t *__restrict__ cls_tab, const BpeSlot *__restrict__ slots,
    uint32_t sh, const uint32_t *__restrict__ merged_of_rank, uint32_t *__restrict__ scratch,
    uint32_t *__restrict__ sent_local, uint32_t *__restrict__ tile_tok, const uint32_t *__restrict__ uslot,
    unsigned long long *__restrict__ rec, unsigned long long *__restrict__ drec, DirectOut direct, uint32_t dbg_arg) {
#ifdef SWT_ABLATION
  const uint32_t dbg = dbg_arg;
#else
  constexpr uint32_t dbg = 0;  // the ablation switches exist in -DSWT_ABLATION builds only
  (void)dbg_arg;
#endif
  constexpr bool kDirect = Mode == 2, kRec = Mode == 1;
  // uslot/rec/drec (dedup path, every "sentence" s is unique word s): the word's token run -- its place in scratch and its
  // length -- goes straight to drec[s] (dense: stays in L2 for the last pass) and length | s to the word's table slot, and
  // nobody needs a scan or a gather of this launch's output.
  constexpr int Blocks = Cap / 64;
  __shared__ BpeLds<Cap> L;
  const int lane = threadIdx.x;
  const unsigned long long lt = (1ull << lane) - 1ull;  // lanes below me
  const unsigned long long le = (2ull << lane) - 1ull;  // me and below
  const uint64_t t = blockIdx.x;
  const uint64_t s_lo = kDirect ? 0 : plan[t], s_hi = kDirect ? direct.n_sent : plan[t + 1];
  if (s_lo == s_hi) {
    if (Mode == 0 && lane == 0) tile_tok[t] = 0;
    return;
  }
  {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (lane < kClsLds / 16) {
      if (cls_tab) v = reinterpret_cast<const uint4 *>(cls_tab)[lane];
      reinterpret_cast<uint4 *>(L.cls_lo)[lane] = v;
    }
  }
  const uint64_t span_base = sent_off[s_lo], span_end = sent_off[s_hi];
  uint32_t *const tile_out = scratch + span_base;
  uint32_t run = 0;        // tokens emitted by this tile so far
  uint64_t s_next = s_lo;  // first sentence whose local offset is not recorded yet
  uint64_t cb = span_base;

  for (;;) {
    const uint64_t abase = cb & ~15ull;
    const uint32_t off0 = (uint32_t)(cb - abase);
    const uint64_t avail = span_end - abase;
    const bool last = avail <= (uint64_t)Cap;
    const uint32_t staged = last ? (uint32_t)avail : (uint32_t)Cap;
    const uint32_t nblk = (staged + 63) >> 6;

    // ---- A. stage [abase, abase+staged): one dwordx4 per lane
    for (uint32_t c = lane * 16; c < staged; c += 64 * 16) {
      const uint64_t g = abase + c;
      if (g + 16 <= n_bytes && ((reinterpret_cast<uintptr_t>(text + g) & 15) == 0)) {
        *reinterpret_cast<uint4 *>(&L.txt[c]) = *reinterpret_cast<const uint4 *>(text + g);
      } else {
        for (int i = 0; i < 16; i++) L.txt[c + i] = (g + i < n_bytes) ? text[g + i] : (uint8_t)' ';
      }
    }
    if (lane <= Blocks) { L.sbits[lane] = 0ull; L.mark[lane] = 0ull; L.tk[lane] = 0ull; L.dead[lane] = 0ull; }
    __syncthreads();
    for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
      const uint64_t o = sent_off[s];
      if (o >= abase + staged) break;
      if (o >= cb) atomicOr(&L.sbits[(o - abase) >> 6], 1ull << ((o - abase) & 63));
    }
    __syncthreads();
    if (dbg & 1) { if (last) break; cb = abase + staged; continue; }  // ablation: staging only

    // ---- B/C. 64 bytes per step.  Scalars carried from block to block:
    uint32_t na = 0;          // active-list length
    bool prev_wb = true;      // the byte before this block belongs to a whitespace/punctuation char (or chunk start)
    uint32_t pend = kNoPos;   // last symbol of the previous block whose word may continue, and its data
    uint32_t pend_head = 0;
    bool pend_listed = false;
    int cut = -1;             // last word boundary (for a span longer than the chunk)
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
      uint8_t c = kClsWs;  // bytes outside the chunk behave as whitespace
      if (inr && lead) c = cp < (uint32_t)kClsLds ? L.cls_lo[cp] : ((cls_tab && cp < kNumCodePoints) ? cls_tab[cp] : (uint8_t)0);
      const unsigned long long INR = __ballot(inr);
      const unsigned long long LEAD = __ballot(lead);
      const unsigned long long WSm = __ballot(lead && (c & kClsWs));
      const unsigned long long PNm = __ballot(lead && (c & kClsPunct));
      const unsigned long long CONT = ~LEAD;
      // bytes that belong to a whitespace/punctuation char (continuation bytes inherit from their lead byte)
      unsigned long long WB = WSm | PNm | ((prev_wb && (CONT & 1ull)) ? 1ull : 0ull);
      WB |= (WB << 1) & CONT;
      WB |= (WB << 1) & CONT;
      WB |= (WB << 1) & CONT;
      const unsigned long long SS = L.sbits[blk];
      const unsigned long long first_bit = blk == 0 ? (1ull << off0) : 0ull;  // off0 < 16
      const unsigned long long before = (WB << 1) | (prev_wb ? 1ull : 0ull) | SS | first_bit;
      const unsigned long long SYM = LEAD & ~WSm & INR;
      const unsigned long long WSTART = SYM & (PNm | before);
      // chunk cut candidates: word boundaries strictly inside, with room for a whole UTF-8 char behind them
      {
        const unsigned long long CUT = LEAD & (WSm | PNm | SS) & __ballot(inr && p > off0 && p + 4 <= staged);
        if (CUT) cut = (int)(blk * 64 + 63 - __builtin_clzll(CUT));
      }
      const bool is_sym = (SYM >> lane) & 1ull;
      const bool wstart = (WSTART >> lane) & 1ull;
      const unsigned long long hm = WSTART & le;
      const uint32_t head = hm ? blk * 64 + 63 - __builtin_clzll(hm) : pend_head;
      const unsigned long long after = SYM & ~le;
      const uint32_t q = after ? (uint32_t)__builtin_ctzll(after) : 64u;
      const bool hasnext = is_sym && q < 64 && !((WSTART >> (q & 63)) & 1ull);
      // does the symbol left pending by the previous block continue into this one?
      const uint32_t f = SYM ? (uint32_t)__builtin_ctzll(SYM) : 64u;
      const bool joins = pend != kNoPos && f < 64 && !((WSTART >> (f & 63)) & 1ull);
      const bool i_join = joins && lane == (int)f;
      if (i_join) L.nxt[pend] = (uint16_t)p;
      L.sym[p] = is_sym ? cp : kInvalidTok;
      L.nxt[p] = hasnext ? (uint16_t)(blk * 64 + q) : (uint16_t)kNoPos;
      // active list: symbols of words with at least two symbols (a lone word start waits for its successor)
      const bool active = is_sym && (!wstart || hasnext);
      const unsigned long long ACT = __ballot(active);
      const uint32_t extra = (joins && !pend_listed) ? 1u : 0u;  // the pending word start now has a successor
      if (i_join && extra) { L.act[na] = pend | (pend_head << 16); L.wm[0][pend_head >> 1] = kNoRank; }
      if (active) {
        L.act[na + extra + __popcll(ACT & lt)] = p | (head << 16);
        if (p == head) L.wm[0][head >> 1] = kNoRank;
      }
      na += extra + __popcll(ACT);
      // carries
      if (SYM) {
        const int li = 63 - __builtin_clzll(SYM);
        const unsigned long long tail = li == 63 ? 0ull : ~((2ull << li) - 1ull);
        const bool open = ((WSm | PNm | SS) & tail) == 0ull && !((PNm >> li) & 1ull);
        pend = open ? blk * 64 + li : kNoPos;
        pend_head = __shfl(head, li);
        pend_listed = ((ACT >> li) & 1ull) != 0ull;
      } else {
        pend = kNoPos;  // a block without symbols holds whitespace: every word ended
      }
      prev_wb = (WB >> 63) & 1ull;
    }
    __syncthreads();

    // ---- chunk end
    uint32_t ce = staged;
    if (!last) {
      if (cut < 0) {
        // a single word longer than the LDS chunk: one lane, global memory
        if (lane == 0) {
          uint64_t s = s_next;
          while (s < s_hi && sent_off[s] <= cb) s++;
          const uint64_t send = sent_off[s];  // s <= s_hi and sent_off[s_hi] = span_end > cb
          L.giant = giant_word(text, cb, send, cls_tab, slots, sh, tile_out + run);
        }
        __syncthreads();
        const GiantResult g = L.giant;
        uint32_t mine = 0;
        for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
          if (sent_off[s] >= g.end) break;
          if (kDirect) direct.off[s] = run; else if (Mode == 0) sent_local[s] = run;
          if (kRec) {
            drec[s] = (unsigned long long)(span_base + run) | ((unsigned long long)g.ntok << 32);
            rec[uslot[s]] = (unsigned long long)s | ((unsigned long long)g.ntok << 32);
          }
          mine++;
        }
        for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
        s_next += mine;
        run += g.ntok;
        cb = g.end;
        __syncthreads();
        continue;
      }
      ce = (uint32_t)cut;
      // drop list entries at or beyond the cut (the next chunk stages them again)
      uint32_t keep = 0;
      for (uint32_t k0 = 0; k0 < na; k0 += 64) {
        const uint32_t k = k0 + lane;
        const uint32_t e = k < na ? L.act[k] : 0u;
        const bool ok = k < na && (e & 0xFFFFu) < ce;
        const unsigned long long M = __ballot(ok);
        __syncthreads();
        if (ok) L.act[keep + __popcll(M & lt)] = e;
        keep += __popcll(M);
        __syncthreads();
      }
      na = keep;
    }

    // ---- C2. the table value of every adjacent pair of every listed symbol: the whole first merge round
    // (bpe.py:211-219), four independent probes in flight per lane; each value also goes into its word's minimum
    for (uint32_t k0 = 0; k0 < na; k0 += 256) {
      uint32_t pl[4], pr[4], hd[4];
      bool want[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t k = k0 + u * 64 + lane;
        want[u] = false;
        pl[u] = pr[u] = hd[u] = 0;
        if (k < na && !(dbg & 4)) {
          const uint32_t e = L.act[k];
          const uint32_t p = e & 0xFFFFu;
          const uint32_t qn = L.nxt[p];
          hd[u] = e >> 17;
          if (qn != kNoPos) {
            want[u] = true;
            pl[u] = L.sym[p];
            pr[u] = L.sym[qn];
          }
        }
      }
      uint32_t vv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) vv[u] = slot_value(slots, sh, pl[u], pr[u]);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t k = k0 + u * 64 + lane;
        const uint32_t v = want[u] ? vv[u] : kNoRank;
        if (v != kNoRank && !(dbg & 8)) atomicMin(&L.wm[0][hd[u]], v);
        if (k < na) L.aval[k] = v;
      }
    }
    __syncthreads();

    // ---- D. merge rounds (bpe.py:210-238), all lanes on the symbols of unfinished words.  Two passes per round:
    //   X  candidates = left symbols of their word's best pair; they publish "taken" (tk) and "consumed" (dead) bits.
    //      A pair of two different symbols cannot overlap itself, so such a candidate always merges; a twin pair
    //      (a,a) in a run like "aaaa" merges left to right, non-overlapping (bpe.py:225-235): every second member,
    //      decided by the run's first member.
    //   Y  every listed symbol updates itself from those bits (merged symbol, next pointer), re-probes the table only
    //      if its pair changed, feeds the next round's per-word minimum, and the list is compacted.
    uint32_t cur = 0, round_no = 0;
    while (na > 0 && !(dbg & 16)) {
      if (na <= 64 && dbg == 0) {
        // ---- register rounds: once the symbols of the unfinished words fit one wave (always, after a few rounds; from the
        // start for the short tiles of the unique-word pass) every entry lives in a lane.  The entries of a word are
        // consecutive lanes in text order, so "my next symbol" is the lane above me, a round is ballots and shuffles, and
        // the dependent LDS chain of the list form (~30 accesses per round) shrinks to the word minimum and the compaction.
        bool valid = (uint32_t)lane < na;
        uint32_t e = valid ? L.act[lane] : 0u, v = valid ? L.aval[lane] : kNoRank;
        uint32_t p = e & 0xFFFFu, head = e >> 16;
        uint32_t sy = valid ? L.sym[p] : 0u;
        for (;;) {
          const unsigned long long VALID = __ballot(valid);
          if (!VALID) break;
          const uint32_t hw = head >> 1;
          const uint32_t m = valid ? L.wm[cur][hw] : kNoRank;
          __syncthreads();
          if (valid && p == head) L.wm[cur ^ 1][hw] = kNoRank;
          const uint32_t head1 = __shfl_down(head, 1), head2 = __shfl_down(head, 2);
          const uint32_t sy1 = __shfl_down(sy, 1), sy2 = __shfl_down(sy, 2);
          const bool same1 = lane < 63 && ((VALID >> (lane + 1)) & 1ull) && head1 == head;
          const bool same2 = same1 && lane < 62 && ((VALID >> (lane + 2)) & 1ull) && head2 == head;
          const bool cand = valid && m != kNoRank && v == m;  // v is the value of (me, lane above me): that lane exists
          const bool twin = cand && sy1 == sy;
          const unsigned long long TW = __ballot(twin);
          bool taken = cand;
          if (twin) {
            // a run of twin pairs ("aaaa") merges left to right, non-overlapping (bpe.py:225-235): every second member,
            // counted from the run's first lane (runs of different words never touch: a word's last symbol is no candidate)
            const unsigned long long low = ~TW & lt;
            const int start = low ? 64 - __builtin_clzll(low) : 0;
            taken = ((lane - start) & 1) == 0;
          }
          const unsigned long long TK = __ballot(taken);
          const bool dead = lane > 0 && ((TK >> (lane - 1)) & 1ull);  // the lane below me merged me in
          const bool tk1 = same1 && ((TK >> (lane + 1)) & 1ull), tk2 = same2 && ((TK >> (lane + 2)) & 1ull);
          uint32_t mg = 0;
          if (valid && m != kNoRank) mg = Packed ? (SWT_SYM_BASE + (m & 0xFFFFu)) : merged_of_rank[m];
          const bool keep = valid && m != kNoRank && !dead;
          if (valid && !keep) L.sym[p] = dead ? kInvalidTok : (p != head ? (sy | SWT_BPE_CONT) : sy);  // consumed, or word finished
          // my symbol and the symbol above me after this round
          const uint32_t sn = taken ? mg : sy;
          const bool has_r = taken ? same2 : same1;
          const uint32_t sr = taken ? (tk2 ? mg : sy2) : (tk1 ? mg : sy1);
          uint32_t vn = v;
          if (!has_r) vn = kNoRank;
          else if (taken || tk1) vn = slot_value(slots, sh, sn, sr);
          __syncthreads();  // the reset of the next round's minima (above) comes before the lanes feed them
          if (keep && vn != kNoRank) atomicMin(&L.wm[cur ^ 1][hw], vn);
          const unsigned long long KEEP = __ballot(keep);
          if (keep) {
            const uint32_t d = (uint32_t)__popcll(KEEP & lt);
            L.act[d] = e;
            L.aval[d] = vn;
            L.sym[p] = sn;
          }
          __syncthreads();
          valid = (uint32_t)lane < (uint32_t)__popcll(KEEP);
          if (valid) {
            e = L.act[lane];
            v = L.aval[lane];
            p = e & 0xFFFFu;
            head = e >> 16;
            sy = L.sym[p];
          }
          cur ^= 1;
        }
        na = 0;
        break;
      }
      round_no++;
      const bool stop_now = (dbg >> 8) && round_no > (dbg >> 8);  // ablation: bounded rounds
      bool twin = false;
      for (uint32_t k = lane; k < na; k += 64) {
        const uint32_t e = L.act[k];
        const uint32_t v = L.aval[k];
        const uint32_t p = e & 0xFFFFu, w = e >> 17;
        const uint32_t m = stop_now ? kNoRank : L.wm[cur][w];
        if (p == (e >> 16)) L.wm[cur ^ 1][w] = kNoRank;
        if (m != kNoRank && v == m) {
          const uint32_t qn = L.nxt[p];
          if (L.sym[p] == L.sym[qn]) {
            atomicOr(&L.mark[qn >> 6], 1ull << (qn & 63));
            twin = true;
          } else {
            atomicOr(&L.tk[p >> 6], 1ull << (p & 63));
            atomicOr(&L.dead[qn >> 6], 1ull << (qn & 63));
          }
        }
      }
      twin = __any(twin);
      __syncthreads();
      if (twin) {
        for (uint32_t k = lane; k < na; k += 64) {
          const uint32_t e = L.act[k];
          uint32_t p = e & 0xFFFFu;
          const uint32_t m = L.wm[cur][e >> 17];
          if (m != kNoRank && L.aval[k] == m && L.sym[p] == L.sym[L.nxt[p]] && !bit_at(L.mark, p)) {
            // first member of a run of twins: take, skip, take, ...  (a member's pair value equals m iff the symbol
            // after it is the same symbol again)
            const uint32_t a = L.sym[p];
            for (;;) {
              const uint32_t qn = L.nxt[p];
              if (qn == kNoPos || L.sym[qn] != a) break;
              atomicOr(&L.tk[p >> 6], 1ull << (p & 63));
              atomicOr(&L.dead[qn >> 6], 1ull << (qn & 63));
              p = L.nxt[qn];
              if (p == kNoPos || L.sym[p] != a) break;
            }
          }
        }
        __syncthreads();
      }
      uint32_t keep = 0;
      for (uint32_t k0 = 0; k0 < na; k0 += 64) {
        const uint32_t k = k0 + lane;
        uint32_t e = 0, v = kNoRank;
        bool stay = false;
        if (k < na) {
          e = L.act[k];
          v = L.aval[k];
          const uint32_t p = e & 0xFFFFu, h = e >> 16;
          const uint32_t m = stop_now ? kNoRank : L.wm[cur][h >> 1];
          if (bit_at(L.dead, p)) {
            L.sym[p] = kInvalidTok;  // consumed by the symbol before it; nobody reads a consumed symbol again
          } else {
            uint32_t sp = L.sym[p];
            if (m == kNoRank) {
              if (p != h) L.sym[p] = sp | SWT_BPE_CONT;  // word finished: '##' on all but its first token (bpe.py:240-241)
            } else {
              stay = true;
              const uint32_t mg = Packed ? (SWT_SYM_BASE + (m & 0xFFFFu)) : merged_of_rank[m];
              uint32_t nq = L.nxt[p];
              bool dirty;
              if (bit_at(L.tk, p)) {
                nq = L.nxt[nq];  // my partner is consumed; its successor becomes mine
                sp = mg;
                L.sym[p] = mg;
                L.nxt[p] = (uint16_t)nq;
                dirty = true;
              } else {
                dirty = nq != kNoPos && bit_at(L.tk, nq);
              }
              if (nq == kNoPos) v = kNoRank;
              else if (dirty) v = slot_value(slots, sh, sp, bit_at(L.tk, nq) ? mg : L.sym[nq]);
              if (v != kNoRank) atomicMin(&L.wm[cur ^ 1][h >> 1], v);
            }
          }
        }
        const unsigned long long M = __ballot(stay);
        __syncthreads();
        if (stay) {
          const uint32_t d = keep + __popcll(M & lt);
          L.act[d] = e;
          L.aval[d] = v;
        }
        keep += __popcll(M);
        __syncthreads();
      }
      na = keep;
      if (lane <= Blocks) { L.tk[lane] = 0ull; L.mark[lane] = 0ull; L.dead[lane] = 0ull; }
      cur ^= 1;
      __syncthreads();
    }

    // ---- E. order-preserving compaction of the valid positions into the tile's output run
    if (dbg & 32) { if (last) break; cb = abase + ce; continue; }
    uint32_t total = 0;
    for (uint32_t blk = 0; blk < nblk; blk++) {
      const uint32_t p = blk * 64 + lane;
      const uint32_t sv = (p >= off0 && p < ce) ? L.sym[p] : kInvalidTok;
      const unsigned long long m = __ballot(sv != kInvalidTok);
      if (lane == 0) { L.vmask[blk] = m; L.blkpre[blk] = total; }
      if (sv != kInvalidTok) tile_out[run + total + __popcll(m & lt)] = sv;
      total += __popcll(m);
    }
    __syncthreads();
    // ---- F. tile-local token offset of every sentence starting in [cb, ce) (and == ce on the last chunk)
    uint32_t mine = 0;
    for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
      const uint64_t rel = sent_off[s] - abase;
      if (rel > ce || (rel == ce && !last)) break;
      uint32_t e = total;
      if (rel < ce && (rel >> 6) < nblk) e = L.blkpre[rel >> 6] + __popcll(L.vmask[rel >> 6] & ((1ull << (rel & 63)) - 1ull));
      if (kDirect) direct.off[s] = run + e; else if (Mode == 0) sent_local[s] = run + e;
      if (kRec && rel < ce) {  // a word never straddles the cut, so its end lies in this chunk too
        const uint64_t rel2 = sent_off[s + 1] - abase;
        uint32_t e2 = total;
        if (rel2 < ce && (rel2 >> 6) < nblk) e2 = L.blkpre[rel2 >> 6] + __popcll(L.vmask[rel2 >> 6] & ((1ull << (rel2 & 63)) - 1ull));
        drec[s] = (unsigned long long)(span_base + run + e) | ((unsigned long long)(e2 - e) << 32);
        rec[uslot[s]] = (unsigned long long)s | ((unsigned long long)(e2 - e) << 32);
      }
      mine++;
    }
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    s_next += mine;
    run += total;
    if (last) break;
    cb = abase + ce;
    __syncthreads();
  }
  if (lane == 0) {
    if (kDirect) { direct.off[s_hi] = run; *direct.n_tokens = run; }
    else if (Mode == 0) tile_tok[t] = run;
  }
}


// ======================================================================================================================
// The word-lane form of the same call (round 3; the default): bpe_lane_kernel.
//
// The kernel above keeps one lane per BYTE through the merge rounds: a round costs the wave the same ~170 instructions
// whether 64 symbols take part or three, and a tile goes through as many rounds as its longest word has merges (the SQ
// counters of round 2: 19.7 scalar + 14 vector instructions per input byte, the scalar pipe bound).  Here the split stays
// byte-parallel (ballots), but it leaves the symbols DENSE (indexed by symbol, not by byte) together with the table value of
// every adjacent pair (probed right there, all lanes at once), and then one lane owns one WORD:
//   W    the multi-symbol words of the chunk, longest class first (9+, 5-8, 2-4 symbols)
//   D    lane = word.  The word's slots stay where the split put them; a 32-bit mask says which are still live.  A round =
//        leftmost minimum over the live slots' cached pair values (four slots per step) -> the pair merges in place, its
//        right slot dies, and the two pairs next to the merged symbol are probed again (four loads in flight).  A lane whose
//        word is finished takes the next word of the list.  On a PROPER table (every pair ranks above the merges that produce
//        its symbols: any trained table) merging one occurrence per round is the reference's "all occurrences of the best
//        pair, left to right" (bpe.py:221-235) -- what is left of the pair is still the minimum and is found leftmost-first
//        in the next round.  Tables without that property, and words beyond 32 symbols, take slow_word(): the same loop in
//        its literal form (all occurrences per round, compaction), one lane per word.
//   E/F  ballot compaction over the SYMBOL space, sentence offsets through the split's symbol masks.
// Measured and dropped (profiles/r03_experiments/bpe_lane_*.txt): one wave running the rounds for the words of four tiles
// (fewer instructions, but three waves of four idle meanwhile: 0.229 against 0.200 ms), tiles that place their own output by a
// decoupled look-back, and tiles that plan themselves (two launches instead of four).
// Token ids, offsets and the launches around the kernel (plan, scan, gather / the dedup records) are those of the kernel above.
template <int Cap>
struct LaneLds {
  static constexpr int Blocks = Cap / 64;
  __attribute__((aligned(16))) uint8_t txt[Cap + 16];  // staged bytes; once the split is done, a single-wave kernel's word list
  uint32_t sym[Cap + 4];      // per symbol: id (| SWT_BPE_CONT unless it opens its word), kInvalidTok once consumed
  uint32_t val[Cap + 4];      // per symbol: table value of (this symbol, next live symbol of the word), kNoRank when none
  uint16_t wl[Cap + 2];       // first symbol of every word, in text order
  unsigned long long sbits[Blocks + 1];    // sentence-start bit per byte
  unsigned long long symmask[Blocks + 1];  // symbol bit per byte
  unsigned long long vmask[Blocks + 1];    // phase E: live-token bit per symbol
  uint32_t sympre[Blocks + 1];             // symbols before each 64-byte block
  uint32_t blkpre[Blocks + 1];             // tokens before each block of 64 symbols
  uint32_t cls2[64];                       // classes of U+0000..U+03FF, two bits each
  GiantResult giant;
};

// What a wave carries from chunk to chunk of its tile, and what it knows about the chunk at hand.
struct LaneTile {
  uint64_t s_lo, s_hi;           // the tile's sentences
  uint64_t span_base, span_end;  // their bytes
  uint64_t s_next;               // first sentence whose local offset is not recorded yet
  uint64_t cb;                   // first byte not encoded yet
  uint32_t *tile_out;            // the tile's run in scratch
  uint32_t run;                  // tokens emitted so far
};
struct LaneChunk {
  uint64_t abase;                // 16-byte aligned base of the staged bytes
  uint32_t off0, staged, nblk;   // first byte of the chunk inside the staged bytes, staged bytes, 64-byte blocks
  uint32_t ce;                   // end of the chunk (the cut, or staged)
  uint32_t nsym, nw;             // symbols and words of the chunk
  int cut;                       // last word boundary (for a span longer than the chunk), -1 when none
  bool last, giant;              // the tile ends with this chunk; the chunk was one word longer than the staged bytes
};

// LDS traffic between the lanes of ONE wave needs no hardware barrier (a wave's LDS instructions execute in order); the
// compiler must not move accesses across the point, that is all.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr uint32_t kDirtyVal = 0xFFFFFFFEu;  // above every table value (ranks stay below 2^32 - 2), below kNoRank

// bpe.py:210-238 on one word whose symbols S[0..n) and pair values V[0..n) (V[n-1] = kNoRank) live in LDS: every round merges
// ALL occurrences of the best pair left to right, compacts the word and probes only the pairs that changed.
template <bool Packed>
__device__ void slow_word(uint32_t *S, uint32_t *V, const uint32_t n0, const BpeSlot *__restrict__ slots, uint32_t sh,
                          const uint32_t *__restrict__ merged_of_rank) {
  uint32_t n = n0;
  for (;;) {
    uint32_t m = kNoRank;
    for (uint32_t i = 0; i + 1 < n; i++) m = min(m, V[i]);
    if (m == kNoRank) break;
    const uint32_t mg = Packed ? (SWT_SYM_BASE + (m & 0xFFFFu)) : merged_of_rank[m];
    uint32_t i = 0, j = 0;
    while (i < n) {
      const uint32_t vi = V[i];
      const bool take = i + 1 < n && vi == m;
      const uint32_t s = take ? mg : (S[i] & ~SWT_BPE_CONT);
      if (take && j) V[j - 1] = kDirtyVal;
      S[j] = j ? (s | SWT_BPE_CONT) : s;
      V[j] = take ? kDirtyVal : vi;
      i += take ? 2u : 1u;
      j++;
    }
    n = j;
    V[n - 1] = kNoRank;
    for (uint32_t k = 0; k + 1 < n; k++)
      if (V[k] == kDirtyVal) V[k] = slot_value(slots, sh, S[k] & ~SWT_BPE_CONT, S[k + 1] & ~SWT_BPE_CONT);
  }
  for (uint32_t k = n; k < n0; k++) S[k] = kInvalidTok;
}

// classes of the first 1,024 code points: 16 per lane, two bits each (SWT_CLS_BERT_WS | SWT_CLS_BERT_PUNCT)
template <int Cap>
__device__ __forceinline__ void lane_classes(LaneLds<Cap> &L, const uint8_t *__restrict__ cls_tab, int lane) {
  uint32_t w = 0;
  if (cls_tab) {
    const uint4 v = reinterpret_cast<const uint4 *>(cls_tab)[lane];
    auto pk = [](uint32_t d) {
      uint32_t x = d & 0x03030303u;
      x = (x | (x >> 6)) & 0x000F000Fu;
      return (x | (x >> 12)) & 0xFFu;
    };
    w = pk(v.x) | (pk(v.y) << 8) | (pk(v.z) << 16) | (pk(v.w) << 24);
  }
  L.cls2[lane] = w;
}

// ---- A + B/C of one chunk: stage the bytes, then 64 bytes per step: classes, word structure from ballot masks (as in the
// kernel above), the dense symbols, the list of word starts, and the table value of every adjacent pair of a word.
template <int Cap>
__device__ __forceinline__ void lane_split(LaneLds<Cap> &L, const uint8_t *__restrict__ text, uint64_t n_bytes,
                                           const uint64_t *__restrict__ sent_off, const uint8_t *__restrict__ cls_tab,
                                           const BpeSlot *__restrict__ slots, uint32_t sh, const LaneTile &T, LaneChunk &C, int lane) {
  constexpr int Blocks = Cap / 64;
  const unsigned long long lt = (1ull << lane) - 1ull;  // lanes below me
  const unsigned long long le = (2ull << lane) - 1ull;  // me and below
  const uint64_t abase = T.cb & ~15ull;
  const uint32_t off0 = (uint32_t)(T.cb - abase);
  const uint64_t avail = T.span_end - abase;
  const bool last = avail <= (uint64_t)Cap;
  const uint32_t staged = last ? (uint32_t)avail : (uint32_t)Cap;
  const uint32_t nblk = (staged + 63) >> 6;

  // ---- A. stage [abase, abase+staged): one dwordx4 per lane
  for (uint32_t c = lane * 16; c < staged; c += 64 * 16) {
    const uint64_t g = abase + c;
    if (g + 16 <= n_bytes && ((reinterpret_cast<uintptr_t>(text + g) & 15) == 0)) {
      *reinterpret_cast<uint4 *>(&L.txt[c]) = *reinterpret_cast<const uint4 *>(text + g);
    } else {
      for (int i = 0; i < 16; i++) L.txt[c + i] = (g + i < n_bytes) ? text[g + i] : (uint8_t)' ';
    }
  }
  if (lane <= Blocks) L.sbits[lane] = 0ull;
  wave_sync();
  for (uint64_t s = T.s_next + lane; s < T.s_hi; s += 64) {
    const uint64_t o = sent_off[s];
    if (o >= abase + staged) break;
    if (o >= T.cb) atomicOr(&L.sbits[(o - abase) >> 6], 1ull << ((o - abase) & 63));
  }
  wave_sync();

  // ---- B/C.  Scalars carried from block to block:
  uint32_t nsym = 0, nw = 0;  // symbols and words so far
  bool prev_wb = true;        // the byte before this block belongs to a whitespace/punctuation char (or chunk start)
  bool pend_open = false;     // the last symbol of the previous block may have its successor in this one
  uint32_t pend_cp = 0;
  int cut = -1;               // last word boundary (for a span longer than the chunk)
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
    uint32_t c = kClsWs;  // bytes outside the chunk behave as whitespace
    if (inr && lead)
      c = cp < 1024u ? ((L.cls2[cp >> 4] >> ((cp & 15u) << 1)) & 3u) : ((cls_tab && cp < kNumCodePoints) ? (cls_tab[cp] & 3u) : 0u);
    const unsigned long long INR = __ballot(inr);
    const unsigned long long LEAD = __ballot(lead);
    const unsigned long long WSm = __ballot(lead && (c & kClsWs));
    const unsigned long long PNm = __ballot(lead && (c & kClsPunct));
    const unsigned long long CONT = ~LEAD;
    unsigned long long WB = WSm | PNm | ((prev_wb && (CONT & 1ull)) ? 1ull : 0ull);
    WB |= (WB << 1) & CONT;
    WB |= (WB << 1) & CONT;
    WB |= (WB << 1) & CONT;
    const unsigned long long SS = L.sbits[blk];
    const unsigned long long first_bit = blk == 0 ? (1ull << off0) : 0ull;  // off0 < 16
    const unsigned long long before = (WB << 1) | (prev_wb ? 1ull : 0ull) | SS | first_bit;
    const unsigned long long SYM = LEAD & ~WSm & INR;
    const unsigned long long WSTART = SYM & (PNm | before);
    {
      const unsigned long long CUT = LEAD & (WSm | PNm | SS) & __ballot(inr && p > off0 && p + 4 <= staged);
      if (CUT) cut = (int)(blk * 64 + 63 - __builtin_clzll(CUT));
    }
    const bool is_sym = (SYM >> lane) & 1ull;
    const bool wstart = (WSTART >> lane) & 1ull;
    const unsigned long long after = SYM & ~le;
    const uint32_t q = after ? (uint32_t)__builtin_ctzll(after) : 64u;
    const bool hasnext = is_sym && q < 64 && !((WSTART >> (q & 63)) & 1ull);
    const uint32_t f = SYM ? (uint32_t)__builtin_ctzll(SYM) : 64u;
    const bool joins = pend_open && f < 64 && !((WSTART >> (f & 63)) & 1ull);
    const uint32_t si = nsym + (uint32_t)__popcll(SYM & lt);
    const uint32_t cpn = __shfl(cp, (int)(q & 63));
    // lane 63 never has a successor inside the block: it probes the pair that straddles the block boundary
    const bool jp = joins && lane == 63;
    const uint32_t cpf = __builtin_amdgcn_readlane(cp, (int)(f & 63));
    const bool want = hasnext || jp;
    uint32_t v = kNoRank;
    if (want) v = slot_value(slots, sh, jp ? pend_cp : cp, jp ? cpf : cpn);
    if (is_sym) {
      L.sym[si] = wstart ? cp : (cp | SWT_BPE_CONT);
      if (!hasnext) L.val[si] = kNoRank;
    }
    if (want) L.val[jp ? nsym - 1 : si] = v;
    if (wstart) L.wl[nw + (uint32_t)__popcll(WSTART & lt)] = (uint16_t)si;
    if (lane == 0) { L.symmask[blk] = SYM; L.sympre[blk] = nsym; }
    nw += (uint32_t)__popcll(WSTART);
    nsym += (uint32_t)__popcll(SYM);
    if (SYM) {
      const int li = 63 - __builtin_clzll(SYM);
      const unsigned long long tail = li == 63 ? 0ull : ~((2ull << li) - 1ull);
      pend_open = ((WSm | PNm | SS) & tail) == 0ull && !((PNm >> li) & 1ull);
      pend_cp = __builtin_amdgcn_readlane(cp, li);
    } else {
      pend_open = false;  // a block without symbols holds whitespace: every word ended
    }
    prev_wb = (WB >> 63) & 1ull;
  }
  wave_sync();
  C.abase = abase;
  C.off0 = off0;
  C.staged = staged;
  C.nblk = nblk;
  C.ce = staged;
  C.nsym = nsym;
  C.nw = nw;
  C.cut = cut;
  C.last = last;
  C.giant = false;
}

// ---- the end of a chunk that does not end the tile: one word longer than the staged bytes goes to the one-lane walker (and the
// chunk is done: C.giant), anything else is cut at its last word boundary.  Closes the word list either way.
template <int Cap, int Mode>
__device__ __forceinline__ void lane_chunk_end(LaneLds<Cap> &L, const uint8_t *__restrict__ text, const uint64_t *__restrict__ sent_off,
                                               const uint8_t *__restrict__ cls_tab, const BpeSlot *__restrict__ slots, uint32_t sh,
                                               LaneTile &T, LaneChunk &C, int lane, uint32_t *__restrict__ sent_local,
                                               const uint32_t *__restrict__ uslot, unsigned long long *__restrict__ rec,
                                               unsigned long long *__restrict__ drec, const DirectOut &direct) {
  constexpr bool kDirect = Mode == 2, kRec = Mode == 1;
  if (!C.last) {
    if (C.cut < 0) {
      // a single word longer than the LDS chunk: one lane, global memory
      if (lane == 0) {
        uint64_t s = T.s_next;
        while (s < T.s_hi && sent_off[s] <= T.cb) s++;
        const uint64_t send = sent_off[s];  // s <= s_hi and sent_off[s_hi] = span_end > cb
        L.giant = giant_word(text, T.cb, send, cls_tab, slots, sh, T.tile_out + T.run);
      }
      wave_sync();
      const GiantResult g = L.giant;
      uint32_t mine = 0;
      for (uint64_t s = T.s_next + lane; s < T.s_hi; s += 64) {
        if (sent_off[s] >= g.end) break;
        if (kDirect) direct.off[s] = T.run; else if (Mode == 0) sent_local[s] = T.run;
        if (kRec) {
          drec[s] = (unsigned long long)(T.span_base + T.run) | ((unsigned long long)g.ntok << 32);
          rec[uslot[s]] = (unsigned long long)s | ((unsigned long long)g.ntok << 32);
        }
        mine++;
      }
      for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
      T.s_next += mine;
      T.run += g.ntok;
      T.cb = g.end;
      C.giant = true;
      C.nw = 0;
      C.nsym = 0;
      wave_sync();
      return;
    }
    // the cut is a word boundary: symbols and words at or beyond it are staged again by the next chunk
    C.ce = (uint32_t)C.cut;
    C.nsym = L.sympre[C.ce >> 6] + (uint32_t)__popcll(L.symmask[C.ce >> 6] & ((1ull << (C.ce & 63)) - 1ull));
    uint32_t keep = 0;
    for (uint32_t k0 = 0; k0 < C.nw; k0 += 64) {
      const uint32_t k = k0 + lane;
      keep += (uint32_t)__popcll(__ballot(k < C.nw && L.wl[k] < C.nsym));
    }
    C.nw = keep;
  }
  if (lane == 0) L.wl[C.nw] = (uint16_t)C.nsym;
  wave_sync();
}

// ---- W. the words with two symbols or more by length class: how many, then their entries (tag | index in wl[]) into a list
// that holds all 9+ words first, then the 5-8, then the 2-4 (o2 / o1 / o0 = where this wave's share of each class begins)
template <int Cap>
__device__ __forceinline__ void lane_words_count(const LaneLds<Cap> &L, uint32_t nw, int lane, uint32_t &c2, uint32_t &c1, uint32_t &c0) {
  c0 = c1 = c2 = 0;
  for (uint32_t k0 = 0; k0 < nw; k0 += 64) {
    const uint32_t k = k0 + lane;
    const uint32_t n = k < nw ? (uint32_t)L.wl[k + 1] - (uint32_t)L.wl[k] : 0u;
    c0 += (uint32_t)__popcll(__ballot(n >= 2 && n <= 4));
    c1 += (uint32_t)__popcll(__ballot(n >= 5 && n <= 8));
    c2 += (uint32_t)__popcll(__ballot(n >= 9));
  }
}
template <int Cap>
__device__ __forceinline__ void lane_words_write(const LaneLds<Cap> &L, uint32_t nw, int lane, uint16_t *list, uint32_t o2, uint32_t o1,
                                                 uint32_t o0, uint32_t tag) {
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (uint32_t k0 = 0; k0 < nw; k0 += 64) {
    const uint32_t k = k0 + lane;
    const uint32_t n = k < nw ? (uint32_t)L.wl[k + 1] - (uint32_t)L.wl[k] : 0u;
    const unsigned long long M0 = __ballot(n >= 2 && n <= 4), M1 = __ballot(n >= 5 && n <= 8), M2 = __ballot(n >= 9);
    if (n >= 9) list[o2 + (uint32_t)__popcll(M2 & lt)] = (uint16_t)(tag | k);
    else if (n >= 5) list[o1 + (uint32_t)__popcll(M1 & lt)] = (uint16_t)(tag | k);
    else if (n >= 2) list[o0 + (uint32_t)__popcll(M0 & lt)] = (uint16_t)(tag | k);
    o0 += (uint32_t)__popcll(M0);
    o1 += (uint32_t)__popcll(M1);
    o2 += (uint32_t)__popcll(M2);
  }
}

// ---- D. merge rounds (bpe.py:210-238) over a list of words, one lane per word.  A lane whose word is finished takes the next
// one of the list, so the wave goes through about as many rounds as its longest word needs -- the short words fill the lanes
// beside it.  An entry is tile:4 | index:12 into that tile's wl[] (LL = the tiles of the workgroup).
//
// The minimum of a round is found as a KEY = rank | slot (packed values: rank:16 | merged:16 -> rank:16 | 0:11 | slot:5; wide
// values: rank << 5 | slot), so one min per slot yields the leftmost smallest rank and its place; a dead slot's value is all
// ones (written when the slot dies) and slots past the word read slot n-1, which never has a pair.  (Scanning for the next
// minimum WHILE the two lookups of a round are in flight -- their slots blanked first, their values joining the minimum on
// arrival -- was measured and is not faster: 0.1905 against 0.1872 ms per call; the rounds are not waiting on the L2.)
template <bool Packed>
__device__ __forceinline__ uint32_t rank_key(uint32_t v, uint32_t slot) {
  return Packed ? ((v & 0xFFFF0000u) | slot) : (v >= (1u << 27) ? (0xFFFFFFE0u | slot) : ((v << 5) | slot));
}
template <bool Packed>
__device__ __forceinline__ uint32_t scan_key(const uint32_t *V, uint32_t n) {
  const uint32_t nm1 = n - 1u;
  uint32_t key = 0xFFFFFFFFu;
  for (uint32_t i0 = 0; i0 < nm1; i0 += 4) {
    const uint32_t i1 = min(i0 + 1u, nm1), i2 = min(i0 + 2u, nm1), i3 = min(i0 + 3u, nm1);
    const uint32_t k0 = rank_key<Packed>(V[i0], i0), k1 = rank_key<Packed>(V[i1], i1);
    const uint32_t k2 = rank_key<Packed>(V[i2], i2), k3 = rank_key<Packed>(V[i3], i3);
    key = min(min(key, k0), min(min(k1, k2), k3));
  }
  return key;
}
// ---- the tail of the rounds.  Two thirds of a tile's rounds run with a handful of words left -- the longest ones, which started
// first -- and a round costs the wave the same whether 60 lanes take part or 5 (S85k-open, simulated from the oracle: 13.8 rounds per
// tile, 8.8 of them with <= 16 words, 6.9 with <= 8).  So the last 64 / TL words get TL lanes each: lane j of a word scans its share
// of the slots, a shuffle finds the word's minimum, every lane of the group follows the merge (same values, LDS broadcasts), lane 0
// writes it and looks the left pair up while lane 1 looks up the right.  LEAD = the lanes that hold a word's state on entry; the
// function returns when at most `stop` words are left, their state in the first lane of each group (and in all of its lanes).
template <bool Packed, uint32_t TL>
__device__ __forceinline__ void lane_tail(uint32_t *sym0, uint32_t *val0, uint32_t *&S, uint32_t *&V, uint32_t &n, uint32_t &alive,
                                          unsigned long long LEAD, uint32_t stop, int lane, const BpeSlot *__restrict__ slots, uint32_t sh,
                                          const uint32_t *__restrict__ merged_of_rank) {
  constexpr uint32_t kNoKey = Packed ? 0xFFFF0000u : 0xFFFFFFE0u;
  // group g of TL lanes takes over the g-th word
  const uint32_t g = (uint32_t)lane / TL, j = (uint32_t)lane % TL;
  unsigned long long mrest = LEAD;
  for (uint32_t i = 0; i < g; i++) mrest &= mrest - 1ull;
  const bool have = mrest != 0ull;
  const int src = have ? __builtin_ctzll(mrest) : 0;
  const uint32_t sidx = (uint32_t)(S - sym0), vidx = (uint32_t)(V - val0);
  const uint32_t t_s = __shfl(sidx, src), t_v = __shfl(vidx, src), t_n = __shfl(n, src), t_alive = __shfl(alive, src);
  S = sym0 + t_s;
  V = val0 + t_v;
  n = have ? t_n : 0u;
  alive = t_alive;
  for (;;) {
    const unsigned long long BUSY = __ballot(n != 0u);
    if ((uint32_t)__popcll(BUSY) <= stop * TL) break;
    if (n != 0u) {
      const uint32_t nm1 = n - 1u;
      uint32_t k = 0xFFFFFFFFu;
#pragma unroll
      for (uint32_t q0 = 0; q0 < 32u / TL; q0 += 4) {  // this lane's share of the (at most 32) slots
        const uint32_t i0 = (32u / TL) * j + q0;
        if (i0 < nm1) {
          const uint32_t i1 = min(i0 + 1u, nm1), i2 = min(i0 + 2u, nm1), i3 = min(i0 + 3u, nm1);
          k = min(k, min(min(rank_key<Packed>(V[i0], i0), rank_key<Packed>(V[i1], i1)), min(rank_key<Packed>(V[i2], i2), rank_key<Packed>(V[i3], i3))));
        }
      }
#pragma unroll
      for (uint32_t d = 1; d < TL; d <<= 1) k = min(k, (uint32_t)__shfl_xor(k, (int)d));
      const uint32_t im = k & 31u;
      const uint32_t hi = alive & (0xFFFFFFFEu << im);
      if (k >= kNoKey || hi == 0u) {
        n = 0u;  // the same decision in all lanes of the group
      } else {
        const uint32_t m = V[im];
        const uint32_t r = (uint32_t)__builtin_ctz(hi);
        const uint32_t hi2 = hi & (hi - 1u);
        const uint32_t lo = alive & ((1u << im) - 1u);
        const bool has_rr = hi2 != 0u, has_pl = lo != 0u;
        const uint32_t rr = has_rr ? (uint32_t)__builtin_ctz(hi2) : 0u, pl = has_pl ? 31u - (uint32_t)__builtin_clz(lo) : 0u;
        alive &= ~(1u << r);
        const uint32_t mg = Packed ? (SWT_SYM_BASE + (m & 0xFFFFu)) : merged_of_rank[m];
        const uint32_t sl = S[pl] & ~SWT_BPE_CONT, sr = S[rr] & ~SWT_BPE_CONT;
        wave_sync();  // every lane has read the old symbols before lane 0 changes them
        const bool left = j == 0u && has_pl, right = j == 1u && has_rr;
        uint32_t v = kNoRank;
        if (left || right) v = slot_value(slots, sh, left ? sl : mg, left ? mg : sr);
        if (j == 0u) {
          S[im] = im ? (mg | SWT_BPE_CONT) : mg;
          S[r] = kInvalidTok;
          V[r] = kNoRank;
          if (has_pl) V[pl] = v;
        }
        if (j == 1u) V[im] = v;  // kNoRank when nothing follows
      }
    }
    wave_sync();
  }
}

template <bool Packed, bool Proper, int Cap>
__device__ __forceinline__ void lane_rounds(LaneLds<Cap> *LL, const uint16_t *list, uint32_t n_list, int lane,
                                            const BpeSlot *__restrict__ slots, uint32_t sh, const uint32_t *__restrict__ merged_of_rank) {
  constexpr uint32_t kNoKey = Packed ? 0xFFFF0000u : 0xFFFFFFE0u;  // keys from here up: no pair
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t next = 0;          // first word of the list no lane has taken (the same in every lane)
  uint32_t n = 0, alive = 0;  // this lane's word: symbols (0: none), live slots
  uint32_t key = 0xFFFFFFFFu; // its smallest rank | the slot that holds it
  uint32_t *S = LL[0].sym, *V = LL[0].val;
  for (;;) {
    const unsigned long long IDLE = __ballot(n == 0u);
    if (IDLE && next < n_list) {
      const uint32_t k = next + (uint32_t)__popcll(IDLE & lt);
      if (n == 0u && k < n_list) {
        const uint32_t e = list[k];
        LaneLds<Cap> &L = LL[e >> 12];
        const uint32_t base = L.wl[e & 0xFFFu];
        n = (uint32_t)L.wl[(e & 0xFFFu) + 1] - base;
        S = &L.sym[base];
        V = &L.val[base];
        alive = n >= 32u ? 0xFFFFFFFFu : (1u << n) - 1u;  // bit i: slot i still holds a symbol
        if (!Proper || n > 32u) {
          slow_word<Packed>(S, V, n, slots, sh, merged_of_rank);
          n = 0u;
        } else {
          key = scan_key<Packed>(V, n);
        }
      }
      next += (uint32_t)__popcll(IDLE);
    }
    const unsigned long long BUSY = __ballot(n != 0u);
    if (BUSY == 0ull) {
      if (next >= n_list) break;
      continue;
    }
    if (next >= n_list && __popcll(BUSY) <= 16) break;  // the last few (long) words: several lanes each, below
    if (n != 0u) {
      // slot im merges with the next live slot r; pl / rr = the live slots either side of the pair
      const uint32_t im = key & 31u;
      const uint32_t hi = alive & (0xFFFFFFFEu << im);
      if (key >= kNoKey || hi == 0u) {
        n = 0u;  // the word is finished (hi == 0 cannot happen: a slot with a pair value has a live successor)
      } else {
        const uint32_t m = V[im];
        const uint32_t r = (uint32_t)__builtin_ctz(hi);
        const uint32_t hi2 = hi & (hi - 1u);
        const uint32_t lo = alive & ((1u << im) - 1u);
        const bool has_rr = hi2 != 0u, has_pl = lo != 0u;
        const uint32_t rr = has_rr ? (uint32_t)__builtin_ctz(hi2) : 0u, pl = has_pl ? 31u - (uint32_t)__builtin_clz(lo) : im;
        alive &= ~(1u << r);
        const uint32_t mg = Packed ? (SWT_SYM_BASE + (m & 0xFFFFu)) : merged_of_rank[m];
        const uint32_t sl = S[pl] & ~SWT_BPE_CONT, sr = S[rr] & ~SWT_BPE_CONT;
        S[im] = im ? (mg | SWT_BPE_CONT) : mg;
        S[r] = kInvalidTok;
        V[r] = kNoRank;
        const uint32_t v1 = slot_value(slots, sh, sl, mg), v2 = has_rr ? slot_value(slots, sh, mg, sr) : kNoRank;
        if (has_pl) V[pl] = v1;
        V[im] = v2;
        key = scan_key<Packed>(V, n);
      }
    }
  }
  // ---- the tail (lane_tail above): four lanes a word once 16 words are left
  {
    const unsigned long long BUSY = __ballot(n != 0u);
    if (BUSY != 0ull) lane_tail<Packed, 4>(LL[0].sym, LL[0].val, S, V, n, alive, BUSY, 0u, lane, slots, sh, merged_of_rank);
  }
}

// ---- E + F of one chunk: order-preserving compaction of the live symbols into the tile's output run, then the tile-local
// token offset of every sentence starting in [cb, ce) (and == ce on the last chunk): byte -> symbol through the split's
// masks, symbol -> token through phase E's.  Advances the tile.
template <int Cap, int Mode>
__device__ __forceinline__ void lane_emit(LaneLds<Cap> &L, const uint64_t *__restrict__ sent_off, LaneTile &T, const LaneChunk &C, int lane,
                                          uint32_t *__restrict__ sent_local, const uint32_t *__restrict__ uslot,
                                          unsigned long long *__restrict__ rec, unsigned long long *__restrict__ drec,
                                          const DirectOut &direct) {
  constexpr bool kDirect = Mode == 2, kRec = Mode == 1;
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t total = 0;
  for (uint32_t j0 = 0; j0 < C.nsym; j0 += 64) {
    const uint32_t j = j0 + lane;
    const uint32_t sv = j < C.nsym ? L.sym[j] : kInvalidTok;
    const unsigned long long m = __ballot(sv != kInvalidTok);
    if (lane == 0) { L.vmask[j0 >> 6] = m; L.blkpre[j0 >> 6] = total; }
    if (sv != kInvalidTok) T.tile_out[T.run + total + (uint32_t)__popcll(m & lt)] = sv;
    total += (uint32_t)__popcll(m);
  }
  wave_sync();
  auto tokens_before = [&](uint64_t rel) -> uint32_t {
    if (rel >= C.ce || (rel >> 6) >= C.nblk) return total;
    const uint32_t sidx = L.sympre[rel >> 6] + (uint32_t)__popcll(L.symmask[rel >> 6] & ((1ull << (rel & 63)) - 1ull));
    if (sidx >= C.nsym) return total;
    return L.blkpre[sidx >> 6] + (uint32_t)__popcll(L.vmask[sidx >> 6] & ((1ull << (sidx & 63)) - 1ull));
  };
  uint32_t mine = 0;
  for (uint64_t s = T.s_next + lane; s < T.s_hi; s += 64) {
    const uint64_t rel = sent_off[s] - C.abase;
    if (rel > C.ce || (rel == C.ce && !C.last)) break;
    const uint32_t e = tokens_before(rel);
    if (kDirect) direct.off[s] = T.run + e; else if (Mode == 0) sent_local[s] = T.run + e;
    if (kRec && rel < C.ce) {  // a word never straddles the cut, so its end lies in this chunk too
      const uint32_t e2 = tokens_before(sent_off[s + 1] - C.abase);
      drec[s] = (unsigned long long)(T.span_base + T.run + e) | ((unsigned long long)(e2 - e) << 32);
      rec[uslot[s]] = (unsigned long long)s | ((unsigned long long)(e2 - e) << 32);
    }
    mine++;
  }
  for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
  T.s_next += mine;
  T.run += total;
  T.cb = C.abase + C.ce;
  wave_sync();
}

// One wave per tile: running text (Mode 0), the unique words of the dedup path (Mode 1), the single-workgroup call (Mode 2).
template <bool Packed, bool Proper, int Cap, int Mode>
__global__ __launch_bounds__(64) void bpe_lane_kernel(
    const uint8_t *__restrict__ text, uint64_t n_bytes, const uint64_t *__restrict__ sent_off,
    const uint64_t *__restrict__ plan, const uint8_t *__restrict__ cls_tab, const BpeSlot *__restrict__ slots,
    uint32_t sh, const uint32_t *__restrict__ merged_of_rank, uint32_t *__restrict__ scratch,
    uint32_t *__restrict__ sent_local, uint32_t *__restrict__ tile_tok, const uint32_t *__restrict__ uslot,
    unsigned long long *__restrict__ rec, unsigned long long *__restrict__ drec, DirectOut direct) {
  constexpr bool kDirect = Mode == 2;
  static_assert(Cap % 64 == 0 && Cap <= 4032, "an entry of the word list is tile:4 | word:12, and wl[] holds 16-bit symbol indices");
  __shared__ LaneLds<Cap> L;
  const int lane = threadIdx.x;
  const uint64_t t = blockIdx.x;
  LaneTile T;
  T.s_lo = kDirect ? 0 : plan[t];
  T.s_hi = kDirect ? direct.n_sent : plan[t + 1];
  if (T.s_lo == T.s_hi) {
    if (Mode == 0 && lane == 0) tile_tok[t] = 0;
    return;
  }
  lane_classes(L, cls_tab, lane);
  T.span_base = sent_off[T.s_lo];
  T.span_end = sent_off[T.s_hi];
  T.tile_out = scratch + T.span_base;
  T.run = 0;
  T.s_next = T.s_lo;
  T.cb = T.span_base;
  uint16_t *const list = reinterpret_cast<uint16_t *>(L.txt);  // the staged bytes are not read again once the split is done
  for (;;) {
    LaneChunk C;
    lane_split(L, text, n_bytes, sent_off, cls_tab, slots, sh, T, C, lane);
    lane_chunk_end<Cap, Mode>(L, text, sent_off, cls_tab, slots, sh, T, C, lane, sent_local, uslot, rec, drec, direct);
    if (C.giant) continue;
    uint32_t c2, c1, c0;
    lane_words_count(L, C.nw, lane, c2, c1, c0);
    lane_words_write(L, C.nw, lane, list, 0u, c2, c2 + c1, 0u);
    wave_sync();
    lane_rounds<Packed, Proper, Cap>(&L, list, c2 + c1 + c0, lane, slots, sh, merged_of_rank);
    wave_sync();
    lane_emit<Cap, Mode>(L, sent_off, T, C, lane, sent_local, uslot, rec, drec, direct);
    if (C.last) break;
  }
  if (lane == 0) {
    if (kDirect) { direct.off[T.s_hi] = T.run; *direct.n_tokens = T.run; }
    else if (Mode == 0) tile_tok[t] = T.run;
  }
}

}  // namespace swt

using namespace swt;

struct swt_bpe_table {
  std::vector<BpeSlot> h_slots;    // built on the host at create; uploaded on first encode
  std::vector<uint32_t> h_merged;  // merged symbol id by rank
  bool packed = false;             // slot value = rank << 16 | (merged - SWT_SYM_BASE)
  bool proper = false;             // every pair ranks above the merges that produce its symbols (any trained table)
  bool lane_kernel = true;         // bpe_lane_kernel (default) or the byte-lane kernel of rounds 1-2 (SWT_BPE_KERNEL=bytes)
  BpeSlot *d_slots = nullptr;
  uint32_t *d_merged = nullptr;
  uint32_t bits = 0;
  uint32_t n_merges = 0;
  TileWorkspace ws;
  DevBuf in_text, in_off, out_ids, out_off, n_tok;  // staging for the host-buffer entry point
  // word-level dedup inside one call
  TileWorkspace ws2;          // workspaces of the encode over the unique words
  DedupEngine dd;
  PinnedBuf pin;       // small host calls: inputs and outputs staged in one pinned buffer, one copy each way
  DevBuf small_in, small_out;
  int opt_unique_tile = 0;  // SWT_OPT_UNIQUE_TILE
};

static int bpe_upload(swt_bpe_table *t) {
  if (t->d_slots) return SWT_OK;
  int rc = ensure_device();
  if (rc) return rc;
  SWT_HIP(hipMalloc((void **)&t->d_slots, t->h_slots.size() * sizeof(BpeSlot)));
  SWT_HIP(hipMemcpy(t->d_slots, t->h_slots.data(), t->h_slots.size() * sizeof(BpeSlot), hipMemcpyHostToDevice));
  SWT_HIP(hipMalloc((void **)&t->d_merged, (t->h_merged.size() + 1) * 4));
  if (!t->h_merged.empty())
    SWT_HIP(hipMemcpy(t->d_merged, t->h_merged.data(), t->h_merged.size() * 4, hipMemcpyHostToDevice));
  return SWT_OK;
}

template <bool Packed, int Cap>
static void launch_encode_kernel_as(swt_bpe_table *t, uint64_t n_tiles, const TileWorkspace &ws, const uint8_t *d_text, uint64_t n_bytes,
                                    const uint64_t *d_sent_off, const uint8_t *d_cls, const uint32_t *d_uslot,
                                    unsigned long long *d_rec, unsigned long long *d_drec, hipStream_t st) {
  const uint32_t sh = 32u - t->bits;
  if (d_rec)
    hipLaunchKernelGGL((bpe_encode_kernel<Packed, Cap, 1>), dim3((unsigned)n_tiles), dim3(64), 0, st, d_text, n_bytes, d_sent_off,
                       ws.plan.as<uint64_t>(), d_cls, t->d_slots, sh, t->d_merged, ws.scratch.as<uint32_t>(),
                       ws.sent_local.as<uint32_t>(), ws.tile_tok.as<uint32_t>(), d_uslot, d_rec, d_drec, DirectOut{nullptr, nullptr, 0},
                       (uint32_t)ablation_knob(0));
  else
    hipLaunchKernelGGL((bpe_encode_kernel<Packed, Cap, 0>), dim3((unsigned)n_tiles), dim3(64), 0, st, d_text, n_bytes, d_sent_off,
                       ws.plan.as<uint64_t>(), d_cls, t->d_slots, sh, t->d_merged, ws.scratch.as<uint32_t>(),
                       ws.sent_local.as<uint32_t>(), ws.tile_tok.as<uint32_t>(), d_uslot, d_rec, d_drec, DirectOut{nullptr, nullptr, 0},
                       (uint32_t)ablation_knob(0));
}

// the word-lane kernel: running text, or the unique words of the dedup path (d_rec)
template <bool Packed, bool Proper, int Cap>
static void launch_lane_kernel_as(swt_bpe_table *t, uint64_t n_tiles, const TileWorkspace &ws, const uint8_t *d_text, uint64_t n_bytes,
                                  const uint64_t *d_sent_off, const uint8_t *d_cls, const uint32_t *d_uslot,
                                  unsigned long long *d_rec, unsigned long long *d_drec, hipStream_t st) {
  const uint32_t sh = 32u - t->bits;
  if (d_rec)
    hipLaunchKernelGGL((bpe_lane_kernel<Packed, Proper, Cap, 1>), dim3((unsigned)n_tiles), dim3(64), 0, st, d_text, n_bytes, d_sent_off,
                       ws.plan.as<uint64_t>(), d_cls, t->d_slots, sh, t->d_merged, ws.scratch.as<uint32_t>(),
                       ws.sent_local.as<uint32_t>(), ws.tile_tok.as<uint32_t>(), d_uslot, d_rec, d_drec, DirectOut{nullptr, nullptr, 0});
  else
    hipLaunchKernelGGL((bpe_lane_kernel<Packed, Proper, Cap, 0>), dim3((unsigned)n_tiles), dim3(64), 0, st, d_text, n_bytes, d_sent_off,
                       ws.plan.as<uint64_t>(), d_cls, t->d_slots, sh, t->d_merged, ws.scratch.as<uint32_t>(),
                       ws.sent_local.as<uint32_t>(), ws.tile_tok.as<uint32_t>(), d_uslot, d_rec, d_drec, DirectOut{nullptr, nullptr, 0});
}

extern "C" {

// diagnostics (not part of include/swt.h): resident workgroups per CU the runtime grants the encode kernel
int swt_debug_occupancy(int which) try {
  int n = -1;
  hipError_t e = which == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bpe_lane_kernel<true, true, kLaneCap, 0>, 64, 0)
                 : which  ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bpe_encode_kernel<false, kBpeCap, 0>, 64, 0)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bpe_encode_kernel<true, kBpeCap, 0>, 64, 0);
  return e == hipSuccess ? n : -(int)e;
} SWT_API_CATCH

// diagnostics (not part of include/swt.h): what swt_bpe_table_create decided.  0: log2 of the slots, 1: packed values,
// 2: proper (every pair ranks above the merges producing its symbols), 3: entries in the table, 4: every entry is found where
// a device lookup looks for it (its first or its second slot) and nowhere else
int swt_debug_bpe_table_info(const swt_bpe_table *t, int which) try {
  if (!t) return -1;
  if (which == 0) return (int)t->bits;
  if (which == 1) return t->packed ? 1 : 0;
  if (which == 2) return t->proper ? 1 : 0;
  uint32_t n = 0;
  bool placed = true;
  const uint32_t sh = 32u - t->bits;
  for (size_t i = 0; i < t->h_slots.size(); i++) {
    const BpeSlot &sl = t->h_slots[i];
    if (sl.key == kEmptyKey) continue;
    n++;
    const uint32_t l = (uint32_t)(sl.key >> 32), r = (uint32_t)sl.key;
    const uint32_t h1 = bpe_hash(l, r, sh, kHash1), h2 = bpe_hash(l, r, sh, kHash2);
    if (i != h1 && i != h2) placed = false;
    if (h1 != h2 && t->h_slots[i == h1 ? h2 : h1].key == sl.key) placed = false;
  }
  return which == 3 ? (int)n : (placed ? 1 : 0);
} SWT_API_CATCH

int swt_bpe_table_create(const uint32_t *left, const uint32_t *right, const uint32_t *merged, uint32_t n_merges,
                         swt_bpe_table **out) try {
  if (!out || (n_merges && (!left || !right || !merged))) return fail(SWT_ERR_INVALID, "null argument");
  for (uint32_t i = 0; i < n_merges; i++)
    if ((left[i] | right[i] | merged[i]) & SWT_BPE_CONT) return fail(SWT_ERR_INVALID, "symbol id out of range at merge %u", i);
  // {pair: i} (bpe.py:257): a later duplicate overwrites, so only the LAST index of a pair goes into the table
  std::unordered_map<uint64_t, uint32_t> last;
  last.reserve((size_t)n_merges * 2 + 16);
  for (uint32_t i = 0; i < n_merges; i++) last[pair_key(left[i], right[i])] = i;
  uint32_t bits = 4;
  while ((1ull << bits) < 2ull * n_merges + 2) bits++;
  auto *t = new swt_bpe_table();
  std::vector<BpeSlot> &slots = t->h_slots;
  // two-choice cuckoo placement (see slot_lookup); a table that does not settle gets twice the slots
  const uint32_t bits0 = bits;
  for (;; bits++) {
    if (bits > 28 || bits > bits0 + 4) { delete t; return fail(SWT_ERR_UNSUPPORTED, "the rank table could not be placed (%u pairs)", n_merges); }
    const uint32_t sh = 32u - bits;
    slots.assign((size_t)1 << bits, BpeSlot{kEmptyKey, 0u, 0u});
    bool ok = true;
    for (uint32_t i = 0; i < n_merges && ok; i++) {
      if (last[pair_key(left[i], right[i])] != i) continue;
      BpeSlot cur{pair_key(left[i], right[i]), i, merged[i]};
      uint32_t avoid = 0xFFFFFFFFu;
      ok = false;
      for (int kick = 0; kick < 2000; kick++) {
        const uint32_t l = (uint32_t)(cur.key >> 32), r = (uint32_t)cur.key;
        const uint32_t h1 = bpe_hash(l, r, sh, kHash1), h2 = bpe_hash(l, r, sh, kHash2);
        if (slots[h1].key == kEmptyKey) { slots[h1] = cur; ok = true; break; }
        if (slots[h2].key == kEmptyKey) { slots[h2] = cur; ok = true; break; }
        const uint32_t j = h1 == avoid ? h2 : h1;  // evict, but not from the slot this entry was just evicted from
        std::swap(cur, slots[j]);
        avoid = j;
      }
    }
    if (ok) break;
  }
  t->bits = bits;
  t->n_merges = n_merges;
  // proper: every pair ranks above every merge that produces one of its symbols, so the pairs a merge creates rank above it
  // and "one occurrence of the best pair per round" equals the reference's "all occurrences" (bpe_lane_kernel)
  {
    std::unordered_map<uint32_t, uint32_t> maxprod;
    for (const auto &sl : slots)
      if (sl.key != kEmptyKey) {
        auto it = maxprod.find(sl.merged);
        if (it == maxprod.end()) maxprod[sl.merged] = sl.rank; else if (sl.rank > it->second) it->second = sl.rank;
      }
    t->proper = true;
    for (const auto &sl : slots)
      if (sl.key != kEmptyKey) {
        for (uint32_t sy : {(uint32_t)(sl.key >> 32), (uint32_t)sl.key}) {
          auto it = maxprod.find(sy);
          if (it != maxprod.end() && it->second >= sl.rank) t->proper = false;
        }
      }
  }
  if (const char *e = getenv("SWT_BPE_KERNEL")) t->lane_kernel = strcmp(e, "bytes") != 0;
  if (const char *e = getenv("SWT_BPE_UTILE")) t->opt_unique_tile = atoi(e);  // measurement knob, as SWT_OPT_UNIQUE_TILE
  if (const char *e = getenv("SWT_BPE_DEDUP")) t->dd.opt_mode = atoi(e);       // measurement knob, as SWT_OPT_DEDUP (0 auto, 1 never, 2 always)
  t->h_merged.assign(merged, merged + n_merges);
  // packed values when every rank and every merged-symbol index fits 16 bits (any realistic table below 65k merges)
  t->packed = n_merges < 0xFFFEu;
  for (uint32_t i = 0; i < n_merges && t->packed; i++)
    if (merged[i] < SWT_SYM_BASE || merged[i] - SWT_SYM_BASE >= 0xFFFFu) t->packed = false;
  if (t->packed)
    for (auto &sl : slots)
      if (sl.key != kEmptyKey) sl.rank = (sl.rank << 16) | (sl.merged - SWT_SYM_BASE);
  *out = t;
  return SWT_OK;
} SWT_API_CATCH

int swt_bpe_table_set_option(swt_bpe_table *t, int option, int value) try {
  if (!t) return fail(SWT_ERR_INVALID, "null table");
  switch (option) {
    case SWT_OPT_DEDUP:
      if (value < 0 || value > 2) return fail(SWT_ERR_INVALID, "SWT_OPT_DEDUP takes 0, 1 or 2");
      t->dd.opt_mode = value;
      return SWT_OK;
    case SWT_OPT_DEDUP_TABLE_BITS:
      if (value != 0 && (value < 4 || value > 24)) return fail(SWT_ERR_INVALID, "SWT_OPT_DEDUP_TABLE_BITS takes 0 or 4..24");
      t->dd.opt_table_bits = (uint32_t)value;
      return SWT_OK;
    case SWT_OPT_UNIQUE_TILE:
      if (value != 0 && value != 64 && value != 128 && value != 256) return fail(SWT_ERR_INVALID, "SWT_OPT_UNIQUE_TILE takes 0, 64, 128 or 256");
      t->opt_unique_tile = value;
      return SWT_OK;
  }
  return fail(SWT_ERR_INVALID, "no such option");
} SWT_API_CATCH

void swt_bpe_table_destroy(swt_bpe_table *t) try {
  if (!t) return;
  if (t->d_slots) (void)hipFree(t->d_slots);
  if (t->d_merged) (void)hipFree(t->d_merged);
  t->ws.release();
  t->ws2.release();
  t->dd.release();
  t->pin.release();
  t->small_in.release();
  t->small_out.release();
  for (DevBuf *b : {&t->in_text, &t->in_off, &t->out_ids, &t->out_off, &t->n_tok}) b->release();
  delete t;
} SWT_API_CATCH_VOID

// the direct path: every word occurrence goes through the merge rounds
// cap = staged bytes per chunk (LDS footprint ~ 20 B per byte): 512 for running text, less for the unique-word pass
static void launch_encode_kernel(swt_bpe_table *t, uint64_t n_tiles, const TileWorkspace &ws, const uint8_t *d_text, uint64_t n_bytes,
                                 const uint64_t *d_sent_off, const uint8_t *d_cls, const uint32_t *d_uslot, unsigned long long *d_rec,
                                 unsigned long long *d_drec, hipStream_t st, int cap = kBpeCap) {
#define SWT_ENC(P, C) launch_encode_kernel_as<P, C>(t, n_tiles, ws, d_text, n_bytes, d_sent_off, d_cls, d_uslot, d_rec, d_drec, st)
#define SWT_LANE(P, R, C) launch_lane_kernel_as<P, R, C>(t, n_tiles, ws, d_text, n_bytes, d_sent_off, d_cls, d_uslot, d_rec, d_drec, st)
#define SWT_LANE_CAPS(P, R) do { if (cap == 128) SWT_LANE(P, R, 128); else if (cap == 256) SWT_LANE(P, R, 256); else SWT_LANE(P, R, 512); } while (0)
  if (t->lane_kernel && !d_rec) {  // running text: the one chunk size the tile was chosen for
    if (t->packed) { if (t->proper) SWT_LANE(true, true, kLaneCap); else SWT_LANE(true, false, kLaneCap); }
    else { if (t->proper) SWT_LANE(false, true, kLaneCap); else SWT_LANE(false, false, kLaneCap); }
    return;
  }
  if (t->lane_kernel) {
    if (t->packed) { if (t->proper) SWT_LANE_CAPS(true, true); else SWT_LANE_CAPS(true, false); }
    else { if (t->proper) SWT_LANE_CAPS(false, true); else SWT_LANE_CAPS(false, false); }
    return;
  }
  if (t->packed) {
    if (cap == 128) SWT_ENC(true, 128); else if (cap == 256) SWT_ENC(true, 256); else SWT_ENC(true, 512);
  } else {
    if (cap == 128) SWT_ENC(false, 128); else if (cap == 256) SWT_ENC(false, 256); else SWT_ENC(false, 512);
  }
#undef SWT_LANE_CAPS
#undef SWT_LANE
#undef SWT_ENC
}

static int bpe_encode_direct(swt_bpe_table *t, TileWorkspace &ws, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off,
                             uint64_t n_sent, uint32_t *d_out_ids, uint64_t *d_out_off, uint64_t *d_n_tokens, const uint8_t *d_cls,
                             hipStream_t st) {
  const uint32_t tile = t->lane_kernel ? (uint32_t)kLaneTile : (uint32_t)kBpeTile;
  const uint64_t n_tiles = tile_count(n_bytes, tile);
  if (n_tiles > 0x7FFFFFFFull)
    return fail(SWT_ERR_UNSUPPORTED, "text too large for one call (%llu bytes)", (unsigned long long)n_bytes);
  int rc;
  if (n_bytes <= kDirectBytes && n_sent <= kDirectSents) {
    // a sentence or a few: one workgroup, one launch, the caller's arrays written by the kernel (DirectOut)
    if ((rc = ws.reserve(64, 0, 1))) return rc;
    const DirectOut direct{d_out_off, d_n_tokens, n_sent};
    const uint32_t sh = 32u - t->bits;
    auto one_lane = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, st, d_text, n_bytes, d_sent_off, (const uint64_t *)nullptr, d_cls, t->d_slots, sh,
                         t->d_merged, d_out_ids, ws.sent_local.as<uint32_t>(), ws.tile_tok.as<uint32_t>(), (const uint32_t *)nullptr,
                         (unsigned long long *)nullptr, (unsigned long long *)nullptr, direct);
    };
    auto one_bytes = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, st, d_text, n_bytes, d_sent_off, (const uint64_t *)nullptr, d_cls, t->d_slots, sh,
                         t->d_merged, d_out_ids, ws.sent_local.as<uint32_t>(), ws.tile_tok.as<uint32_t>(), (const uint32_t *)nullptr,
                         (unsigned long long *)nullptr, (unsigned long long *)nullptr, direct, 0u);
    };
    if (t->lane_kernel) {
      if (t->packed) { if (t->proper) one_lane(bpe_lane_kernel<true, true, kLaneCap, 2>); else one_lane(bpe_lane_kernel<true, false, kLaneCap, 2>); }
      else { if (t->proper) one_lane(bpe_lane_kernel<false, true, kLaneCap, 2>); else one_lane(bpe_lane_kernel<false, false, kLaneCap, 2>); }
    } else {
      if (t->packed) one_bytes(bpe_encode_kernel<true, kBpeCap, 2>); else one_bytes(bpe_encode_kernel<false, kBpeCap, 2>);
    }
    SWT_HIP(hipGetLastError());
    return SWT_OK;
  }
  if ((rc = ws.reserve(n_bytes, n_sent, n_tiles))) return rc;
  prof_begin(st, 2);
  launch_plan(d_sent_off, n_sent, n_tiles, tile, ws.plan.as<uint64_t>(), st);
  prof_begin(st);
  launch_encode_kernel(t, n_tiles, ws, d_text, n_bytes, d_sent_off, d_cls, nullptr, nullptr, nullptr, st, t->lane_kernel ? kLaneCap : kBpeCap);
  prof_end(st);
  launch_scan_gather(d_sent_off, n_sent, n_tiles, ws, d_out_ids, d_out_off, d_n_tokens, st);
  prof_end(st, 2);
  SWT_HIP(hipGetLastError());
  return SWT_OK;
}

// The dedup path (swt_dedup.h): eight launches, no host round trip -- the number of unique words stays on the device, so
// the unique-word encode has a fixed number of workgroups and its tile size follows on the device (ureg_kernel writes its plan).
// Returns 1 when the batch is too large for the 32-bit fields of this path (the caller takes the direct path).
constexpr int kUTile = 128;            // smallest tile of the unique-word pass (chunk = 2 such tiles); measured: 64 -> 84 us, 128 -> 72 us, 256 -> 88 us
constexpr uint64_t kUMaxTiles = 8192;  // its launch size: 256 CUs x 32 single-wave workgroups
static int bpe_encode_dedup(swt_bpe_table *t, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off, uint64_t n_sent,
                            uint32_t *d_out_ids, uint64_t *d_out_off, uint64_t *d_n_tokens, const uint8_t *d_cls, hipStream_t st) {
  int rc;
  // the word-lane kernel wants a batch of words per tile: 256-byte tiles (S85k-lex: 64 -> 0.182, 128 -> 0.174, 256 -> 0.168 ms per call)
  const uint32_t tile2 = t->opt_unique_tile == 256 ? 256u : (t->opt_unique_tile == 64 ? 64u : (t->opt_unique_tile == 128 || !t->lane_kernel ? (uint32_t)kUTile : 256u));
  uint64_t n_tiles2 = tile_count(n_bytes, tile2);  // the unique words together are no longer than the text
  if (n_tiles2 > kUMaxTiles) n_tiles2 = kUMaxTiles;
  if (n_bytes > kDedupMaxBytes) return 1;
  if ((rc = t->ws2.reserve(n_bytes, n_bytes / 2 + 2, n_tiles2))) return rc;
  prof_begin(st, 2);
  if ((rc = dedup_front(t->dd, t->ws, d_text, n_bytes, d_sent_off, n_sent, d_cls, kDedupBpe, st, t->ws2.plan.as<uint64_t>(), n_tiles2, tile2)))
    return rc;
  // encode the unique words once (raw-word mode: each one is a "sentence"); their token runs stay in ws2.scratch and
  // phase F of the kernel leaves count | place in rec[slot]
  prof_begin(st);
  launch_encode_kernel(t, n_tiles2, t->ws2, t->dd.utext.as<uint8_t>(), n_bytes, t->dd.uoff.as<uint64_t>(), nullptr,
                       t->dd.uslot.as<uint32_t>(), t->dd.rec_ptr(), t->dd.drec_ptr(), st, (int)(2 * tile2));
  prof_end(st);
  rc = dedup_back(t->dd, t->ws, d_sent_off, n_sent, n_bytes, t->ws2.scratch.as<uint32_t>(), kDedupBpe, nullptr, d_out_ids, d_out_off,
                  d_n_tokens, st);
  prof_end(st, 2);
  return rc;
}

int swt_bpe_encode_dev(swt_bpe_table *t, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off,
                       uint64_t n_sent, uint32_t *d_out_ids, uint64_t *d_out_off, uint64_t *d_n_tokens, uint32_t flags,
                       void *stream) try {
  if (!t || !d_sent_off || !d_out_off || !d_n_tokens || (n_bytes && (!d_text || !d_out_ids)))
    return fail(SWT_ERR_INVALID, "null argument");
  int rc = bpe_upload(t);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const uint8_t *d_cls = nullptr;
  if ((rc = device_class_table(&d_cls))) return rc;
  if (n_sent == 0) {
    SWT_HIP(hipMemsetAsync(d_out_off, 0, 8, st));
    SWT_HIP(hipMemsetAsync(d_n_tokens, 0, 8, st));
    return SWT_OK;
  }
  const bool raw = (flags & SWT_BPE_RAW_WORDS) != 0;
  // debug knob 1: bit 0 = never dedup, bit 1 = dedup whatever the batch size (tests)
  if (!raw && !(flags & SWT_BPE_NO_DEDUP) && t->dd.opt_mode != 1 &&
      (t->dd.opt_mode == 2 || (n_bytes >= kDedupMinBytes && t->dd.pays(n_bytes)))) {
    rc = bpe_encode_dedup(t, d_text, n_bytes, d_sent_off, n_sent, d_out_ids, d_out_off, d_n_tokens, d_cls, st);
    if (rc == 0 && t->dd.opt_mode == 0) t->dd.note(n_bytes, st);
    if (rc <= 0) return rc;  // done, or a real error
  }
  // raw-word mode: no classes, so nothing splits and nothing is dropped
  return bpe_encode_direct(t, t->ws, d_text, n_bytes, d_sent_off, n_sent, d_out_ids, d_out_off, d_n_tokens, raw ? nullptr : d_cls, st);
} SWT_API_CATCH

// text and offsets on the device -> ids, offsets and the count in the caller's host arrays
static int bpe_encode_to_host(swt_bpe_table *t, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_off, uint64_t n_sent,
                              uint32_t *out_ids, uint64_t out_cap, uint64_t *out_off, uint64_t *n_tokens, uint32_t flags) {
  int rc;
  if ((rc = t->out_ids.reserve((n_bytes + 64) * 4))) return rc;
  if ((rc = t->out_off.reserve((n_sent + 1) * 8))) return rc;
  if ((rc = t->n_tok.reserve(8))) return rc;
  rc = swt_bpe_encode_dev(t, d_text, n_bytes, d_off, n_sent, t->out_ids.as<uint32_t>(), t->out_off.as<uint64_t>(), t->n_tok.as<uint64_t>(),
                          flags, nullptr);
  if (rc) return rc;
  uint64_t nt = 0;
  SWT_HIP(hipMemcpy(&nt, t->n_tok.p, 8, hipMemcpyDeviceToHost));
  *n_tokens = nt;
  SWT_HIP(hipMemcpy(out_off, t->out_off.p, (n_sent + 1) * 8, hipMemcpyDeviceToHost));
  if (nt > out_cap)
    return fail(SWT_ERR_CAPACITY, "out_ids too small: need %llu ids, have %llu", (unsigned long long)nt, (unsigned long long)out_cap);
  if (nt) SWT_HIP(hipMemcpy(out_ids, t->out_ids.p, nt * 4, hipMemcpyDeviceToHost));
  return SWT_OK;
}

int swt_bpe_encode(swt_bpe_table *t, const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent, uint32_t *out_ids,
                   uint64_t out_cap, uint64_t *out_off, uint64_t *n_tokens, uint32_t flags) try {
  if (!t || !sent_off || !out_off || !n_tokens) return fail(SWT_ERR_INVALID, "null argument");
  int rc = bpe_upload(t);
  if (rc) return rc;
  const uint64_t n_bytes = sent_off[n_sent];
  if (sent_off[0] != 0) return fail(SWT_ERR_INVALID, "sent_off[0] must be 0");
  for (uint64_t s = 0; s < n_sent; s++)
    if (sent_off[s] > sent_off[s + 1])
      return fail(SWT_ERR_INVALID, "sentence offsets must be non-decreasing (at %llu)", (unsigned long long)s);
  if (n_bytes && !text) return fail(SWT_ERR_INVALID, "null text");
  if (n_bytes <= kDirectBytes && n_sent <= kDirectSents && n_sent > 0) {
    // tokenize(text) on one sentence (bpe.py:245): the single workgroup of the direct form reads the text and the offsets from
    // pinned host memory and writes the count, the offsets and the ids there -- one launch and one synchronisation, no copy
    // call at all.  Measured: 43.1 -> 40.7 us per call from Python; a launch + hipStreamSynchronize is 11 us here, spinning on
    // a host word instead would save 4.5 of them (tools/micro/sync_probe.hip), the rest is the kernel's own latency chain
    // (~10 merge rounds, one L2 probe each) and ctypes.
    const size_t off_bytes = ((n_sent + 1) * 8 + 15) & ~(size_t)15, text_bytes = (n_bytes + 64 + 15) & ~(size_t)15;
    const size_t out_at = off_bytes + text_bytes;
    if ((rc = t->pin.reserve(out_at + 16 + off_bytes + (n_bytes + 64) * 4))) return rc;
    uint8_t *h = t->pin.as<uint8_t>();
    memcpy(h, sent_off, (n_sent + 1) * 8);
    if (n_bytes) memcpy(h + off_bytes, text, n_bytes);
    memset(h + off_bytes + n_bytes, ' ', 64);
    uint8_t *o = h + out_at;
    rc = swt_bpe_encode_dev(t, h + off_bytes, n_bytes, reinterpret_cast<const uint64_t *>(h), n_sent, reinterpret_cast<uint32_t *>(o + 16 + off_bytes),
                            reinterpret_cast<uint64_t *>(o + 16), reinterpret_cast<uint64_t *>(o), flags, nullptr);
    if (rc) return rc;
    SWT_HIP(hipStreamSynchronize(0));
    const uint64_t nt = *reinterpret_cast<const volatile uint64_t *>(o);
    *n_tokens = nt;
    memcpy(out_off, o + 16, (n_sent + 1) * 8);
    if (nt > out_cap)
      return fail(SWT_ERR_CAPACITY, "out_ids too small: need %llu ids, have %llu", (unsigned long long)nt, (unsigned long long)out_cap);
    if (nt) memcpy(out_ids, o + 16 + off_bytes, nt * 4);
    return SWT_OK;
  }
  if (n_bytes <= kSmallCallBytes && n_sent <= kSmallCallSents) {
    // The reference-style call (one sentence, or a few): five small copies and their synchronisations cost more than the
    // kernels.  Offsets + text go up in ONE copy from pinned memory, token count + offsets + ids come back in ONE.
    const size_t off_bytes = ((n_sent + 1) * 8 + 15) & ~(size_t)15;
    const size_t in_bytes = off_bytes + n_bytes + 64;
    const size_t out_bytes = 16 + off_bytes + (n_bytes + 64) * 4;
    if ((rc = t->pin.reserve(in_bytes > out_bytes ? in_bytes : out_bytes)) || (rc = t->small_in.reserve(in_bytes)) ||
        (rc = t->small_out.reserve(out_bytes)))
      return rc;
    uint8_t *h = t->pin.as<uint8_t>();
    memcpy(h, sent_off, (n_sent + 1) * 8);
    if (n_bytes) memcpy(h + off_bytes, text, n_bytes);
    SWT_HIP(hipMemcpyAsync(t->small_in.p, h, off_bytes + n_bytes, hipMemcpyHostToDevice, 0));
    uint8_t *d_in = t->small_in.as<uint8_t>(), *d_out = t->small_out.as<uint8_t>();
    rc = swt_bpe_encode_dev(t, d_in + off_bytes, n_bytes, reinterpret_cast<const uint64_t *>(d_in), n_sent,
                            reinterpret_cast<uint32_t *>(d_out + 16 + off_bytes), reinterpret_cast<uint64_t *>(d_out + 16),
                            reinterpret_cast<uint64_t *>(d_out), flags, nullptr);
    if (rc) return rc;
    SWT_HIP(hipMemcpyAsync(h, d_out, 16 + off_bytes + (n_bytes + 64) * 4, hipMemcpyDeviceToHost, 0));
    SWT_HIP(hipStreamSynchronize(0));
    const uint64_t nt = *reinterpret_cast<const uint64_t *>(h);
    *n_tokens = nt;
    memcpy(out_off, h + 16, (n_sent + 1) * 8);
    if (nt > out_cap)
      return fail(SWT_ERR_CAPACITY, "out_ids too small: need %llu ids, have %llu", (unsigned long long)nt, (unsigned long long)out_cap);
    if (nt) memcpy(out_ids, h + 16 + off_bytes, nt * 4);
    return SWT_OK;
  }
  if ((rc = t->in_text.reserve(n_bytes + 64))) return rc;
  if ((rc = t->in_off.reserve((n_sent + 1) * 8))) return rc;
  if (n_bytes) SWT_HIP(hipMemcpyAsync(t->in_text.p, text, n_bytes, hipMemcpyHostToDevice, 0));
  SWT_HIP(hipMemcpyAsync(t->in_off.p, sent_off, (n_sent + 1) * 8, hipMemcpyHostToDevice, 0));
  return bpe_encode_to_host(t, t->in_text.as<uint8_t>(), n_bytes, t->in_off.as<uint64_t>(), n_sent, out_ids, out_cap, out_off, n_tokens, flags);
} SWT_API_CATCH

// list[str] joined with U+0000 -> ids without the prepared text ever coming back to the host (swt_utf8_prepare_joined +
// swt_bpe_encode in one call).  *n_tokens = UINT64_MAX on return: a sentence needs the host's str.lower() (need_host says
// which) and nothing was encoded.
int swt_bpe_encode_joined(swt_bpe_table *t, const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint32_t *out_ids, uint64_t out_cap,
                          uint64_t *out_off, uint64_t *n_tokens, uint8_t *need_host, uint32_t flags) try {
  if (!t || !out_off || !n_tokens || (n_sent && !need_host) || (n_joined && !joined)) return fail(SWT_ERR_INVALID, "null argument");
  int rc = bpe_upload(t);
  if (rc) return rc;
  *n_tokens = UINT64_MAX;
  struct Ctx { swt_bpe_table *t; uint64_t n_sent; uint32_t *out_ids; uint64_t out_cap; uint64_t *out_off, *n_tokens; uint32_t flags; };
  Ctx c{t, n_sent, out_ids, out_cap, out_off, n_tokens, flags};
  bool consumed = false;
  return with_prepared_joined(joined, n_joined, n_sent, need_host, &consumed,
      [](void *p, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_off) {
        Ctx *c = static_cast<Ctx *>(p);
        return bpe_encode_to_host(c->t, d_text, n_bytes, d_off, c->n_sent, c->out_ids, c->out_cap, c->out_off, c->n_tokens, c->flags);
      }, &c);
} SWT_API_CATCH

}  // extern "C"
