// swt_bpe_encode.hip -- FastBPE batched encode on gfx950.
//
// Replaces, for a whole batch of sentences at once:
//   SubwordTokenizer.preprocessing   /root/reference/source/utils.py:26-29   (split only; lower() is the caller's)
//   FastBPE.tokenize                 /root/reference/source/bpe.py:245-249
//   FastBPE.encode_word / _pairs     /root/reference/source/bpe.py:202-243
//   the rank dict                    /root/reference/source/bpe.py:200,257
//
// Runs on the tile/chunk skeleton of swt_tile.h.  Encoder-specific phases:
//   B  per byte: decode the code point at every UTF-8 lead byte, look its pre-tokenizer class up
//   C  per byte: word starts (whitespace removed, every punctuation code point its own word) -> per-wave lists
//   D  one lane per word: gather the word's code points, run the lowest-rank merge loop against the
//      device rank table (open-addressing hash, one 16-byte slot per probe), set the '##' flag
// A span longer than kCap is cut at word boundaries; a single word longer than kCap falls to a one-lane
// global-memory path (correct, slow, pathological inputs only).
#include "swt_tile.h"

namespace swt {

struct alignas(16) BpeSlot {
  uint64_t key;     // left << 32 | right, kEmptyKey when free
  uint32_t rank;    // index in merges_list (last duplicate wins)
  uint32_t merged;  // symbol id of left+right
};

constexpr uint8_t kClsWs = 1, kClsPunct = 2, kClsCont = 0x80;
constexpr uint32_t kNoRank = 0xFFFFFFFFu, kDirtyRank = 0xFFFFFFFEu;
constexpr int kClsLds = 1024;

__device__ __forceinline__ bool slot_lookup(const BpeSlot *__restrict__ slots, uint32_t bits, uint32_t l, uint32_t r,
                                            uint32_t &rank, uint32_t &merged) {
  const uint64_t key = pair_key(l, r);
  const uint32_t mask = (1u << bits) - 1u;
  uint32_t h = hash_slot(key, bits);
  for (;;) {
    const uint4 raw = *reinterpret_cast<const uint4 *>(&slots[h]);
    const uint64_t k = ((uint64_t)raw.y << 32) | raw.x;
    if (k == key) { rank = raw.z; merged = raw.w; return true; }
    if (k == kEmptyKey) return false;
    h = (h + 1) & mask;
  }
}

// FastBPE.encode_word on a symbol array (bpe.py:210-238): repeat { lowest-rank adjacent pair; replace all
// of its occurrences left to right, non-overlapping }.  Returns the new length.
template <class Ptr>
__device__ __forceinline__ uint32_t merge_word(Ptr s, uint32_t n, const BpeSlot *__restrict__ slots, uint32_t bits) {
  while (n >= 2) {
    uint32_t best = 0xFFFFFFFFu, bm = 0, bl = 0, br = 0;
    uint32_t a = s[0];
    for (uint32_t i = 0; i + 1 < n; i++) {
      const uint32_t b = s[i + 1];
      uint32_t rk, mg;
      if (slot_lookup(slots, bits, a, b, rk, mg) && rk < best) { best = rk; bm = mg; bl = a; br = b; }
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
                                  const uint8_t *__restrict__ cls_tab, const BpeSlot *__restrict__ slots, uint32_t bits,
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
  n = merge_word(out, n, slots, bits);
  for (uint32_t i = 1; i < n; i++) out[i] |= SWT_BPE_CONT;
  r.ntok = n;
  return r;
}

// first probe of two independent lookups issued back to back (one L2 round trip for both); a probe that lands on
// another key falls back to the ordinary probe loop (rare at load factor <= 1/2)
__device__ __forceinline__ void slot_lookup2(const BpeSlot *__restrict__ slots, uint32_t bits, uint32_t l0, uint32_t r0,
                                             uint32_t l1, uint32_t r1, bool second, uint32_t &v0, uint32_t &v1) {
  const uint64_t k0 = pair_key(l0, r0), k1 = pair_key(l1, r1);
  const uint32_t h0 = hash_slot(k0, bits), h1 = second ? hash_slot(k1, bits) : h0;
  const uint4 a = *reinterpret_cast<const uint4 *>(&slots[h0]);
  const uint4 b = *reinterpret_cast<const uint4 *>(&slots[h1]);
  const uint64_t ka = ((uint64_t)a.y << 32) | a.x, kb = ((uint64_t)b.y << 32) | b.x;
  uint32_t dummy;
  if (ka == k0) v0 = a.z;
  else if (ka == kEmptyKey) v0 = kNoRank;
  else v0 = slot_lookup(slots, bits, l0, r0, v0, dummy) ? v0 : kNoRank;
  if (second) {
    if (kb == k1) v1 = b.z;
    else if (kb == kEmptyKey) v1 = kNoRank;
    else v1 = slot_lookup(slots, bits, l1, r1, v1, dummy) ? v1 : kNoRank;
  }
}

// Packed = true: the cached value of a pair is (rank << 16 | merged - SWT_SYM_BASE), so a merge round needs no
// table access to learn the merged symbol (tables of < 65535 merges); false: the value is the rank and the merged
// symbol comes from merged_of_rank[].
template <bool Packed>
__global__ __launch_bounds__(kThreads) void bpe_encode_kernel(
    const uint8_t *__restrict__ text, uint64_t n_bytes, const uint64_t *__restrict__ sent_off,
    const uint64_t *__restrict__ plan, const uint8_t *__restrict__ cls_tab, const BpeSlot *__restrict__ slots,
    uint32_t bits, const uint32_t *__restrict__ merged_of_rank, uint32_t *__restrict__ scratch,
    uint32_t *__restrict__ sent_local, uint32_t *__restrict__ tile_tok, uint32_t dbg) {
  __shared__ TileLds L;
  __shared__ uint32_t rk[kCap];  // rank of the pair (symbol here, next symbol of the word); kNoRank when none
  __shared__ uint16_t wl[kCap];  // word starts, one list per wave (its quarter of the chunk)
  __shared__ uint32_t wnext[kWaves];
  __shared__ GiantResult s_giant;
  __shared__ __attribute__((aligned(16))) uint8_t cls_lo[kClsLds];  // classes of U+0000..U+03FF: no global trip for Latin text

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const uint64_t t = blockIdx.x;
  const uint64_t s_lo = plan[t], s_hi = plan[t + 1];
  if (s_lo == s_hi) {
    if (tid == 0) tile_tok[t] = 0;
    return;
  }
  if (tid < kClsLds / 16) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (cls_tab) v = reinterpret_cast<const uint4 *>(cls_tab)[tid];
    reinterpret_cast<uint4 *>(cls_lo)[tid] = v;
  }
  const uint64_t span_base = sent_off[s_lo], span_end = sent_off[s_hi];
  uint32_t *const tile_out = scratch + span_base;
  uint32_t run = 0;        // tokens emitted by this tile so far (block-uniform)
  uint64_t s_next = s_lo;  // first sentence whose local offset is not recorded yet (block-uniform)
  uint64_t cb = span_base;

  for (;;) {
    const uint64_t abase = cb & ~15ull;
    const uint32_t off0 = (uint32_t)(cb - abase);
    const uint64_t avail = span_end - abase;
    const bool last = avail <= (uint64_t)kCap;
    const uint32_t staged = last ? (uint32_t)avail : (uint32_t)kCap;

    tile_stage(L, text, n_bytes, abase, staged);
    if (dbg & 1) { if (last) break; cb = abase + staged; continue; }  // ablation: staging only

    // ---- B. per byte: code point + pre-tokenizer class at every lead byte; sentence-start bits
    for (uint32_t p = tid; p < staged; p += kThreads) {
      const uint8_t b = L.txt[p];
      uint32_t sv = kInvalidTok;
      uint8_t cv = kClsCont;
      if (!utf8_is_cont(b) && p >= off0) {
        int len = utf8_len(b);
        if (p + len > staged) len = (int)(staged - p);
        uint32_t cp = b;
        if (b >= 0x80 && len > 1) {
          cp = b & (0xFF >> (len + 1));
          for (int i = 1; i < len; i++) cp = (cp << 6) | (L.txt[p + i] & 0x3F);
        }
        uint8_t c = 0;
        if (cp < (uint32_t)kClsLds) c = cls_lo[cp];
        else if (cls_tab && cp < kNumCodePoints) c = cls_tab[cp];
        cv = c & (kClsWs | kClsPunct);
        if (!(c & kClsWs)) sv = cp;
      }
      L.sym[p] = sv;
      L.cls[p] = cv;
    }
    tile_mark_sentences(L, sent_off, s_next, s_hi, cb, abase, staged);
    __syncthreads();

    // ---- chunk end: the whole rest of the span, or the last word boundary that fits
    uint32_t ce = staged;
    if (!last) {
      int best = -1;
      for (uint32_t p = tid; p + 4 <= staged; p += kThreads) {
        if (p <= off0) continue;
        const uint8_t c = L.cls[p];
        if (c & kClsCont) continue;
        if ((c & (kClsWs | kClsPunct)) || tile_sbit(L, p)) best = (int)p;
      }
      if (best >= 0) atomicMax(&L.cut, best);
      __syncthreads();
      if (L.cut < 0) {
        // a single word longer than the LDS chunk: one lane, global memory
        if (tid == 0) {
          uint64_t s = s_next;
          while (s < s_hi && sent_off[s] <= cb) s++;
          const uint64_t send = sent_off[s];  // s <= s_hi and sent_off[s_hi] = span_end > cb
          s_giant = giant_word(text, cb, send, cls_tab, slots, bits, tile_out + run);
        }
        __syncthreads();
        const GiantResult g = s_giant;
        for (uint64_t s = s_next + tid; s < s_hi; s += kThreads) {
          if (sent_off[s] >= g.end) break;
          sent_local[s] = run;
          atomicAdd(&L.cnt, 1u);
        }
        __syncthreads();
        s_next += L.cnt;
        run += g.ntok;
        cb = g.end;
        __syncthreads();
        continue;
      }
      ce = (uint32_t)L.cut;
    }

    // ---- C. per byte position, all lanes busy: (1) word starts (utils.py:27 split) into per-wave lists,
    // (2) the table value of the pair this symbol forms with the NEXT symbol of its word -- every lookup of the
    // first merge round is issued here, four independent probes in flight per lane, instead of serially per word
    uint32_t nwords = 0;
    uint16_t *const mywl = wl + wave * kQuarter;
    for (int r = 0; r < kQuarter / 64; r += 4) {
      uint32_t pl[4], pr[4];
      bool want[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t p = wave * kQuarter + (r + u) * 64 + lane;
        bool is = false;
        want[u] = false;
        pl[u] = pr[u] = 0;
        if (p >= off0 && p < ce) {
          const uint8_t c = L.cls[p];
          if (!(c & (kClsCont | kClsWs))) {
            if (c & kClsPunct) is = true;
            else if (p == off0 || tile_sbit(L, p)) is = true;
            else {
              uint32_t q = p - 1;
              while (q > off0 && (L.cls[q] & kClsCont)) q--;
              is = (L.cls[q] & (kClsWs | kClsPunct | kClsCont)) != 0;
            }
            if (!(c & kClsPunct)) {
              uint32_t q = p + utf8_len(L.txt[p]);
              while (q < ce && (L.cls[q] & kClsCont)) q++;
              if (q < ce && !tile_sbit(L, q) && !(L.cls[q] & (kClsWs | kClsPunct)) && !(dbg & 4)) {
                want[u] = true;
                pl[u] = L.sym[p];
                pr[u] = L.sym[q];
              }
            }
          }
        }
        const unsigned long long m = __ballot(is);
        if (is) mywl[nwords + __popcll(m & lt)] = (uint16_t)p;
        nwords += __popcll(m);
      }
      // four first probes back to back, then resolve
      uint32_t hh[4];
      uint4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; u++) hh[u] = want[u] ? hash_slot(pair_key(pl[u], pr[u]), bits) : 0u;
#pragma unroll
      for (int u = 0; u < 4; u++) raw[u] = *reinterpret_cast<const uint4 *>(&slots[hh[u]]);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t p = wave * kQuarter + (r + u) * 64 + lane;
        uint32_t val = kNoRank;
        if (want[u]) {
          const uint64_t key = pair_key(pl[u], pr[u]);
          const uint64_t k = ((uint64_t)raw[u].y << 32) | raw[u].x;
          uint32_t dummy;
          if (k == key) val = raw[u].z;
          else if (k != kEmptyKey && !slot_lookup(slots, bits, pl[u], pr[u], val, dummy)) val = kNoRank;
        }
        rk[p] = val;
      }
    }
    if (lane == 0) wnext[wave] = 64;
    // (rk[] of a word is written by the lanes of this wave or the next one: a word may cross the quarter)
    __syncthreads();

    // ---- D. one lane per word, words handed out dynamically inside the wave: gather symbols + cached ranks,
    // merge loop (bpe.py:205-243) touching the table only for pairs a merge created, '##' flag, invalidate the tail
    if (!(dbg & 16))
    for (uint32_t k = lane; k < nwords; k = atomicAdd(&wnext[wave], 1u)) {
      const uint32_t ws = mywl[k];
      uint32_t n = 0, p = ws;
      if (L.cls[ws] & kClsPunct) {
        n = 1;
        p = ws + utf8_len(L.txt[ws]);
      } else {
        for (;;) {
          L.sym[ws + n] = L.sym[p];
          rk[ws + n] = rk[p];
          n++;
          p += utf8_len(L.txt[p]);
          while (p < ce && (L.cls[p] & kClsCont)) p++;
          if (p >= ce) break;
          if (tile_sbit(L, p)) break;
          if (L.cls[p] & (kClsWs | kClsPunct)) break;
        }
      }
      if (p > ce) p = ce;
      uint32_t *const s = &L.sym[ws];
      uint32_t *const rr = &rk[ws];
      while (n >= 2 && !(dbg & 8)) {
        uint32_t best = kNoRank;
        for (uint32_t i = 0; i + 1 < n; i++) best = min(best, rr[i]);
        if (best == kNoRank) break;
        const uint32_t mg = Packed ? (SWT_SYM_BASE + (best & 0xFFFFu)) : merged_of_rank[best];
        // replace every occurrence left to right (equal value <=> equal pair); values next to a merge go stale
        uint32_t i = 0, j = 0;
        while (i < n) {
          if (i + 1 < n && rr[i] == best) {
            s[j] = mg;
            rr[j] = kDirtyRank;
            i += 2;
          } else {
            const bool next_taken = (i + 2 < n) && rr[i + 1] == best;
            const uint32_t keep = rr[i];
            s[j] = s[i];
            rr[j] = next_taken ? kDirtyRank : keep;
            i += 1;
          }
          j++;
        }
        n = j;
        rr[n - 1] = kNoRank;
        // refresh the stale values two at a time: both probes in flight together (one L2 round trip per round
        // in the common case of a single occurrence)
        uint32_t q = 0;
        for (;;) {
          while (q + 1 < n && rr[q] != kDirtyRank) q++;
          if (q + 1 >= n) break;
          const uint32_t a0 = q;
          uint32_t b0 = q + 1;
          while (b0 + 1 < n && rr[b0] != kDirtyRank) b0++;
          const bool second = b0 + 1 < n;
          uint32_t v0 = kNoRank, v1 = kNoRank;
          slot_lookup2(slots, bits, s[a0], s[a0 + 1], second ? s[b0] : 0u, second ? s[b0 + 1] : 0u, second, v0, v1);
          rr[a0] = v0;
          if (second) rr[b0] = v1;
          q = (second ? b0 : a0) + 1;
        }
      }
      for (uint32_t i = 1; i < n; i++) s[i] |= SWT_BPE_CONT;
      for (uint32_t q = ws + n; q < p; q++) L.sym[q] = kInvalidTok;
    }
    __syncthreads();

    // ---- E, F
    if (dbg & 32) { if (last) break; cb = abase + ce; continue; }
    const uint32_t total = tile_compact(L, off0, ce, tile_out + run);
    s_next += tile_record(L, sent_off, sent_local, s_next, s_hi, abase, ce, last, run, total);
    run += total;
    if (last) break;
    cb = abase + ce;
  }
  if (tid == 0) tile_tok[t] = run;
}

}  // namespace swt

using namespace swt;

struct swt_bpe_table {
  std::vector<BpeSlot> h_slots;  // built on the host at create; uploaded on first encode
  std::vector<uint32_t> h_merged;  // merged symbol id by rank
  bool packed = false;             // slot value = rank << 16 | (merged - SWT_SYM_BASE)
  BpeSlot *d_slots = nullptr;
  uint32_t *d_merged = nullptr;
  uint32_t bits = 0;
  uint32_t n_merges = 0;
  TileWorkspace ws;
  DevBuf in_text, in_off, out_ids, out_off, n_tok;  // staging for the host-buffer entry point
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

extern "C" {

// diagnostics (not part of include/swt.h): resident workgroups per CU the runtime grants the encode kernel
int swt_debug_occupancy(int which) {
  int n = -1;
  hipError_t e = which ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bpe_encode_kernel<false>, kThreads, 0)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bpe_encode_kernel<true>, kThreads, 0);
  return e == hipSuccess ? n : -(int)e;
}

int swt_bpe_table_create(const uint32_t *left, const uint32_t *right, const uint32_t *merged, uint32_t n_merges,
                         swt_bpe_table **out) {
  if (!out || (n_merges && (!left || !right || !merged))) return fail(SWT_ERR_INVALID, "null argument");
  uint32_t bits = 4;
  while ((1ull << bits) < 2ull * n_merges + 2) bits++;
  const size_t cap = (size_t)1 << bits;
  auto *t = new swt_bpe_table();
  t->bits = bits;
  t->n_merges = n_merges;
  std::vector<BpeSlot> &slots = t->h_slots;
  slots.resize(cap);
  for (auto &s : slots) { s.key = kEmptyKey; s.rank = 0; s.merged = 0; }
  const uint32_t mask = (uint32_t)cap - 1;
  for (uint32_t i = 0; i < n_merges; i++) {
    if ((left[i] | right[i] | merged[i]) & SWT_BPE_CONT) {
      delete t;
      return fail(SWT_ERR_INVALID, "symbol id out of range at merge %u", i);
    }
    const uint64_t key = pair_key(left[i], right[i]);
    uint32_t h = hash_slot(key, bits);
    while (slots[h].key != kEmptyKey && slots[h].key != key) h = (h + 1) & mask;
    slots[h].key = key;  // {pair: i}: a later duplicate overwrites (bpe.py:257)
    slots[h].rank = i;
    slots[h].merged = merged[i];
  }
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
}

void swt_bpe_table_destroy(swt_bpe_table *t) {
  if (!t) return;
  if (t->d_slots) (void)hipFree(t->d_slots);
  if (t->d_merged) (void)hipFree(t->d_merged);
  t->ws.release();
  for (DevBuf *b : {&t->in_text, &t->in_off, &t->out_ids, &t->out_off, &t->n_tok}) b->release();
  delete t;
}

int swt_bpe_encode_dev(swt_bpe_table *t, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off,
                       uint64_t n_sent, uint32_t *d_out_ids, uint64_t *d_out_off, uint64_t *d_n_tokens, uint32_t flags,
                       void *stream) {
  if (!t || !d_sent_off || !d_out_off || !d_n_tokens || (n_bytes && (!d_text || !d_out_ids)))
    return fail(SWT_ERR_INVALID, "null argument");
  int rc = bpe_upload(t);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const uint8_t *d_cls = nullptr;
  if ((rc = device_class_table(&d_cls))) return rc;
  if (flags & SWT_BPE_RAW_WORDS) d_cls = nullptr;  // no classes: nothing splits, nothing is dropped
  const uint64_t n_tiles = tile_count(n_bytes);
  if (n_tiles > 0x7FFFFFFFull)
    return fail(SWT_ERR_UNSUPPORTED, "text too large for one call (%llu bytes)", (unsigned long long)n_bytes);
  if ((rc = t->ws.reserve(n_bytes, n_sent, n_tiles))) return rc;
  if (n_sent == 0) {
    SWT_HIP(hipMemsetAsync(d_out_off, 0, 8, st));
    SWT_HIP(hipMemsetAsync(d_n_tokens, 0, 8, st));
    return SWT_OK;
  }
  launch_plan(d_sent_off, n_sent, n_tiles, t->ws.plan.as<uint64_t>(), st);
  prof_begin(st);
  if (t->packed)
    hipLaunchKernelGGL(bpe_encode_kernel<true>, dim3((unsigned)n_tiles), dim3(kThreads), 0, st, d_text, n_bytes, d_sent_off,
                       t->ws.plan.as<uint64_t>(), d_cls, t->d_slots, t->bits, t->d_merged, t->ws.scratch.as<uint32_t>(),
                       t->ws.sent_local.as<uint32_t>(), t->ws.tile_tok.as<uint32_t>(), (uint32_t)debug_knob(0));
  else
    hipLaunchKernelGGL(bpe_encode_kernel<false>, dim3((unsigned)n_tiles), dim3(kThreads), 0, st, d_text, n_bytes, d_sent_off,
                       t->ws.plan.as<uint64_t>(), d_cls, t->d_slots, t->bits, t->d_merged, t->ws.scratch.as<uint32_t>(),
                       t->ws.sent_local.as<uint32_t>(), t->ws.tile_tok.as<uint32_t>(), (uint32_t)debug_knob(0));
  prof_end(st);
  launch_scan_gather(d_sent_off, n_sent, n_tiles, t->ws, d_out_ids, d_out_off, d_n_tokens, st);
  SWT_HIP(hipGetLastError());
  return SWT_OK;
}

int swt_bpe_encode(swt_bpe_table *t, const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent, uint32_t *out_ids,
                   uint64_t out_cap, uint64_t *out_off, uint64_t *n_tokens, uint32_t flags) {
  if (!t || !sent_off || !out_off || !n_tokens) return fail(SWT_ERR_INVALID, "null argument");
  int rc = bpe_upload(t);
  if (rc) return rc;
  const uint64_t n_bytes = sent_off[n_sent];
  if (sent_off[0] != 0) return fail(SWT_ERR_INVALID, "sent_off[0] must be 0");
  for (uint64_t s = 0; s < n_sent; s++)
    if (sent_off[s] > sent_off[s + 1])
      return fail(SWT_ERR_INVALID, "sentence offsets must be non-decreasing (at %llu)", (unsigned long long)s);
  if (n_bytes && !text) return fail(SWT_ERR_INVALID, "null text");
  if ((rc = t->in_text.reserve(n_bytes + 64))) return rc;
  if ((rc = t->in_off.reserve((n_sent + 1) * 8))) return rc;
  if ((rc = t->out_ids.reserve((n_bytes + 64) * 4))) return rc;
  if ((rc = t->out_off.reserve((n_sent + 1) * 8))) return rc;
  if ((rc = t->n_tok.reserve(8))) return rc;
  if (n_bytes) SWT_HIP(hipMemcpyAsync(t->in_text.p, text, n_bytes, hipMemcpyHostToDevice, 0));
  SWT_HIP(hipMemcpyAsync(t->in_off.p, sent_off, (n_sent + 1) * 8, hipMemcpyHostToDevice, 0));
  rc = swt_bpe_encode_dev(t, t->in_text.as<uint8_t>(), n_bytes, t->in_off.as<uint64_t>(), n_sent, t->out_ids.as<uint32_t>(),
                          t->out_off.as<uint64_t>(), t->n_tok.as<uint64_t>(), flags, nullptr);
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

}  // extern "C"
