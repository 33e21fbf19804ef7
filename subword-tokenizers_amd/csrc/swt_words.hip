// swt_words.hip -- the training corpus as unique words, on the device.
//
// Replaces, for the whole corpus at once (lowercased UTF-8 already resident in HBM):
//   SubwordTokenizer.preprocessing   /root/reference/source/utils.py:26-29   (the BertPreTokenizer split)
//   new_words / Counter / symbolise  /root/reference/source/bpe.py:73-81     (unique words in first-occurrence
//                                                                             order, their frequency, list(word))
//
//   census   one wave per tile (the chunk walk and the ballot-mask word split of swt_bpe_encode.hip); one lane per
//            word: FNV-1a over its bytes, then find-or-insert in a global open-addressing table of packed 64-bit
//            slots [tag:16 | byte length:8 | offset of a representative occurrence:40].  A slot matches only after an
//            exact byte compare with its representative, so the hash is never trusted.  first[slot] = atomicMin of the
//            occurrence offsets (bpe.py:77: Counter order = first occurrence), cnt[slot] = atomicAdd.
//   collect  used slots -> (first occurrence, slot) pairs; rocPRIM radix sort by first occurrence
//   emit     per unique word: code-point count -> exclusive scan -> list(word) as uint32 symbols; bitmap of the
//            code points seen (the initial vocab, bpe.py:75)
// Words of 255 bytes or more are never matched against each other (each occurrence becomes its own entry with
// frequency 1): counts are frequency-weighted sums and first-occurrence order is preserved, so the merges are the
// same (the dedup is an optimisation of the reference's, not part of its semantics).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "swt_tile.h"
#include "swt_words.h"

namespace swt {

constexpr uint8_t kWClsWs = 1, kWClsPunct = 2;
constexpr int kWTile = 512;
constexpr int kWCap = 1024;
constexpr int kWBlocks = kWCap / 64;
constexpr unsigned long long kSlotEmpty = ~0ull;
constexpr unsigned long long kOffMask = (1ull << 40) - 1ull;

struct WordTable {
  unsigned long long *slot;   // tag:16 | len:8 | representative offset:40
  unsigned long long *first;  // smallest occurrence offset
  uint32_t *cnt;              // occurrences
  uint32_t *biglen;           // byte length of entries whose len field is 255
  uint32_t bits;
};

__device__ __forceinline__ unsigned long long word_hash(const uint8_t *p, uint32_t n) {
  unsigned long long h = 0xcbf29ce484222325ull;
  for (uint32_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
  h ^= h >> 29; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 32;
  return h;
}

// find-or-insert one occurrence: `mine` = its bytes (LDS or global), gpos = its global byte offset
__device__ void word_insert(const WordTable &T, const uint8_t *__restrict__ text, const uint8_t *mine, uint32_t len, uint64_t gpos) {
  const unsigned long long h = word_hash(mine, len);
  const uint32_t mask = (1u << T.bits) - 1u;
  const uint32_t lf = len < 255u ? len : 255u;
  const unsigned long long head = ((h >> 48) << 48) | ((unsigned long long)lf << 40);
  const unsigned long long packed = head | gpos;
  uint32_t idx = (uint32_t)h & mask;
  for (;;) {
    unsigned long long v = __hip_atomic_load(&T.slot[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == kSlotEmpty) {
      v = atomicCAS(&T.slot[idx], kSlotEmpty, packed);
      if (v == kSlotEmpty) {
        if (lf == 255u) T.biglen[idx] = len;
        break;  // this occurrence is the representative
      }
    }
    if (lf != 255u && (v & ~kOffMask) == head) {
      const uint8_t *rep = text + (v & kOffMask);
      bool same = true;
      for (uint32_t i = 0; i < len; i++)
        if (rep[i] != mine[i]) { same = false; break; }
      if (same) break;
    }
    idx = (idx + 1) & mask;
  }
  atomicMin(&T.first[idx], (unsigned long long)gpos);
  atomicAdd(&T.cnt[idx], 1u);
}

struct CensusLds {
  __attribute__((aligned(16))) uint8_t txt[kWCap + 16];
  uint16_t wl[kWCap];
  unsigned long long sbits[kWBlocks + 1];
  unsigned long long endm[kWBlocks + 1];  // positions where a word cannot continue (whitespace, next word start, outside)
  __attribute__((aligned(16))) uint8_t cls_lo[1024];
  uint64_t giant_end;
};

// count_only: only tally the number of word occurrences (sizes the table); otherwise insert every word
__global__ __launch_bounds__(64) void census_kernel(const uint8_t *__restrict__ text, uint64_t n_bytes,
                                                    const uint64_t *__restrict__ sent_off, const uint64_t *__restrict__ plan,
                                                    const uint8_t *__restrict__ cls_tab, WordTable T, bool count_only,
                                                    unsigned long long *__restrict__ n_words_out) {
  __shared__ CensusLds L;
  const int lane = threadIdx.x;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const uint64_t t = blockIdx.x;
  const uint64_t s_lo = plan[t], s_hi = plan[t + 1];
  if (s_lo == s_hi) return;
  reinterpret_cast<uint4 *>(L.cls_lo)[lane] = reinterpret_cast<const uint4 *>(cls_tab)[lane];
  const uint64_t span_base = sent_off[s_lo], span_end = sent_off[s_hi];
  uint64_t s_next = s_lo;
  uint64_t cb = span_base;
  uint32_t tile_words = 0;

  for (;;) {
    const uint64_t abase = cb & ~15ull;
    const uint32_t off0 = (uint32_t)(cb - abase);
    const uint64_t avail = span_end - abase;
    const bool last = avail <= (uint64_t)kWCap;
    const uint32_t staged = last ? (uint32_t)avail : (uint32_t)kWCap;
    const uint32_t nblk = (staged + 63) >> 6;
    for (uint32_t c = lane * 16; c < staged; c += 64 * 16) {
      const uint64_t g = abase + c;
      if (g + 16 <= n_bytes && ((reinterpret_cast<uintptr_t>(text + g) & 15) == 0)) {
        *reinterpret_cast<uint4 *>(&L.txt[c]) = *reinterpret_cast<const uint4 *>(text + g);
      } else {
        for (int i = 0; i < 16; i++) L.txt[c + i] = (g + i < n_bytes) ? text[g + i] : (uint8_t)' ';
      }
    }
    if (lane <= kWBlocks) L.sbits[lane] = 0ull;
    __syncthreads();
    for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
      const uint64_t o = sent_off[s];
      if (o >= abase + staged) break;
      if (o >= cb) atomicOr(&L.sbits[(o - abase) >> 6], 1ull << ((o - abase) & 63));
    }
    __syncthreads();

    // the ballot-mask word split of swt_bpe_encode.hip (phase B/C), keeping only word starts and stop positions
    uint32_t nw = 0;
    bool prev_wb = true;
    int cut = -1;
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
      uint8_t c = kWClsWs;
      if (inr && lead) c = cp < 1024u ? L.cls_lo[cp] : (cp < kNumCodePoints ? cls_tab[cp] : (uint8_t)0);
      const unsigned long long INR = __ballot(inr);
      const unsigned long long LEAD = __ballot(lead);
      const unsigned long long WSm = __ballot(lead && (c & kWClsWs));
      const unsigned long long PNm = __ballot(lead && (c & kWClsPunct));
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
      if (lane == 0) L.endm[blk] = WSm | WSTART | ~INR;
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
          while (e < send) {
            const uint8_t b = text[e];
            int len = utf8_len(b);
            if (e + len > send) len = (int)(send - e);
            uint32_t cp = b;
            if (b >= 0x80 && len > 1) {
              cp = b & (0xFF >> (len + 1));
              for (int i = 1; i < len; i++) cp = (cp << 6) | (text[e + i] & 0x3F);
            }
            const uint8_t c = utf8_is_cont(b) ? kWClsWs : (cp < kNumCodePoints ? cls_tab[cp] : (uint8_t)0);
            if (c & kWClsWs) { if (first) { e += len; has_word = false; } break; }
            if (c & kWClsPunct) { if (first) e += len; break; }
            e += len;
            first = false;
          }
          if (has_word && e > cb && !count_only) word_insert(T, text, text + cb, (uint32_t)(e - cb), cb);
          L.giant_end = e | ((has_word && e > cb) ? (1ull << 63) : 0ull);
        }
        __syncthreads();
        const uint64_t ge = L.giant_end;
        tile_words += (uint32_t)(ge >> 63);
        cb = ge & ~(1ull << 63);
        // sentences that start before the new cb are done with
        uint32_t mine = 0;
        for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
          if (sent_off[s] >= cb) break;
          mine++;
        }
        for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
        s_next += mine;
        __syncthreads();
        if (cb >= span_end) break;
        continue;
      }
      ce = (uint32_t)cut;
    }

    // ---- one lane per word that starts before the chunk end
    uint32_t chunk_words = 0;
    for (uint32_t k0 = 0; k0 < nw; k0 += 64) {
      const uint32_t k = k0 + lane;
      const uint32_t s = k < nw ? L.wl[k] : 0xFFFFu;
      const bool mine = k < nw && s < ce;
      if (mine && !count_only) {
        // end of the word: the next stop position after s
        uint32_t w = s >> 6;
        unsigned long long m = (s & 63) == 63 ? 0ull : (L.endm[w] & ~((2ull << (s & 63)) - 1ull));
        while (!m && w + 1 < nblk) m = L.endm[++w];
        uint32_t e = m ? w * 64 + (uint32_t)__builtin_ctzll(m) : ce;
        if (e > ce) e = ce;
        word_insert(T, text, &L.txt[s], e - s, abase + s);
      }
      chunk_words += __popcll(__ballot(mine));
    }
    tile_words += chunk_words;
    if (last) break;
    cb = abase + ce;
    // sentences that start before the new cb are done with
    uint32_t gone = 0;
    for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
      if (sent_off[s] >= cb) break;
      gone++;
    }
    for (int d = 32; d >= 1; d >>= 1) gone += __shfl_xor(gone, d);
    s_next += gone;
    __syncthreads();
  }
  if (lane == 0 && tile_words) atomicAdd(n_words_out, (unsigned long long)tile_words);
}

__global__ void words_fill_kernel(unsigned long long *a, unsigned long long *b, uint32_t *c, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    a[i] = kSlotEmpty;
    b[i] = ~0ull;
    c[i] = 0;
  }
}

__global__ void words_collect_kernel(WordTable T, unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals,
                                     unsigned long long *__restrict__ n_out) {
  const uint64_t cap = 1ull << T.bits;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
    if (T.slot[i] != kSlotEmpty) {
      const unsigned long long k = atomicAdd(n_out, 1ull);
      keys[k] = T.first[i];
      vals[k] = (uint32_t)i;
    }
  }
}

// per unique word (in first-occurrence order): its code-point count and frequency
__global__ void words_measure_kernel(WordTable T, const uint8_t *__restrict__ text, const uint32_t *__restrict__ order,
                                     uint64_t n_uniq, uint32_t *__restrict__ nchar, uint32_t *__restrict__ freq) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_uniq) return;
  const uint32_t idx = order[i];
  const unsigned long long v = T.slot[idx];
  const uint32_t lf = (uint32_t)(v >> 40) & 0xFFu;
  const uint32_t len = lf == 255u ? T.biglen[idx] : lf;
  const uint8_t *p = text + (v & kOffMask);
  uint32_t n = 0;
  for (uint32_t j = 0; j < len; j++) n += !utf8_is_cont(p[j]);
  nchar[i] = n;
  freq[i] = T.cnt[idx];
}

// list(word) for every unique word; bitmap of the code points seen (bpe.py:75, :79-81)
__global__ void words_emit_kernel(WordTable T, const uint8_t *__restrict__ text, const uint32_t *__restrict__ order, uint64_t n_uniq,
                                  const uint64_t *__restrict__ woff, uint32_t *__restrict__ sym, uint32_t *__restrict__ seen) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_uniq) return;
  const uint32_t idx = order[i];
  const unsigned long long v = T.slot[idx];
  const uint32_t lf = (uint32_t)(v >> 40) & 0xFFu;
  const uint32_t len = lf == 255u ? T.biglen[idx] : lf;
  const uint8_t *p = text + (v & kOffMask);
  uint64_t o = woff[i];
  uint32_t j = 0;
  while (j < len) {
    const uint8_t b = p[j];
    int n = utf8_len(b);
    if (j + n > len) n = (int)(len - j);
    uint32_t cp = b;
    if (b >= 0x80 && n > 1) {
      cp = b & (0xFF >> (n + 1));
      for (int q = 1; q < n; q++) cp = (cp << 6) | (p[j + q] & 0x3F);
    }
    if (!utf8_is_cont(b)) {
      sym[o++] = cp;
      // test first: a letter is seen by every word, and atomics on one address serialise chip-wide (7 ms here before)
      if (cp < kNumCodePoints && !(__hip_atomic_load(&seen[cp >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (1u << (cp & 31))))
        atomicOr(&seen[cp >> 5], 1u << (cp & 31));
    }
    j += n < 1 ? 1 : n;
  }
}

__global__ void widen_scan_kernel(const uint32_t *__restrict__ in, uint64_t *__restrict__ out, uint64_t n) {
  // out[i] = in[i] as uint64 (the scan itself is done in place by rocPRIM on the widened array)
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[i];
}

static unsigned grid_of(uint64_t n, unsigned threads, unsigned cap) {
  uint64_t g = (n + threads - 1) / threads;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

int device_words_from_text(const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off, uint64_t n_sent, DeviceWords *out) {
  int rc = ensure_device();
  if (rc) return rc;
  if (n_bytes >= (1ull << 40)) return fail(SWT_ERR_UNSUPPORTED, "corpus shard larger than 1 TiB");
  const uint8_t *d_cls = nullptr;
  if ((rc = device_class_table(&d_cls))) return rc;
  out->d_sym = nullptr; out->d_woff = nullptr; out->d_freq = nullptr;
  out->n_words = 0; out->n_syms = 0; out->base_syms.clear();
  DevBuf plan, counter, seen;
  WordTable T{nullptr, nullptr, nullptr, nullptr, 0};
  unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
  uint32_t *d_vals = nullptr, *d_vals2 = nullptr, *d_nchar = nullptr;
  void *d_tmp = nullptr;
  auto cleanup = [&]() {
    plan.release(); counter.release(); seen.release();
    for (void *p : {(void *)T.slot, (void *)T.first, (void *)T.cnt, (void *)T.biglen, (void *)d_keys, (void *)d_keys2, (void *)d_vals,
                    (void *)d_vals2, (void *)d_nchar, d_tmp})
      if (p) (void)hipFree(p);
  };
  auto drop_out = [&]() {  // an error: the caller gets nothing it would have to free
    for (void *p : {(void *)out->d_sym, (void *)out->d_woff, (void *)out->d_freq})
      if (p) (void)hipFree(p);
    out->d_sym = nullptr; out->d_woff = nullptr; out->d_freq = nullptr;
  };
#define W_HIP(expr)                                                                                         \
  do {                                                                                                      \
    hipError_t _e = (expr);                                                                                 \
    if (_e != hipSuccess) {                                                                                 \
      cleanup();                                                                                            \
      drop_out();                                                                                           \
      return fail(SWT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
    }                                                                                                       \
  } while (0)
  const uint64_t n_tiles = tile_count(n_bytes, kWTile);
  if ((rc = plan.reserve((n_tiles + 1) * 8)) || (rc = counter.reserve(64)) || (rc = seen.reserve(kNumCodePoints / 8 + 64))) {
    cleanup();
    return rc;
  }
  unsigned long long *d_cnt = counter.as<unsigned long long>();
  W_HIP(hipMemset(d_cnt, 0, 64));
  uint64_t n_occ = 0;
  if (n_sent && n_bytes) {
    launch_plan(d_sent_off, n_sent, n_tiles, kWTile, plan.as<uint64_t>(), 0);
    // pass 1: how many word occurrences (sizes the table)
    hipLaunchKernelGGL(census_kernel, dim3((unsigned)n_tiles), dim3(64), 0, 0, d_text, n_bytes, d_sent_off, plan.as<uint64_t>(), d_cls,
                       T, true, d_cnt);
    W_HIP(hipMemcpy(&n_occ, d_cnt, 8, hipMemcpyDeviceToHost));
  }
  uint64_t n_uniq = 0;
  if (n_occ) {
    uint32_t bits = 10;
    while ((1ull << bits) < 2 * n_occ + 16) bits++;
    if (bits > 32) { cleanup(); return fail(SWT_ERR_UNSUPPORTED, "too many words for one shard"); }
    const uint64_t cap = 1ull << bits;
    T.bits = bits;
    W_HIP(hipMalloc((void **)&T.slot, cap * 8));
    W_HIP(hipMalloc((void **)&T.first, cap * 8));
    W_HIP(hipMalloc((void **)&T.cnt, cap * 4));
    W_HIP(hipMalloc((void **)&T.biglen, cap * 4));
    hipLaunchKernelGGL(words_fill_kernel, dim3(grid_of(cap, 256, 8192)), dim3(256), 0, 0, T.slot, T.first, T.cnt, cap);
    // pass 2: the census
    hipLaunchKernelGGL(census_kernel, dim3((unsigned)n_tiles), dim3(64), 0, 0, d_text, n_bytes, d_sent_off, plan.as<uint64_t>(), d_cls,
                       T, false, d_cnt + 1);
    // unique words, ordered by first occurrence
    W_HIP(hipMalloc((void **)&d_keys, (n_occ + 1) * 8));
    W_HIP(hipMalloc((void **)&d_keys2, (n_occ + 1) * 8));
    W_HIP(hipMalloc((void **)&d_vals, (n_occ + 1) * 4));
    W_HIP(hipMalloc((void **)&d_vals2, (n_occ + 1) * 4));
    hipLaunchKernelGGL(words_collect_kernel, dim3(grid_of(cap, 256, 8192)), dim3(256), 0, 0, T, d_keys, d_vals, d_cnt + 2);
    W_HIP(hipMemcpy(&n_uniq, d_cnt + 2, 8, hipMemcpyDeviceToHost));
    size_t tmp_bytes = 0;
    W_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys, d_keys2, d_vals, d_vals2, (size_t)n_uniq, 0u, 40u));
    W_HIP(hipMalloc(&d_tmp, tmp_bytes + 16));
    W_HIP(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_vals, d_vals2, (size_t)n_uniq, 0u, 40u));
    (void)hipFree(d_tmp);
    d_tmp = nullptr;
    // symbol counts, offsets, symbols
    W_HIP(hipMalloc((void **)&d_nchar, (n_uniq + 1) * 4));
    W_HIP(hipMalloc((void **)&out->d_freq, (n_uniq + 1) * 4));
    W_HIP(hipMalloc((void **)&out->d_woff, (n_uniq + 2) * 8));
    hipLaunchKernelGGL(words_measure_kernel, dim3(grid_of(n_uniq, 256, 1u << 22)), dim3(256), 0, 0, T, d_text, d_vals2, n_uniq, d_nchar,
                       out->d_freq);
    W_HIP(hipMemset(d_nchar + n_uniq, 0, 4));
    hipLaunchKernelGGL(widen_scan_kernel, dim3(grid_of(n_uniq + 1, 256, 8192)), dim3(256), 0, 0, d_nchar, out->d_woff, n_uniq + 1);
    W_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, out->d_woff, out->d_woff, (uint64_t)0, (size_t)(n_uniq + 1), rocprim::plus<uint64_t>()));
    W_HIP(hipMalloc(&d_tmp, tmp_bytes + 16));
    W_HIP(rocprim::exclusive_scan(d_tmp, tmp_bytes, out->d_woff, out->d_woff, (uint64_t)0, (size_t)(n_uniq + 1), rocprim::plus<uint64_t>()));
    uint64_t n_syms = 0;
    W_HIP(hipMemcpy(&n_syms, out->d_woff + n_uniq, 8, hipMemcpyDeviceToHost));
    W_HIP(hipMalloc((void **)&out->d_sym, (n_syms + 16) * 4));
    W_HIP(hipMemset(seen.p, 0, kNumCodePoints / 8 + 64));
    hipLaunchKernelGGL(words_emit_kernel, dim3(grid_of(n_uniq, 256, 1u << 22)), dim3(256), 0, 0, T, d_text, d_vals2, n_uniq,
                       out->d_woff, out->d_sym, seen.as<uint32_t>());
    std::vector<uint32_t> bitmap(kNumCodePoints / 32);
    W_HIP(hipMemcpy(bitmap.data(), seen.p, kNumCodePoints / 8, hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < kNumCodePoints; c++)
      if (bitmap[c >> 5] & (1u << (c & 31))) out->base_syms.push_back(c);
    out->n_words = n_uniq;
    out->n_syms = n_syms;
  } else {
    W_HIP(hipMalloc((void **)&out->d_sym, 64));
    W_HIP(hipMalloc((void **)&out->d_woff, 16));
    W_HIP(hipMalloc((void **)&out->d_freq, 16));
    W_HIP(hipMemset(out->d_woff, 0, 16));
  }
  W_HIP(hipDeviceSynchronize());
  cleanup();
#undef W_HIP
  return SWT_OK;
}

}  // namespace swt
