// swt_bpe_train.hip -- BPE / WordPiece training merge loop on gfx950.
//
// Replaces NaiveBPE.train's loop (FastBPE.train inherits it):
//   word dedup + symbolisation   /root/reference/source/bpe.py:73-81   (swt_words.hip, on the device)
//   pair histogram               /root/reference/source/bpe.py:90-95
//   argmax with first-seen tie   /root/reference/source/bpe.py:98-102
//   merge-apply (_replace_pair)  /root/reference/source/bpe.py:25-48, 108-111
//
// The reference recounts every pair and rewrites every word on every merge.  Here NOTHING is rescanned per merge:
//
//   sym[]    packed uint32 symbol stream of the UNIQUE words, word w at [woff[w], woff[w+1]).  A merge writes the merged
//            symbol over `left` and a HOLE over `right`: addresses never move, so (word, offset) is a stable first-occurrence
//            key (bpe.py:102) and an index of word ids never goes stale by position.
//   pair histogram   open-addressing hash keys[] (left<<32|right) + cnt[] (64-bit, weighted by word frequency), built once
//            (LDS-staged per-wave pair counters, flushed with one global atomic per distinct key), then kept exact
//            incrementally: a merge subtracts the pairs it destroys and adds the pairs it creates.
//   inverted index   pair -> words that hold it.  Every occurrence of a pair (x, y) is created in ONE step -- the step that
//            created the later-born of x and y -- and only disappears afterwards.  So the index is: a static table grouped by
//            key for the pairs of initial symbols (k0), plus one log segment per merge step holding (tag, word) for every
//            new-pair occurrence that step created.  apply_kernel visits only those words (a stale entry is a word that no
//            longer holds the pair: skipped); a word is claimed by atomicMax on its stamp and rewritten by one lane.
//   candidates   (BPE) the slots whose count is >= theta.  Counts only fall, except for the pairs a merge creates, whose
//            count is caught as it crosses theta; so the maximum of the table is the maximum of the candidate list while
//            that maximum stays >= theta.  The argmax scans a few thousand candidates instead of the table; when the list
//            runs dry (or overflows) the step reports it and the host re-plans theta from a histogram of the counts.
//            WordPiece scores rise and fall with the symbol frequencies, so that mode keeps the full-table argmax.
//   tie-break    bpe.py:102: among the pairs holding the maximum, the one whose first occurrence in scan order comes first.
//            Words are scanned from a cursor: while the maximum stays at one level c, every pair with count c lies at or
//            after the word where the last tie-break found its winner (new pairs appear only in words a merge touched, and
//            the cursor is pulled back to the first touched word), so a plateau of k tied pairs costs one sweep, not k.
//
// Per merge, enqueued back to back with no host round trip (swt_bpe_train_run, up to 256 merges per batch):
//   cand_argmax (or argmax_full) -> tie_kernel -> decide_kernel -> apply_kernel
// Workgroup partials are combined redundantly by the consumers (a kernel boundary is the barrier; no tickets, no
// __threadfence chains).
//
// WordPiece mode (NaiveWP.train, /root/reference/source/wordpiece.py:29-103): the same stream, histogram,
// index and merge-apply; symbols are the word's first character and "##c" (= 0x110000 + c) for the others; a dense
// symbol-frequency array is kept exact beside the pair histogram, and the argmax key is the likelihood score
// freq / (f_left * f_right) -- Python's int / int, i.e. the correctly rounded quotient -- compared as a bit pattern.
//
// Sharded training (one process per GPU, swt_dist.hip): every rank keeps the histogram of the WHOLE corpus.  apply_kernel
// then accumulates its deltas per table slot in pend[] and lists the touched slots; the runner packs them into a fixed-size
// record block, all-gathers the blocks (RCCL) and every rank adds every block to its replica.
#include <algorithm>
#include <cstdlib>
#include <ctime>
#include <rocprim/device/device_scan.hpp>

#include "swt_common.h"
#include "swt_train.h"
#include "swt_words.h"

namespace swt {

constexpr int kTrainThreads = 256;
constexpr uint32_t kHole = 0xFFFFFFFFu;

// ---- WordPiece score (wordpiece.py:84-87) --------------------------------------------------------------------------
constexpr uint32_t kWpCont = 0x110000u;         // "##c" = kWpCont + c
constexpr uint32_t kWpMergedBase = 0x220000u;   // merged symbols of a WordPiece trainer start here
constexpr uint64_t kWpSymCap = (uint64_t)kWpMergedBase + (1u << 21);

// RN(cnt / (fl * fr)) as the double's bit pattern.  Below 2^53 both operands are exact doubles and the IEEE division rounds
// once; above, the mantissa comes from a bitwise long division of the exact integers.
__device__ __forceinline__ unsigned long long wp_score_bits(unsigned long long cnt, unsigned long long fl, unsigned long long fr) {
  if (cnt == 0 || fl == 0 || fr == 0) return 0ull;
  const unsigned __int128 d0 = (unsigned __int128)fl * fr;
  if (d0 < ((unsigned __int128)1 << 53) && cnt < (1ull << 53))
    return (unsigned long long)__double_as_longlong((double)cnt / (double)(unsigned long long)d0);
  unsigned __int128 d = d0, r = cnt;
  int e = 0;
  while (r < d) { r <<= 1; e--; }
  while (r >= 2 * d) { d <<= 1; e++; }
  unsigned long long mant = 1;
  r -= d;
  for (int i = 0; i < 52; i++) {
    r <<= 1;
    mant <<= 1;
    if (r >= d) { r -= d; mant |= 1; }
  }
  r <<= 1;
  const bool rb = r >= d;
  if (rb) r -= d;
  if (rb && (r != 0 || (mant & 1))) mant++;
  if (mant == (1ull << 53)) { mant >>= 1; e++; }
  return ((unsigned long long)(e + 1023) << 52) | (mant & ((1ull << 52) - 1ull));
}

// the value the argmax maximises for a live pair: its count (BPE) or its score (WordPiece)
__device__ __forceinline__ unsigned long long pair_value(unsigned long long key, long long cnt, const long long *__restrict__ sfreq) {
  if (!sfreq) return (unsigned long long)cnt;
  return wp_score_bits((unsigned long long)cnt, (unsigned long long)sfreq[key >> 32], (unsigned long long)sfreq[(uint32_t)key]);
}

// ---- pair table --------------------------------------------------------------------------------------------------------
// Insert-or-find; returns the slot.  Keys are never removed (a rehash drops the dead ones).
// The host keeps the load factor below 1/2 (ensure_room), so a probe sequence is short; should that bound ever be wrong, a
// full table must not become a hang: after one whole turn the insert gives up, raises kFlagTableFull (the host turns it into
// SWT_ERR_STATE at its next look) and hands back the home slot -- a valid index, so nothing faults; the counts are void.
__device__ __forceinline__ uint32_t table_slot(const PairTable &T, unsigned long long key, TrainState *st) {
  const uint32_t mask = (uint32_t)((1ull << T.bits) - 1ull);
  uint32_t h = hash_slot(key, T.bits);
  for (uint32_t turn = 0; turn <= mask; turn++) {
    unsigned long long k = __hip_atomic_load(&T.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == kEmptyKey) {
      k = atomicCAS(&T.keys[h], kEmptyKey, key);
      if (k == kEmptyKey) {
        atomicAdd(&st->n_used, 1ull);
        k = key;
      }
    }
    if (k == key) return h;
    h = (h + 1) & mask;
  }
  atomicOr(&st->flags, kFlagTableFull);
  return hash_slot(key, T.bits);
}

constexpr uint32_t kCidxPending = 0xFFFFFFFEu;  // cidx[slot]: the slot was listed by the exchange's settle pass; its mirror comes with the next step

__device__ __forceinline__ void cand_push(const TrainCtx &C, uint32_t slot) {
  // past the capacity the list has lost a candidate: n_cand says so, and the next argmax refuses to answer (the flag is not
  // raised here: workgroups of the apply launch that is running must not see it change)
  const unsigned long long k = atomicAdd(&C.st->n_cand, 1ull);
  if (k < C.cand_cap) C.cand[k] = slot;
}

// count += delta with the candidate invariant kept: a count that crosses theta upwards joins the list (only the pairs a
// merge creates ever rise)
__device__ __forceinline__ void count_add(const TrainCtx &C, uint32_t slot, long long delta) {
  if (C.cidx) {  // a listed pair keeps its compact copy in step (kCidxPending: listed, not mirrored yet)
    const uint32_t ci = C.cidx[slot];
    if (ci < kCidxPending)
      (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&C.ccnt[ci]), (unsigned long long)delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (delta > 0 && C.theta) {
    const long long o = (long long)atomicAdd(reinterpret_cast<unsigned long long *>(&C.T.cnt[slot]), (unsigned long long)delta);
    if (o < (long long)C.theta && o + delta >= (long long)C.theta) cand_push(C, slot);
  } else {
    (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&C.T.cnt[slot]), (unsigned long long)delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// one histogram delta of a merge: straight into the replica (single GPU), or into pend[] + the touched list (sharded: the
// exchange adds every rank's record block, this rank's included, to every replica)
__device__ __forceinline__ void table_add(const TrainCtx &C, unsigned long long key, long long delta) {
  const uint32_t h = table_slot(C.T, key, C.st);
  if (!C.pend) {
    count_add(C, h, delta);
    return;
  }
  atomicAdd(reinterpret_cast<unsigned long long *>(&C.pend[h]), (unsigned long long)delta);
  if (atomicMax(&C.tstamp[h], C.step) < C.step) {
    const unsigned long long k = atomicAdd(&C.st->n_touched, 1ull);
    if (k < C.touched_cap) C.touched[k] = h;
  }
}

__device__ __forceinline__ long long table_get(const PairTable &T, unsigned long long key) {
  const uint32_t mask = (uint32_t)((1ull << T.bits) - 1ull);
  uint32_t h = hash_slot(key, T.bits);
  for (uint32_t turn = 0; turn <= mask; turn++) {
    const unsigned long long k = T.keys[h];
    if (k == key) return T.cnt[h];
    if (k == kEmptyKey) return 0;
    h = (h + 1) & mask;
  }
  return 0;  // a full table without the key (see table_slot)
}

__global__ void table_clear_kernel(unsigned long long *keys, long long *cnt, uint64_t cap) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
    keys[i] = kEmptyKey;
    cnt[i] = 0;
  }
}

// ---- bpe.py:90-95 once -------------------------------------------------------------------------------------------------
// Every adjacent pair of every unique word, weighted by the word's frequency.  One lane per word, one wave per run of 64
// words; the wave's pair counters are staged in LDS (a 512-slot table per wave, linear probing): the pairs of natural text
// are Zipfian, so most additions meet an LDS counter and only the distinct keys of a wave's words reach the global table,
// one atomicAdd each.
constexpr int kHistSlots = 512;
__global__ __launch_bounds__(kTrainThreads) void hist_build_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                                   const uint32_t *__restrict__ freq, uint64_t n_words, TrainCtx C) {
  __shared__ unsigned long long lk[kTrainThreads / 64][kHistSlots];
  __shared__ unsigned long long lc[kTrainThreads / 64][kHistSlots];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = lane; i < kHistSlots; i += 64) { lk[wave][i] = kEmptyKey; lc[wave][i] = 0; }
  __syncthreads();
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n_words) {
    const uint64_t b0 = woff[w], b1 = woff[w + 1];
    const unsigned long long f = freq[w];
    uint32_t a = b0 < b1 ? sym[b0] : 0;
    for (uint64_t i = b0 + 1; i < b1; i++) {
      const uint32_t b = sym[i];
      const unsigned long long key = pair_key(a, b);
      a = b;
      uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 55);  // 9 bits
      bool done = false;
      for (int probe = 0; probe < 8 && !done; probe++) {
        unsigned long long k = lk[wave][h];
        if (k == kEmptyKey) {
          k = atomicCAS(&lk[wave][h], kEmptyKey, key);
          if (k == kEmptyKey) k = key;
        }
        if (k == key) { atomicAdd(&lc[wave][h], f); done = true; }
        h = (h + 1) & (kHistSlots - 1);
      }
      if (!done) count_add(C, table_slot(C.T, key, C.st), (long long)f);  // the wave's table is crowded: straight to global
    }
  }
  __syncthreads();
  for (int i = lane; i < kHistSlots; i += 64) {
    const unsigned long long k = lk[wave][i];
    if (k != kEmptyKey) count_add(C, table_slot(C.T, k, C.st), (long long)lc[wave][i]);
  }
}

// ---- the static index of the initial pairs (k0) ------------------------------------------------------------------------
__device__ __forceinline__ uint32_t k0_find(const K0Index &K, unsigned long long key) {
  const uint32_t mask = (uint32_t)((1ull << K.bits) - 1ull);
  uint32_t h = hash_slot(key, K.bits);
  for (uint32_t turn = 0; turn <= mask; turn++) {
    const unsigned long long k = K.keys[h];
    if (k == key) return h;
    if (k == kEmptyKey) return 0xFFFFFFFFu;
    h = (h + 1) & mask;
  }
  return 0xFFFFFFFFu;
}

// pass 0: occurrences per key; pass 1: fill the lists (start[] holds each list's base by then).  One lane per word, four
// pairs per lane and trip.  The pairs of natural text are Zipfian: a lane-per-occurrence atomicAdd on fill[] serialised on the
// hot keys (1.26 ms per pass on S85k-open, against 0.1-0.3 ms for hist_build_kernel, which visits the same pairs).  So, like
// there, a wave stages its trip's keys in an LDS table (512 slots, linear probing): the LDS atomic gives every occurrence its
// rank among the wave's occurrences of that key, ONE global atomicAdd per distinct key reserves the run, and the lanes write
// their words at run + rank.  A key that finds no LDS slot in eight probes goes to the global table directly.
constexpr int kK0Slots = 512, kK0Fan = 4;
__global__ __launch_bounds__(kTrainThreads) void k0_pass_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                                uint64_t n_words, K0Index K, int pass) {
  __shared__ unsigned long long lk[kTrainThreads / 64][kK0Slots];
  __shared__ uint32_t lc[kTrainThreads / 64][kK0Slots];   // occurrences in this trip
  __shared__ uint32_t lb[kTrainThreads / 64][kK0Slots];   // pass 1: where the wave's run of this key starts in K.words
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = lane; i < kK0Slots; i += 64) { lk[wave][i] = kEmptyKey; lc[wave][i] = 0; }
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t b0 = 0, b1 = 0;
  if (w < n_words) { b0 = woff[w]; b1 = woff[w + 1]; }
  __syncthreads();
  for (uint64_t base = b0;; base += kK0Fan) {
    const bool more = base + 1 < b1;
    if (!__syncthreads_or(more)) break;  // every wave of the workgroup makes the same number of trips (the barriers below)
    unsigned long long key[kK0Fan];
    int slot[kK0Fan];
    uint32_t rank[kK0Fan];
    uint32_t s5[kK0Fan + 1];
#pragma unroll
    for (int u = 0; u <= kK0Fan; u++) s5[u] = more && base + u < b1 ? sym[base + u] : kHole;
#pragma unroll
    for (int u = 0; u < kK0Fan; u++) {
      key[u] = more && base + u + 1 < b1 ? pair_key(s5[u], s5[u + 1]) : kEmptyKey;
      slot[u] = -1;
      rank[u] = 0;
      if (key[u] == kEmptyKey) continue;
      uint32_t h = (uint32_t)((key[u] * 0x9E3779B97F4A7C15ull) >> 55);  // 9 bits
      for (int probe = 0; probe < 8; probe++) {
        unsigned long long k = lk[wave][h];
        if (k == kEmptyKey) {
          k = atomicCAS(&lk[wave][h], kEmptyKey, key[u]);
          if (k == kEmptyKey) k = key[u];
        }
        if (k == key[u]) {
          slot[u] = (int)h;
          rank[u] = atomicAdd(&lc[wave][h], 1u);
          break;
        }
        h = (h + 1) & (kK0Slots - 1);
      }
    }
    __syncthreads();
    // one lane per staged key: its slot in the index, one global atomic for the wave's occurrences
    for (int i = lane; i < kK0Slots; i += 64) {
      const unsigned long long k = lk[wave][i];
      if (k == kEmptyKey) continue;
      const uint32_t h = k0_find(K, k);
      uint32_t at = 0xFFFFFFFFu;
      if (h != 0xFFFFFFFFu) {
        const uint32_t got = atomicAdd(&K.fill[h], lc[wave][i]);
        if (pass == 1) at = K.start[h] + got;
      }
      lb[wave][i] = at;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kK0Fan; u++) {
      if (key[u] == kEmptyKey) continue;
      if (slot[u] >= 0) {
        if (pass == 1 && lb[wave][slot[u]] != 0xFFFFFFFFu) K.words[(uint64_t)lb[wave][slot[u]] + rank[u]] = (uint32_t)w;
      } else {  // the wave's table was crowded: straight to the global one
        const uint32_t h = k0_find(K, key[u]);
        if (h != 0xFFFFFFFFu) {
          const uint32_t got = atomicAdd(&K.fill[h], 1u);
          if (pass == 1) K.words[(uint64_t)K.start[h] + got] = (uint32_t)w;
        }
      }
    }
    __syncthreads();
    for (int i = lane; i < kK0Slots; i += 64) { lk[wave][i] = kEmptyKey; lc[wave][i] = 0; }
    __syncthreads();
  }
}

// list bases by wave-aggregated allocation (the order of the lists is irrelevant): len[] = fill[] of pass 0
__global__ __launch_bounds__(kTrainThreads) void k0_alloc_kernel(K0Index K, unsigned long long *cursor) {
  const uint64_t cap = 1ull << K.bits;
  const int lane = threadIdx.x & 63;
  for (uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; i0 < cap; i0 += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = i0 + lane;
    const uint32_t n = i < cap ? K.fill[i] : 0u;
    uint32_t x = n;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    const uint32_t total = __shfl(x, 63);
    unsigned long long base = 0;
    if (lane == 0 && total) base = atomicAdd(cursor, (unsigned long long)total);
    base = __shfl(base, 0);
    if (i < cap) {
      K.start[i] = (uint32_t)(base + x - n);
      K.len[i] = n;
      K.fill[i] = 0;
    }
  }
}

// ---- argmax ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void arg_combine(unsigned long long &m, unsigned long long &c, unsigned long long &k,
                                            unsigned long long m2, unsigned long long c2, unsigned long long k2) {
  if (m2 > m) { m = m2; c = c2; k = k2; }
  else if (m2 == m) { c += c2; k = k2 < k ? k2 : k; }
}

__device__ __forceinline__ void arg_wave_reduce(unsigned long long &m, unsigned long long &c, unsigned long long &k) {
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long m2 = __shfl_xor(m, d), c2 = __shfl_xor(c, d), k2 = __shfl_xor(k, d);
    arg_combine(m, c, k, m2, c2, k2);
  }
}

// the workgroup's (max, pairs holding it, smallest such key) -> parts[blockIdx.x]
__device__ __forceinline__ void arg_publish(unsigned long long m, unsigned long long c, unsigned long long k, ArgPart *__restrict__ parts) {
  __shared__ unsigned long long sm[4], sc[4], sk[4];
  arg_wave_reduce(m, c, k);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sm[wave] = m; sc[wave] = c; sk[wave] = k; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) arg_combine(m, c, k, sm[w], sc[w], sk[w]);
    parts[blockIdx.x].mx = m;
    parts[blockIdx.x].cnt = c;
    parts[blockIdx.x].key = k;
  }
}

// every consumer combines the producer's partials itself (one wave, n_parts <= kArgParts): the kernel boundary is the barrier
__device__ __forceinline__ void arg_collect(const ArgPart *__restrict__ parts, uint32_t n_parts, unsigned long long &m,
                                            unsigned long long &c, unsigned long long &k) {
  const int lane = threadIdx.x & 63;
  m = 0; c = 0; k = kEmptyKey;
  for (uint32_t j = lane; j < n_parts; j += 64) {
    const unsigned long long m2 = parts[j].mx;
    if (m2 > 0) arg_combine(m, c, k, m2, parts[j].cnt, parts[j].key);
  }
  arg_wave_reduce(m, c, k);
}

// the candidate list cannot answer: it overflowed, or its maximum fell below theta (theta 1 lists every live pair)
__device__ __forceinline__ bool cand_dry(const TrainCtx &C, unsigned long long mx) {
  return C.theta && ((C.st->flags & kFlagReplan) || C.st->n_cand > C.cand_cap || (mx < C.theta && C.theta > 1));
}

// BPE: maximum over the candidate list (the slots whose count is >= theta)
__global__ __launch_bounds__(256) void cand_argmax_kernel(TrainCtx C, ArgPart *__restrict__ parts) {
  unsigned long long m = 0, c = 0, k = kEmptyKey;
  unsigned long long n = C.st->n_cand;
  if (n > C.cand_cap) n = C.cand_cap;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t slot = C.cand[i];
    const long long v = C.T.cnt[slot];
    if (v > 0 && (unsigned long long)v >= m) arg_combine(m, c, k, (unsigned long long)v, 1ull, C.T.keys[slot]);
  }
  arg_publish(m, c, k, parts);
}

// WordPiece: maximum SCORE over the list of every live pair (theta = 1), read from its compact mirror.  The score of a pair
// moves whenever the frequency of one of its symbols does (wordpiece.py:84-87), so no count threshold bounds it -- but the
// LIST of live pairs is still far shorter than the table it lives in (S85k-lex: ~0.5 M pairs in 4 M slots of 16 B), and its
// mirror is two dense streams: the argmax went from a 67 MB table scan per merge to ~10 MB.  The entries pushed since the
// last step (the pairs the last merge created) are gathered from the table here and mirrored on the way -- no count moves
// while this launch runs -- and decide_kernel moves n_synced up afterwards.
__global__ __launch_bounds__(256) void wp_list_argmax_kernel(TrainCtx C, ArgPart *__restrict__ parts) {
  unsigned long long m = 0, c = 0, k = kEmptyKey;
  unsigned long long n = C.st->n_cand;
  if (n > C.cand_cap) n = C.cand_cap;
  const unsigned long long n_synced = C.st->n_synced < n ? C.st->n_synced : n;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
    long long v[4];
    unsigned long long key[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {  // four independent streams of loads in flight
      const uint64_t i = i0 + u * stride;
      v[u] = 0;
      key[u] = kEmptyKey;
      if (i < n_synced) { v[u] = C.ccnt[i]; key[u] = C.ckey[i]; }
      else if (i < n) {
        const uint32_t slot = C.cand[i];
        v[u] = C.T.cnt[slot];
        key[u] = C.T.keys[slot];
        C.ccnt[i] = v[u];
        C.ckey[i] = key[u];
        C.cidx[slot] = (uint32_t)i;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (v[u] > 0 && key[u] != kEmptyKey) {
        const unsigned long long val = pair_value(key[u], v[u], C.sfreq);
        if (val >= m) arg_combine(m, c, k, val, 1ull, key[u]);
      }
    }
  }
  arg_publish(m, c, k, parts);
}

// The BPE fallback (and a WordPiece table whose list does not fit): maximum over the whole table.  The counts are streamed two per load, four loads in
// flight per lane (cap is a power of two >= 1024); a key is only fetched for a count that can still win.
__global__ __launch_bounds__(256) void argmax_full_kernel(const unsigned long long *__restrict__ keys, const long long *__restrict__ cnt,
                                                          uint64_t cap, ArgPart *__restrict__ parts, const long long *__restrict__ sfreq) {
  unsigned long long m = 0, c = 0, k = kEmptyKey;
  const uint64_t n2 = cap >> 1;
  const longlong2 *cnt2 = reinterpret_cast<const longlong2 *>(cnt);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n2; i0 += 4 * stride) {
    longlong2 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t i = i0 + u * stride;
      v[u] = i < n2 ? cnt2[i] : make_longlong2(0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t i = i0 + u * stride;
      const long long vv[2] = {v[u].x, v[u].y};
#pragma unroll
      for (int h = 0; h < 2; h++) {
        if (vv[h] > 0 && (sfreq || (unsigned long long)vv[h] >= m)) {
          const unsigned long long key = keys[2 * i + h];
          if (key != kEmptyKey) {
            const unsigned long long val = pair_value(key, vv[h], sfreq);
            if (val >= m) arg_combine(m, c, k, val, 1ull, key);
          }
        }
      }
    }
  }
  arg_publish(m, c, k, parts);
}

// ---- bpe.py:102 tie-break ----------------------------------------------------------------------------------------------
// The earliest (word, offset) whose pair holds the maximum.  An untied step costs a handful of workgroups that return at once.
// best_pos = word << 32 | offset of the pair's left symbol inside the word.
__global__ __launch_bounds__(kTrainThreads) void tie_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                            uint64_t n_words, TrainCtx C, const ArgPart *__restrict__ parts, uint32_t n_parts) {
  __shared__ unsigned long long s_mx, s_tied;
  if (threadIdx.x < 64) {
    unsigned long long m, c, k;
    arg_collect(parts, n_parts, m, c, k);
    if (threadIdx.x == 0) { s_mx = m; s_tied = c; }
  }
  __syncthreads();
  if (s_tied < 2 || s_mx == 0) return;
  if (cand_dry(C, s_mx)) return;  // decide_kernel reports it
  const unsigned long long mx = s_mx;
  // plateau cursor (BPE only: a WordPiece score moves whenever a symbol frequency does)
  const uint64_t start = (!C.sfreq && C.st->plateau == mx) ? (uint64_t)C.st->cursor_w : 0ull;
  for (uint64_t w = start + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
    if ((w << 32) >= __hip_atomic_load(&C.st->best_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // words only get later
    const uint64_t b0 = woff[w], b1 = woff[w + 1];
    uint32_t a = kHole;
    uint64_t ai = 0;
    for (uint64_t i = b0; i < b1; i++) {
      const uint32_t b = sym[i];
      if (b == kHole) continue;
      if (a != kHole) {
        const unsigned long long key = pair_key(a, b);
        if (pair_value(key, table_get(C.T, key), C.sfreq) == mx) {
          atomicMin(&C.st->best_pos, (unsigned long long)((w << 32) | (ai - b0)));
          break;
        }
      }
      a = b;
      ai = i;
    }
  }
}

// WordPiece: the symbol frequencies follow the merge.  A pair of two different symbols cannot overlap itself, so the merge
// happens exactly count(l, r) times (weighted); a twin pair (a, a) is counted by apply_kernel, occurrence by occurrence.
__device__ __forceinline__ void wp_move_freq(const PairTable &T, uint32_t l, uint32_t r, uint32_t m, long long *sfreq) {
  if (l == r) return;
  const long long c = table_get(T, pair_key(l, r));
  sfreq[l] -= c;
  sfreq[r] -= c;
  sfreq[m] += c;
}

__device__ __forceinline__ uint32_t seg_of_symbol(const TrainCtx &C, uint32_t s) {
  if (s < C.id_base) return 0u;
  const uint32_t k = s - C.id_base;
  if (k >= C.seg_cap) return 0u;
  const uint32_t v = C.seg_of[k];
  return v == kSegBase ? 0u : v;
}

// the pair at a tie position: the symbol there and the next live one of its word
__device__ __forceinline__ unsigned long long pair_at(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff, unsigned long long pos) {
  const uint64_t w = pos >> 32, b1 = woff[w + 1];
  uint64_t i = woff[w] + (uint32_t)pos;
  const uint32_t a = sym[i];
  uint32_t b = kHole;
  for (i++; i < b1; i++) {
    b = sym[i];
    if (b != kHole) break;
  }
  return pair_key(a, b);
}

// a step begins: its index segment starts where the log stands (and ends where the next step's begins); the merged symbol is
// born in this step
__device__ __forceinline__ void open_step(const TrainCtx &C, uint32_t merged, bool valid) {
  TrainState *st = C.st;
  C.seg_start[C.step] = st->idx_cursor;
  st->last_open = C.step;
  if (st->idx_cursor > C.idx_cap) st->flags |= kFlagIndexBroken;  // entries were dropped: whole-stream applies from here on
  if (!valid) return;
  if (merged >= C.id_base && merged - C.id_base < C.seg_cap && C.seg_of[merged - C.id_base] == 0) C.seg_of[merged - C.id_base] = C.step;
  else st->flags |= kFlagIndexBroken;  // the id already names a symbol: its pairs are no longer born in one step
}

// One wave: the step's decision.  Combines the partials, resolves a tie from best_pos, writes the merge command for
// apply_kernel and the step's log line, registers the merged symbol's index segment, moves the plateau cursor.
// cmd == nullptr: host-driven swt_bpe_train_best -- only the result fields are written.
__global__ __launch_bounds__(64) void decide_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff, TrainCtx C,
                                                    const ArgPart *__restrict__ parts, uint32_t n_parts, StepCmd *cmd, StepLog *log,
                                                    uint32_t log_i, uint32_t merged) {
  unsigned long long mx, tied, key;
  arg_collect(parts, n_parts, mx, tied, key);
  if (threadIdx.x != 0) return;
  TrainState *st = C.st;
  const bool dry = cand_dry(C, mx);
  unsigned long long pos = kEmptyKey;
  if (!dry && mx && tied >= 2) {
    pos = st->best_pos;
    key = pos != kEmptyKey ? pair_at(sym, woff, pos) : kEmptyKey;  // kEmptyKey: none of the tied pairs occurs in this shard
  }
  st->max_count = dry ? 0 : mx;
  st->n_tied = (dry || !mx) ? 0 : tied;
  st->best_key = key;
  st->win_key = key;
  st->res_pos = pos;
  st->best_pos = kEmptyKey;
  if (dry) st->flags |= kFlagReplan;
  if (C.sfreq && C.theta) st->n_synced = st->n_cand < C.cand_cap ? st->n_cand : C.cand_cap;  // wp_list_argmax_kernel mirrored up to here
  if (!dry && !C.sfreq) {
    if (tied >= 2 && pos != kEmptyKey) { st->plateau = mx; st->cursor_w = (uint32_t)(pos >> 32); }
    else if (st->plateau != mx) { st->plateau = mx; st->cursor_w = 0; }
  }
  if (!cmd) return;
  const bool ok = !dry && mx > 0 && key != kEmptyKey;
  cmd->l = (uint32_t)(key >> 32);
  cmd->r = (uint32_t)key;
  cmd->m = merged;
  cmd->valid = ok ? 1u : 0u;
  open_step(C, merged, ok);
  if (ok && C.sfreq) wp_move_freq(C.T, cmd->l, cmd->r, merged, C.sfreq);
  log[log_i].l = cmd->l;
  log[log_i].r = cmd->r;
  log[log_i].count = mx;
  log[log_i].flag = ok ? 0ull : (dry ? 3ull : 2ull);
  log[log_i].n_syms = st->n_syms;
  log[log_i].n_tied = tied;
  log[log_i].n_cand = st->n_cand;
}

// host-driven step (swt_bpe_train_apply): the command comes from the caller
__global__ void set_cmd_kernel(TrainCtx C, StepCmd *cmd, uint32_t l, uint32_t r, uint32_t m) {
  cmd->l = l; cmd->r = r; cmd->m = m; cmd->valid = 1u;
  open_step(C, m, true);
  if (C.sfreq) wp_move_freq(C.T, l, r, m, C.sfreq);
}

// wordpiece.py:78-81 once: symbol frequencies, weighted by the word's frequency
__global__ __launch_bounds__(kTrainThreads) void sym_hist_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                                 const uint32_t *__restrict__ freq, uint64_t n_words,
                                                                 long long *__restrict__ sfreq, uint64_t sym_cap, unsigned int *__restrict__ bad) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  const unsigned long long f = freq[w];
  for (uint64_t i = woff[w]; i < woff[w + 1]; i++) {
    if (sym[i] < sym_cap) atomicAdd(reinterpret_cast<unsigned long long *>(&sfreq[sym[i]]), f);
    else *bad = 1u;
  }
}

// wordpiece.py:54-57: [word[0]] + ["##" + c for c in word[1:]]
__global__ __launch_bounds__(kTrainThreads) void wp_symbolise_kernel(uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                                     uint64_t n_words) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  for (uint64_t i = woff[w] + 1; i < woff[w + 1]; i++) sym[i] += kWpCont;
}

__global__ void wp_live_symbols_kernel(const long long *__restrict__ sfreq, uint64_t cap, uint32_t *__restrict__ out, uint32_t out_cap,
                                       unsigned int *__restrict__ n_out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x)
    if (sfreq[i] != 0) {
      const unsigned int k = atomicAdd(n_out, 1u);
      if (k < out_cap) out[k] = (uint32_t)i;
    }
}

// ---- bpe.py:108-111 + _replace_pair (bpe.py:25-48), with the histogram kept exact ---------------------------------------
//   an old pair (x[i],x[i+1]) disappears iff x[i] or x[i+1] is consumed by an occurrence;
//   a new pair (y[j],y[j+1]) appears iff y[j] or y[j+1] is a freshly merged symbol.
// One lane per index entry (a word that held the pair when the pair was born).  The lane claims the word and walks it once,
// left to right through the holes, rewriting it in place; the histogram deltas it meets are parked in LDS (kEmitCap per
// lane), because each one costs a dependent trip to the pair table (~0.2 us: the table lives in the Infinity Cache) and
// issued one by one they would be the whole cost of a merge.  After the walk the wave reserves its new index entries with
// one atomic, and every lane flushes its parked deltas four at a time: four independent probes, then four atomics.
// ---- in-kernel time stamps (diagnostic builds only: SWT_EXTRA_FLAGS=-DSWT_STAMPS, tools/gpu_train_stamps.py) ---------------
// s_memrealtime is the chip-wide 100 MHz counter: per merge step the first start / last end over all workgroups of the two
// launches (= what the step costs on the device, without the launch), and the time per phase on one lane's critical path.
#ifdef SWT_STAMPS
constexpr uint32_t kSpanSteps = 16384;
__device__ unsigned long long g_span[4][kSpanSteps];  // tie first start, tie last end, apply first start, apply last end
__device__ unsigned long long g_phase[3][16];  // [2]: apply launch, every flushing lane of steps 1..64 (the big merges)
//        // [0] tie launch, workgroup 0; [1] apply launch, per step the first lane done flushing; [.][15] = samples
__device__ unsigned int g_reported[kSpanSteps];
__device__ unsigned long long g_pmax[6][kSpanSteps];  // per step: latest apply_body entry (absolute), then the longest of each of its 5 phases over the lanes
__device__ unsigned int g_kstep[kSpanSteps];  // merges the step carried
__device__ unsigned long long g_why[32];  // fast_apply_kernel, tied steps: [K] steps that merged K pairs; [16 + reason] why the batch ended
struct StampSpan {
  int k; uint32_t step;
  __device__ StampSpan(int k_, uint32_t step_) : k(k_), step(step_ & (kSpanSteps - 1)) {
    if (threadIdx.x == 0) atomicMin(&g_span[2 * k][step], __builtin_amdgcn_s_memrealtime());
  }
  __device__ ~StampSpan() {
    if (threadIdx.x == 0) atomicMax(&g_span[2 * k + 1][step], __builtin_amdgcn_s_memrealtime());
  }
};
__device__ __forceinline__ unsigned long long stamp_now() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // what was requested so far has arrived
  return __builtin_amdgcn_s_memrealtime();
}
#define SWT_SPAN(k, step) StampSpan span_((k), (step))
#define SWT_COUNT(i) do { if (C.step > 1000) atomicAdd(&g_why[(i)], 1ull); } while (0)
#define SWT_STAMP(arr, i) (arr)[(i)] = stamp_now()
#else
#define SWT_SPAN(k, step) do { } while (0)
#define SWT_COUNT(i) do { } while (0)
#define SWT_STAMP(arr, i) do { } while (0)
#endif

constexpr int kEmitCap = 10;
constexpr int kStage = 24;      // stream slots of a word staged in LDS before its walk
constexpr int kEntryFan = 8;   // index entries a lane looks at per trip of an apply launch
constexpr int kFlushBatch = 8;
#ifndef SWT_BIG_WORDS
#define SWT_BIG_WORDS 128
#endif
constexpr int kBigWords = SWT_BIG_WORDS;   // ... when a workgroup holds at least this many words of the merge in a trip
#ifndef SWT_BIG_MERGE
#define SWT_BIG_MERGE 8192
#endif
constexpr int kBigMerge = SWT_BIG_MERGE;  // index entries from which an apply launch sums its deltas per pair in LDS first
constexpr int kAggSlots = kTrainThreads * kStage * 4 / 16;  // (key, sum) slots the staging area holds  // parked deltas whose table probes go out together
constexpr unsigned long long kEmitNew = 1ull << 63;  // symbol ids stay below 2^31, so bit 63 of a pair key is free

// the slot of a key that is very likely in the table already: one plain load; anything else goes the insert-or-find way
__device__ __forceinline__ uint32_t slot_hint(const PairTable &T, unsigned long long key) { return hash_slot(key, T.bits); }

// What one apply launch merges: K pairs (l[q], r[q]) -> first_m + q, in this order.  K > 1 only on the fast path, for pairs
// that cannot affect each other (fast_apply_kernel says when); unused members hold kHole, which no live symbol equals.
struct BatchPlan {
  uint32_t K, first_m;
  uint32_t l[kMaxBatch], r[kMaxBatch];
  const uint32_t *list[kMaxBatch];  // member q's words: word ids ...
  const uint32_t *tags[kMaxBatch];  // ... and, in a log segment, the tags that tell its entries from other pairs'
  uint32_t want[kMaxBatch];
  unsigned long long ent0[kMaxBatch + 1];  // running sum of the lists' lengths
};

// where the words of member q are listed: the static index of the initial pairs, or the log segment of the step that made the
// later-born of its two symbols (every occurrence of a pair is created in that ONE step)
__device__ __forceinline__ TiedPlan plan_lookup(const TrainCtx &C, uint32_t l, uint32_t r) {
  const uint32_t sl = seg_of_symbol(C, l), sr = seg_of_symbol(C, r);
  const uint32_t seg = sl > sr ? sl : sr;
  TiedPlan t{0, 0, 0, 0};
  if (seg == 0) {
    const uint32_t h = k0_find(C.K, pair_key(l, r));
    if (h != 0xFFFFFFFFu) { t.kind = 1; t.start = C.K.start[h]; t.n_ent = C.K.len[h]; }
  } else {
    const uint64_t s0 = C.seg_start[seg], s1 = C.seg_start[seg + 1];
    t.kind = 2;
    t.start = s0;
    t.n_ent = s1 - s0;
    t.want = (sl >= sr) ? ((r << 1) | 1u) : (l << 1);  // the later-born symbol is the segment's m; (m, m') counts as m left
  }
  return t;
}

// member q of the plan takes the list `t`; returns its length
__device__ __forceinline__ unsigned long long plan_take(const TrainCtx &C, BatchPlan &P, int q, const TiedPlan &t) {
  P.list[q] = t.kind == 2 ? C.idx_word + t.start : C.K.words + t.start;
  P.tags[q] = t.kind == 2 ? C.idx_tag + t.start : nullptr;
  P.want[q] = t.want;
  return t.kind ? t.n_ent : 0ull;
}

__device__ __forceinline__ unsigned long long plan_member(const TrainCtx &C, BatchPlan &P, int q) {
  return plan_take(C, P, q, plan_lookup(C, P.l[q], P.r[q]));
}

// the index entry of a new pair: the symbol beside the merged one, and the side the merged one is on.  When both symbols were
// made in this step the left one counts as "the merged one" -- the same rule plan_member looks entries up by.
__device__ __forceinline__ void index_entry(const TrainCtx &C, uint64_t at, unsigned long long key, uint32_t first_m, uint32_t K, uint32_t w) {
  if (at >= C.idx_cap) return;
  const uint32_t a = (uint32_t)(key >> 32), b = (uint32_t)key;
  C.idx_tag[at] = (a - first_m < K) ? ((b << 1) | 1u) : (a << 1);
  C.idx_word[at] = w;
}

// walk one claimed word: rewrite in place, park the deltas.  Returns merges done; n_parked / n_new_parked by reference.
// Deltas beyond kEmitCap are applied on the spot (a new pair then takes its index entry with an atomic of its own).
// `stage` holds the word's first kStage stream slots as they were before the walk (one LDS column per lane, loaded in one go:
// the walk never reads a slot again after writing it, so the copy stays good).
// L / R are the plan's pairs in registers.  With one pair its own delta is NOT parked: the caller sends it once per wave
// (every lane of a big merge has it); a batch parks it like any other.
__device__ __forceinline__ uint32_t walk_word(uint32_t *__restrict__ sym, uint64_t b0, uint64_t b1, const uint32_t (&L)[kMaxBatch],
                                              const uint32_t (&R)[kMaxBatch], uint32_t K, uint32_t first_m, long long f, const TrainCtx &C,
                                              uint32_t w, uint32_t *stage, unsigned long long *park, int &n_park, int &n_new) {
  uint32_t n_merged = 0;
  uint32_t po = 0, pn = 0;  // previous old / new symbol
  bool po_cov = false, pn_new = false, have = false;
  uint64_t i = b0;
  // the staged window [sb, sb + kStage): reads only move forward, and everything at or behind a read position is still as it
  // was (the walk writes behind itself), so a word longer than the window refills it there -- one round trip per kStage slots
  uint64_t sb = b0;
  auto sym_at = [&](uint64_t at) -> uint32_t {
    if (at - sb >= (uint64_t)kStage) {
      sb = at;
      uint32_t pre[kStage];
#pragma unroll
      for (int u = 0; u < kStage; u++) pre[u] = sb + u < b1 ? sym[sb + u] : kHole;
#pragma unroll
      for (int u = 0; u < kStage; u++) stage[u * kTrainThreads] = pre[u];
    }
    return stage[(at - sb) * kTrainThreads];
  };
#define SYM(i_) sym_at(i_)
  while (i < b1 && SYM(i) == kHole) i++;
#define EMIT(a_, b_, new_)                                                                   \
  do {                                                                                       \
    const unsigned long long key_ = pair_key((a_), (b_));                                    \
    if (n_park < kEmitCap) {                                                                 \
      park[(n_park++) * kTrainThreads] = key_ | ((new_) ? kEmitNew : 0ull);                  \
      n_new += (new_) ? 1 : 0;                                                               \
    } else {                                                                                 \
      SWT_COUNT(27);                                                                         \
      table_add(C, key_, (new_) ? f : -f);                                                   \
      if (new_) index_entry(C, atomicAdd(&C.st->idx_cursor, 1ull), key_, first_m, K, w);     \
    }                                                                                        \
  } while (0)
  while (i < b1) {
    const uint32_t x = SYM(i);
    uint64_t j = i + 1;
    uint32_t y = kHole;
    while (j < b1 && (y = SYM(j)) == kHole) j++;
    int q = -1;  // the plan's pair here (the members share no symbol: at most one)
    if (j < b1) {
#pragma unroll
      for (int u = 0; u < (int)kMaxBatch; u++) q = (x == L[u] && y == R[u]) ? u : q;
    }
    if (q >= 0) {
      const uint32_t m = first_m + (uint32_t)q;
      if (have) EMIT(po, x, false);    // (prev, l): l is consumed
      if (K > 1) EMIT(x, y, false);    // (l, r) itself
      if (have) EMIT(pn, m, true);     // (prev_new, merged)
      po = y; po_cov = true; pn = m; pn_new = true; have = true;
      sym[i] = m;
      sym[j] = kHole;
      n_merged++;
      i = j + 1;
      while (i < b1 && SYM(i) == kHole) i++;
    } else {
      if (have) {
        if (po_cov) EMIT(po, x, false);  // (r, x): r was consumed
        if (pn_new) EMIT(pn, x, true);   // (merged, x)
      }
      po = x; po_cov = false; pn = x; pn_new = false; have = true;
      i = j;
    }
  }
#undef EMIT
#undef SYM
  return n_merged;
}

// One apply launch over the plan P (in LDS, complete before the call).
__device__ __forceinline__ void apply_body(uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff, const uint32_t *__restrict__ freq,
                                           uint64_t n_words, const TrainCtx &C, const BatchPlan &P) {
  __shared__ unsigned long long park_s[kTrainThreads * kEmitCap];  // [delta][lane]: a wave's lanes sit side by side
  __shared__ uint32_t stage_s[kTrainThreads * kStage];             // [slot][lane]
  __shared__ uint32_t queue_s[kTrainThreads / 64][64 * kEntryFan];  // per wave: the words it found in a trip's entries
  unsigned long long *park = park_s + threadIdx.x;
  uint32_t *stage = stage_s + threadIdx.x;
  uint32_t *queue = queue_s[threadIdx.x >> 6];
  TrainState *st = C.st;
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
#ifdef SWT_STAMPS
  unsigned long long ts[8];
#endif
  SWT_STAMP(ts, 0);
  const uint32_t K = P.K, first_m = P.first_m;
  uint32_t L[kMaxBatch], R[kMaxBatch];
#pragma unroll
  for (int u = 0; u < (int)kMaxBatch; u++) { L[u] = P.l[u]; R[u] = P.r[u]; }
  const bool whole = (st->flags & kFlagIndexBroken) != 0;  // every word (only after a symbol id was reused: never on trained tables)
  const uint64_t n_ent = whole ? n_words : P.ent0[K];
  if (blockIdx.x == 0 && threadIdx.x == 0) st->ent_scanned += n_ent;
  // A log segment lists every pair its step made, and a lane keeps the entries of ITS pairs (by tag): of a segment born early
  // in training that is one in hundreds.  So a lane looks at kEntryFan entries per trip -- all their loads in flight together
  // -- the wave packs the matches into a queue (ballots), and goes through it 64 at a time (sparse matches: one round; a list
  // without tags, where everything matches: kEntryFan rounds, the trips it would have made anyway).
  // (A lane's entries lie a whole grid apart, as the trips did: neighbours in a list stay with neighbouring lanes, so a dense
  // list still spreads over the launch.)
  const uint64_t lanes = (uint64_t)gridDim.x * blockDim.x, stride = lanes * kEntryFan;
  unsigned long long removed = 0, self_delta = 0, inserted = 0;
  uint32_t min_w = 0xFFFFFFFFu;
  // A big merge (the first few hundred of a training run: tens of thousands of words each) is bound by the table: most of its
  // deltas go to a few hot pairs.  There the workgroup first sums its deltas per pair in LDS (the staging area, free once the
  // walks are done) and sends every pair once.
  const bool maybe_big = n_ent >= (uint64_t)kBigMerge && !C.pend;  // (a long list may still hold few words of THESE pairs)
  __shared__ unsigned int blk_rounds, blk_matches;
  unsigned long long *agg_key = reinterpret_cast<unsigned long long *>(stage_s);  // [kAggSlots] keys, then [kAggSlots] sums
  long long *agg_sum = reinterpret_cast<long long *>(stage_s) + kAggSlots;
  const uint32_t tmask = (uint32_t)((1ull << C.T.bits) - 1ull);
  // One batch of table updates: `n_items` (<= kFlushBatch) items, item j = (key_of(j) -- kEmptyKey: none --, delta_of(j)).
  // Three rounds, each with all its memory operations in flight together: probe, add, follow up.  index_at: where the index
  // entries of the new pairs among them go (word w), or nullptr when the caller has written them.
  auto flush_batch = [&](int n_items, auto key_of, auto delta_of, uint64_t *index_at, uint32_t w) {
    unsigned long long key[kFlushBatch], seen[kFlushBatch], seen2[kFlushBatch];
    long long was[kFlushBatch], dl[kFlushBatch];
    uint32_t h[kFlushBatch], ci[kFlushBatch], ci2[kFlushBatch];
    bool on[kFlushBatch];
#pragma unroll
    for (int u = 0; u < kFlushBatch; u++) {
      // the home slot AND the one behind it (linear probing: most keys that are not at home are there)
      key[u] = u < n_items ? key_of(u) : kEmptyKey;
      on[u] = key[u] != kEmptyKey;
      dl[u] = on[u] ? delta_of(u) : 0;
      h[u] = slot_hint(C.T, key[u] & ~kEmitNew);
      const uint32_t h2 = (h[u] + 1) & tmask;
      seen[u] = on[u] ? C.T.keys[h[u]] : 0ull;
      seen2[u] = on[u] ? C.T.keys[h2] : 0ull;
      ci[u] = on[u] && C.cidx ? C.cidx[h[u]] : 0xFFFFFFFFu;
      ci2[u] = on[u] && C.cidx ? C.cidx[h2] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < kFlushBatch; u++) {
      const unsigned long long k = key[u] & ~kEmitNew;
      if (on[u] && seen[u] != k && seen[u] != kEmptyKey && (seen2[u] == k || seen2[u] == kEmptyKey)) {
        h[u] = (h[u] + 1) & tmask;  // home is taken by another key: this pair is, or goes, one slot on
        seen[u] = seen2[u];
        ci[u] = ci2[u];
      }
    }
    // a pair the merge has just made is usually not in the table: its home slot is taken here, all of them at once
    unsigned long long got[kFlushBatch];
#pragma unroll
    for (int u = 0; u < kFlushBatch; u++)
      got[u] = (on[u] && seen[u] == kEmptyKey) ? atomicCAS(&C.T.keys[h[u]], kEmptyKey, key[u] & ~kEmitNew) : 1ull;
#pragma unroll
    for (int u = 0; u < kFlushBatch; u++) {
      if (!on[u]) continue;
      const unsigned long long k = key[u] & ~kEmitNew;
      if (seen[u] == kEmptyKey) {
        if (got[u] == kEmptyKey) { inserted++; seen[u] = k; }
        else seen[u] = got[u];  // another lane was first (with this pair, or with another one)
      }
      if (seen[u] != k) {  // not at its home slot
        h[u] = table_slot(C.T, k, st);
        ci[u] = C.cidx ? C.cidx[h[u]] : 0xFFFFFFFFu;
      }
    }
#pragma unroll
    for (int u = 0; u < kFlushBatch; u++) {
      was[u] = 0;
      if (!on[u]) continue;
      const unsigned long long delta = (unsigned long long)dl[u];
      const uint32_t slot = h[u];
      if (!C.pend) {  // count_add, with the compact copy's index already here
        if (ci[u] < kCidxPending)
          (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&C.ccnt[ci[u]]), delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (dl[u] > 0 && C.theta)  // only the pairs a merge creates ever rise: they may cross theta
          was[u] = (long long)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&C.T.cnt[slot]), delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&C.T.cnt[slot]), delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&C.pend[slot]), delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        was[u] = (long long)atomicMax(&C.tstamp[slot], C.step);
      }
    }
#pragma unroll
    for (int u = 0; u < kFlushBatch; u++) {
      if (!on[u]) continue;
      if (!C.pend) {
        if (dl[u] > 0 && C.theta && was[u] < (long long)C.theta && was[u] + dl[u] >= (long long)C.theta) cand_push(C, h[u]);
      } else if ((uint32_t)was[u] < C.step) {
        const unsigned long long q = atomicAdd(&st->n_touched, 1ull);
        if (q < C.touched_cap) C.touched[q] = h[u];
      }
      if (index_at && (key[u] & kEmitNew)) index_entry(C, (*index_at)++, key[u] & ~kEmitNew, first_m, K, w);
    }
  };
  for (uint64_t eb = (uint64_t)blockIdx.x * blockDim.x; eb < n_ent; eb += stride) {  // the same trips for the whole workgroup
    const uint64_t e0 = eb + (uint64_t)(threadIdx.x - lane);
    int n_q = 0;  // the wave's matches of this trip, packed into its queue in entry order
    {
      uint32_t we[kEntryFan], tg[kEntryFan], wt[kEntryFan];
#pragma unroll
      for (int u = 0; u < kEntryFan; u++) {
        const uint64_t e = e0 + (uint64_t)lane + (uint64_t)u * lanes;
        we[u] = 0xFFFFFFFFu;
        tg[u] = 0;
        wt[u] = 0;
        if (e < n_ent) {
          if (whole) {
            we[u] = (uint32_t)e;
          } else {
            uint32_t q = 0;
            while (q + 1 < K && e >= P.ent0[q + 1]) q++;
            const uint64_t at = e - P.ent0[q];
            we[u] = P.list[q][at];  // both loads go out together
            if (P.tags[q]) { tg[u] = P.tags[q][at]; wt[u] = P.want[q]; }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kEntryFan; u++) {
        const bool match = we[u] != 0xFFFFFFFFu && tg[u] == wt[u];
        const unsigned long long mm = __ballot(match);
        if (match) queue[n_q + __popcll(mm & lt)] = we[u];
        n_q += __popcll(mm);
      }
    }
   int n_rounds = (n_q + 63) >> 6;
   bool big = false;
   if (maybe_big) {  // the workgroup's words of this trip: many -> the summing rounds, which end in workgroup barriers
     if (threadIdx.x == 0) { blk_rounds = 0; blk_matches = 0; }
     __syncthreads();
     if (lane == 0) { atomicMax(&blk_rounds, (unsigned int)n_rounds); atomicAdd(&blk_matches, (unsigned int)n_q); }
     __syncthreads();
     big = blk_matches >= (unsigned int)kBigWords;
     if (big) n_rounds = (int)blk_rounds;
     __syncthreads();  // (the next trip resets the two)
   }
   for (int round = 0; round < n_rounds; round++) {
    const int qi = round * 64 + lane;
    uint32_t w = qi < n_q ? queue[qi] : 0xFFFFFFFFu;
    int n_park = 0, n_new = 0;
    SWT_STAMP(ts, 1);
    if (w != 0xFFFFFFFFu) {
      // the word's bounds and frequency travel beside the claim
      const uint64_t b0 = woff[w], b1 = woff[w + 1];
      const long long f = freq[w];
      const uint32_t claimed = atomicMax(&C.wstamp[w], C.step);
      uint32_t pre[kStage];  // the word's first slots: all loads in flight at once instead of one per step of the walk
#pragma unroll
      for (int u = 0; u < kStage; u++) pre[u] = b0 + u < b1 ? sym[b0 + u] : kHole;
#pragma unroll
      for (int u = 0; u < kStage; u++) stage[u * kTrainThreads] = pre[u];
      SWT_STAMP(ts, 2);
      if (claimed < C.step) {  // one lane per word and step: it merges every pair of the plan there
        const uint32_t nm = walk_word(sym, b0, b1, L, R, K, first_m, f, C, w, stage, park, n_park, n_new);
        if (nm) {
          removed += nm;
          self_delta += (unsigned long long)nm * (unsigned long long)f;
          min_w = w < min_w ? w : min_w;
          if (C.sfreq && L[0] == R[0]) {  // twin pair: nm merges happened in this word (see wp_move_freq)
            const unsigned long long d = (unsigned long long)nm * (unsigned long long)f;
            atomicAdd(reinterpret_cast<unsigned long long *>(&C.sfreq[L[0]]), (unsigned long long)0 - 2 * d);
            atomicAdd(reinterpret_cast<unsigned long long *>(&C.sfreq[first_m]), d);
          }
        }
      } else {
        w = 0xFFFFFFFFu;  // another lane has this word
      }
    }
    SWT_STAMP(ts, 3);
    // room in the index log for the wave's new entries: one atomic per wave (lanes without a word take part with zero)
    uint32_t x = (uint32_t)n_new;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    const uint32_t total = __shfl(x, 63);
    unsigned long long base = 0;
    if (lane == 0 && total) base = atomicAdd(&st->idx_cursor, (unsigned long long)total);
    base = __shfl(base, 0);
    SWT_STAMP(ts, 4);
    if (big) {
      // the index entries of the new pairs, then the deltas into the workgroup's LDS sums
      const long long fw = (w != 0xFFFFFFFFu && n_park) ? (long long)freq[w] : 0;
      uint64_t at = base + x - (uint32_t)n_new;
      for (int j = 0; j < n_park; j++) {
        const unsigned long long pk = park[j * kTrainThreads];
        if (pk & kEmitNew) index_entry(C, at++, pk & ~kEmitNew, first_m, K, w);
      }
      __syncthreads();  // every walk of the round is done: the staging area is free
      for (int i = threadIdx.x; i < kAggSlots; i += blockDim.x) { agg_key[i] = kEmptyKey; agg_sum[i] = 0; }
      __syncthreads();
      int n_left = 0;  // deltas that found no room in the sums: they stay parked (in front) and go out one by one
      for (int j = 0; j < n_park; j++) {
        const unsigned long long pk = park[j * kTrainThreads], k = pk & ~kEmitNew;
        const long long d = (pk & kEmitNew) ? fw : -fw;
        uint32_t hs = (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 40) % (uint32_t)kAggSlots;
        bool put = false;
        for (int probe = 0; probe < 12 && !put; probe++) {
          unsigned long long old = agg_key[hs];
          if (old == kEmptyKey) {
            old = atomicCAS(&agg_key[hs], kEmptyKey, k);
            if (old == kEmptyKey) old = k;
          }
          if (old == k) { atomicAdd(reinterpret_cast<unsigned long long *>(&agg_sum[hs]), (unsigned long long)d); put = true; }
          hs = hs + 1 == (uint32_t)kAggSlots ? 0u : hs + 1;
        }
        if (!put) park[(n_left++) * kTrainThreads] = pk;
      }
      __syncthreads();
      // every pair of the sums once (kAggSlots / 256 slots a lane, one batch)
      flush_batch(kAggSlots / kTrainThreads,
                  [&](int u) -> unsigned long long { const long long sm = agg_sum[threadIdx.x + u * kTrainThreads]; return sm ? agg_key[threadIdx.x + u * kTrainThreads] : kEmptyKey; },
                  [&](int u) -> long long { return agg_sum[threadIdx.x + u * kTrainThreads]; }, nullptr, 0u);
      for (int j0 = 0; j0 < n_left; j0 += kFlushBatch)
        flush_batch(n_left - j0 < kFlushBatch ? n_left - j0 : kFlushBatch,
                    [&](int u) -> unsigned long long { return park[(j0 + u) * kTrainThreads] & ~kEmitNew; },
                    [&](int u) -> long long { return (park[(j0 + u) * kTrainThreads] & kEmitNew) ? fw : -fw; }, nullptr, 0u);
      __syncthreads();  // the next round stages into the area again
    } else if (w != 0xFFFFFFFFu && n_park) {
      SWT_COUNT(28);
      if (n_park > kFlushBatch) SWT_COUNT(26);
      const long long f = freq[w];  // L1: loaded a moment ago
      uint64_t at = base + x - (uint32_t)n_new;
      for (int j0 = 0; j0 < n_park; j0 += kFlushBatch)
        flush_batch(n_park - j0 < kFlushBatch ? n_park - j0 : kFlushBatch,
                    [&](int u) -> unsigned long long { return park[(j0 + u) * kTrainThreads]; },
                    [&](int u) -> long long { return (park[(j0 + u) * kTrainThreads] & kEmitNew) ? f : -f; }, &at, w);
#ifdef SWT_STAMPS
      SWT_STAMP(ts, 5);
      if (e0 < stride && C.step > 512 && atomicAdd(&g_reported[C.step & (kSpanSteps - 1)], 1u) == 0) {  // ts[0] is the kernel's start
        for (int q = 0; q < 5; q++) atomicAdd(&g_phase[1][q], ts[q + 1] - ts[q]);
        atomicAdd(&g_phase[1][8], (unsigned long long)n_park);
        atomicAdd(&g_phase[1][9], (unsigned long long)n_new);
        atomicAdd(&g_phase[1][15], 1ull);
      }
      if (e0 < stride) {
        atomicMax(&g_pmax[0][C.step & (kSpanSteps - 1)], ts[0]);
        for (int q = 0; q < 5; q++) atomicMax(&g_pmax[1 + q][C.step & (kSpanSteps - 1)], ts[q + 1] - ts[q]);
      }
      if (C.step <= 64) {
        for (int q = 0; q < 5; q++) atomicAdd(&g_phase[2][q], ts[q + 1] - ts[q]);
        atomicAdd(&g_phase[2][8], (unsigned long long)n_park);
        atomicAdd(&g_phase[2][9], (unsigned long long)n_new);
        atomicAdd(&g_phase[2][15], 1ull);
      }
#endif
    }
   }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    removed += __shfl_xor(removed, d);
    self_delta += __shfl_xor(self_delta, d);
    inserted += __shfl_xor(inserted, d);
    const uint32_t o = __shfl_xor(min_w, d);
    min_w = o < min_w ? o : min_w;
  }
  if (lane == 0 && inserted) atomicAdd(&st->n_used, inserted);
  if (lane == 0 && removed) {
    if (K == 1) table_add(C, pair_key(L[0], R[0]), -(long long)self_delta);  // the pair's own count: once per wave, not once per word
    atomicAdd(&st->n_syms, (unsigned long long)0 - removed);
    atomicMin(&st->cursor_w, min_w);  // new pairs were born in these words: the tie scan may not start after them
  }
}

// The tied pairs' first positions through the index: the waves of the launch share the list of live pairs; a wave that meets
// a pair whose score is `mx` goes through that pair's words.  Entries below n_mirrored are read from the mirror, the others
// from the table (wp_step_kernel calls this before the step's new candidates are mirrored).
__device__ __forceinline__ void wp_tie_by_index(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff, uint64_t n_words,
                                                const TrainCtx &C, unsigned long long mx, unsigned long long n, unsigned long long n_mirrored) {
  TrainState *st = C.st;
  const int lane = threadIdx.x & 63;
  const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  const uint64_t gw = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  for (uint64_t base = gw * 64; base < n; base += n_waves * 64) {
    const uint64_t i = base + lane;
    unsigned long long key = kEmptyKey;
    bool hit = false;
    if (i < n) {
      long long v;
      if (i < n_mirrored) { v = C.ccnt[i]; key = C.ckey[i]; }
      else { const uint32_t slot = C.cand[i]; v = C.T.cnt[slot]; key = C.T.keys[slot]; }
      hit = v > 0 && key != kEmptyKey && pair_value(key, v, C.sfreq) == mx;
    }
    unsigned long long H = __ballot(hit);
    while (H) {  // one tied pair at a time, the whole wave on its word list
      const int src = __builtin_ctzll(H);
      H &= H - 1ull;
      const unsigned long long k = __shfl(key, src);
      const uint32_t l = (uint32_t)(k >> 32), r = (uint32_t)k;
      TiedPlan t = plan_lookup(C, l, r);
      if (!t.kind) continue;
      if (t.kind == 2) {
        // This launch runs BEFORE the step's decide kernel opens the step's segment: seg_start[] is written up to last_open
        // only, and the newest segment ends where the log stands (no apply is in flight).  (Reading seg_start[seg + 1] here
        // took a stale 0 for the end of the previous step's segment -- a list "length" of 2^64 - start, and a GPU memory
        // fault on its first GPU run: gpurun_out/r03b_pytest.log.)
        const uint32_t sl = seg_of_symbol(C, l), sr = seg_of_symbol(C, r);
        const uint32_t seg = sl > sr ? sl : sr;
        const unsigned long long end = seg >= st->last_open ? st->idx_cursor : C.seg_start[seg + 1];
        t.n_ent = end > t.start ? end - t.start : 0ull;
        if (t.start + t.n_ent > C.idx_cap) t.n_ent = t.start < C.idx_cap ? C.idx_cap - t.start : 0ull;  // (entries past the log were dropped)
      }
      const uint32_t *list = t.kind == 2 ? C.idx_word + t.start : C.K.words + t.start;
      const uint32_t *tags = t.kind == 2 ? C.idx_tag + t.start : nullptr;
      for (uint64_t e = lane; e < t.n_ent; e += 64) {
        if (tags && tags[e] != t.want) continue;
        const uint64_t w = list[e];
        if (w >= n_words) continue;
        if ((w << 32) >= __hip_atomic_load(&st->best_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;  // a later word cannot win
        const uint64_t b0 = woff[w], b1 = woff[w + 1];
        uint32_t a = kHole;
        uint64_t ai = 0;
        for (uint64_t j = b0; j < b1; j++) {
          const uint32_t b = sym[j];
          if (b == kHole) continue;
          if (a == l && b == r) {
            atomicMin(&st->best_pos, (unsigned long long)((w << 32) | (ai - b0)));
            break;
          }
          a = b;
          ai = j;
        }
      }
    }
  }
}

// ---- wordpiece.py:92 tie-break through the inverted index -----------------------------------------------------------------
// A WordPiece score moves whenever a symbol frequency does, so the plateau cursor of the BPE scan does not hold and
// tie_kernel had to read the stream from word 0 at every tied step: ~0.5 MB and ~25 us per merge on S85k-lex, most of the
// step.  But the tied pairs are few, and the index knows the words of each: the waves go through the list of live pairs once
// more (its mirror is complete: wp_list_argmax_kernel ran), and a wave that meets a pair holding the maximum looks its words
// up (k0 list or log segment, as an apply would), finds the pair's first position in each and keeps the minimum -- the same
// best_pos the stream scan leaves.  An index that was abandoned (kFlagIndexBroken) falls back to that scan, in this kernel.
__global__ __launch_bounds__(kTrainThreads) void wp_tie_index_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                                     uint64_t n_words, TrainCtx C, const ArgPart *__restrict__ parts,
                                                                     uint32_t n_parts) {
  __shared__ unsigned long long s_mx, s_tied;
  if (threadIdx.x < 64) {
    unsigned long long m, c, k;
    arg_collect(parts, n_parts, m, c, k);
    if (threadIdx.x == 0) { s_mx = m; s_tied = c; }
  }
  __syncthreads();
  if (s_tied < 2 || s_mx == 0) return;
  if (cand_dry(C, s_mx)) return;  // decide_kernel reports it
  const unsigned long long mx = s_mx;
  TrainState *st = C.st;
  if (st->flags & kFlagIndexBroken) {  // the stream scan of tie_kernel
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
      if ((w << 32) >= __hip_atomic_load(&st->best_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      const uint64_t b0 = woff[w], b1 = woff[w + 1];
      uint32_t a = kHole;
      uint64_t ai = 0;
      for (uint64_t i = b0; i < b1; i++) {
        const uint32_t b = sym[i];
        if (b == kHole) continue;
        if (a != kHole) {
          const unsigned long long key = pair_key(a, b);
          if (pair_value(key, table_get(C.T, key), C.sfreq) == mx) {
            atomicMin(&st->best_pos, (unsigned long long)((w << 32) | (ai - b0)));
            break;
          }
        }
        a = b;
        ai = i;
      }
    }
    return;
  }
  unsigned long long n = st->n_cand;
  if (n > C.cand_cap) n = C.cand_cap;
  wp_tie_by_index(sym, woff, n_words, C, mx, n, n);
}

// a plan of one pair (every path but the fast one)
__device__ __forceinline__ void plan_single(const TrainCtx &C, BatchPlan &P, uint32_t l, uint32_t r, uint32_t m) {
  if (threadIdx.x == 0) {
    P.K = 1;
    P.first_m = m;
    for (int u = 0; u < (int)kMaxBatch; u++) { P.l[u] = kHole; P.r[u] = kHole; }
    P.l[0] = l;
    P.r[0] = r;
    P.ent0[0] = 0;
    P.ent0[1] = (C.st->flags & kFlagIndexBroken) ? 0ull : plan_member(C, P, 0);
  }
  __syncthreads();
}

__global__ __launch_bounds__(kTrainThreads) void apply_kernel(uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                              const uint32_t *__restrict__ freq, uint64_t n_words, TrainCtx C,
                                                              const StepCmd *__restrict__ cmd) {
  __shared__ BatchPlan P;
  if (!cmd->valid) return;
  plan_single(C, P, cmd->l, cmd->r, cmd->m);
  apply_body(sym, woff, freq, n_words, C, P);
}

// ---- the fast path of unsharded BPE: two launches per STEP, one or more merges per step ---------------------------------
// The candidate list is short (kCandTarget at a re-plan), so EVERY workgroup of the tie launch finds the maximum over it by
// itself -- no argmax launch, no partials, no decide launch:
//   fast_tie_kernel    workgroup argmax (workgroup 0 publishes it); when the maximum is tied, the tied pairs go into an LDS
//                      set and the words are scanned from the plateau cursor: every tied pair gets the earliest position seen
//                      of it (gpos[], by its place in the candidate list; the earliest of all in best2[step parity])
//   fast_apply_kernel  reads the published maximum and the positions -- it may not look at the counts itself, its own
//                      workgroups are already moving them -- and decides what the step merges; then apply_body; workgroup 0
//                      also writes the log lines, the plateau cursor, the merged symbols' birth step, and resets the other best2
//
// Several merges in one step.  On a plateau (pairs tied at the maximum c) the reference (bpe.py:102) takes the tied pairs one
// at a time, each time the one whose first occurrence comes first.  Let P1 < P2 < ... be the tied pairs in the order of their
// first positions.  After merging P1 the reference picks P2 next PROVIDED
//   (1) P2 still counts c: a merge only lowers the counts of pairs that share a symbol with it, so it suffices that P2 shares
//       no symbol with P1 (then all of P2's occurrences, and so its first position and its rank among the others, stand), and
//   (2) no pair the merge of P1 = (a, b) creates reaches c.  A new pair (x, ab) occurs at most as often as (x, a) did, so it
//       reaches c only if (x, a) is itself tied at c; likewise (ab, y) needs (b, y) tied, and (ab, ab) needs (b, a) tied.
//       So: P1 is "dangerous" iff some tied pair has a on its right or b on its left -- decided from the tied set alone.
// By induction a step may merge the longest prefix P1..PK whose members share no symbol, ending at (and including) the first
// dangerous one -- what the reference does in K iterations -- as long as the positions are those of a region EVERY workgroup
// scanned (the scan's first trip: win_end); beyond it only the overall minimum is certain, and the step merges that alone.
struct BlockArg {
  unsigned long long mx, tied, key;
};

#ifndef SWT_TIE_WORDS
#define SWT_TIE_WORDS 16
#endif
constexpr int kTieWords = SWT_TIE_WORDS;  // words a wave scans per trip of the tie scan: 16 lanes per word, 4 words at a time
constexpr int kTieStage = 8;              // coalesced loads per lane that stage those words (512 stream slots; beyond: the stream)
constexpr int kTieSetSlots = 1024;        // LDS set of the tied pairs: kTieSet keys at most, load factor 1/4
constexpr int kCandRegs = 8;              // candidates a lane holds in registers: kTrainThreads * kCandRegs = kCandHigh

// the tied pairs of a step, in LDS
struct TieSets {
  unsigned long long key[kTieSetSlots];  // the pairs (open addressing)
  uint32_t idx[kTieSetSlots];            // its place in the candidate list
  uint32_t lefts[kTieSetSlots];          // the symbols some tied pair has on its left ...
  uint32_t rights[kTieSetSlots];         // ... and on its right (for "dangerous", see above)
};

// 32-bit mixing: a 64-bit multiply is a dozen quarter-rate instructions, and the scan hashes every live symbol it passes
__device__ __forceinline__ uint32_t tset_hash(unsigned long long key) {
  return (((uint32_t)(key >> 32) * 0x9E3779B1u) ^ ((uint32_t)key * 0x85EBCA6Bu)) >> 22;
}

__device__ __forceinline__ void symset_insert(uint32_t *set, uint32_t s) {
  uint32_t h = (s * 0x9E3779B1u) >> 22;
  for (;;) {
    const uint32_t old = atomicCAS(&set[h], kHole, s);
    if (old == kHole || old == s) break;
    h = (h + 1) & (kTieSetSlots - 1);
  }
}

__device__ __forceinline__ bool symset_has(const uint32_t *set, uint32_t s) {
  uint32_t h = (s * 0x9E3779B1u) >> 22;
  for (;;) {
    const uint32_t k = set[h];
    if (k == s) return true;
    if (k == kHole) return false;
    h = (h + 1) & (kTieSetSlots - 1);
  }
}

// (the symbol sets are the planner's: `with_syms`)
__device__ __forceinline__ void tset_clear(TieSets &S, bool with_syms) {
  for (int i = threadIdx.x; i < kTieSetSlots; i += blockDim.x) {
    S.key[i] = kEmptyKey;
    if (with_syms) {
      S.lefts[i] = kHole;
      S.rights[i] = kHole;
    }
  }
}

__device__ __forceinline__ void tset_insert(TieSets &S, unsigned long long key, uint32_t cand_i, bool with_syms) {
  uint32_t h = tset_hash(key);
  for (;;) {  // a slot may be listed twice: the set takes a key once
    const unsigned long long old = atomicCAS(&S.key[h], kEmptyKey, key);
    if (old == kEmptyKey) {
      S.idx[h] = cand_i;
      if (with_syms) {
        symset_insert(S.lefts, (uint32_t)(key >> 32));
        symset_insert(S.rights, (uint32_t)key);
      }
      break;
    }
    if (old == key) break;
    h = (h + 1) & (kTieSetSlots - 1);
  }
}

// the slot of `key`, -1 when it is not a tied pair
__device__ __forceinline__ int tset_find(const TieSets &S, unsigned long long key) {
  uint32_t h = tset_hash(key);
  for (;;) {
    const unsigned long long k = S.key[h];
    if (k == key) return (int)h;
    if (k == kEmptyKey) return -1;
    h = (h + 1) & (kTieSetSlots - 1);
  }
}

__device__ __forceinline__ BlockArg block_reduce(unsigned long long m, unsigned long long c, unsigned long long k) {
  __shared__ unsigned long long sm[4], sc[4], sk[4];
  arg_wave_reduce(m, c, k);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sm[wave] = m; sc[wave] = c; sk[wave] = k; }
  __syncthreads();
  m = sm[0]; c = sc[0]; k = sk[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); w++) arg_combine(m, c, k, sm[w], sc[w], sk[w]);
  __syncthreads();
  return BlockArg{m, c, k};
}

// The workgroup's maximum over the candidate list, and (when the maximum is tied among at most kTieSet pairs) the tied pairs
// in the LDS sets.  The latency of a dependent access to the pair table is what this costs, so a lane takes its candidates as
// rounds of independent loads instead of a chain per candidate.
// `spec` = cand[threadIdx.x + u * kTrainThreads], requested by the caller together with the state (before n_cand was known:
// entries past n_cand are stale and unused), which takes the gather of the unmirrored tail from three round trips to two.
// The sets are empty when this is called (tset_clear, a barrier ago); `with_syms`: the symbol sets are wanted too.
__device__ __forceinline__ BlockArg block_argmax(const TrainCtx &C, unsigned long long n_cand, unsigned long long n_synced, TieSets &S,
                                                 const uint32_t (&spec)[kCandRegs], bool with_syms) {
  if (n_cand <= (unsigned long long)kTrainThreads * kCandRegs) {
    long long v[kCandRegs];
    unsigned long long key[kCandRegs];
    uint32_t slot[kCandRegs];
    // the mirrored part of the list is two coalesced streams, counts and keys, requested together; the few candidates
    // pushed since the last step are gathered from the table in a second round
    long long lm = 0;
#pragma unroll
    for (int u = 0; u < kCandRegs; u++) {
      const uint64_t i = threadIdx.x + (uint64_t)u * kTrainThreads;
      v[u] = 0;
      key[u] = kEmptyKey;
      slot[u] = 0xFFFFFFFFu;
      if (i < n_synced) { v[u] = C.ccnt[i]; key[u] = C.ckey[i]; }
      else if (i < n_cand) slot[u] = spec[u];
    }
#pragma unroll
    for (int u = 0; u < kCandRegs; u++) {
      if (slot[u] != 0xFFFFFFFFu) { v[u] = C.T.cnt[slot[u]]; key[u] = C.T.keys[slot[u]]; }
      lm = v[u] > lm ? v[u] : lm;
    }
    unsigned long long m = 0, c = 0, k = kEmptyKey;
#pragma unroll
    for (int u = 0; u < kCandRegs; u++) {
      if (!(lm > 0 && v[u] == lm)) key[u] = kEmptyKey;
      if (key[u] != kEmptyKey) arg_combine(m, c, k, (unsigned long long)lm, 1ull, key[u]);
    }
    const BlockArg a = block_reduce(m, c, k);
    if (a.tied >= 2 && a.tied <= kTieSet && a.mx) {
      if ((unsigned long long)lm == a.mx) {
#pragma unroll
        for (int u = 0; u < kCandRegs; u++)
          if (key[u] != kEmptyKey) tset_insert(S, key[u], (uint32_t)(threadIdx.x + u * kTrainThreads), with_syms);
      }
      __syncthreads();
    }
    return a;
  }
  // a list that pushes have grown past the register budget (the host re-plans it at the next batch boundary)
  unsigned long long m = 0, c = 0, k = kEmptyKey;
  for (uint64_t i = threadIdx.x; i < n_cand; i += blockDim.x) {
    const uint32_t slot = C.cand[i];
    const long long v = C.T.cnt[slot];
    if (v > 0 && (unsigned long long)v >= m) arg_combine(m, c, k, (unsigned long long)v, 1ull, C.T.keys[slot]);
  }
  const BlockArg a = block_reduce(m, c, k);
  if (a.tied >= 2 && a.tied <= kTieSet && a.mx) {
    for (uint64_t i = threadIdx.x; i < n_cand; i += blockDim.x) {
      const uint32_t slot = C.cand[i];
      if ((unsigned long long)C.T.cnt[slot] == a.mx) tset_insert(S, C.T.keys[slot], (uint32_t)i, with_syms);
    }
    __syncthreads();
  }
  return a;
}

// `limit`: merges this host round trip may log (st->run_done counts them); a step past it is a no-op
__global__ __launch_bounds__(kTrainThreads) void fast_tie_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                                 uint64_t n_words, TrainCtx C, uint32_t limit) {
  __shared__ TieSets S;
  TrainState *st = C.st;
  SWT_SPAN(0, C.step);
#ifdef SWT_STAMPS
  unsigned long long ts[8];
#endif
  SWT_STAMP(ts, 0);
  const unsigned par = C.step & 1u;
  // what this launch itself may change (workgroup 0 raises flags / halt) is read once per workgroup: its lanes must agree
  __shared__ unsigned long long hdr[3];
  if (threadIdx.x == 0) { hdr[0] = st->flags; hdr[1] = st->run_done[par]; hdr[2] = st->halt; }
  const unsigned long long n_cand = st->n_cand, n_synced = st->n_synced, plateau = st->plateau, idx_cursor = st->idx_cursor;
  const unsigned long long n_old = st->n_list[par ^ 1u], n_syms_now = st->n_syms;
  tset_clear(S, blockIdx.x == gridDim.x - 1);  // while the state travels (the barrier below covers it)
  const uint64_t cursor_w = st->cursor_w;
  uint32_t cand_spec[kCandRegs];  // see block_argmax
#pragma unroll
  for (int u = 0; u < kCandRegs; u++) {
    const uint64_t i = threadIdx.x + (uint64_t)u * kTrainThreads;
    cand_spec[u] = i < C.cand_cap ? C.cand[i] : 0xFFFFFFFFu;
  }
  __syncthreads();
  const unsigned int flags = (unsigned int)hdr[0];
  // The LAST workgroup scans no words: it is the step's planner -- it publishes the maximum, mirrors the new candidates, and
  // lists the tied pairs with where their words are (what fast_apply_kernel would otherwise look up, two round trips, after
  // it has chosen) -- all of it beside the scan, not in front of it.
  // The one before it is the housekeeper: the candidates the last merge pushed get their compact copy, the other half of
  // gpos[] is wiped for the next step.
  const bool planner = blockIdx.x == gridDim.x - 1, keeper = blockIdx.x == gridDim.x - 2;
  const bool lead = planner && threadIdx.x == 0;
  const unsigned int n_scan = gridDim.x - 2;  // workgroups that scan
  // the step's index segment begins where the log stands (no apply is in flight).  A step that does nothing says so too: its
  // number is taken, and the segment before it ends where this one begins
  if (lead) C.seg_start[C.step] = idx_cursor;
  if ((flags & kFlagReplan) || hdr[2] || hdr[1] >= limit) return;  // nothing runs until the host has looked (fast_apply_kernel: too)
  if (C.halt_ext && *C.halt_ext) return;  // sharded: an exchange block overflowed (the same on every rank)
  SWT_STAMP(ts, 1);
  if (lead) {
    // a merged id that was reused in the last step voids the index from this step on
    if (idx_cursor > C.idx_cap || (flags & kFlagBrokenPending)) atomicOr(&st->flags, kFlagIndexBroken);
    st->n_list[par] = 0;
    st->step_syms = n_syms_now;
  }
  if (n_cand > C.cand_cap) {  // the list lost a candidate
    if (lead) { atomicOr(&st->flags, kFlagReplan); st->halt = 3; }
    return;
  }
  // The tie scan: trips of gridDim.x * 4 * kTieWords words from the plateau cursor.  What bounds this launch is the instruction
  // stream of its slowest wave, so a word gets SIXTEEN lanes, one stream slot each (a lane per word walked its slots one after
  // the other, holes included: 6 us of the 14), and a wave takes only kTieWords consecutive words: one contiguous piece of the
  // stream, fetched with coalesced loads that are all in flight together and parked in LDS.  The first trip is requested NOW
  // -- from the cursor: the common case late in training -- so that it travels while the argmax waits for the candidate list.
  __shared__ uint32_t wbuf_s[kTrainThreads / 64][kTieStage * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *wbuf = wbuf_s[wave];
  const uint64_t trip_words = (uint64_t)n_scan * (kTrainThreads / 64) * kTieWords;
  const uint64_t wave_off = ((uint64_t)blockIdx.x * (kTrainThreads / 64) + wave) * kTieWords;
  uint64_t wo = 0, s0 = 0, s1 = 0;  // wo: lanes 0..kTieWords hold the bounds of the wave's words
  uint32_t pre[kTieStage];
#define SWT_TIE_FETCH(w_)                                                         \
  do {                                                                            \
    const uint64_t wl_ = (w_) + (uint64_t)lane;                                   \
    wo = woff[wl_ < n_words ? wl_ : n_words];  /* lanes past kTieWords: unused */  \
    s0 = __shfl(wo, 0);                                                           \
    s1 = __shfl(wo, kTieWords);                                                   \
    _Pragma("unroll") for (int k_ = 0; k_ < kTieStage; k_++) {                    \
      const uint64_t at_ = s0 + (uint64_t)(k_ * 64 + lane);                       \
      pre[k_] = at_ < s1 ? sym[at_] : kHole;                                      \
    }                                                                             \
  } while (0)
  if (!planner && !keeper) SWT_TIE_FETCH(cursor_w + wave_off);
  if (keeper) {
    // the candidates the last merge pushed get their compact copy now (no count moves while this launch runs); every
    // workgroup of THIS launch still gathers them from the table, fast_apply_kernel then moves n_synced up
    for (uint64_t i = n_synced + threadIdx.x; i < n_cand; i += blockDim.x) {
      const uint32_t slot = C.cand[i];
      C.ccnt[i] = C.T.cnt[slot];
      C.ckey[i] = C.T.keys[slot];
      C.cidx[slot] = (uint32_t)i;
    }
    if (threadIdx.x == 0) st->n_synced_next = n_cand;
    // the positions the step before the last left in the other half of gpos[] have been read: that half is the next step's
    for (uint64_t i = threadIdx.x; i < n_old; i += blockDim.x) {
      const size_t ci = (size_t)(par ^ 1u) * C.cand_cap + (C.tied_idx[(par ^ 1u) * kTieSet + i] & 0x7FFFFFFFu);
      C.gpos[ci] = kEmptyKey;
      C.gnb_min[2 * ci] = 0xFFFFFFFFu;
      C.gnb_min[2 * ci + 1] = 0xFFFFFFFFu;
      C.gnb_max[2 * ci] = 0u;
      C.gnb_max[2 * ci + 1] = 0u;
    }
    return;
  }
  SWT_STAMP(ts, 2);
  const BlockArg a = block_argmax(C, n_cand, n_synced, S, cand_spec, planner);
  SWT_STAMP(ts, 3);
  const bool dry = a.mx < C.theta && C.theta > 1;
  const unsigned long long mx = a.mx;
  const bool use_set = a.tied <= kTieSet;
  const uint64_t start = plateau == mx ? cursor_w : 0ull;
  if (lead) {
    st->max_count = a.mx;
    st->n_tied = a.mx ? a.tied : 0;
    st->best_key = a.key;
    const uint64_t we = start + trip_words;
    st->win_end = (we < n_words ? we : n_words) << 32;
    if (!dry && a.tied >= 2 && a.mx) st->tie_words += (we < n_words ? we : n_words) - start;
    if (dry) { atomicOr(&st->flags, kFlagReplan); st->halt = 3; }
  }
#ifdef SWT_STAMPS
  const bool stamper = blockIdx.x == 0 && threadIdx.x == 0;  // a scanning workgroup's view
  if (stamper && (dry || a.tied < 2 || a.mx == 0)) {  // no tie: the launch ends here
    for (int q = 0; q < 3; q++) atomicAdd(&g_phase[0][5 + q], ts[q + 1] - ts[q]);
    atomicAdd(&g_phase[0][14], 1ull);
  }
  int n_trips = 0;
#endif
  if (planner) {
    if (!dry && a.mx && a.tied < 2) {
      // one pair holds the maximum: its list, ready for fast_apply_kernel
      if (threadIdx.x == 0) C.tied_plan[par * kTieSet] = plan_lookup(C, (uint32_t)(a.key >> 32), (uint32_t)a.key);
    } else if (!dry && a.mx && use_set) {
      // the tied pairs as a list: place in the candidate list | dangerous << 31, the key, where its words are
      __shared__ unsigned int n_listed;
      if (threadIdx.x == 0) n_listed = 0;
      __syncthreads();
      for (int i = threadIdx.x; i < kTieSetSlots; i += blockDim.x) {
        const unsigned long long key = S.key[i];
        if (key == kEmptyKey) continue;
        const uint32_t qa = (uint32_t)(key >> 32), qb = (uint32_t)key;
        const bool danger = symset_has(S.rights, qa) || symset_has(S.lefts, qb);  // a twin pair is its own witness
        const unsigned int k = atomicAdd(&n_listed, 1u);
        C.tied_idx[par * kTieSet + k] = S.idx[i] | (danger ? 0x80000000u : 0u);
        C.tied_key[par * kTieSet + k] = key;
        C.tied_plan[par * kTieSet + k] = plan_lookup(C, qa, qb);
      }
      __syncthreads();
      if (threadIdx.x == 0) st->n_list[par] = n_listed;
    }
    return;
  }
  if (dry || a.tied < 2 || a.mx == 0) return;
  unsigned long long *best = &st->best2[par];
  __shared__ unsigned long long blk_seen;
  __shared__ unsigned int blk_hit;
  if (threadIdx.x == 0) blk_hit = 0;
  const int grp = lane >> 4, j = lane & 15;
  for (uint64_t t0 = start; t0 < n_words; t0 += trip_words) {  // the same trips for every lane of the workgroup
    const uint64_t w_wave = t0 + wave_off;
    if (t0 != start) {
      // words only get later: once a hit before this trip's words is in, the workgroup is done
      if (threadIdx.x == 0) blk_seen = __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (((t0 + (uint64_t)blockIdx.x * (kTrainThreads / 64) * kTieWords) << 32) >= blk_seen) break;
    }
#ifdef SWT_STAMPS
    if (n_trips == 0) { SWT_STAMP(ts, 4); ts[7] = ts[4]; }
    n_trips++;
#endif
    if (t0 != start || start != cursor_w) SWT_TIE_FETCH(w_wave);  // not the trip that is here already
#pragma unroll
    for (int k = 0; k < kTieStage; k++) wbuf[k * 64 + lane] = pre[k];
#ifdef SWT_STAMPS
    if (n_trips == 1) SWT_STAMP(ts, 7);
#endif
    __syncthreads();
    unsigned long long mine = kEmptyKey;  // this lane's earliest hit of the trip
    for (int r = 0; r < kTieWords / 4; r++) {  // four words at a time, in word order
      const int wi = r * 4 + grp;
      const uint64_t w = w_wave + (uint64_t)wi;
      const uint64_t b0 = __shfl(wo, wi), b1 = __shfl(wo, wi + 1);
      const uint32_t o0 = (uint32_t)(b0 - s0), len = w < n_words ? (uint32_t)(b1 - b0) : 0u;
      uint32_t carry = kHole, carry_i = 0;  // the word's last live symbol before this round of sixteen slots
      uint32_t carry2 = kHole;              // ... and the live symbol before that one
      bool done = false;                    // table-probe mode: this word has its hit
      for (uint32_t base = 0; __any(base < len && !done); base += 16) {
        const uint32_t p = base + (uint32_t)j;
        uint32_t y = kHole;
        if (p < len && !done) y = o0 + p < (uint32_t)(kTieStage * 64) ? wbuf[o0 + p] : sym[b0 + p];
        const unsigned long long live = __ballot(y != kHole);
        const uint32_t gm = (uint32_t)(live >> (grp * 16)) & 0xFFFFu;
        const uint32_t below = gm & ((1u << j) - 1u);
        // the live symbol before mine: a lane of my group, or the carry
        const int src = below ? 31 - __builtin_clz(below) : j;
        const uint32_t from = __shfl(y, grp * 16 + src);
        const uint32_t px = below ? from : carry;
        const uint32_t pxi = below ? base + (uint32_t)src : carry_i;
        // the neighbours of the pair (px, y): the live symbol before px, and the one after y when this round shows it
        const uint32_t px_of_src = __shfl(px, grp * 16 + src);
        const uint32_t nb_l = below ? px_of_src : carry2;
        const uint32_t above = gm & ~((2u << j) - 1u);
        const int nxt = above ? __builtin_ctz(above) : j;
        const uint32_t y_of_nxt = __shfl(y, grp * 16 + nxt);
        const bool r_known = above != 0 || base + 16 >= len;  // else the word goes on behind this round
        const uint32_t nb_r = above ? y_of_nxt : kHole;
        bool hit = false;  // table-probe mode only
        unsigned long long key = kEmptyKey;
        const unsigned long long at = (unsigned long long)((w << 32) | pxi);
        if (y != kHole && px != kHole) {
          key = pair_key(px, y);
          if (use_set) {
            // every tied pair keeps the earliest position seen of it (here: by this workgroup)
            const uint32_t h = tset_hash(key);
            const unsigned long long got = S.key[h];
            const int slot = got == key ? (int)h : (got == kEmptyKey ? -1 : tset_find(S, key));  // a collision on the first probe: walk on
            if (slot >= 0) {
              const size_t ci = (size_t)par * C.cand_cap + S.idx[slot];
              atomicMin(&C.gpos[ci], at);  // one address per tied pair: no answer is waited for
              mine = at < mine ? at : mine;
              // What clears a "dangerous" pair (a, b): a new pair (x, ab) reaches the maximum only if EVERY occurrence has x
              // before it.  An occurrence that starts its word, or whose left neighbour x makes (x, a) a pair that is not
              // tied, says no for every x (both extremes go in); otherwise x itself goes in, and two different ones say no
              // as well (min < max).  The same on the right.
              const bool l_free = nb_l == kHole || tset_find(S, pair_key(nb_l, px)) < 0;
              atomicMin(&C.gnb_min[2 * ci], l_free ? 0u : nb_l);
              atomicMax(&C.gnb_max[2 * ci], l_free ? 0xFFFFFFFFu : nb_l);
              if (r_known) {
                const bool r_free = nb_r == kHole || tset_find(S, pair_key(y, nb_r)) < 0;
                atomicMin(&C.gnb_min[2 * ci + 1], r_free ? 0u : nb_r);
                atomicMax(&C.gnb_max[2 * ci + 1], r_free ? 0xFFFFFFFFu : nb_r);
              }
            }
          } else {
            hit = (unsigned long long)table_get(C.T, key) == mx;  // a plateau wider than the set: membership by table probe
          }
        }
        if (!use_set) {
          // ... and only the word's first hit, its pair left in wkey[word]: such a step merges one pair
          const uint32_t hm = (uint32_t)(__ballot(hit) >> (grp * 16)) & 0xFFFFu;
          if (hm) {
            if (hit && (hm & ((1u << j) - 1u)) == 0) {
              C.wkey[w] = key;
              mine = at < mine ? at : mine;
            }
            done = true;
          }
        }
        // the group's last live symbol of this round is the next round's carry
        const int top = gm ? 31 - __builtin_clz(gm) : j;
        const uint32_t last = __shfl(y, grp * 16 + top), before_last = __shfl(px, grp * 16 + top);
        if (gm) { carry = last; carry_i = base + (uint32_t)top; carry2 = before_last; }
      }
      if (!use_set && __any(done)) break;  // the wave's later words are later
    }
    // the wave's earliest hit goes out as ONE atomic (same-address device atomics from hundreds of lanes cost microseconds:
    // tools/micro/atomic_probe.hip)
    if (__any(mine != kEmptyKey)) {
      for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(mine, d);
        mine = o < mine ? o : mine;
      }
      if (lane == 0) { atomicMin(best, mine); blk_hit = 1; }
    }
#ifdef SWT_STAMPS
    if (n_trips == 1) SWT_STAMP(ts, 5);
#endif
    __syncthreads();
    if (blk_hit) break;  // later trips only hold later words
  }
#undef SWT_TIE_FETCH
#ifdef SWT_STAMPS
  SWT_STAMP(ts, 6);
  if (stamper) {  // phases: state loads, mirror + prefetch, argmax (+ set), -, first word scanned, the rest of the trips
    if (n_trips == 0) { ts[4] = ts[3]; ts[5] = ts[3]; ts[7] = ts[3]; }
    for (int q = 0; q < 3; q++) atomicAdd(&g_phase[0][q], ts[q + 1] - ts[q]);
    atomicAdd(&g_phase[0][3], ts[5] - ts[3]);
    atomicAdd(&g_phase[0][4], ts[6] - ts[5]);
    atomicAdd(&g_phase[0][8], ts[4] - ts[3]);   // argmax done -> scan begins (the lead's stores)
    atomicAdd(&g_phase[0][9], ts[7] - ts[4]);   // the first word's symbols
    atomicAdd(&g_phase[0][10], ts[5] - ts[7]);  // its probes, the hit
    atomicAdd(&g_phase[0][13], (unsigned long long)n_trips);
    atomicAdd(&g_phase[0][15], 1ull);
  }
#endif
}

// `first_merged`: the symbol id of the round trip's first merge; `limit`: merges the round trip may log
__global__ __launch_bounds__(kTrainThreads) void fast_apply_kernel(uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                                   const uint32_t *__restrict__ freq, uint64_t n_words, TrainCtx C,
                                                                   StepLog *__restrict__ log, uint32_t first_merged, uint32_t limit) {
  __shared__ BatchPlan P;
  __shared__ unsigned long long f_pos[kTieSet], f_key[kTieSet], p1_pos;
  __shared__ uint32_t f_dng[kTieSet], ord[kMaxBatch];
  __shared__ unsigned int n_found;
  TrainState *st = C.st;
  SWT_SPAN(1, C.step);
  const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
  const unsigned par = C.step & 1u;
  // what this launch itself may change (workgroup 0 raises flags / halt) is read once per workgroup: its lanes must agree
  __shared__ unsigned long long hdr[3];
  if (threadIdx.x == 0) { hdr[0] = st->flags; hdr[1] = st->run_done[par]; hdr[2] = st->halt; }
  const unsigned long long mx = st->max_count, tied = st->n_tied, n_list = st->n_list[par], win_end = st->win_end;
  __syncthreads();
  const unsigned int flags = (unsigned int)hdr[0];
  const unsigned long long run_done = hdr[1];
  if ((flags & kFlagReplan) || hdr[2] || run_done >= limit) {  // nothing runs until the host has looked
    if (lead) st->run_done[par ^ 1u] = run_done;
    return;
  }
  // ---- what this step merges (every workgroup decides for itself, from what the tie launch left) ----
  if (threadIdx.x == 0) {
    n_found = 0;
    p1_pos = kEmptyKey;
    P.K = 0;
    P.first_m = first_merged + (uint32_t)run_done;
    for (int u = 0; u < (int)kMaxBatch; u++) { P.l[u] = kHole; P.r[u] = kHole; }
  }
  __syncthreads();
  TiedPlan my_plan{0, 0, 0, 0};  // lane t: the list of tied pair t (the tie launch's planner has looked it up) ...
  unsigned int my_rank = 0xFFFFFFFFu;  // ... and its place among the pairs that were seen, by position
  if (mx && tied >= 2 && n_list) {
    unsigned int my_k = 0xFFFFFFFFu;
    unsigned long long my_pos = kEmptyKey;
    if (threadIdx.x < n_list) {
      const uint32_t info = C.tied_idx[par * kTieSet + threadIdx.x];
      const unsigned long long key = C.tied_key[par * kTieSet + threadIdx.x];
      my_plan = C.tied_plan[par * kTieSet + threadIdx.x];
      const size_t ci = (size_t)par * C.cand_cap + (info & 0x7FFFFFFFu);
      my_pos = C.gpos[ci];
      const uint32_t l_lo = C.gnb_min[2 * ci], l_hi = C.gnb_max[2 * ci], r_lo = C.gnb_min[2 * ci + 1], r_hi = C.gnb_max[2 * ci + 1];
      if (my_pos != kEmptyKey) {
        my_k = atomicAdd(&n_found, 1u);
        f_pos[my_k] = my_pos;
        f_key[my_k] = key;
        // dangerous by the symbol sets, unless the occurrences the scan saw clear it on both sides (twins stay dangerous)
        f_dng[my_k] = (info >> 31) && !(l_lo < l_hi && r_lo < r_hi && (uint32_t)(key >> 32) != (uint32_t)key);
      }
    }
    __syncthreads();
    const unsigned int nf = n_found;
    if (my_k != 0xFFFFFFFFu) {
      unsigned int rank = 0;
      for (unsigned int u = 0; u < nf; u++) rank += f_pos[u] < my_pos ? 1u : 0u;  // positions are distinct
      my_rank = rank;
      if (rank < kMaxBatch) ord[rank] = my_k;
    }
    __syncthreads();
    if (threadIdx.x < 64 && nf) {
      // the longest prefix whose members share no symbol, up to and including the first dangerous one (see above): the first
      // wave holds one candidate per lane, in order, and settles it with shuffles and ballots
      const unsigned int nsel = nf < kMaxBatch ? nf : kMaxBatch;
      const unsigned int rk = threadIdx.x;
      uint32_t a = kHole, b = kHole, dng = 0;
      unsigned long long pos = kEmptyKey;
      if (rk < nsel) {
        const unsigned int e = ord[rk];
        a = (uint32_t)(f_key[e] >> 32);
        b = (uint32_t)f_key[e];
        pos = f_pos[e];
        dng = f_dng[e];
      }
      bool clash = false;  // with any pair before it (were one of those left out, the prefix would end there anyway)
#pragma unroll
      for (int jj = 0; jj < (int)kMaxBatch; jj++) {
        const uint32_t aj = __shfl(a, jj), bj = __shfl(b, jj);
        if ((unsigned int)jj < rk && rk < nsel) clash |= a == aj || a == bj || b == aj || b == bj;
      }
      const bool far = rk > 0 && rk < nsel && pos >= win_end;  // not every workgroup scanned that far
      const unsigned long long m_clash = __ballot(clash), m_far = __ballot(far), m_dng = __ballot(dng != 0 && rk < nsel);
      const unsigned long long room = (unsigned long long)limit - run_done;
      uint32_t K = nsel;
      int why = nf > kMaxBatch ? 5 : 0;  // 0: every pair seen is in, 1 beyond the window, 2 shared symbol, 3 dangerous, 4 the round trip's cap, 5 kMaxBatch
      if (m_far && (uint32_t)__builtin_ctzll(m_far) < K) { K = (uint32_t)__builtin_ctzll(m_far); why = 1; }
      if (m_clash && (uint32_t)__builtin_ctzll(m_clash) < K) { K = (uint32_t)__builtin_ctzll(m_clash); why = 2; }
      if (m_dng && (uint32_t)__builtin_ctzll(m_dng) + 1 < K) { K = (uint32_t)__builtin_ctzll(m_dng) + 1; why = 3; }
      else if (m_dng && (uint32_t)__builtin_ctzll(m_dng) + 1 == K) why = 3;
      if ((unsigned long long)K > room) { K = (uint32_t)room; why = 4; }
      if (rk < K) { P.l[rk] = a; P.r[rk] = b; }
      if (rk == 0) {
        P.K = K;
        p1_pos = pos;
#ifdef SWT_STAMPS
        if (blockIdx.x == 0) { atomicAdd(&g_why[K], 1ull); atomicAdd(&g_why[16 + why], 1ull); atomicAdd(&g_why[24], (unsigned long long)nf); atomicAdd(&g_why[25], n_list); }
#else
        (void)why;
#endif
      }
    }
  } else if (threadIdx.x == 0 && mx) {
    unsigned long long key = st->best_key;
    if (tied >= 2) {  // a plateau wider than the tie set: the scan left the earliest pair in wkey[]
      const unsigned long long pos = st->best2[par];
      key = pos != kEmptyKey ? C.wkey[pos >> 32] : kEmptyKey;
      p1_pos = pos;
    }
    if (key != kEmptyKey) {
      P.l[0] = (uint32_t)(key >> 32);
      P.r[0] = (uint32_t)key;
      P.K = 1;
      P.ent0[1] = (flags & kFlagIndexBroken) ? 0ull : (tied >= 2 ? plan_member(C, P, 0) : plan_take(C, P, 0, C.tied_plan[par * kTieSet]));
    }
  }
  __syncthreads();
  const uint32_t K = P.K;
  if (K == 0) {  // bpe.py:98-99: no pair left
    if (lead) { st->halt = 2; st->run_done[par ^ 1u] = run_done; }
    return;
  }
  if (my_rank < K) P.ent0[my_rank + 1] = (flags & kFlagIndexBroken) ? 0ull : plan_take(C, P, (int)my_rank, my_plan);  // lengths ...
  __syncthreads();
  if (threadIdx.x == 0) {  // ... to running sums
    P.ent0[0] = 0;
    for (uint32_t q = 0; q < K; q++) P.ent0[q + 1] += P.ent0[q];
  }
#ifdef SWT_STAMPS
  if (lead) g_kstep[C.step & (kSpanSteps - 1)] = K;
#endif
  if (lead) {
    const unsigned long long pos = p1_pos;
    st->res_pos = pos;
    st->win_key = pair_key(P.l[0], P.r[0]);
    st->best2[par ^ 1u] = kEmptyKey;    // the next step's scan starts from a clean minimum
    st->n_synced = st->n_synced_next;  // the tie launch mirrored the candidates up to there
    // the plateau cursor: a plain store beside the other workgroups' atomicMin of the words they touch -- both orders leave a
    // valid lower bound (the first member's first word is the first word this step touches)
    if (tied >= 2 && pos != kEmptyKey) { st->plateau = mx; st->cursor_w = (uint32_t)(pos >> 32); }
    else if (st->plateau != mx) { st->plateau = mx; st->cursor_w = 0; }
    st->run_done[par ^ 1u] = run_done + K;
    st->run_active += 1;
  }
  if (blockIdx.x == 0 && threadIdx.x < K) {  // one lane per member: its birth step, its log line
    const uint32_t q = threadIdx.x, merged = P.first_m + q;
    const unsigned long long n_syms = st->step_syms, n_cand = st->n_cand;
    if (merged >= C.id_base && merged - C.id_base < C.seg_cap && C.seg_of[merged - C.id_base] == 0) C.seg_of[merged - C.id_base] = C.step;
    else atomicOr(&st->flags, kFlagBrokenPending);  // the next step's tie launch turns it into kFlagIndexBroken
    StepLog &row = log[run_done + q];
    row.l = P.l[q];
    row.r = P.r[q];
    row.count = mx;
    row.flag = 0ull;
    row.n_syms = n_syms;  // of the step: the members after the first saw a few symbols less
    row.n_tied = tied;
    row.n_cand = n_cand;
  }
  __syncthreads();
  apply_body(sym, woff, freq, n_words, C, P);
}

// ---- the fast path, corpus-sharded (swt_dist.hip drives it) ----------------------------------------------------------------
// Every rank holds the histogram of the WHOLE corpus, so fast_tie_kernel finds the same maximum and the same tied set on
// every rank; what differs is where the tied pairs OCCUR.  bpe.py:102 orders them by first occurrence over the whole
// corpus = (rank, word, offset), ranks being ordered by their sentence ranges.  Per step:
//   fast_tie_kernel            as unsharded: local window scan from the local plateau cursor
//   tie_pack_kernel            this rank's TieMsg: its <= kMaxBatch earliest tied pairs inside its window (with the
//                              neighbour evidence), its earliest tied occurrence anywhere, whether the window was everything
//   [all-gather of the TieMsgs]
//   fast_apply_sharded_kernel  every workgroup of every rank derives the SAME batch from the gathered messages, then
//                              apply_body on the local words (deltas into pend[], as the generic sharded step)
//   pack_records -> [all-gather of the record blocks] -> add_blocks -> finish_exchange        (once per STEP, not per merge)
// What is certain about the global order: the pairs a rank saw inside its window are its earliest ones; a pair it did not
// see there may still occur behind the window -- unless the window was the whole rest of the shard (`exhausted`).  So the
// batch is taken from: everything the ranks 0..r*-1 reported (all exhausted), plus the window of r*, the first rank that is
// not; a pair seen by several ranks belongs to the first.  If that leaves nothing (r* saw no tied pair inside its window),
// the step merges the one pair that is certain: r*'s earliest occurrence anywhere (its scan goes on until it has one).
__global__ __launch_bounds__(kTrainThreads) void tie_pack_kernel(TrainCtx C, uint64_t n_words, uint32_t limit) {
  __shared__ unsigned long long f_pos[kTieSet], f_key[kTieSet];
  __shared__ uint32_t f_info[kTieSet][5];
  __shared__ unsigned int n_found, n_inwin;
  TrainState *st = C.st;
  TieMsg *msg = C.tie_msg;
  const unsigned par = C.step & 1u;
  const unsigned int flags = st->flags;
  const unsigned long long run_done = st->run_done[par], halt = st->halt;
  const unsigned long long mx = st->max_count, tied = st->n_tied, n_list = st->n_list[par], win_end = st->win_end;
  const bool idle = (flags & kFlagReplan) || halt || run_done >= limit || (C.halt_ext && *C.halt_ext);
  if (threadIdx.x == 0) { n_found = 0; n_inwin = 0; }
  __syncthreads();
  if (idle || !mx) {  // fast_apply_sharded_kernel does nothing either (the same state on every rank)
    if (threadIdx.x == 0) { msg->win_end = 0; msg->min_pos = kEmptyKey; msg->min_key = kEmptyKey; msg->n = 0; msg->exhausted = 1; }
    return;
  }
  if (tied >= 2 && n_list) {
    unsigned int my_k = 0xFFFFFFFFu;
    unsigned long long my_pos = kEmptyKey;
    if (threadIdx.x < n_list) {
      const uint32_t info = C.tied_idx[par * kTieSet + threadIdx.x];
      const size_t ci = (size_t)par * C.cand_cap + (info & 0x7FFFFFFFu);
      my_pos = C.gpos[ci];
      if (my_pos != kEmptyKey) {
        my_k = atomicAdd(&n_found, 1u);
        f_pos[my_k] = my_pos;
        f_key[my_k] = C.tied_key[par * kTieSet + threadIdx.x];
        f_info[my_k][0] = C.gnb_min[2 * ci];
        f_info[my_k][1] = C.gnb_min[2 * ci + 1];
        f_info[my_k][2] = C.gnb_max[2 * ci];
        f_info[my_k][3] = C.gnb_max[2 * ci + 1];
        f_info[my_k][4] = info >> 31;
        if (my_pos < win_end) atomicAdd(&n_inwin, 1u);
      }
    }
    __syncthreads();
    const unsigned int nf = n_found;
    if (my_k != 0xFFFFFFFFu) {
      unsigned int rank = 0;
      for (unsigned int u = 0; u < nf; u++) rank += f_pos[u] < my_pos ? 1u : 0u;  // positions are distinct
      if (rank < kMaxBatch && my_pos < win_end) {
        TieEntry &e = msg->e[rank];
        e.pos = my_pos;
        e.key = f_key[my_k];
        e.nb_lo[0] = f_info[my_k][0]; e.nb_lo[1] = f_info[my_k][1];
        e.nb_hi[0] = f_info[my_k][2]; e.nb_hi[1] = f_info[my_k][3];
        e.danger = f_info[my_k][4];
        e.pad = 0;
      }
      if (rank == 0) { msg->min_pos = my_pos; msg->min_key = f_key[my_k]; }
    }
    if (threadIdx.x == 0) {
      if (!nf) { msg->min_pos = kEmptyKey; msg->min_key = kEmptyKey; }
      msg->win_end = win_end;
      msg->n = n_inwin < kMaxBatch ? n_inwin : kMaxBatch;  // (the pairs inside the window are a prefix of the order by position)
      msg->exhausted = ((win_end >> 32) >= n_words || !nf) ? 1u : 0u;
    }
  } else if (threadIdx.x == 0) {
    unsigned long long pos = kEmptyKey, key = st->best_key;  // one pair holds the maximum: every rank names it
    if (tied >= 2) {  // a plateau wider than the tie set: the scan left this shard's earliest pair in wkey[]
      pos = st->best2[par];
      key = pos != kEmptyKey ? C.wkey[pos >> 32] : kEmptyKey;
    }
    msg->win_end = win_end;
    msg->min_pos = pos;
    msg->min_key = key;
    msg->n = 0;
    msg->exhausted = (tied >= 2 && pos != kEmptyKey) ? 0u : 1u;  // wide plateau: only the earliest occurrence is known
  }
}

__global__ __launch_bounds__(kTrainThreads) void fast_apply_sharded_kernel(uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                                           const uint32_t *__restrict__ freq, uint64_t n_words, TrainCtx C,
                                                                           StepLog *__restrict__ log, uint32_t first_merged, uint32_t limit) {
  __shared__ BatchPlan P;
  __shared__ unsigned long long g_key[kMaxBatch];
  __shared__ uint32_t g_lo[kMaxBatch][2], g_hi[kMaxBatch][2], g_dng[kMaxBatch];
  __shared__ unsigned int g_n;
  TrainState *st = C.st;
  const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
  const unsigned par = C.step & 1u;
  __shared__ unsigned long long hdr[3];
  if (threadIdx.x == 0) { hdr[0] = st->flags; hdr[1] = st->run_done[par]; hdr[2] = st->halt; }
  const unsigned long long mx = st->max_count, tied_here = st->n_tied, n_list = st->n_list[par];
  __syncthreads();
  const unsigned int flags = (unsigned int)hdr[0];
  const unsigned long long run_done = hdr[1];
  if ((flags & kFlagReplan) || hdr[2] || run_done >= limit || (C.halt_ext && *C.halt_ext)) {
    if (lead) st->run_done[par ^ 1u] = run_done;
    return;
  }
  if (threadIdx.x == 0) {
    g_n = 0;
    P.K = 0;
    P.first_m = first_merged + (uint32_t)run_done;
    for (int u = 0; u < (int)kMaxBatch; u++) { P.l[u] = kHole; P.r[u] = kHole; }
  }
  __syncthreads();
  // ---- the batch, from the gathered messages alone (so every workgroup of every rank decides alike) ----
  // ranks in order; the first wave collects, the workgroup's barriers carry g_n from rank to rank (uniform trip count: the
  // loop looks at nothing but the gathered messages and g_n)
  __shared__ unsigned int g_lone;
  if (threadIdx.x == 0) g_lone = 0;
  __syncthreads();
  if (mx) {
    for (uint32_t r = 0; r < C.world; r++) {
      const TieMsg &m = C.tie_all[r];
      const uint32_t nr = m.n < kMaxBatch ? m.n : kMaxBatch;
      if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned int gn = g_n;
        unsigned long long key = kEmptyKey;
        bool fresh = false;
        if ((uint32_t)lane < nr) {
          key = m.e[lane].key;
          fresh = true;
          for (unsigned int u = 0; u < gn; u++) fresh &= g_key[u] != key;  // a lower rank holds it earlier
        }
        const unsigned long long F = __ballot(fresh);
        const unsigned int at = gn + (unsigned int)__popcll(F & ((1ull << lane) - 1ull));
        if (fresh && at < kMaxBatch) {
          g_key[at] = key;
          g_lo[at][0] = 0xFFFFFFFFu; g_lo[at][1] = 0xFFFFFFFFu;
          g_hi[at][0] = 0u; g_hi[at][1] = 0u;
          g_dng[at] = m.e[lane].danger;
        }
        if (lane == 0) {
          const unsigned int tot = gn + (unsigned int)__popcll(F);
          g_n = tot < kMaxBatch ? tot : kMaxBatch;
        }
      }
      __syncthreads();
      const unsigned int gn = g_n;
      // an untied step (every rank names the one pair that holds the maximum), or nothing certain but this rank's
      // earliest occurrence behind its window: the batch is that single pair
      const bool untied = m.min_pos == kEmptyKey && m.min_key != kEmptyKey && m.n == 0;
      const bool behind = !m.exhausted && m.min_key != kEmptyKey;
      if (gn == 0 && (untied || behind)) {
        if (threadIdx.x == 0) {
          g_key[0] = m.min_key; g_dng[0] = 1u;
          g_lo[0][0] = g_lo[0][1] = 0xFFFFFFFFu; g_hi[0][0] = g_hi[0][1] = 0u;
          g_n = 1; g_lone = 1;
        }
        break;
      }
      if (!m.exhausted || gn >= kMaxBatch) break;  // what the later ranks hold may come after something unseen here
    }
    __syncthreads();
    // the evidence every rank saw of the collected pairs (any occurrence anywhere is a fact)
    if (!g_lone && threadIdx.x < 64) {
      const unsigned int gn = g_n;
      for (uint32_t r = 0; r < C.world; r++) {
        const TieMsg &m = C.tie_all[r];
        const uint32_t nr = m.n < kMaxBatch ? m.n : kMaxBatch;
        if (threadIdx.x < nr) {
          const unsigned long long key = m.e[threadIdx.x].key;
          for (unsigned int u = 0; u < gn; u++)
            if (g_key[u] == key) {
              atomicMin(&g_lo[u][0], m.e[threadIdx.x].nb_lo[0]);
              atomicMin(&g_lo[u][1], m.e[threadIdx.x].nb_lo[1]);
              atomicMax(&g_hi[u][0], m.e[threadIdx.x].nb_hi[0]);
              atomicMax(&g_hi[u][1], m.e[threadIdx.x].nb_hi[1]);
            }
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 64 && mx) {
    const int lane = threadIdx.x;
    const unsigned int gn = g_n;
    // the longest prefix whose members share no symbol, up to and including the first dangerous one (fast_apply_kernel)
    const unsigned int nsel = gn;
    const unsigned int rk = lane;
    uint32_t a = kHole, b = kHole, dng = 0;
    if (rk < nsel) {
      a = (uint32_t)(g_key[rk] >> 32);
      b = (uint32_t)g_key[rk];
      dng = g_dng[rk] && !(g_lo[rk][0] < g_hi[rk][0] && g_lo[rk][1] < g_hi[rk][1] && a != b);
    }
    bool clash = false;
#pragma unroll
    for (int jj = 0; jj < (int)kMaxBatch; jj++) {
      const uint32_t aj = __shfl(a, jj), bj = __shfl(b, jj);
      if ((unsigned int)jj < rk && rk < nsel) clash |= a == aj || a == bj || b == aj || b == bj;
    }
    const unsigned long long m_clash = __ballot(clash), m_dng = __ballot(dng != 0 && rk < nsel);
    const unsigned long long room = (unsigned long long)limit - run_done;
    uint32_t K = nsel;
    if (m_clash && (uint32_t)__builtin_ctzll(m_clash) < K) K = (uint32_t)__builtin_ctzll(m_clash);
    if (m_dng && (uint32_t)__builtin_ctzll(m_dng) + 1 < K) K = (uint32_t)__builtin_ctzll(m_dng) + 1;
    if ((unsigned long long)K > room) K = (uint32_t)room;
    if (rk < K) { P.l[rk] = a; P.r[rk] = b; }
    if (rk == 0) P.K = K;
  }
  __syncthreads();
  const uint32_t K = P.K;
  if (K == 0) {  // bpe.py:98-99: no pair left in any shard
    if (lead) { st->halt = 2; st->run_done[par ^ 1u] = run_done; }
    return;
  }
  // ---- where this rank's words of the K pairs are listed: its own tied list knows (the planner looked them up), else a lookup
  if (threadIdx.x < kMaxBatch) P.ent0[threadIdx.x + 1] = ~0ull;
  __syncthreads();
  if (!(flags & kFlagIndexBroken)) {
    if (tied_here >= 2 && threadIdx.x < n_list) {
      const unsigned long long key = C.tied_key[par * kTieSet + threadIdx.x];
      for (uint32_t q = 0; q < K; q++)
        if (pair_key(P.l[q], P.r[q]) == key) P.ent0[q + 1] = plan_take(C, P, (int)q, C.tied_plan[par * kTieSet + threadIdx.x]);
    } else if (tied_here < 2 && threadIdx.x == 0 && K == 1 && pair_key(P.l[0], P.r[0]) == st->best_key) {
      P.ent0[1] = plan_take(C, P, 0, C.tied_plan[par * kTieSet]);
    }
  }
  __syncthreads();
  if (threadIdx.x < K && P.ent0[threadIdx.x + 1] == ~0ull)
    P.ent0[threadIdx.x + 1] = (flags & kFlagIndexBroken) ? 0ull : plan_member(C, P, (int)threadIdx.x);
  __syncthreads();
  if (threadIdx.x == 0) {
    P.ent0[0] = 0;
    for (uint32_t q = 0; q < K; q++) P.ent0[q + 1] += P.ent0[q];
  }
  if (lead) {
    const unsigned long long my_pos = st->best2[par];  // this shard's earliest tied occurrence (the scan went on until it had one)
    st->res_pos = my_pos;
    st->win_key = pair_key(P.l[0], P.r[0]);
    st->best2[par ^ 1u] = kEmptyKey;
    st->n_synced = st->n_synced_next;
    // the plateau cursor is local: every tied pair of this shard lies at or after its earliest tied occurrence
    if (tied_here >= 2 && my_pos != kEmptyKey) { st->plateau = mx; st->cursor_w = (uint32_t)(my_pos >> 32); }
    else if (st->plateau != mx) { st->plateau = mx; st->cursor_w = 0; }
    st->run_done[par ^ 1u] = run_done + K;
    st->run_active += 1;
  }
  if (blockIdx.x == 0 && threadIdx.x < K) {
    const uint32_t q = threadIdx.x, merged = P.first_m + q;
    const unsigned long long n_syms = st->step_syms, n_cand = st->n_cand;
    if (merged >= C.id_base && merged - C.id_base < C.seg_cap && C.seg_of[merged - C.id_base] == 0) C.seg_of[merged - C.id_base] = C.step;
    else atomicOr(&st->flags, kFlagBrokenPending);
    StepLog &row = log[run_done + q];
    row.l = P.l[q];
    row.r = P.r[q];
    row.count = mx;
    row.flag = 0ull;
    row.n_syms = n_syms;
    row.n_tied = g_n;
    row.n_cand = n_cand;
  }
  __syncthreads();
  apply_body(sym, woff, freq, n_words, C, P);
}

// ---- WordPiece: argmax + tie-break + decision in ONE launch ----------------------------------------------------------------
// The generic step costs four launches (argmax, tie, decide, apply) and what a WordPiece merge costs on the device IS its
// launches: ~8 us each, the work inside them is small (S85k-lex: ~2,400 live pairs, a few dozen words per merge).  While the
// list of live pairs is short (kWpStepList), every workgroup takes the maximum score over it by itself -- as fast_tie_kernel
// does for BPE -- the waves share the tied pairs' index lookups (wp_tie_by_index), and the workgroup that finishes LAST (a
// ticket) reads the final first position and does what decide_kernel does: the merge command for apply_kernel, the log line,
// the step's index segment, the symbol frequencies.  No apply is in flight during this launch, so the stream, the counts and
// the frequencies stand still.  Two launches per merge instead of four.
__global__ __launch_bounds__(kTrainThreads) void wp_step_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                                uint64_t n_words, TrainCtx C, StepCmd *cmd, StepLog *log, uint32_t log_i,
                                                                uint32_t merged) {
  TrainState *st = C.st;
  __shared__ unsigned int s_last;
  const unsigned long long n_all = st->n_cand;
  const unsigned long long n = n_all < C.cand_cap ? n_all : C.cand_cap;
  const unsigned long long n_synced = st->n_synced < n ? st->n_synced : n;
  const unsigned int flags = st->flags;
  unsigned long long m = 0, c = 0, k = kEmptyKey;
  // eight entries per lane and trip, each stage's loads in flight together: counts + keys (two coalesced streams), then the
  // two symbol frequencies of every pair (gathers), then the divisions.  Entry by entry this loop was a chain of dependent L2
  // round trips -- most of what the launch cost.
  constexpr int kU = 8;
  for (uint64_t i0 = threadIdx.x; i0 < n; i0 += (uint64_t)blockDim.x * kU) {
    long long v[kU], fl[kU], fr[kU];
    unsigned long long key[kU];
    uint32_t slot[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const uint64_t i = i0 + (uint64_t)u * blockDim.x;
      v[u] = 0; key[u] = kEmptyKey; slot[u] = 0xFFFFFFFFu;
      if (i < n_synced) { v[u] = C.ccnt[i]; key[u] = C.ckey[i]; }
      else if (i < n) slot[u] = C.cand[i];
    }
#pragma unroll
    for (int u = 0; u < kU; u++)
      if (slot[u] != 0xFFFFFFFFu) { v[u] = C.T.cnt[slot[u]]; key[u] = C.T.keys[slot[u]]; }
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const bool on = v[u] > 0 && key[u] != kEmptyKey;
      fl[u] = on ? C.sfreq[key[u] >> 32] : 0;
      fr[u] = on ? C.sfreq[(uint32_t)key[u]] : 0;
      if (slot[u] != 0xFFFFFFFFu && blockIdx.x == 0) {  // the pairs the last merge made get their mirror (the others read them from the table meanwhile)
        const uint64_t i = i0 + (uint64_t)u * blockDim.x;
        C.ccnt[i] = v[u];
        C.ckey[i] = key[u];
        C.cidx[slot[u]] = (uint32_t)i;
      }
    }
#pragma unroll
    for (int u = 0; u < kU; u++) {
      if (v[u] > 0 && key[u] != kEmptyKey) {
        const unsigned long long val = wp_score_bits((unsigned long long)v[u], (unsigned long long)fl[u], (unsigned long long)fr[u]);
        if (val >= m) arg_combine(m, c, k, val, 1ull, key[u]);
      }
    }
  }
  const BlockArg a = block_reduce(m, c, k);
  const bool dry = (flags & kFlagReplan) || n_all > C.cand_cap;  // cand_dry at theta 1: the list lost a pair, or was voided
  if (!dry && a.mx && a.tied >= 2) {
    if (flags & kFlagIndexBroken) {  // no index: the stream scan (tie_kernel)
      for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        if ((w << 32) >= __hip_atomic_load(&st->best_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        const uint64_t b0 = woff[w], b1 = woff[w + 1];
        uint32_t x = kHole;
        uint64_t xi = 0;
        for (uint64_t i = b0; i < b1; i++) {
          const uint32_t y = sym[i];
          if (y == kHole) continue;
          if (x != kHole) {
            const unsigned long long key = pair_key(x, y);
            if (pair_value(key, table_get(C.T, key), C.sfreq) == a.mx) {
              atomicMin(&st->best_pos, (unsigned long long)((w << 32) | (xi - b0)));
              break;
            }
          }
          x = y;
          xi = i;
        }
      }
    } else {
      wp_tie_by_index(sym, woff, n_words, C, a.mx, n, n_synced);
    }
  }
  // ---- the last workgroup to get here decides.  Everything the workgroups tell each other in this launch travels in device-scope
  // atomics (best_pos, the ticket), so all the ticket needs is that this workgroup's atomics have been performed: a workgroup-scope
  // fence (a wait for the outstanding memory operations) and the barrier.  __threadfence() -- a device-scope fence -- writes the
  // XCD's L2 back and invalidates it, once per workgroup and step (SWT_WP_THREADFENCE=1 in the environment of the build: the old form).
#ifdef SWT_WP_THREADFENCE
  __threadfence();
#else
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&st->ticket, 1ull) == (unsigned long long)gridDim.x - 1ull ? 1u : 0u;
  __syncthreads();
  if (!s_last || threadIdx.x != 0) return;
#ifdef SWT_WP_THREADFENCE
  __threadfence();
#else
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
  st->ticket = 0;
  unsigned long long key = a.key, pos = kEmptyKey;
  if (!dry && a.mx && a.tied >= 2) {
    pos = __hip_atomic_load(&st->best_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    key = pos != kEmptyKey ? pair_at(sym, woff, pos) : kEmptyKey;
  }
  st->max_count = dry ? 0 : a.mx;
  st->n_tied = (dry || !a.mx) ? 0 : a.tied;
  st->best_key = key;
  st->win_key = key;
  st->res_pos = pos;
  st->best_pos = kEmptyKey;
  if (dry) st->flags |= kFlagReplan;
  st->n_synced = n;
  const bool ok = !dry && a.mx > 0 && key != kEmptyKey;
  cmd->l = (uint32_t)(key >> 32);
  cmd->r = (uint32_t)key;
  cmd->m = merged;
  cmd->valid = ok ? 1u : 0u;
  open_step(C, merged, ok);
  if (ok) wp_move_freq(C.T, cmd->l, cmd->r, merged, C.sfreq);
  log[log_i].l = cmd->l;
  log[log_i].r = cmd->r;
  log[log_i].count = a.mx;
  log[log_i].flag = ok ? 0ull : (dry ? 3ull : 2ull);
  log[log_i].n_syms = st->n_syms;
  log[log_i].n_tied = a.tied;
  log[log_i].n_cand = n_all;
}

// ---- candidates --------------------------------------------------------------------------------------------------------
// histogram of the live counts over 8 sub-buckets per octave: the host picks theta so that ~kCandTarget pairs pass it
__device__ __forceinline__ uint32_t count_bucket(unsigned long long c) {
  const int e = 63 - __builtin_clzll(c);
  const uint32_t sub = e >= 3 ? (uint32_t)((c >> (e - 3)) & 7u) : (uint32_t)((c << (3 - e)) & 7u);
  return (uint32_t)e * 8u + sub;
}

__global__ __launch_bounds__(256) void cand_hist_kernel(const long long *__restrict__ cnt, uint64_t cap, unsigned long long *__restrict__ buckets) {
  __shared__ unsigned int lb[512];
  for (int i = threadIdx.x; i < 512; i += blockDim.x) lb[i] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
    const long long c = cnt[i];
    if (c > 0) atomicAdd(&lb[count_bucket((unsigned long long)c)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += blockDim.x)
    if (lb[i]) atomicAdd(&buckets[i], (unsigned long long)lb[i]);
}

__global__ __launch_bounds__(256) void cand_build_kernel(TrainCtx C, uint64_t cap) {
  const int lane = threadIdx.x & 63;
  for (uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; i0 < cap; i0 += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = i0 + lane;
    const bool in = i < cap && C.T.cnt[i] >= (long long)C.theta && C.T.keys[i] != kEmptyKey;
    const unsigned long long mask = __ballot(in);
    if (!mask) continue;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&C.st->n_cand, (unsigned long long)__popcll(mask));
    base = __shfl(base, 0);
    if (in) {
      const unsigned long long k = base + __popcll(mask & ((1ull << lane) - 1ull));
      if (k < C.cand_cap) {
        C.cand[k] = (uint32_t)i;
        C.ccnt[k] = C.T.cnt[i];
        C.ckey[k] = C.T.keys[i];
        C.cidx[i] = (uint32_t)k;
      }
    }
  }
}

// ---- table maintenance, export -----------------------------------------------------------------------------------------
__global__ void add_records_kernel(const DeltaRec *__restrict__ recs, uint64_t n, TrainCtx C) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    if (recs[i].delta != 0) count_add(C, table_slot(C.T, recs[i].key, C.st), recs[i].delta);
}

// live entries -> (key, count) records
__global__ void table_export_kernel(const unsigned long long *__restrict__ keys, const long long *__restrict__ cnt, uint64_t cap,
                                    DeltaRec *__restrict__ out, uint64_t out_cap, unsigned long long *n_out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
    if (keys[i] != kEmptyKey && cnt[i] != 0) {
      const unsigned long long k = atomicAdd(n_out, 1ull);
      if (k < out_cap) { out[k].key = keys[i]; out[k].delta = cnt[i]; }
    }
  }
}

__global__ void table_rehash_kernel(const unsigned long long *__restrict__ keys, const long long *__restrict__ cnt, uint64_t cap,
                                    PairTable dst, TrainState *st) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x)
    if (keys[i] != kEmptyKey && cnt[i] != 0)
      atomicAdd(reinterpret_cast<unsigned long long *>(&dst.cnt[table_slot(dst, keys[i], st)]), (unsigned long long)cnt[i]);
}

// ---- sharded training: the exchange of one step's deltas ----------------------------------------------------------------
// the slots this rank's apply touched -> one record block.  block[0] is the header: key = records that follow (or were
// wanted), delta != 0 when they did not fit.
__global__ void pack_records_kernel(TrainCtx C, DeltaRec *__restrict__ block, uint64_t block_cap) {
  const unsigned long long n = C.st->n_touched;
  const bool over = n > C.touched_cap || n + 1 > block_cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) { block[0].key = n; block[0].delta = over ? 1 : 0; }
  if (over) return;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t h = C.touched[i];
    block[i + 1].key = C.T.keys[h];
    block[i + 1].delta = C.pend[h];
  }
}

// Every rank's block into this replica -- unless a block overflowed somewhere: then nothing is added on any rank, the halt
// flag stops the following steps, and the host grows the blocks and repeats the exchange of this step.
__global__ void add_blocks_kernel(const DeltaRec *__restrict__ blocks, uint32_t world, uint64_t block_cap, TrainCtx C, unsigned int *halt) {
  __shared__ int bad;
  if (threadIdx.x == 0) {
    int b = *halt != 0;
    for (uint32_t r = 0; r < world; r++)
      if (blocks[(uint64_t)r * block_cap].delta != 0) b = 1;
    bad = b;
  }
  __syncthreads();
  if (bad) return;
  for (uint32_t r = 0; r < world; r++) {
    const DeltaRec *blk = blocks + (uint64_t)r * block_cap;
    const uint64_t n = blk[0].key;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
      if (blk[i + 1].delta != 0) count_add(C, table_slot(C.T, blk[i + 1].key, C.st), blk[i + 1].delta);
  }
}

// after add_blocks_kernel: pend[] of the touched slots back to zero and the list emptied -- or the halt raised.
// Also the candidate list's new members.  add_blocks_kernel adds the ranks' blocks concurrently, so the ORDER in which the
// deltas of one pair land differs from rank to rank: a push "when the count crosses theta" would list a pair twice on one
// rank and once on another, the lists would differ in LENGTH, and a host decision that looks at the length (re-plan above
// kCandHigh) would split the ranks.  So add_blocks_kernel adds without listing, and the listing happens here from the FINAL
// counts, which are the same everywhere: every record's pair that now counts >= theta and is not listed yet is pushed once
// (cidx[slot] goes from "none" to kCidxPending by CAS; the next step's housekeeper mirrors it).
__global__ void finish_exchange_kernel(const DeltaRec *__restrict__ blocks, uint32_t world, uint64_t block_cap, TrainCtx C, unsigned int *halt) {
  __shared__ int bad;
  if (threadIdx.x == 0) {
    int b = *halt != 0;
    for (uint32_t r = 0; r < world; r++)
      if (blocks[(uint64_t)r * block_cap].delta != 0) b = 1;
    bad = b;
  }
  __syncthreads();
  if (bad) {
    if (threadIdx.x == 0) *halt = 1u;
    return;
  }
  const unsigned long long n = C.st->n_touched;
  for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) C.pend[C.touched[i]] = 0;
  if (C.theta && C.cidx)
    for (uint32_t r = 0; r < world; r++) {
      const DeltaRec *blk = blocks + (uint64_t)r * block_cap;
      const uint64_t nr = blk[0].key;
      for (uint64_t i = threadIdx.x; i < nr; i += blockDim.x) {
        if (blk[i + 1].delta <= 0) continue;  // only a pair that gained can have risen to theta
        const uint32_t slot = table_slot(C.T, blk[i + 1].key, C.st);  // (the key is there: add_blocks_kernel put it)
        if (C.T.cnt[slot] >= (long long)C.theta && atomicCAS(&C.cidx[slot], 0xFFFFFFFFu, kCidxPending) == 0xFFFFFFFFu) cand_push(C, slot);
      }
    }
  __syncthreads();
  if (threadIdx.x == 0) C.st->n_touched = 0;
}

// recovery after an overflow: the touched list is rebuilt from pend[] itself (the list may have dropped slots)
__global__ void rebuild_touched_kernel(TrainCtx C, uint64_t cap) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x)
    if (C.pend[i] != 0) {
      const unsigned long long k = atomicAdd(&C.st->n_touched, 1ull);
      if (k < C.touched_cap) C.touched[k] = (uint32_t)i;
    }
}

// sharded tie-break: this rank's (first position, pair) for the all-gather
__global__ __launch_bounds__(64) void tie_send_kernel(TrainCtx C, const ArgPart *__restrict__ parts, uint32_t n_parts,
                                                      const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                      unsigned long long *__restrict__ line) {
  unsigned long long mx, tied, key;
  arg_collect(parts, n_parts, mx, tied, key);
  if (threadIdx.x != 0) return;
  unsigned long long pos = kEmptyKey, k = kEmptyKey;
  if (mx && tied >= 2 && !cand_dry(C, mx)) {
    pos = C.st->best_pos;
    if (pos != kEmptyKey) k = pair_at(sym, woff, pos);
  }
  line[0] = pos;
  line[1] = k;
}

// ... and the decision from every rank's line: ranks are ordered by their sentence ranges, so the first rank that holds a
// tied pair holds the earliest occurrence
__global__ __launch_bounds__(64) void decide_sharded_kernel(TrainCtx C, const ArgPart *__restrict__ parts, uint32_t n_parts,
                                                            const unsigned long long *__restrict__ lines, uint32_t world, uint32_t rank,
                                                            StepCmd *cmd, StepLog *log, uint32_t log_i, uint32_t merged,
                                                            const unsigned int *__restrict__ halt) {
  unsigned long long mx, tied, key;
  arg_collect(parts, n_parts, mx, tied, key);
  if (threadIdx.x != 0) return;
  TrainState *st = C.st;
  const bool halted = *halt != 0;
  const bool dry = cand_dry(C, mx);
  if (!dry && !halted && mx && tied >= 2) {
    key = kEmptyKey;
    for (uint32_t r = 0; r < world; r++)
      if (lines[2 * r] != kEmptyKey) { key = lines[2 * r + 1]; break; }
    const unsigned long long mine = lines[2 * rank];
    if (mine != kEmptyKey) { st->plateau = mx; st->cursor_w = (uint32_t)(mine >> 32); }
    else if (st->plateau != mx) { st->plateau = mx; st->cursor_w = 0; }
  } else if (!dry && !halted && st->plateau != mx) {
    st->plateau = mx; st->cursor_w = 0;
  }
  st->best_pos = kEmptyKey;
  st->max_count = dry ? 0 : mx;
  st->n_tied = (dry || !mx) ? 0 : tied;
  st->best_key = key;
  if (dry) st->flags |= kFlagReplan;
  const bool ok = !dry && !halted && mx > 0 && key != kEmptyKey;
  cmd->l = (uint32_t)(key >> 32);
  cmd->r = (uint32_t)key;
  cmd->m = merged;
  cmd->valid = ok ? 1u : 0u;
  open_step(C, merged, ok);
  log[log_i].l = cmd->l;
  log[log_i].r = cmd->r;
  log[log_i].count = mx;
  log[log_i].flag = ok ? 0ull : (halted ? 4ull : (dry ? 3ull : 2ull));
  log[log_i].n_syms = st->n_syms;
  log[log_i].n_tied = tied;
  log[log_i].n_cand = st->n_cand;
}

}  // namespace swt

using namespace swt;

static double host_now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static unsigned grid_for(uint64_t n, int threads, unsigned cap = 1u << 20) {
  uint64_t g = (n + threads - 1) / threads;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

// ---- squeezing the holes out -------------------------------------------------------------------------------------------
// Merges leave holes (stable addresses are what lets a step touch only the words it changes), and late in training two slots
// in three are holes: every walk and every tie scan steps over them.  Nothing but woff[] refers to a stream address between
// steps (the index lists WORDS), so at a re-plan the host may rewrite the stream without them: count, scan, copy.
__global__ void live_count_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff, uint64_t n_words,
                                  unsigned long long *__restrict__ out) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w > n_words) return;
  unsigned long long n = 0;
  if (w < n_words)
    for (uint64_t i = woff[w], e = woff[w + 1]; i < e; i++) n += sym[i] != kHole ? 1 : 0;
  out[w] = n;  // out[n_words] = 0: the exclusive sum leaves the total there
}

__global__ void squeeze_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff, const uint64_t *__restrict__ new_woff,
                               uint64_t n_words, uint32_t *__restrict__ out) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  uint64_t o = new_woff[w];
  for (uint64_t i = woff[w], e = woff[w + 1]; i < e; i++) {
    const uint32_t x = sym[i];
    if (x != kHole) out[o++] = x;
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------

TrainCtx swt_bpe_trainer::ctx() const {
  TrainCtx C{};
  C.T = T;
  C.K = K;
  C.st = d_st;
  C.sfreq = d_sfreq;
  C.theta = theta;
  C.cand = d_cand;
  C.cand_cap = cand_cap;
  C.ccnt = d_ccnt;
  C.ckey = d_ckey;
  C.cidx = cand_valid && theta ? d_cidx : nullptr;
  C.idx_tag = d_idx_tag;
  C.idx_word = d_idx_word;
  C.idx_cap = idx_cap;
  C.seg_start = d_seg_start;
  C.seg_of = d_seg_of;
  C.seg_cap = seg_cap;
  C.id_base = id_base;
  C.wstamp = d_wstamp;
  C.wkey = d_wkey;
  C.tied_idx = d_tied_idx;
  C.tied_key = d_tied_key;
  C.tied_plan = d_tied_plan;
  C.gpos = d_gpos;
  C.gnb_min = d_gnb_min;
  C.gnb_max = d_gnb_max;
  C.step = step_no;
  C.pend = sharded ? d_pend : nullptr;
  C.tstamp = d_tstamp;
  C.touched = d_touched;
  C.touched_cap = touched_cap;
  C.halt_ext = sharded ? d_halt : nullptr;
  C.tie_msg = d_tie_msg;
  C.tie_all = d_tie_msgs;
  C.world = world;
  C.rank = rank;
  return C;
}

static int table_alloc(PairTable &T, uint32_t bits, hipStream_t st) {
  const size_t cap = (size_t)1 << bits;
  T.bits = bits;
  SWT_HIP(hipMalloc((void **)&T.keys, cap * 8));
  SWT_HIP(hipMalloc((void **)&T.cnt, cap * 8));
  hipLaunchKernelGGL(table_clear_kernel, dim3(grid_for(cap, 256, 4096)), dim3(256), 0, st, T.keys, T.cnt, (uint64_t)cap);
  return SWT_OK;
}

static void table_free(PairTable &T) {
  if (T.keys) (void)hipFree(T.keys);
  if (T.cnt) (void)hipFree(T.cnt);
  T.keys = nullptr;
  T.cnt = nullptr;
}

int swt_bpe_trainer::sync_state() {
  SWT_HIP(hipMemcpyAsync(&h_st, d_st, sizeof(TrainState), hipMemcpyDeviceToHost, stream));
  SWT_HIP(hipStreamSynchronize(stream));
  return SWT_OK;
}

int swt_bpe_trainer::ready() const {
  if (broken) return fail(SWT_ERR_STATE, "the trainer failed a capacity check earlier and cannot go on");
  if (!hist_ready || !T.keys || !T.cnt || !T.bits || !d_st || !d_steplog || !d_parts || !d_cmd)
    return fail(SWT_ERR_STATE, "the trainer has no pair histogram (the handle was not completely built)");
  if (n_words && (!K.keys || !K.words || !d_idx_tag || !d_idx_word || !d_wstamp || !d_wkey || !d_seg_start || !d_seg_of))
    return fail(SWT_ERR_STATE, "the trainer has no inverted index (the handle was not completely built)");
  if (n_words && (!d_sym || !d_woff || !d_freq)) return fail(SWT_ERR_STATE, "the trainer has no symbol stream");
  return SWT_OK;
}

// What the kernels rely on without checking, looked at whenever the host has the state anyway (h_st is fresh):
//   the pair table is at most half full (table_slot's probe sequences end; ensure_room's bound held);
//   no insert gave up (kFlagTableFull);
//   the candidate list, the index log and the step numbers are inside what was allocated (the device drops what does not
//   fit and says so in flags -- kFlagReplan, kFlagIndexBroken -- so these two only bound what the HOST sized).
int swt_bpe_trainer::check_state() {
  const uint64_t cap = 1ull << T.bits;
  const char *what = nullptr;
  if (h_st.flags & kFlagTableFull) what = "an insert found the pair table full";
  else if (2 * h_st.n_used > cap) what = "the pair table is more than half full (the head-room bound of a round trip was wrong)";
  else if (h_st.n_syms > n_syms0) what = "more live symbols than the stream has slots";
  else if ((uint64_t)step_no + 2 > seg_start_cap) what = "more steps than index segments were allocated for";
  if (!what) return SWT_OK;
  broken = true;
  return fail(SWT_ERR_STATE, "trainer capacity check failed: %s (keys %llu of %llu slots, candidates %llu, index entries %llu of %llu, step %u)", what,
              (unsigned long long)h_st.n_used, (unsigned long long)cap, (unsigned long long)h_st.n_cand,
              (unsigned long long)h_st.idx_cursor, (unsigned long long)idx_cap, step_no);
}

// per-slot arrays of the sharded mode follow the table's size (called between exchanges: pend[] is all zero then)
static int sharded_arrays(swt_bpe_trainer *t) {
  if (!t->sharded) return SWT_OK;
  const size_t cap = (size_t)1 << t->T.bits;
  if (t->d_pend) (void)hipFree(t->d_pend);
  if (t->d_tstamp) (void)hipFree(t->d_tstamp);
  t->d_pend = nullptr; t->d_tstamp = nullptr;
  SWT_HIP(hipMalloc((void **)&t->d_pend, cap * 8));
  SWT_HIP(hipMalloc((void **)&t->d_tstamp, cap * 4));
  SWT_HIP(hipMemsetAsync(t->d_pend, 0, cap * 8, t->stream));
  SWT_HIP(hipMemsetAsync(t->d_tstamp, 0, cap * 4, t->stream));
  return SWT_OK;
}

// Rebuild the table at `bits` from its live entries (drops zero-count keys).  Slots move: the candidate list is void.
static int table_resize(swt_bpe_trainer *t, uint32_t bits) {
  if (bits > 31) return fail(SWT_ERR_UNSUPPORTED, "pair table would exceed 2^31 slots");
  PairTable nt{nullptr, nullptr, 0};
  int rc = table_alloc(nt, bits, t->stream);
  if (rc) return rc;
  SWT_HIP(hipMemsetAsync(&t->d_st->n_used, 0, 8, t->stream));
  const uint64_t cap = 1ull << t->T.bits;
  hipLaunchKernelGGL(table_rehash_kernel, dim3(grid_for(cap, 256, 4096)), dim3(256), 0, t->stream, t->T.keys, t->T.cnt, cap, nt, t->d_st);
  SWT_HIP(hipStreamSynchronize(t->stream));
  table_free(t->T);
  t->T = nt;
  t->cand_valid = false;
  return sharded_arrays(t);
}

// the stream without its holes (between steps, unsharded BPE; the caller has synchronised h_st)
static int squeeze_stream(swt_bpe_trainer *t) {
  if (!t->n_words || t->sharded) return SWT_OK;
  if (!t->d_sym_alt) {
    SWT_HIP(hipMalloc((void **)&t->d_sym_alt, (size_t)(t->n_syms0 + 16) * 4));
    SWT_HIP(hipMalloc((void **)&t->d_woff_alt, (size_t)(t->n_words + 2) * 8));
  }
  const unsigned g = grid_for(t->n_words + 1, 256);
  unsigned long long *lens = reinterpret_cast<unsigned long long *>(t->d_woff_alt);
  hipLaunchKernelGGL(live_count_kernel, dim3(g), dim3(256), 0, t->stream, t->d_sym, t->d_woff, t->n_words, lens);
  size_t tmp_bytes = 0;
  SWT_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, lens, lens, 0ull, (size_t)(t->n_words + 1), rocprim::plus<unsigned long long>(), t->stream));
  int rc = t->tmp.reserve(tmp_bytes + 16);
  if (rc) return rc;
  SWT_HIP(rocprim::exclusive_scan(t->tmp.p, tmp_bytes, lens, lens, 0ull, (size_t)(t->n_words + 1), rocprim::plus<unsigned long long>(), t->stream));
  hipLaunchKernelGGL(squeeze_kernel, dim3(g), dim3(256), 0, t->stream, t->d_sym, t->d_woff, t->d_woff_alt, t->n_words, t->d_sym_alt);
  unsigned long long total = 0;
  SWT_HIP(hipMemcpyAsync(&total, t->d_woff_alt + t->n_words, 8, hipMemcpyDeviceToHost, t->stream));
  SWT_HIP(hipStreamSynchronize(t->stream));
  std::swap(t->d_sym, t->d_sym_alt);
  std::swap(t->d_woff, t->d_woff_alt);
  t->extent = total;
  t->n_squeezes++;
  return SWT_OK;
}

// theta from the histogram of the counts, then the list of the slots that pass it
// (re)allocate the candidate arrays for `want` entries (the fast path's per-candidate arrays only for BPE)
static int cand_alloc(swt_bpe_trainer *t, uint64_t want) {
  if (t->d_cand && want <= t->cand_cap) return SWT_OK;
  SWT_HIP(hipStreamSynchronize(t->stream));
  for (void *p : {(void *)t->d_cand, (void *)t->d_ccnt, (void *)t->d_ckey, (void *)t->d_gpos, (void *)t->d_gnb_min, (void *)t->d_gnb_max})
    if (p) (void)hipFree(p);
  t->d_cand = nullptr; t->d_ccnt = nullptr; t->d_ckey = nullptr; t->d_gpos = nullptr; t->d_gnb_min = nullptr; t->d_gnb_max = nullptr;
  t->cand_cap = 0;
  SWT_HIP(hipMalloc((void **)&t->d_cand, (size_t)want * 4));
  SWT_HIP(hipMalloc((void **)&t->d_ccnt, (size_t)want * 8));
  SWT_HIP(hipMalloc((void **)&t->d_ckey, (size_t)want * 8));
  if (!t->d_sfreq) {
    SWT_HIP(hipMalloc((void **)&t->d_gpos, 2 * (size_t)want * 8));
    SWT_HIP(hipMalloc((void **)&t->d_gnb_min, 4 * (size_t)want * 4));
    SWT_HIP(hipMalloc((void **)&t->d_gnb_max, 4 * (size_t)want * 4));
  }
  t->cand_cap = want;
  return SWT_OK;
}

int swt_bpe_trainer::replan() {
  int rc;
  if (!d_buckets) {
    SWT_HIP(hipMalloc((void **)&d_buckets, 512 * 8));
    SWT_HIP(hipMalloc((void **)&d_tied_idx, 2 * kTieSet * 4));
    SWT_HIP(hipMalloc((void **)&d_tied_key, 2 * kTieSet * 8));
    SWT_HIP(hipMalloc((void **)&d_tied_plan, 2 * kTieSet * sizeof(TiedPlan)));
  }
  if (!d_sfreq && (rc = cand_alloc(this, kCandCap))) return rc;
  if (!d_cidx || cidx_bits != T.bits) {  // one place per table slot
    if (d_cidx) (void)hipFree(d_cidx);
    d_cidx = nullptr;
    SWT_HIP(hipMalloc((void **)&d_cidx, ((size_t)1 << T.bits) * 4));
    cidx_bits = T.bits;
  }
  SWT_HIP(hipMemsetAsync(d_cidx, 0xFF, ((size_t)1 << T.bits) * 4, stream));
  const uint64_t cap = 1ull << T.bits;
  SWT_HIP(hipMemsetAsync(d_buckets, 0, 512 * 8, stream));
  hipLaunchKernelGGL(cand_hist_kernel, dim3(grid_for(cap, 256 * 8, 1024)), dim3(256), 0, stream, T.cnt, cap, d_buckets);
  unsigned long long hb[512];
  SWT_HIP(hipMemcpyAsync(hb, d_buckets, sizeof hb, hipMemcpyDeviceToHost, stream));
  if ((rc = sync_state())) return rc;
  if (d_sfreq) {
    // WordPiece: every live pair is listed (theta = 1), with room for the pairs the next merges create; a list that would
    // not fit 2^28 entries falls back to the full-table argmax (theta = 0)
    unsigned long long live = 0;
    for (int b = 0; b < 512; b++) live += hb[b];
    const uint64_t want = 2 * live + 65536;
    const unsigned int wflags = h_st.flags & ~kFlagReplan;
    SWT_HIP(hipMemsetAsync(&d_st->n_cand, 0, 8, stream));
    SWT_HIP(hipMemcpyAsync(&d_st->flags, &wflags, 4, hipMemcpyHostToDevice, stream));
    if (want > (1ull << 28)) {
      theta = 0;
      cand_valid = true;
      SWT_HIP(hipStreamSynchronize(stream));  // `wflags` is a stack variable
      return SWT_OK;
    }
    if (want > cand_cap || !d_cand) {
      if ((rc = cand_alloc(this, want + want / 2))) return rc;
    }
    theta = 1;
    cand_built = live;
    h_st.n_cand = live;  // (the host's copy is from before the build: the fused step is chosen by the list's length)
    since_replan = 0;
    cand_valid = true;
    hipLaunchKernelGGL(cand_build_kernel, dim3(grid_for(cap, 256, 2048)), dim3(256), 0, stream, ctx(), cap);
    SWT_HIP(hipMemcpyAsync(&d_st->n_synced, &d_st->n_cand, 8, hipMemcpyDeviceToDevice, stream));
    SWT_HIP(hipMemcpyAsync(&d_st->n_synced_next, &d_st->n_cand, 8, hipMemcpyDeviceToDevice, stream));
    SWT_HIP(hipStreamSynchronize(stream));
    n_replans++;
    return SWT_OK;
  }
  // from the top: whole buckets while at most kCandTarget pairs pass (the highest non-empty bucket always passes)
  unsigned long long above = 0;
  int lowest = 512;  // lowest bucket taken
  for (int b = 511; b >= 0; b--) {
    if (!hb[b]) continue;
    if (above && above + hb[b] > kCandTarget) break;
    above += hb[b];
    lowest = b;
  }
  unsigned long long th = 1;
  if (lowest < 512) {
    const int e = lowest / 8, sub = lowest % 8;  // bucket `lowest` starts at (8 + sub) * 2^(e - 3)
    th = e >= 3 ? ((unsigned long long)(8 + sub) << (e - 3)) : (((unsigned long long)(8 + sub) + ((1u << (3 - e)) - 1)) >> (3 - e));
    if (th < 1) th = 1;
  }
  bool any_lower = false;
  for (int b = 0; b < lowest && b < 512; b++) any_lower |= hb[b] != 0;
  if (!any_lower) th = 1;  // everything passes: the list is the whole table, and an empty list means no pair is left
  // a plateau wider than the list (say a million pairs of count 1): full-table argmax until the next re-plan
  theta = above > cand_cap / 2 ? 0 : th;
  cand_built = above;
  since_replan = 0;
  const unsigned int flags = h_st.flags & ~kFlagReplan;
  SWT_HIP(hipMemsetAsync(&d_st->n_cand, 0, 8, stream));
  SWT_HIP(hipMemcpyAsync(&d_st->flags, &flags, 4, hipMemcpyHostToDevice, stream));
  cand_valid = true;  // ctx() hands the mirror out from here on
  if (theta) hipLaunchKernelGGL(cand_build_kernel, dim3(grid_for(cap, 256, 2048)), dim3(256), 0, stream, ctx(), cap);
  // everything the build listed is mirrored
  SWT_HIP(hipMemcpyAsync(&d_st->n_synced, &d_st->n_cand, 8, hipMemcpyDeviceToDevice, stream));
  SWT_HIP(hipMemcpyAsync(&d_st->n_synced_next, &d_st->n_cand, 8, hipMemcpyDeviceToDevice, stream));
  SWT_HIP(hipStreamSynchronize(stream));  // `flags` is a stack variable
  n_replans++;
  return SWT_OK;
}

static int build_histogram(swt_bpe_trainer *t) {
  // size for the worst case first (every position a distinct pair), then shrink to what is used
  uint32_t bits = 10;
  while ((1ull << bits) < 2 * t->n_syms0 + 16 && bits < 31) bits++;
  int rc = table_alloc(t->T, bits, t->stream);
  if (rc) return rc;
  TrainCtx C = t->ctx();
  C.theta = 0;
  C.pend = nullptr;
  if (t->n_words)
    hipLaunchKernelGGL(hist_build_kernel, dim3(grid_for(t->n_words, kTrainThreads)), dim3(kTrainThreads), 0, t->stream, t->d_sym, t->d_woff,
                       t->d_freq, t->n_words, C);
  if ((rc = t->sync_state())) return rc;
  uint32_t want = 10;
  while ((1ull << want) < 4 * t->h_st.n_used + 1024) want++;
  if (want < bits) {
    if ((rc = table_resize(t, want))) return rc;
    if ((rc = t->sync_state())) return rc;
  }
  t->hist_ready = true;
  return SWT_OK;
}

// k0: the words of every initial pair, grouped by key; the log of the later segments; the word stamps
static int build_index(swt_bpe_trainer *t) {
  K0Index &K = t->K;
  K.bits = t->T.bits;
  const size_t cap = (size_t)1 << K.bits;
  SWT_HIP(hipMalloc((void **)&K.keys, cap * 8));
  SWT_HIP(hipMalloc((void **)&K.start, cap * 4));
  SWT_HIP(hipMalloc((void **)&K.len, cap * 4));
  SWT_HIP(hipMalloc((void **)&K.fill, cap * 4));
  SWT_HIP(hipMalloc((void **)&K.words, (size_t)(t->n_syms0 + 16) * 4));
  SWT_HIP(hipMemcpyAsync(K.keys, t->T.keys, cap * 8, hipMemcpyDeviceToDevice, t->stream));
  SWT_HIP(hipMemsetAsync(K.fill, 0, cap * 4, t->stream));
  SWT_HIP(hipMemsetAsync(K.start, 0, cap * 4, t->stream));
  SWT_HIP(hipMemsetAsync(K.len, 0, cap * 4, t->stream));
  unsigned long long *cursor = &t->d_st->scratch;
  SWT_HIP(hipMemsetAsync(cursor, 0, 8, t->stream));
  if (t->n_words) {
    const unsigned g = grid_for(t->n_words, kTrainThreads);
    hipLaunchKernelGGL(k0_pass_kernel, dim3(g), dim3(kTrainThreads), 0, t->stream, t->d_sym, t->d_woff, t->n_words, K, 0);
    hipLaunchKernelGGL(k0_alloc_kernel, dim3(grid_for(cap, 256, 2048)), dim3(256), 0, t->stream, K, cursor);
    hipLaunchKernelGGL(k0_pass_kernel, dim3(g), dim3(kTrainThreads), 0, t->stream, t->d_sym, t->d_woff, t->n_words, K, 1);
  }
  // a merge removes one symbol and adds at most two entries
  t->idx_cap = 2 * t->n_syms0 + 1024;
  SWT_HIP(hipMalloc((void **)&t->d_idx_tag, (size_t)t->idx_cap * 4));
  SWT_HIP(hipMalloc((void **)&t->d_idx_word, (size_t)t->idx_cap * 4));
  SWT_HIP(hipMalloc((void **)&t->d_wstamp, (size_t)(t->n_words + 1) * 4));
  SWT_HIP(hipMemsetAsync(t->d_wstamp, 0, (size_t)(t->n_words + 1) * 4, t->stream));
  SWT_HIP(hipMalloc((void **)&t->d_wkey, (size_t)(t->n_words + 1) * 8));
  return SWT_OK;
}

// seg_of[] (merged symbol -> the step that created it) and seg_start[] grow with the number of steps
static int ensure_steps(swt_bpe_trainer *t, uint64_t more_steps, uint32_t max_merged_id) {
  const uint64_t need_seg = (uint64_t)t->step_no + more_steps + 4;
  if (need_seg > t->seg_start_cap) {
    uint64_t cap = t->seg_start_cap ? t->seg_start_cap : 4096;
    while (cap < need_seg) cap *= 2;
    unsigned long long *p = nullptr;
    SWT_HIP(hipMalloc((void **)&p, cap * 8));
    SWT_HIP(hipMemsetAsync(p, 0, cap * 8, t->stream));
    if (t->d_seg_start) {
      SWT_HIP(hipMemcpyAsync(p, t->d_seg_start, t->seg_start_cap * 8, hipMemcpyDeviceToDevice, t->stream));
      SWT_HIP(hipStreamSynchronize(t->stream));
      (void)hipFree(t->d_seg_start);
    }
    t->d_seg_start = p;
    t->seg_start_cap = cap;
  }
  uint64_t need_ids = 1024;
  if (max_merged_id >= t->id_base && max_merged_id != 0xFFFFFFFFu) need_ids = (uint64_t)(max_merged_id - t->id_base) + 1;
  if (need_ids > (1ull << 26)) return fail(SWT_ERR_UNSUPPORTED, "merged symbol id %u is beyond the trainer's range", max_merged_id);
  if (need_ids > t->seg_cap) {
    uint64_t cap = t->seg_cap ? t->seg_cap : 65536;
    while (cap < need_ids) cap *= 2;
    uint32_t *p = nullptr;
    SWT_HIP(hipMalloc((void **)&p, cap * 4));
    SWT_HIP(hipMemsetAsync(p, 0, cap * 4, t->stream));
    if (t->d_seg_of) {
      SWT_HIP(hipMemcpyAsync(p, t->d_seg_of, (size_t)t->seg_cap * 4, hipMemcpyDeviceToDevice, t->stream));
      SWT_HIP(hipStreamSynchronize(t->stream));
      (void)hipFree(t->d_seg_of);
    }
    t->d_seg_of = p;
    t->seg_cap = (uint32_t)cap;
  }
  return SWT_OK;
}

static int alloc_common(swt_bpe_trainer *t) {
  SWT_HIP(hipMalloc((void **)&t->d_st, sizeof(TrainState)));
  SWT_HIP(hipMalloc((void **)&t->d_cmd, sizeof(StepCmd)));
  SWT_HIP(hipMemsetAsync(t->d_cmd, 0, sizeof(StepCmd), t->stream));
  SWT_HIP(hipMalloc((void **)&t->d_steplog, kMaxRunSteps * sizeof(StepLog)));
  SWT_HIP(hipMalloc((void **)&t->d_parts, kArgParts * sizeof(ArgPart)));
  SWT_HIP(hipMemsetAsync(t->d_parts, 0, kArgParts * sizeof(ArgPart), t->stream));
  TrainState init{};
  init.n_syms = t->n_syms0;
  init.best_pos = kEmptyKey;
  init.res_pos = kEmptyKey;
  init.best2[0] = init.best2[1] = kEmptyKey;
  init.plateau = ~0ull;
  SWT_HIP(hipMemcpyAsync(t->d_st, &init, sizeof init, hipMemcpyHostToDevice, t->stream));
  SWT_HIP(hipStreamSynchronize(t->stream));
  return SWT_OK;
}

// ids at or above id_base that are already in the stream are initial symbols: they have no birth step, and a merge may not
// take their id
static int mark_initial_ids(swt_bpe_trainer *t, const std::vector<uint32_t> &ids) {
  uint32_t hi = 0;
  for (uint32_t s : ids) if (s >= t->id_base && s > hi) hi = s;
  if (!hi) return SWT_OK;
  int rc = ensure_steps(t, 0, hi);
  if (rc) return rc;
  std::vector<uint32_t> seg(t->seg_cap, 0u);
  for (uint32_t s : ids) if (s >= t->id_base) seg[s - t->id_base] = kSegBase;
  SWT_HIP(hipMemcpy(t->d_seg_of, seg.data(), (size_t)t->seg_cap * 4, hipMemcpyHostToDevice));
  return SWT_OK;
}

static int finish_create(swt_bpe_trainer *t) {
  int rc;
  if ((rc = build_histogram(t))) return rc;
  if ((rc = build_index(t))) return rc;
  if ((rc = ensure_steps(t, kRunBatch, 0xFFFFFFFFu))) return rc;
  return t->sync_state();
}

// take ownership of a unique-word stream that is already on the device (swt_words.hip)
static int trainer_adopt(swt_bpe_trainer *t, DeviceWords &dw) {
  t->n_words = dw.n_words;
  t->n_syms0 = dw.n_syms;
  t->extent = dw.n_syms;
  t->d_sym = dw.d_sym;
  t->d_woff = dw.d_woff;
  t->d_freq = dw.d_freq;
  dw.d_sym = nullptr; dw.d_woff = nullptr; dw.d_freq = nullptr;
  if (t->n_words >= 0xFFFFFFFFull || t->n_syms0 >= 0xFFFFFFFFull) return fail(SWT_ERR_UNSUPPORTED, "more than 2^32 - 1 unique words or symbols");
  int rc = alloc_common(t);
  if (rc) return rc;
  t->base_syms = dw.base_syms;
  t->n_base = (uint32_t)t->base_syms.size();
  return SWT_OK;
}

static int trainer_upload(swt_bpe_trainer *t, const uint32_t *syms, const uint64_t *word_off, const uint32_t *freq, uint64_t n_words) {
  int rc = ensure_device();
  if (rc) return rc;
  const uint64_t n_syms = word_off[n_words];
  t->n_words = n_words;
  t->n_syms0 = n_syms;
  t->extent = n_syms;
  if (n_words >= 0xFFFFFFFFull || n_syms >= 0xFFFFFFFFull) return fail(SWT_ERR_UNSUPPORTED, "more than 2^32 - 1 unique words or symbols");
  for (uint64_t w = 0; w < n_words; w++) {
    if (word_off[w + 1] < word_off[w]) return fail(SWT_ERR_INVALID, "word offsets must be non-decreasing");
    if (word_off[w + 1] - word_off[w] > 0xFFFFFFFFull) return fail(SWT_ERR_UNSUPPORTED, "word too long");
  }
  for (uint64_t i = 0; i < n_syms; i++)
    if (syms[i] == kHole) return fail(SWT_ERR_INVALID, "symbol id 0xFFFFFFFF is reserved");
  SWT_HIP(hipMalloc((void **)&t->d_sym, (n_syms + 16) * 4));
  SWT_HIP(hipMalloc((void **)&t->d_woff, (n_words + 1) * 8));
  SWT_HIP(hipMalloc((void **)&t->d_freq, (n_words + 1) * 4));
  if ((rc = alloc_common(t))) return rc;
  if (n_syms) SWT_HIP(hipMemcpy(t->d_sym, syms, n_syms * 4, hipMemcpyHostToDevice));
  SWT_HIP(hipMemcpy(t->d_woff, word_off, (n_words + 1) * 8, hipMemcpyHostToDevice));
  if (n_words) SWT_HIP(hipMemcpy(t->d_freq, freq, n_words * 4, hipMemcpyHostToDevice));
  // distinct code points (the initial vocab, bpe.py:75)
  std::vector<uint8_t> seen(kNumCodePoints, 0);
  std::vector<uint32_t> other;  // ids that are not code points (a caller-supplied stream may already hold merged symbols)
  for (uint64_t i = 0; i < n_syms; i++) {
    if (syms[i] < kNumCodePoints) seen[syms[i]] = 1;
    else other.push_back(syms[i]);
  }
  std::vector<uint32_t> b;
  for (uint32_t c = 0; c < kNumCodePoints; c++) if (seen[c]) b.push_back(c);
  std::sort(other.begin(), other.end());
  other.erase(std::unique(other.begin(), other.end()), other.end());
  b.insert(b.end(), other.begin(), other.end());
  t->base_syms = b;
  t->n_base = (uint32_t)b.size();
  if ((rc = mark_initial_ids(t, other))) return rc;
  return finish_create(t);
}

// New pairs one merge can create: two per occurrence (occurrences <= the pair's count, counts never grow), and never
// more than (x, m) / (m, y) over the distinct symbols x, y plus (m, m).
static uint64_t new_pairs_bound(const swt_bpe_trainer *t, uint64_t count_bound) {
  if (t->d_sfreq) count_bound = 0;  // WordPiece: max_count holds a score, not a count
  const uint64_t n_base = t->sharded ? t->n_base_global : t->n_base;
  const uint64_t by_symbols = 2 * (n_base + t->n_applied + 1) + 1;
  uint64_t by_count = count_bound ? 2 * count_bound : by_symbols;
  if (!t->sharded && t->h_st.n_syms && 2 * t->h_st.n_syms < by_count) by_count = 2 * t->h_st.n_syms;
  return by_symbols < by_count ? by_symbols : by_count;
}

static int ensure_room(swt_bpe_trainer *t, uint64_t extra) {
  const uint64_t cap = 1ull << t->T.bits;
  if (2 * (t->h_st.n_used + extra + 64) <= cap) return SWT_OK;
  uint32_t bits = t->T.bits;
  while ((1ull << bits) < 4 * (t->h_st.n_used + extra + 64)) bits++;
  int rc = table_resize(t, bits);
  if (rc) return rc;
  return t->sync_state();
}

// argmax -> tie scan (the caller enqueues its decide kernel behind them)
void swt_bpe_trainer::enqueue_argmax() {
  const TrainCtx C = ctx();
  if (theta && d_sfreq) {
    n_parts = kArgParts;
    hipLaunchKernelGGL(wp_list_argmax_kernel, dim3(kArgParts), dim3(256), 0, stream, C, d_parts);
  } else if (theta) {
    n_parts = kCandBlocks;
    hipLaunchKernelGGL(cand_argmax_kernel, dim3(kCandBlocks), dim3(256), 0, stream, C, d_parts);
  } else {
    const uint64_t cap = 1ull << T.bits;
    n_parts = grid_for(cap, 256 * 8, kArgParts);
    hipLaunchKernelGGL(argmax_full_kernel, dim3(n_parts), dim3(256), 0, stream, T.keys, T.cnt, cap, d_parts, (const long long *)d_sfreq);
  }
  if (!n_words) return;
  if (theta && d_sfreq)  // WordPiece: the tied pairs' first positions through the index (stream scan inside, if the index is void)
    hipLaunchKernelGGL(wp_tie_index_kernel, dim3(64), dim3(kTrainThreads), 0, stream, d_sym, d_woff, n_words, C, d_parts, n_parts);
  else
    hipLaunchKernelGGL(tie_kernel, dim3(grid_for(n_words, kTrainThreads, kTieBlocks)), dim3(kTrainThreads), 0, stream, d_sym, d_woff,
                       n_words, C, d_parts, n_parts);
}

// unsharded BPE: tie scan + apply, each with its own workgroup-level argmax over the (short) candidate list
void swt_bpe_trainer::enqueue_fast_step(uint32_t first_merged, uint32_t limit) {
  const TrainCtx C = ctx();
  if (!n_words) return;
  // every workgroup of the tie launch reads the whole candidate list: few of them for a small corpus, kTieBlocks at most
  // a trip of the tie scan covers 64 words per workgroup (16 lanes a word, kTieWords words a wave)
  const unsigned tie_blocks = grid_for(n_words, 64, kTieBlocks);
  hipLaunchKernelGGL(fast_tie_kernel, dim3(tie_blocks + 2), dim3(kTrainThreads), 0, stream, d_sym, d_woff, n_words, C, limit);  // + housekeeper, planner
  hipLaunchKernelGGL(fast_apply_kernel, dim3(kFastApplyBlocks), dim3(kTrainThreads), 0, stream, d_sym, d_woff, d_freq, n_words, C,
                     d_steplog, first_merged, limit);
}

void swt_bpe_trainer::enqueue_apply() {
  if (n_words)
    hipLaunchKernelGGL(apply_kernel, dim3(kApplyBlocks), dim3(kTrainThreads), 0, stream, d_sym, d_woff, d_freq, n_words, ctx(), d_cmd);
}

#ifdef SWT_STAMPS
// diagnostic builds only (not in include/swt.h): read = 0 resets the stamps, 1 copies spans (4 * 16384) then phases (3 * 16), then g_why (32)
extern "C" int swt_debug_stamps(int read, unsigned long long *out) try {
  if (!read) {
    static unsigned long long init[4][kSpanSteps];
    for (uint32_t i = 0; i < kSpanSteps; i++) { init[0][i] = ~0ull; init[1][i] = 0; init[2][i] = ~0ull; init[3][i] = 0; }
    static unsigned long long zero[3][16];
    static unsigned int zero_r[kSpanSteps];
    static unsigned long long zero_w[32];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_why), zero_w, sizeof zero_w) != hipSuccess) return -4;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_kstep), zero_r, sizeof zero_r) != hipSuccess) return -4;
    { void *pm = nullptr; if (hipGetSymbolAddress(&pm, HIP_SYMBOL(g_pmax)) != hipSuccess || hipMemset(pm, 0, sizeof g_pmax) != hipSuccess) return -4; }
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_span), init, sizeof init) != hipSuccess) return -4;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_reported), zero_r, sizeof zero_r) != hipSuccess) return -4;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof zero) == hipSuccess ? 0 : -4;
  }
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(unsigned long long) * 4 * kSpanSteps) != hipSuccess) return -4;
  if (hipMemcpyFromSymbol(out + 4 * kSpanSteps, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 48) != hipSuccess) return -4;
  if (hipMemcpyFromSymbol(out + 4 * kSpanSteps + 48, HIP_SYMBOL(g_why), sizeof(unsigned long long) * 32) != hipSuccess) return -4;
  if (hipMemcpyFromSymbol(out + 4 * kSpanSteps + 80, HIP_SYMBOL(g_kstep), sizeof(unsigned int) * kSpanSteps) != hipSuccess) return -4;
  return hipMemcpyFromSymbol(out + 4 * kSpanSteps + 80 + kSpanSteps / 2, HIP_SYMBOL(g_pmax), sizeof g_pmax) == hipSuccess ? 0 : -4;
} SWT_API_CATCH
#endif

extern "C" {

int swt_bpe_train_create_words(const uint32_t *syms, const uint64_t *word_off, const uint32_t *freq, uint64_t n_words,
                               swt_bpe_trainer **out) try {
  if (!out || !word_off || (n_words && (!freq || (word_off[n_words] && !syms)))) return fail(SWT_ERR_INVALID, "null argument");
  if (word_off[0] != 0) return fail(SWT_ERR_INVALID, "word_off[0] must be 0");
  auto *t = new swt_bpe_trainer();
  int rc = trainer_upload(t, syms, word_off, freq, n_words);
  if (rc) { swt_bpe_train_destroy(t); return rc; }
  *out = t;
  return SWT_OK;
} SWT_API_CATCH

static int words_from_text(const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent, swt_bpe_trainer **out) {
  if (!out || !sent_off || (n_sent && sent_off[n_sent] && !text)) return fail(SWT_ERR_INVALID, "null argument");
  if (sent_off[0] != 0) return fail(SWT_ERR_INVALID, "sent_off[0] must be 0");
  for (uint64_t s = 0; s < n_sent; s++)
    if (sent_off[s] > sent_off[s + 1]) return fail(SWT_ERR_INVALID, "sentence offsets must be non-decreasing");
  int rc = ensure_device();
  if (rc) return rc;
  const uint64_t n_bytes = sent_off[n_sent];
  DevBuf d_text, d_off;
  struct Guard { DevBuf &a, &b; ~Guard() { a.release(); b.release(); } } guard{d_text, d_off};
  if ((rc = d_text.reserve(n_bytes + 64)) || (rc = d_off.reserve((n_sent + 1) * 8))) return rc;
  if (n_bytes) SWT_HIP(hipMemcpy(d_text.p, text, n_bytes, hipMemcpyHostToDevice));
  SWT_HIP(hipMemcpy(d_off.p, sent_off, (n_sent + 1) * 8, hipMemcpyHostToDevice));
  DeviceWords dw;
  rc = device_words_from_text(d_text.as<uint8_t>(), n_bytes, d_off.as<uint64_t>(), n_sent, &dw);
  if (rc) return rc;
  auto *t = new swt_bpe_trainer();
  rc = trainer_adopt(t, dw);
  if (rc) { swt_bpe_train_destroy(t); return rc; }
  *out = t;
  return SWT_OK;
}

static int trainer_from_joined(const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint8_t *need_host, bool wordpiece,
                               swt_bpe_trainer **out);

// The same from the texts joined with U+0000 (swt_utf8_prepare_joined's input): offsets, lowercase and the word census without
// the prepared text travelling to the host and back.  need_host[s] = 1: sentence s holds a code point only the host lowercases;
// then *out stays NULL and the caller takes the host-array way.
int swt_bpe_train_create_joined(const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint8_t *need_host, swt_bpe_trainer **out) try {
  if (!out || (n_sent && !need_host) || (n_joined && !joined)) return fail(SWT_ERR_INVALID, "null argument");
  *out = nullptr;
  return trainer_from_joined(joined, n_joined, n_sent, need_host, false, out);
} SWT_API_CATCH

// bpe.py:70-81 on the device (swt_words.hip): split (utils.py:27), Counter(words) in first-occurrence order, symbolise.
int swt_bpe_train_create_text(const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent, swt_bpe_trainer **out) try {
  swt_bpe_trainer *t = nullptr;
  int rc = words_from_text(text, sent_off, n_sent, &t);
  if (rc) return rc;
  if ((rc = finish_create(t))) { swt_bpe_train_destroy(t); return rc; }
  *out = t;
  return SWT_OK;
} SWT_API_CATCH

// wordpiece.py:44-63 on the device: the same split and Counter, then [word[0]] + ["##" + c ...] and the symbol frequencies.
// The handle is used with the swt_bpe_train_* calls; `count` outputs carry the winning score's bit pattern.
// everything after the word census: '##' symbols, histogram and index, symbol frequencies, the initial vocabulary.  Owns t.
static int wp_finish_create(swt_bpe_trainer *t, swt_bpe_trainer **out) {
  int rc;
  auto bail = [&](int code) { swt_bpe_train_destroy(t); return code; };
  t->id_base = kWpMergedBase;
  if (t->n_words)
    hipLaunchKernelGGL(wp_symbolise_kernel, dim3(grid_for(t->n_words, kTrainThreads)), dim3(kTrainThreads), 0, t->stream, t->d_sym,
                       t->d_woff, t->n_words);
  if ((rc = finish_create(t))) return bail(rc);  // the histogram build counts plain pairs: the frequencies come after it
  if (hipMalloc((void **)&t->d_sfreq, kWpSymCap * 8) != hipSuccess) return bail(fail(SWT_ERR_HIP, "hipMalloc of the symbol frequencies failed"));
  if (hipMemsetAsync(t->d_sfreq, 0, kWpSymCap * 8, t->stream) != hipSuccess) return bail(fail(SWT_ERR_HIP, "hipMemset failed"));
  DevBuf live, flag;
  struct Guard { DevBuf &a, &b; ~Guard() { a.release(); b.release(); } } guard{live, flag};
  const uint32_t live_cap = 2 * kWpCont;
  if ((rc = live.reserve((size_t)live_cap * 4)) || (rc = flag.reserve(8))) return bail(rc);
  unsigned int *d_flag = flag.as<unsigned int>();
  if (hipMemsetAsync(d_flag, 0, 8, t->stream) != hipSuccess) return bail(fail(SWT_ERR_HIP, "hipMemset failed"));
  if (t->n_words)
    hipLaunchKernelGGL(sym_hist_kernel, dim3(grid_for(t->n_words, kTrainThreads)), dim3(kTrainThreads), 0, t->stream, t->d_sym, t->d_woff,
                       t->d_freq, t->n_words, t->d_sfreq, kWpSymCap, d_flag);
  // the initial vocabulary (wordpiece.py:62-63) = the symbols that occur
  hipLaunchKernelGGL(wp_live_symbols_kernel, dim3(1024), dim3(256), 0, t->stream, (const long long *)t->d_sfreq, (uint64_t)kWpMergedBase,
                     live.as<uint32_t>(), live_cap, d_flag + 1);
  unsigned int h[2] = {0, 0};
  if (hipMemcpyAsync(h, d_flag, 8, hipMemcpyDeviceToHost, t->stream) != hipSuccess || hipStreamSynchronize(t->stream) != hipSuccess)
    return bail(fail(SWT_ERR_HIP, "reading the symbol census failed"));
  if (h[0]) return bail(fail(SWT_ERR_UNSUPPORTED, "symbol id out of range for a WordPiece trainer"));
  t->base_syms.resize(h[1]);
  if (h[1] && hipMemcpy(t->base_syms.data(), live.p, (size_t)h[1] * 4, hipMemcpyDeviceToHost) != hipSuccess)
    return bail(fail(SWT_ERR_HIP, "reading the initial symbols failed"));
  std::sort(t->base_syms.begin(), t->base_syms.end());
  t->n_base = h[1];
  *out = t;
  return SWT_OK;
}

// Both trainers from the joined texts.  The prepared text lives in the calling thread's prepare workspace only while the word
// census reads it (with_prepared_joined: the workspace's guard releases it on every error path and above its keep limit --
// a 1 GiB corpus does not stay pinned in the thread); the trainer owns copies of nothing but the census' own arrays.
// A handle leaves this function only after finish_create (histogram + index): swt_bpe_trainer::ready() refuses any other.
static int trainer_from_joined(const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint8_t *need_host, bool wordpiece,
                               swt_bpe_trainer **out) {
  struct Ctx { uint64_t n_sent; DeviceWords dw; } c{n_sent, {}};
  bool consumed = false;
  int rc = with_prepared_joined(joined, n_joined, n_sent, need_host, &consumed,
      [](void *p, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_off) {
        Ctx *c = static_cast<Ctx *>(p);
        return device_words_from_text(d_text, n_bytes, d_off, c->n_sent, &c->dw);
      }, &c);
  auto drop = [&]() { for (void *q : {(void *)c.dw.d_sym, (void *)c.dw.d_woff, (void *)c.dw.d_freq}) if (q) (void)hipFree(q); };
  if (rc) { drop(); return rc; }
  if (!consumed) return SWT_OK;  // a sentence needs the host's str.lower(): *out stays NULL
  auto *t = new swt_bpe_trainer();
  if ((rc = trainer_adopt(t, c.dw))) { drop(); swt_bpe_train_destroy(t); return rc; }
  if (wordpiece) return wp_finish_create(t, out);
  if ((rc = finish_create(t))) { swt_bpe_train_destroy(t); return rc; }  // histogram, index
  *out = t;
  return SWT_OK;
}

int swt_wp_train_create_text(const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent, swt_bpe_trainer **out) try {
  swt_bpe_trainer *t = nullptr;
  int rc = words_from_text(text, sent_off, n_sent, &t);
  if (rc) return rc;
  return wp_finish_create(t, out);
} SWT_API_CATCH

// the WordPiece trainer from the joined texts: see swt_bpe_train_create_joined
int swt_wp_train_create_joined(const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint8_t *need_host, swt_bpe_trainer **out) try {
  if (!out || (n_sent && !need_host) || (n_joined && !joined)) return fail(SWT_ERR_INVALID, "null argument");
  *out = nullptr;
  return trainer_from_joined(joined, n_joined, n_sent, need_host, true, out);
} SWT_API_CATCH

void swt_bpe_train_destroy(swt_bpe_trainer *t) try {
  if (!t) return;
  (void)hipStreamSynchronize(t->stream);
  for (void *p : {(void *)t->d_sym, (void *)t->d_woff, (void *)t->d_freq, (void *)t->d_st, (void *)t->d_parts, (void *)t->d_cmd,
                  (void *)t->d_steplog, (void *)t->d_sfreq, (void *)t->d_cand, (void *)t->d_ccnt, (void *)t->d_ckey, (void *)t->d_cidx, (void *)t->d_buckets, (void *)t->d_idx_tag,
                  (void *)t->d_idx_word, (void *)t->d_wstamp, (void *)t->d_wkey, (void *)t->d_tied_idx, (void *)t->d_tied_key, (void *)t->d_tied_plan, (void *)t->d_gpos, (void *)t->d_gnb_min, (void *)t->d_gnb_max, (void *)t->d_sym_alt, (void *)t->d_woff_alt, (void *)t->d_seg_start, (void *)t->d_seg_of, (void *)t->d_pend,
                  (void *)t->d_tstamp, (void *)t->d_touched, (void *)t->d_block, (void *)t->d_blocks_all, (void *)t->d_tie_line,
                  (void *)t->d_tie_all, (void *)t->d_halt, (void *)t->d_tie_msg, (void *)t->d_tie_msgs, (void *)t->K.keys, (void *)t->K.start, (void *)t->K.len, (void *)t->K.fill,
                  (void *)t->K.words})
    if (p) (void)hipFree(p);
  table_free(t->T);
  t->tmp.release();
  delete t;
} SWT_API_CATCH_VOID

int swt_bpe_train_set_pos_base(swt_bpe_trainer *t, uint64_t pos_base) try {
  if (!t) return fail(SWT_ERR_INVALID, "null trainer");
  t->pos_base = pos_base;
  return SWT_OK;
} SWT_API_CATCH

int swt_bpe_train_info(const swt_bpe_trainer *t, uint64_t *n_words, uint64_t *n_symbols, uint32_t *n_base_symbols, uint64_t *n_pairs) try {
  if (!t) return fail(SWT_ERR_INVALID, "null trainer");
  if (!t->d_st) return fail(SWT_ERR_STATE, "the trainer has no device state");
  TrainState r;
  SWT_HIP(hipMemcpy(&r, t->d_st, sizeof r, hipMemcpyDeviceToHost));
  if (n_words) *n_words = t->n_words;
  if (n_symbols) *n_symbols = r.n_syms;
  if (n_base_symbols) *n_base_symbols = t->n_base;
  if (n_pairs) *n_pairs = r.n_used;
  return SWT_OK;
} SWT_API_CATCH

int swt_bpe_train_stats(const swt_bpe_trainer *t, uint64_t *out, uint32_t n) try {
  if (!t || !out) return fail(SWT_ERR_INVALID, "null argument");
  if (!t->d_st || !t->T.bits) return fail(SWT_ERR_STATE, "the trainer has no device state");
  TrainState r;
  SWT_HIP(hipMemcpy(&r, t->d_st, sizeof r, hipMemcpyDeviceToHost));
  const uint64_t v[10] = {t->n_replans, t->theta, r.n_cand, r.idx_cursor, (uint64_t)(1ull << t->T.bits), r.flags, t->step_no, r.n_used,
                          r.ent_scanned, r.tie_words};
  for (uint32_t i = 0; i < n && i < 10; i++) out[i] = v[i];
  return SWT_OK;
} SWT_API_CATCH

int swt_bpe_train_trace(const swt_bpe_trainer *t, uint64_t *rows, uint64_t cap_rows, uint64_t *n_rows) try {
  if (!t || !n_rows) return fail(SWT_ERR_INVALID, "null argument");
  *n_rows = t->trace.size();
  for (uint64_t i = 0; rows && i < t->trace.size() && i < cap_rows; i++) {
    rows[4 * i] = t->trace[i].count;
    rows[4 * i + 1] = t->trace[i].n_tied;
    rows[4 * i + 2] = t->trace[i].n_cand;
    rows[4 * i + 3] = t->trace[i].n_syms;
  }
  return SWT_OK;
} SWT_API_CATCH

int swt_bpe_train_base_symbols(const swt_bpe_trainer *t, uint32_t *out, uint32_t cap) try {
  if (!t) return fail(SWT_ERR_INVALID, "null trainer");
  if (cap < t->n_base) return fail(SWT_ERR_CAPACITY, "need room for %u symbols", t->n_base);
  std::copy(t->base_syms.begin(), t->base_syms.end(), out);
  return SWT_OK;
} SWT_API_CATCH

int swt_bpe_train_best(swt_bpe_trainer *t, uint32_t *left, uint32_t *right, uint64_t *count, uint64_t *n_tied,
                       uint64_t *first_pos) try {
  if (!t || !left || !right || !count) return fail(SWT_ERR_INVALID, "null argument");
  int rc = ensure_device();
  if (rc) return rc;
  if ((rc = t->ready())) return rc;
  for (int attempt = 0;; attempt++) {
    if (!t->cand_valid && (rc = t->replan())) return rc;
    t->enqueue_argmax();
    hipLaunchKernelGGL(decide_kernel, dim3(1), dim3(64), 0, t->stream, t->d_sym, t->d_woff, t->ctx(), t->d_parts, t->n_parts,
                       (StepCmd *)nullptr, (StepLog *)nullptr, 0u, 0u);
    if ((rc = t->sync_state()) || (rc = t->check_state())) return rc;
    if (!(t->h_st.flags & kFlagReplan)) break;
    if (attempt > 64) return fail(SWT_ERR_STATE, "the candidate list cannot be rebuilt");
    t->cand_valid = false;  // the list ran dry or overflowed: new theta, again
  }
  const TrainState &r = t->h_st;
  *count = r.max_count;
  if (n_tied) *n_tied = r.n_tied;
  if (r.max_count == 0) { *left = *right = 0; if (first_pos) *first_pos = kEmptyKey; return SWT_OK; }
  const unsigned long long key = r.best_key;
  unsigned long long pos = kEmptyKey;
  if (r.n_tied >= 2 && r.res_pos != kEmptyKey) {
    // the caller compares positions across shards: stream address of the pair's left symbol, after pos_base
    uint64_t wo = 0;
    SWT_HIP(hipMemcpy(&wo, t->d_woff + (r.res_pos >> 32), 8, hipMemcpyDeviceToHost));
    pos = t->pos_base + wo + (uint32_t)r.res_pos;
  }
  *left = (uint32_t)(key >> 32);
  *right = (uint32_t)key;
  if (first_pos) *first_pos = pos;
  return SWT_OK;
} SWT_API_CATCH

int swt_bpe_train_apply(swt_bpe_trainer *t, uint32_t left, uint32_t right, uint32_t merged) try {
  if (!t) return fail(SWT_ERR_INVALID, "null trainer");
  int rc = ensure_device();
  if (rc) return rc;
  if (t->sharded) return fail(SWT_ERR_STATE, "a sharded trainer is stepped by swt_bpe_train_run_sharded");
  if ((rc = t->ready())) return rc;
  if (merged == kHole) return fail(SWT_ERR_INVALID, "symbol id 0xFFFFFFFF is reserved");
  if (t->d_sfreq && (left >= kWpSymCap || right >= kWpSymCap || merged >= kWpSymCap))
    return fail(SWT_ERR_UNSUPPORTED, "WordPiece symbol id beyond %llu", (unsigned long long)kWpSymCap);
  // keep the load factor below 1/2 whatever this merge creates
  const uint64_t occ = new_pairs_bound(t, t->h_st.max_count);
  if ((rc = ensure_room(t, occ))) return rc;
  if ((rc = ensure_steps(t, 2, merged))) return rc;
  if (!t->cand_valid && (rc = t->replan())) return rc;
  t->step_no++;
  hipLaunchKernelGGL(set_cmd_kernel, dim3(1), dim3(1), 0, t->stream, t->ctx(), t->d_cmd, left, right, merged);
  t->enqueue_apply();
  SWT_HIP(hipGetLastError());
  // n_used may have grown; the next best() refreshes h_st.  Be conservative until then.
  t->h_st.n_used += occ;
  t->n_applied++;
  return SWT_OK;
} SWT_API_CATCH

// Up to max_steps iterations of {argmax, tie-break, decide, apply} enqueued back to back: the pair of step i stays on the
// device (decide_kernel -> apply_kernel), only the log comes back.  Step i merges into symbol first_merged + i.
int swt_bpe_train_run(swt_bpe_trainer *t, uint32_t max_steps, uint32_t first_merged, uint32_t *left, uint32_t *right,
                      uint64_t *count, uint32_t *n_done) try {
  if (!t || !left || !right || !count || !n_done) return fail(SWT_ERR_INVALID, "null argument");
  if (t->sharded) return fail(SWT_ERR_STATE, "swt_bpe_train_run is for unsharded training (see swt_bpe_train_run_sharded)");
  if (t->d_sfreq && (uint64_t)first_merged + max_steps > kWpSymCap)
    return fail(SWT_ERR_UNSUPPORTED, "WordPiece symbol id beyond %llu", (unsigned long long)kWpSymCap);
  if ((uint64_t)first_merged + max_steps >= 0xFFFFFFFFull) return fail(SWT_ERR_UNSUPPORTED, "merged symbol ids would reach the reserved id");
  int rc = ensure_device();
  if (rc) return rc;
  *n_done = 0;
  if ((rc = t->ready())) return rc;
  std::vector<StepLog> hlog(kMaxRunSteps);
  uint32_t done = 0;
  bool exhausted = false;
  int dry_runs = 0;
  double per_step = 1.0;  // merges a step of the fast path has carried lately: sizes the next round trip
  while (done < max_steps && !exhausted) {
    const uint32_t remaining = max_steps - done;
    // One round trip: `steps` launch sequences that may log up to `cap` merges.  A step of the fast path carries one merge or
    // several (fast_apply_kernel); every other path carries exactly one.
    const bool maybe_fast = !t->d_sfreq && t->n_words;
    uint32_t steps = remaining < kRunBatch ? remaining : kRunBatch;
    uint32_t cap = steps;
    if (maybe_fast) {
      const double want = (double)remaining / per_step * 1.05 + 1.0;
      if (want < (double)steps) steps = (uint32_t)want;
      if (steps < 8) steps = remaining < 8 ? remaining : 8;
      double c = (double)steps * per_step * 1.5 + 8.0;
      if (c > (double)remaining) c = (double)remaining;
      if (c > (double)kMaxRunSteps) c = (double)kMaxRunSteps;
      cap = (uint32_t)c;
      if (cap < steps) cap = steps;
    }
    // room for everything the round trip may create (the symbol count grows by one per merge, counts never grow)
    // New pairs per merge: two per occurrence (occurrences <= the pair's count, and counts never grow: the maximum at the start
    // of the trip bounds every merge of it), and never more than (x, m) / (m, y) over the distinct symbols plus (m, m) -- the
    // symbols there are when the LAST merge of the trip happens, not now.  (Until round 3 the second bound was taken with the
    // symbol count at the start of the trip: a trip of many merges over a tiny alphabet outgrew it, and the table passed load
    // 1/2 -- silently, until check_state() looked: tools/gpu_soak.py seed 12787955, profiles/r03h_soak.txt.)
    const uint64_t by_count = (!t->d_sfreq && t->h_st.max_count) ? 2 * t->h_st.max_count : ~0ull;  // (WordPiece: max_count is a score)
    uint64_t extra = 0;
    for (uint32_t k2 = 0; k2 < cap; k2++) {  // merge k2 of the trip sees n_base + n_applied + k2 symbols
      const uint64_t by_sym = 2 * (t->n_base + t->n_applied + k2 + 1) + 1;
      extra += by_sym < by_count ? by_sym : by_count;
    }
    // (every new pair costs the stream a symbol: a round trip cannot make more than two per live symbol)
    if (!t->sharded && t->h_st.n_syms && 2 * t->h_st.n_syms + 64 < extra) extra = 2 * t->h_st.n_syms + 64;
    if ((rc = ensure_room(t, extra))) return rc;
    if ((rc = ensure_steps(t, steps, first_merged + done + cap))) return rc;
    // BPE re-plans once pushes have grown the (short) list past kCandHigh; WordPiece lists every live pair and re-plans when
    // three quarters of its room are used (dead entries are dropped on the way; a list that overflows in mid trip says so
    // on the device -- cand_dry -- and the host comes back here)
    const bool list_full = t->d_sfreq ? (t->theta && 4 * t->h_st.n_cand > 3 * t->cand_cap) : t->h_st.n_cand > kCandHigh;
    if (!t->cand_valid || (!t->theta && !t->d_sfreq) || list_full) {
      // a host stop anyway: when three slots in ten are holes, the stream is rewritten without them
      if (maybe_fast && t->h_st.n_syms && t->h_st.n_syms * 10 < t->extent * 7 && (rc = squeeze_stream(t))) return rc;
      if ((rc = t->replan())) return rc;
    }
    const bool fast = t->theta && maybe_fast;
    if (fast && t->theta > 1 && t->cand_built) {
      // A list that runs dry in mid trip turns the rest of the trip into steps that do nothing (two launches each).  A list of
      // n pairs has been good for about dry_ratio * n merges: the trip ends there, and the next one starts with a new threshold.
      const double budget = t->dry_ratio * (double)t->cand_built - (double)t->since_replan;
      if (budget < 24.0 && (double)remaining > budget && t->since_replan) {
        if (t->dry_ratio < 2.0) t->dry_ratio *= 1.1;  // it never ran dry: the list may be good for more than was thought
        t->cand_valid = false;
        continue;  // re-plan now (a cheaper stop than a dry trip)
      }
      const double most = budget / per_step + 1.0;
      if (most < (double)steps) {
        steps = most < 8.0 ? 8u : (uint32_t)most;
        if (steps > remaining) steps = remaining;
        if (cap > steps * kMaxBatch) cap = steps * kMaxBatch;
      }
    }
    if (!fast) {
      if (steps > remaining) steps = remaining;
      cap = steps;
    } else {
      // the step counters of the fast path start from zero, and no position of an earlier round trip is left
      SWT_HIP(hipMemsetAsync(&t->d_st->run_done[0], 0, 8 * 8, t->stream));
      SWT_HIP(hipMemsetAsync(&t->d_st->best2[0], 0xFF, 2 * 8, t->stream));  // steps that did nothing may have left either parity behind
      SWT_HIP(hipMemsetAsync(t->d_gpos, 0xFF, 2 * (size_t)t->cand_cap * 8, t->stream));
      SWT_HIP(hipMemsetAsync(t->d_gnb_min, 0xFF, 4 * (size_t)t->cand_cap * 4, t->stream));
      SWT_HIP(hipMemsetAsync(t->d_gnb_max, 0, 4 * (size_t)t->cand_cap * 4, t->stream));
    }
    // (the list may grow past the limit within the trip: the kernel's loops are strided, only slower then)
    const bool wp_fused = t->d_sfreq && t->theta && t->n_words && t->h_st.n_cand <= kWpStepList && !getenv("SWT_WP_GENERIC");
    prof_begin(t->stream);  // one bracket around the whole batch of merge steps: bench.py divides by the merges done
    const double enq0 = getenv("SWT_TRAIN_DEBUG") ? host_now() : 0.0;
    for (uint32_t i = 0; i < steps; i++) {
      t->step_no++;
      if (fast) {
        t->enqueue_fast_step(first_merged + done, cap);
        continue;
      }
      if (wp_fused) {  // WordPiece, a short list: argmax + tie-break + decision in one launch
        hipLaunchKernelGGL(wp_step_kernel, dim3(kWpStepBlocks), dim3(kTrainThreads), 0, t->stream, t->d_sym, t->d_woff, t->n_words, t->ctx(),
                           t->d_cmd, t->d_steplog, i, first_merged + done + i);
        hipLaunchKernelGGL(apply_kernel, dim3(kFastApplyBlocks), dim3(kTrainThreads), 0, t->stream, t->d_sym, t->d_woff, t->d_freq, t->n_words,
                           t->ctx(), t->d_cmd);
        continue;
      }
      t->enqueue_argmax();
      hipLaunchKernelGGL(decide_kernel, dim3(1), dim3(64), 0, t->stream, t->d_sym, t->d_woff, t->ctx(), t->d_parts, t->n_parts, t->d_cmd,
                         t->d_steplog, i, first_merged + done + i);
      t->enqueue_apply();
    }
    prof_end(t->stream);
    const double enq1 = getenv("SWT_TRAIN_DEBUG") ? host_now() : 0.0;
    SWT_HIP(hipGetLastError());
    if (cap > kMaxRunSteps) return fail(SWT_ERR_STATE, "a round trip was sized beyond the step log (%u rows)", cap);
    SWT_HIP(hipMemcpyAsync(hlog.data(), t->d_steplog, cap * sizeof(StepLog), hipMemcpyDeviceToHost, t->stream));
    if ((rc = t->sync_state()) || (rc = t->check_state())) return rc;
    uint32_t good = 0;
    unsigned long long stop = 0;  // why the device stopped before `cap`: 0 (it did not), 2 no pair left, 3 re-plan
    if (fast) {
      const unsigned long long logged = t->h_st.run_done[(t->step_no + 1) & 1u];
      if (logged > cap) return fail(SWT_ERR_STATE, "the step log overran its round trip");
      good = (uint32_t)logged;
      stop = t->h_st.halt;
      for (uint32_t i = 0; i < good; i++)
        if (hlog[i].flag != 0) return fail(SWT_ERR_STATE, "the step log has a hole");
    } else {
      while (good < cap && hlog[good].flag == 0) good++;
      if (good < cap) stop = hlog[good].flag;
    }
    for (uint32_t i = 0; i < good; i++) {
      left[done] = hlog[i].l;
      right[done] = hlog[i].r;
      count[done] = hlog[i].count;
      t->trace.push_back(hlog[i]);
      done++;
    }
    if (getenv("SWT_TRAIN_DEBUG"))
      fprintf(stderr, "trip: steps %u (host enqueue %.1f us) cap %u -> %u merges, stop %llu, per_step %.2f, listed %llu since %llu ratio %.2f, max %llu theta %llu\n", steps,
              (enq1 - enq0) * 1e6, cap, good, stop, per_step, (unsigned long long)t->cand_built, (unsigned long long)t->since_replan, t->dry_ratio,
              (unsigned long long)(good ? hlog[good - 1].count : 0), (unsigned long long)t->theta);
    t->n_applied += good;
    t->since_replan += good;
    if (good && !t->d_sfreq) t->h_st.max_count = hlog[good - 1].count;  // counts never grow: bound for the next batch
    if (stop == 3) {  // the candidate list ran dry: later steps of the batch were no-ops; new theta, go on
      if (fast && t->theta > 1 && t->cand_built && t->since_replan) {
        t->dry_ratio = 0.9 * (double)t->since_replan / (double)t->cand_built;
        t->dry_ratio = t->dry_ratio < 0.3 ? 0.3 : (t->dry_ratio > 2.0 ? 2.0 : t->dry_ratio);
      }
      t->cand_valid = false;
      if (!good && ++dry_runs > 64) return fail(SWT_ERR_STATE, "the candidate list cannot be rebuilt");
    } else if (stop) {
      exhausted = true;  // bpe.py:98-99: no pair left
    }
    if (fast && good && t->h_st.run_active) {
      // what a working step carried (the steps after a dry point or a full log do nothing and do not count)
      per_step = (double)good / (double)t->h_st.run_active;
      if (per_step < 1.0) per_step = 1.0;
      if (per_step > (double)kMaxBatch) per_step = (double)kMaxBatch;
    }
    if (good) dry_runs = 0;
  }
  *n_done = done;
  return SWT_OK;
} SWT_API_CATCH

int swt_bpe_train_export(swt_bpe_trainer *t, uint32_t *syms, uint64_t syms_cap, uint64_t *word_off, uint32_t *freq) try {
  if (!t || !word_off) return fail(SWT_ERR_INVALID, "null argument");
  if (!t->d_st || (t->n_words && (!t->d_sym || !t->d_woff))) return fail(SWT_ERR_STATE, "the trainer has no symbol stream");
  SWT_HIP(hipStreamSynchronize(t->stream));
  std::vector<uint32_t> all(t->extent + 1);
  std::vector<uint64_t> woff(t->n_words + 1);
  SWT_HIP(hipMemcpy(woff.data(), t->d_woff, (t->n_words + 1) * 8, hipMemcpyDeviceToHost));
  if (t->extent) SWT_HIP(hipMemcpy(all.data(), t->d_sym, t->extent * 4, hipMemcpyDeviceToHost));
  uint64_t o = 0;
  for (uint64_t w = 0; w < t->n_words; w++) {
    word_off[w] = o;
    for (uint64_t i = woff[w]; i < woff[w + 1]; i++) {
      if (all[i] == kHole) continue;
      if (o >= syms_cap) return fail(SWT_ERR_CAPACITY, "syms buffer too small");
      syms[o++] = all[i];
    }
  }
  word_off[t->n_words] = o;
  if (freq && t->n_words) SWT_HIP(hipMemcpy(freq, t->d_freq, t->n_words * 4, hipMemcpyDeviceToHost));
  return SWT_OK;
} SWT_API_CATCH

int swt_bpe_train_histogram(swt_bpe_trainer *t, uint64_t *keys, uint64_t *counts, uint64_t cap, uint64_t *n) try {
  if (!t || !n) return fail(SWT_ERR_INVALID, "null argument");
  int rc;
  if ((rc = t->ready())) return rc;
  if ((rc = t->tmp.reserve(cap * 16 + 32))) return rc;
  unsigned long long *d_n = t->tmp.as<unsigned long long>();
  DeltaRec *d_r = reinterpret_cast<DeltaRec *>(d_n + 2);
  SWT_HIP(hipMemsetAsync(d_n, 0, 8, t->stream));
  const uint64_t tcap = 1ull << t->T.bits;
  hipLaunchKernelGGL(table_export_kernel, dim3(grid_for(tcap, 256, 4096)), dim3(256), 0, t->stream, t->T.keys, t->T.cnt, tcap, d_r, cap, d_n);
  unsigned long long got = 0;
  SWT_HIP(hipMemcpyAsync(&got, d_n, 8, hipMemcpyDeviceToHost, t->stream));
  SWT_HIP(hipStreamSynchronize(t->stream));
  *n = got;
  if (got > cap) return fail(SWT_ERR_CAPACITY, "histogram holds %llu live pairs", got);
  if (got) {
    std::vector<DeltaRec> h(got);
    SWT_HIP(hipMemcpy(h.data(), d_r, got * sizeof(DeltaRec), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < got; i++) { keys[i] = h[i].key; counts[i] = (uint64_t)h[i].delta; }
  }
  return SWT_OK;
} SWT_API_CATCH

}  // extern "C"

// ---- sharded training: the pieces swt_dist.hip drives --------------------------------------------------------------------

namespace swt {

int trainer_set_block_cap(swt_bpe_trainer *t, uint64_t block_cap) {
  if (block_cap < 64) block_cap = 64;
  if (block_cap <= t->block_cap) return SWT_OK;
  SWT_HIP(hipStreamSynchronize(t->stream));
  for (void *p : {(void *)t->d_touched, (void *)t->d_block, (void *)t->d_blocks_all}) if (p) (void)hipFree(p);
  t->d_touched = nullptr; t->d_block = nullptr; t->d_blocks_all = nullptr;
  t->touched_cap = block_cap - 1;
  t->block_cap = block_cap;
  SWT_HIP(hipMalloc((void **)&t->d_touched, (size_t)t->touched_cap * 4));
  SWT_HIP(hipMalloc((void **)&t->d_block, (size_t)block_cap * sizeof(DeltaRec)));
  SWT_HIP(hipMalloc((void **)&t->d_blocks_all, (size_t)block_cap * sizeof(DeltaRec) * t->world));
  // whatever was pending is listed again from pend[] itself
  SWT_HIP(hipMemsetAsync(&t->d_st->n_touched, 0, 8, t->stream));
  const uint64_t cap = 1ull << t->T.bits;
  hipLaunchKernelGGL(rebuild_touched_kernel, dim3(grid_for(cap, 256, 2048)), dim3(256), 0, t->stream, t->ctx(), cap);
  SWT_HIP(hipGetLastError());
  return SWT_OK;
}

int trainer_enter_sharded(swt_bpe_trainer *t, uint32_t world, uint64_t block_cap) {
  if (t->d_sfreq) return fail(SWT_ERR_UNSUPPORTED, "sharded training is built for BPE");
  if (t->sharded) return fail(SWT_ERR_STATE, "the trainer is already sharded");
  t->sharded = true;
  t->world = world;
  int rc = sharded_arrays(t);
  if (rc) return rc;
  SWT_HIP(hipMalloc((void **)&t->d_halt, 8));
  SWT_HIP(hipMemsetAsync(t->d_halt, 0, 8, t->stream));
  SWT_HIP(hipMalloc((void **)&t->d_tie_line, 16));
  SWT_HIP(hipMalloc((void **)&t->d_tie_all, (size_t)world * 16));
  SWT_HIP(hipMalloc((void **)&t->d_tie_msg, sizeof(TieMsg)));
  SWT_HIP(hipMalloc((void **)&t->d_tie_msgs, (size_t)world * sizeof(TieMsg)));
  SWT_HIP(hipMemsetAsync(t->d_tie_msg, 0, sizeof(TieMsg), t->stream));
  SWT_HIP(hipMemsetAsync(t->d_tie_msgs, 0, (size_t)world * sizeof(TieMsg), t->stream));
  return trainer_set_block_cap(t, block_cap);
}

// the local histogram as one record list (the one-off reduction at start)
int trainer_export_records(swt_bpe_trainer *t, DeltaRec *d_out, uint64_t cap, uint64_t *n) {
  unsigned long long *d_n = &t->d_st->scratch;
  SWT_HIP(hipMemsetAsync(d_n, 0, 8, t->stream));
  const uint64_t tcap = 1ull << t->T.bits;
  hipLaunchKernelGGL(table_export_kernel, dim3(grid_for(tcap, 256, 4096)), dim3(256), 0, t->stream, t->T.keys, t->T.cnt, tcap, d_out, cap, d_n);
  unsigned long long got = 0;
  SWT_HIP(hipMemcpyAsync(&got, d_n, 8, hipMemcpyDeviceToHost, t->stream));
  SWT_HIP(hipStreamSynchronize(t->stream));
  *n = got;
  return SWT_OK;
}

int trainer_add_records(swt_bpe_trainer *t, const DeltaRec *d_recs, uint64_t n) {
  if (!n) return SWT_OK;
  int rc = t->sync_state();
  if (rc) return rc;
  if ((rc = ensure_room(t, n))) return rc;
  TrainCtx C = t->ctx();
  C.pend = nullptr;
  C.theta = 0;  // the candidate list is built after the replicas are whole
  t->cand_valid = false;
  hipLaunchKernelGGL(add_records_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, t->stream, d_recs, n, C);
  SWT_HIP(hipGetLastError());
  return t->sync_state();
}

int trainer_prepare_batch(swt_bpe_trainer *t, uint32_t k, uint32_t max_merged) {
  int rc = t->sync_state();
  if (rc) return rc;
  // every rank's new pairs land in every replica: the bound is over the whole corpus
  const uint64_t by_sym = 2 * (t->n_base_global + t->n_applied + k + 1) + 1;
  uint64_t per = t->h_st.max_count ? 2 * t->h_st.max_count : by_sym;
  if (by_sym < per) per = by_sym;
  if ((rc = ensure_room(t, per * k))) return rc;
  if ((rc = ensure_steps(t, k, max_merged))) return rc;
  if ((!t->cand_valid || !t->theta) && (rc = t->replan())) return rc;
  return SWT_OK;
}

void trainer_enqueue_tie_send(swt_bpe_trainer *t) {
  t->enqueue_argmax();
  hipLaunchKernelGGL(tie_send_kernel, dim3(1), dim3(64), 0, t->stream, t->ctx(), t->d_parts, t->n_parts, t->d_sym, t->d_woff, t->d_tie_line);
}

void trainer_enqueue_pack(swt_bpe_trainer *t) {
  hipLaunchKernelGGL(pack_records_kernel, dim3(kPackBlocks), dim3(256), 0, t->stream, t->ctx(), t->d_block, t->block_cap);
}

void trainer_enqueue_decide_apply(swt_bpe_trainer *t, uint32_t rank, uint32_t log_i, uint32_t merged) {
  hipLaunchKernelGGL(decide_sharded_kernel, dim3(1), dim3(64), 0, t->stream, t->ctx(), t->d_parts, t->n_parts, t->d_tie_all, t->world, rank,
                     t->d_cmd, t->d_steplog, log_i, merged, t->d_halt);
  t->enqueue_apply();
  trainer_enqueue_pack(t);
}

// ---- the sharded fast path: the host side of one round trip (the sizing rules of swt_bpe_train_run, over the whole corpus) ----
// Every number below is a function of state that is the same on every rank (the counts, theta, the candidate list's length,
// the merges done) -- except whether this rank's table had to grow, which voids its candidate list: that is why the runner
// ORs `replan_first` over the ranks and all of them re-plan together, or their dry-point estimates would drift apart and
// they would enqueue different numbers of steps.
static void fast_trip_size(const swt_bpe_trainer *t, uint32_t remaining, double per_step, ShardTrip *trip) {
  uint32_t steps = remaining < kRunBatch ? remaining : kRunBatch;
  const double want = (double)remaining / per_step * 1.05 + 1.0;
  if (want < (double)steps) steps = (uint32_t)want;
  if (steps < 8) steps = remaining < 8 ? remaining : 8;
  double c = (double)steps * per_step * 1.5 + 8.0;
  if (c > (double)remaining) c = (double)remaining;
  if (c > (double)kMaxRunSteps) c = (double)kMaxRunSteps;
  uint32_t cap = (uint32_t)c;
  if (cap < steps) cap = steps;
  trip->steps = steps;
  trip->cap = cap;
}

int trainer_fast_room(swt_bpe_trainer *t, uint32_t remaining, double per_step, ShardTrip *trip) {
  int rc;
  fast_trip_size(t, remaining, per_step, trip);
  // every rank's new pairs land in every replica: the bound is over the whole corpus (trainer_prepare_batch)
  const uint64_t by_sym = 2 * (t->n_base_global + t->n_applied + trip->cap + 1) + 1;
  uint64_t per = t->h_st.max_count ? 2 * t->h_st.max_count : by_sym;
  if (by_sym < per) per = by_sym;
  if ((rc = ensure_room(t, per * trip->cap))) return rc;
  trip->replan_first = !t->cand_valid || !t->theta || t->h_st.n_cand > kCandHigh;
  return SWT_OK;
}

int trainer_fast_plan(swt_bpe_trainer *t, uint32_t remaining, double per_step, bool replan, uint32_t first_id, ShardTrip *trip) {
  int rc;
  for (int pass = 0;; pass++) {
    if (replan && (rc = t->replan())) return rc;
    fast_trip_size(t, remaining, per_step, trip);
    trip->fast = t->theta != 0;
    if (!trip->fast) {  // a plateau wider than the list: the generic step with the full-table argmax until the next re-plan
      if (trip->steps > remaining) trip->steps = remaining;
      trip->cap = trip->steps;
      return ensure_steps(t, trip->steps, first_id + trip->cap);
    }
    if (t->theta > 1 && t->cand_built && pass == 0) {  // the dry point of the list (swt_bpe_train_run)
      const double budget = t->dry_ratio * (double)t->cand_built - (double)t->since_replan;
      if (budget < 24.0 && (double)remaining > budget && t->since_replan) {
        if (t->dry_ratio < 2.0) t->dry_ratio *= 1.1;
        replan = true;
        continue;
      }
      const double most = budget / per_step + 1.0;
      if (most < (double)trip->steps) {
        trip->steps = most < 8.0 ? 8u : (uint32_t)most;
        if (trip->steps > remaining) trip->steps = remaining;
        if (trip->cap > trip->steps * kMaxBatch) trip->cap = trip->steps * kMaxBatch;
      }
    }
    return ensure_steps(t, trip->steps, first_id + trip->cap);
  }
}

// the step counters of the fast path start from zero, and no position of an earlier round trip is left
int trainer_fast_begin(swt_bpe_trainer *t) {
  SWT_HIP(hipMemsetAsync(&t->d_st->run_done[0], 0, 8 * 8, t->stream));
  SWT_HIP(hipMemsetAsync(&t->d_st->best2[0], 0xFF, 2 * 8, t->stream));
  SWT_HIP(hipMemsetAsync(t->d_gpos, 0xFF, 2 * (size_t)t->cand_cap * 8, t->stream));
  SWT_HIP(hipMemsetAsync(t->d_gnb_min, 0xFF, 4 * (size_t)t->cand_cap * 4, t->stream));
  SWT_HIP(hipMemsetAsync(t->d_gnb_max, 0, 4 * (size_t)t->cand_cap * 4, t->stream));
  return SWT_OK;
}

void trainer_enqueue_fast_tie(swt_bpe_trainer *t, uint32_t limit) {
  const TrainCtx C = t->ctx();
  // (an empty shard runs the launches too: they carry its part of the exchange)
  const unsigned tie_blocks = grid_for(t->n_words, 64, kTieBlocks);
  hipLaunchKernelGGL(fast_tie_kernel, dim3(tie_blocks + 2), dim3(kTrainThreads), 0, t->stream, t->d_sym, t->d_woff, t->n_words, C, limit);
  hipLaunchKernelGGL(tie_pack_kernel, dim3(1), dim3(kTrainThreads), 0, t->stream, C, t->n_words, limit);
}

void trainer_enqueue_fast_apply(swt_bpe_trainer *t, uint32_t first_merged, uint32_t limit) {
  hipLaunchKernelGGL(fast_apply_sharded_kernel, dim3(kFastApplyBlocks), dim3(kTrainThreads), 0, t->stream, t->d_sym, t->d_woff, t->d_freq,
                     t->n_words, t->ctx(), t->d_steplog, first_merged, limit);
  trainer_enqueue_pack(t);
}

void trainer_enqueue_add_blocks(swt_bpe_trainer *t) {
  TrainCtx C = t->ctx();
  C.pend = nullptr;  // the blocks go into the counts themselves ...
  C.theta = 0;       // ... and nothing is listed on the way: finish_exchange_kernel lists from the final counts
  hipLaunchKernelGGL(add_blocks_kernel, dim3(kPackBlocks), dim3(256), 0, t->stream, t->d_blocks_all, t->world, t->block_cap, C, t->d_halt);
  hipLaunchKernelGGL(finish_exchange_kernel, dim3(1), dim3(256), 0, t->stream, t->d_blocks_all, t->world, t->block_cap, t->ctx(), t->d_halt);
}

}  // namespace swt
