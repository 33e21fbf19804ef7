// swt_bpe_train.hip -- BPE training merge loop on gfx950.
//
// Replaces NaiveBPE.train's loop (FastBPE.train inherits it):
//   word dedup + symbolisation   /root/reference/source/bpe.py:73-81   (host C++ here; device later)
//   pair histogram               /root/reference/source/bpe.py:90-95
//   argmax with first-seen tie   /root/reference/source/bpe.py:98-102
//   merge-apply (_replace_pair)  /root/reference/source/bpe.py:25-48, 108-111
//
// Device state
//   sym[]   packed uint32 symbol stream of the UNIQUE words, word w at [woff[w], woff[w]+wlen[w]); a merge
//           rewrites the word in place (its slot keeps its offset, only wlen shrinks), so stream order = the
//           reference's scan order and a position woff[w]+i is a valid first-occurrence key.
//   pair histogram: open-addressing hash, keys[] (left<<32|right) + cnt[] (64-bit, weighted by word
//           frequency).  Built once by a full scan, then kept exact incrementally: a merge only touches pairs
//           adjacent to its occurrences, so each step subtracts the pairs it destroys and adds the pairs it
//           creates (atomicAdd on the table) -- the same counts the reference recomputes from scratch.
// Per merge: argmax over cnt[] (max via atomicMax, then tie census); only when the maximum is tied, one read-only
// scan finds the earliest (word, position) among the tied pairs (bpe.py:102: Counter.most_common(1) returns
// the first-inserted maximum).  One host round trip per merge: the caller owns the string set and the stop test.
//
// WordPiece mode (NaiveWP.train, /root/reference/source/wordpiece.py:29-103): the same stream, histogram
// and merge-apply; symbols are the word's first character and "##c" (= 0x110000 + c) for the others; a dense
// symbol-frequency array is kept exact beside the pair histogram, and the argmax key is the likelihood score
// freq / (f_left * f_right) -- Python's int / int, i.e. the correctly rounded quotient -- compared as a bit pattern.
#include <algorithm>
#include <unordered_map>

#include "swt_common.h"
#include "swt_words.h"

namespace swt {

constexpr int kTrainThreads = 256;
constexpr int kArgBlocks = 1024;
constexpr uint32_t kMaxRunSteps = 512;
constexpr uint32_t kRunBatch = 256;

struct TrainResult {
  unsigned long long max_count;
  unsigned long long n_tied;
  unsigned long long best_pos;   // local stream position of the winner (kEmptyKey: none found locally)
  unsigned long long best_key;   // a key with count == max (the winner when n_tied == 1)
  unsigned long long n_used;     // distinct keys ever inserted in the table
  unsigned long long n_log;      // delta-log entries of the last apply
  unsigned long long n_syms;     // live symbols after the last apply
  unsigned long long win_key;    // pair at best_pos (tie winner)
};

struct StepCmd {
  uint32_t l, r, m, valid;
};

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
struct StepLog {
  uint32_t l, r;
  unsigned long long count;
  unsigned long long flag;  // 0 merged, 1 tied maximum (the host breaks the tie), 2 no pair left
};

struct PairTable {
  unsigned long long *keys;
  long long *cnt;
  uint32_t bits;
};

__device__ __forceinline__ void table_add(const PairTable &T, unsigned long long key, long long delta, TrainResult *res) {
  const uint32_t mask = (1u << T.bits) - 1u;
  uint32_t h = hash_slot(key, T.bits);
  for (;;) {
    unsigned long long k = __hip_atomic_load(&T.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == kEmptyKey) {
      k = atomicCAS(&T.keys[h], kEmptyKey, key);
      if (k == kEmptyKey) {
        atomicAdd(&res->n_used, 1ull);
        k = key;
      }
    }
    if (k == key) {
      atomicAdd(reinterpret_cast<unsigned long long *>(&T.cnt[h]), (unsigned long long)delta);
      return;
    }
    h = (h + 1) & mask;
  }
}

__device__ __forceinline__ long long table_get(const PairTable &T, unsigned long long key) {
  const uint32_t mask = (1u << T.bits) - 1u;
  uint32_t h = hash_slot(key, T.bits);
  for (;;) {
    const unsigned long long k = T.keys[h];
    if (k == key) return T.cnt[h];
    if (k == kEmptyKey) return 0;
    h = (h + 1) & mask;
  }
}

__global__ void table_clear_kernel(unsigned long long *keys, long long *cnt, uint64_t cap) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
    keys[i] = kEmptyKey;
    cnt[i] = 0;
  }
}

// bpe.py:90-95 -- every adjacent pair of every unique word, weighted by the word's frequency
__global__ __launch_bounds__(kTrainThreads) void hist_build_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                                   const uint32_t *__restrict__ wlen, const uint32_t *__restrict__ freq,
                                                                   uint64_t n_words, PairTable T, TrainResult *res) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  const uint32_t n = wlen[w];
  if (n < 2) return;
  const uint32_t *s = sym + woff[w];
  const long long f = freq[w];
  uint32_t a = s[0];
  for (uint32_t i = 1; i < n; i++) {
    const uint32_t b = s[i];
    table_add(T, pair_key(a, b), f, res);
    a = b;
  }
}

struct ArgPart {
  unsigned long long mx, cnt, key;
};

__device__ __forceinline__ void arg_combine(unsigned long long &m, unsigned long long &c, unsigned long long &k,
                                            unsigned long long m2, unsigned long long c2, unsigned long long k2) {
  if (m2 > m) { m = m2; c = c2; k = k2; }
  else if (m2 == m) { c += c2; k = k2 < k ? k2 : k; }
}

// bpe.py:98-102 in one launch: maximum count, how many pairs hold it, and the smallest such key.  Every workgroup
// reduces its share of the table and publishes a partial; the last one to arrive (ticket) combines the partials
// and resets the tie-break fields of the result.
__global__ __launch_bounds__(256) void argmax_kernel(const unsigned long long *__restrict__ keys, const long long *__restrict__ cnt,
                                                     uint64_t cap, ArgPart *__restrict__ parts, unsigned int *__restrict__ ticket,
                                                     TrainResult *res, const long long *__restrict__ sfreq) {
  __shared__ unsigned long long sm[4], sc[4], sk[4];
  __shared__ bool is_last;
  unsigned long long m = 0, c = 0, k = kEmptyKey;
  // the counts are streamed two per load, four loads in flight per lane (cap is a power of two >= 1024); a key is only
  // fetched for a count that can still win
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
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long m2 = __shfl_xor(m, d), c2 = __shfl_xor(c, d), k2 = __shfl_xor(k, d);
    arg_combine(m, c, k, m2, c2, k2);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sm[wave] = m; sc[wave] = c; sk[wave] = k; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) arg_combine(m, c, k, sm[w], sc[w], sk[w]);
    parts[blockIdx.x].mx = m;
    parts[blockIdx.x].cnt = c;
    parts[blockIdx.x].key = k;
    __threadfence();  // agent-scope release before the ticket
    is_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();  // agent-scope acquire
  m = 0; c = 0; k = kEmptyKey;
  for (uint32_t j = threadIdx.x; j < gridDim.x; j += blockDim.x) {
    const unsigned long long m2 = __hip_atomic_load(&parts[j].mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long c2 = __hip_atomic_load(&parts[j].cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long k2 = __hip_atomic_load(&parts[j].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m2 > 0) arg_combine(m, c, k, m2, c2, k2);
  }
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long m2 = __shfl_xor(m, d), c2 = __shfl_xor(c, d), k2 = __shfl_xor(k, d);
    arg_combine(m, c, k, m2, c2, k2);
  }
  __syncthreads();
  if (lane == 0) { sm[wave] = m; sc[wave] = c; sk[wave] = k; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) arg_combine(m, c, k, sm[w], sc[w], sk[w]);
    res->max_count = m;
    res->n_tied = m ? c : 0;
    res->best_key = k;
    res->best_pos = kEmptyKey;
    res->win_key = kEmptyKey;
    *ticket = 0;
  }
}

// bpe.py:102 tie-break: the earliest (word, position) whose pair holds the maximum count.  Grid-stride over the
// words, so an untied step costs a handful of workgroups that return at once.
// Device-driven mode (cmd != null, swt_bpe_train_run): this kernel also turns the result into the step's merge command
// for apply_kernel and logs it -- workgroup 0 when the maximum is unique, the last workgroup of the scan when tied.
// WordPiece: the symbol frequencies follow the merge.  A pair of two different symbols cannot overlap itself, so the merge
// happens exactly count(l, r) times (weighted); a twin pair (a, a) is counted by apply_kernel, occurrence by occurrence.
__device__ __forceinline__ void wp_move_freq(const PairTable &T, uint32_t l, uint32_t r, uint32_t m, long long *sfreq) {
  if (l == r) return;
  const long long c = table_get(T, pair_key(l, r));
  sfreq[l] -= c;
  sfreq[r] -= c;
  sfreq[m] += c;
}

__device__ __forceinline__ void write_cmd(unsigned long long key, unsigned long long mx, StepCmd *cmd, StepLog *log, uint32_t step,
                                          uint32_t merged, const PairTable &T, long long *sfreq) {
  const bool ok = mx > 0 && key != kEmptyKey;
  cmd->l = (uint32_t)(key >> 32);
  cmd->r = (uint32_t)key;
  cmd->m = merged;
  cmd->valid = ok ? 1u : 0u;
  if (ok && sfreq) wp_move_freq(T, cmd->l, cmd->r, merged, sfreq);
  log[step].l = cmd->l;
  log[step].r = cmd->r;
  log[step].count = mx;
  log[step].flag = ok ? 0ull : 2ull;
}

__global__ __launch_bounds__(kTrainThreads) void first_pos_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                                  const uint32_t *__restrict__ wlen, uint64_t n_words, PairTable T,
                                                                  TrainResult *res, StepCmd *cmd, StepLog *log, uint32_t step,
                                                                  uint32_t merged, unsigned int *ticket, long long *sfreq) {
  __shared__ bool is_last;
  if (res->n_tied < 2) {
    if (cmd && blockIdx.x == 0 && threadIdx.x == 0) write_cmd(res->best_key, res->max_count, cmd, log, step, merged, T, sfreq);
    return;
  }
  const unsigned long long mx = res->max_count;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t n = wlen[w];
    if (n < 2) continue;
    const uint64_t base = woff[w];
    if (base >= __hip_atomic_load(&res->best_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // words only get later
    const uint32_t *s = sym + base;
    uint32_t a = s[0];
    for (uint32_t i = 1; i < n; i++) {
      const uint32_t b = s[i];
      const unsigned long long key = pair_key(a, b);
      if (pair_value(key, table_get(T, key), sfreq) == mx) {
        atomicMin(&res->best_pos, (unsigned long long)(base + i - 1));
        break;
      }
      a = b;
    }
  }
  if (!cmd) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    is_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last || threadIdx.x != 0) return;
  __threadfence();
  const unsigned long long pos = __hip_atomic_load(&res->best_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long key = pos != kEmptyKey ? pair_key(sym[pos], sym[pos + 1]) : kEmptyKey;
  res->win_key = key;
  write_cmd(key, res->max_count, cmd, log, step, merged, T, sfreq);
  *ticket = 0;
}

// host-driven WordPiece step (swt_bpe_train_apply): the frequency move of write_cmd as a launch of its own, BEFORE apply_kernel
__global__ void wp_move_freq_kernel(PairTable T, uint32_t l, uint32_t r, uint32_t m, long long *sfreq) { wp_move_freq(T, l, r, m, sfreq); }

// wordpiece.py:78-81 once: symbol frequencies, weighted by the word's frequency
__global__ __launch_bounds__(kTrainThreads) void sym_hist_kernel(const uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                                 const uint32_t *__restrict__ wlen, const uint32_t *__restrict__ freq,
                                                                 uint64_t n_words, long long *__restrict__ sfreq, uint64_t sym_cap,
                                                                 unsigned int *__restrict__ bad) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  const uint32_t n = wlen[w];
  const uint32_t *s = sym + woff[w];
  const unsigned long long f = freq[w];
  for (uint32_t i = 0; i < n; i++) {
    if (s[i] < sym_cap) atomicAdd(reinterpret_cast<unsigned long long *>(&sfreq[s[i]]), f);
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

__global__ void winner_kernel(const uint32_t *__restrict__ sym, TrainResult *res) {
  if (res->n_tied < 2 || res->best_pos == kEmptyKey) return;
  res->win_key = pair_key(sym[res->best_pos], sym[res->best_pos + 1]);
}

// bpe.py:108-111 + _replace_pair (bpe.py:25-48), with the histogram kept exact:
//   an old pair (x[i],x[i+1]) disappears iff x[i] or x[i+1] is consumed by an occurrence;
//   a new pair (y[j],y[j+1]) appears iff y[j] or y[j+1] is a freshly merged symbol.
__global__ __launch_bounds__(kTrainThreads) void apply_kernel(uint32_t *__restrict__ sym, const uint64_t *__restrict__ woff,
                                                              uint32_t *__restrict__ wlen, const uint32_t *__restrict__ freq,
                                                              uint64_t n_words, uint32_t l, uint32_t r, uint32_t m, PairTable T,
                                                              TrainResult *res, unsigned long long *__restrict__ log_keys,
                                                              long long *__restrict__ log_vals, uint64_t log_cap,
                                                              const StepCmd *__restrict__ cmd, long long *__restrict__ sfreq) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  if (cmd) {  // device-driven step: the pair comes from decide_kernel
    if (!cmd->valid) return;
    l = cmd->l; r = cmd->r; m = cmd->m;
  }
  const uint32_t n = wlen[w];
  if (n < 2) return;
  uint32_t *s = sym + woff[w];
  // cheap reject: does the word hold an occurrence at all?
  bool any = false;
  {
    uint32_t a = s[0];
    for (uint32_t i = 1; i < n; i++) {
      const uint32_t b = s[i];
      any |= (a == l) & (b == r);
      a = b;
    }
  }
  if (!any) return;
  const long long f = freq[w];
#define EMIT(key, delta)                                              \
  do {                                                                \
    table_add(T, (key), (delta), res);                                \
    if (log_keys) {                                                   \
      const unsigned long long k_ = atomicAdd(&res->n_log, 1ull);     \
      if (k_ < log_cap) { log_keys[k_] = (key); log_vals[k_] = (delta); } \
    }                                                                 \
  } while (0)
  uint32_t i = 0, j = 0;
  uint32_t po = 0, pn = 0;      // previous old / new symbol
  bool po_cov = false, pn_new = false, have = false;
  while (i < n) {
    const uint32_t x = s[i];
    const bool occ = (i + 1 < n) && x == l && s[i + 1] == r;
    if (occ) {
      if (have) EMIT(pair_key(po, x), -f);      // (prev, l): l is consumed
      EMIT(pair_key(x, r), -f);                 // (l, r) itself
      if (have) EMIT(pair_key(pn, m), f);       // (prev_new, merged)
      po = r; po_cov = true; pn = m; pn_new = true; have = true;
      s[j++] = m;
      i += 2;
    } else {
      if (have) {
        if (po_cov) EMIT(pair_key(po, x), -f);  // (r, x): r was consumed
        if (pn_new) EMIT(pair_key(pn, x), f);   // (merged, x)
      }
      po = x; po_cov = false; pn = x; pn_new = false; have = true;
      s[j++] = x;
      i += 1;
    }
  }
#undef EMIT
  if (sfreq && l == r) {  // twin pair: (n - j) merges happened in this word (see wp_move_freq)
    const unsigned long long d = (unsigned long long)(n - j) * (unsigned long long)f;
    atomicAdd(reinterpret_cast<unsigned long long *>(&sfreq[l]), (unsigned long long)0 - 2 * d);
    atomicAdd(reinterpret_cast<unsigned long long *>(&sfreq[m]), d);
  }
  wlen[w] = j;
  atomicAdd(&res->n_syms, (unsigned long long)0 - (unsigned long long)(n - j));
}

__global__ void add_remote_kernel(const unsigned long long *__restrict__ keys, const long long *__restrict__ vals, uint64_t n,
                                  PairTable T, TrainResult *res) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    if (vals[i] != 0) table_add(T, keys[i], vals[i], res);
}

// live entries -> (keys, counts) list
__global__ void table_export_kernel(const unsigned long long *__restrict__ keys, const long long *__restrict__ cnt, uint64_t cap,
                                    unsigned long long *__restrict__ out_keys, long long *__restrict__ out_vals, uint64_t out_cap,
                                    unsigned long long *n_out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
    if (keys[i] != kEmptyKey && cnt[i] != 0) {
      const unsigned long long k = atomicAdd(n_out, 1ull);
      if (k < out_cap) { out_keys[k] = keys[i]; out_vals[k] = cnt[i]; }
    }
  }
}

__global__ void table_rehash_kernel(const unsigned long long *__restrict__ keys, const long long *__restrict__ cnt, uint64_t cap,
                                    PairTable dst, TrainResult *res) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x)
    if (keys[i] != kEmptyKey && cnt[i] != 0) table_add(dst, keys[i], cnt[i], res);
}

}  // namespace swt

using namespace swt;

struct swt_bpe_trainer {
  uint64_t n_words = 0, n_syms0 = 0;
  uint32_t n_base = 0;
  std::vector<uint32_t> base_syms;
  uint32_t *d_sym = nullptr;
  uint64_t *d_woff = nullptr;
  uint32_t *d_wlen = nullptr;
  uint32_t *d_freq = nullptr;
  PairTable T{nullptr, nullptr, 0};
  TrainResult *d_res = nullptr;
  ArgPart *d_parts = nullptr;     // per-workgroup argmax partials + the ticket behind them
  StepCmd *d_cmd = nullptr;       // device-driven steps
  StepLog *d_steplog = nullptr;
  unsigned int *d_halt = nullptr;  // ticket of the tie-break scan
  uint64_t n_applied = 0;         // merges applied so far (bounds the number of distinct symbols)
  TrainResult h_res{};
  uint64_t pos_base = 0;
  bool hist_ready = false;
  long long *d_sfreq = nullptr;  // WordPiece mode: symbol frequencies, dense by symbol id (kWpSymCap entries)
  // delta log (sharded training)
  bool logging = false;
  unsigned long long *d_log_keys = nullptr;
  long long *d_log_vals = nullptr;
  uint64_t log_cap = 0;
  DevBuf tmp;
};

static unsigned grid_for(uint64_t n, int threads, unsigned cap = 1u << 20) {
  uint64_t g = (n + threads - 1) / threads;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

static int table_alloc(PairTable &T, uint32_t bits) {
  const size_t cap = (size_t)1 << bits;
  T.bits = bits;
  SWT_HIP(hipMalloc((void **)&T.keys, cap * 8));
  SWT_HIP(hipMalloc((void **)&T.cnt, cap * 8));
  hipLaunchKernelGGL(table_clear_kernel, dim3(grid_for(cap, 256, 4096)), dim3(256), 0, 0, T.keys, T.cnt, (uint64_t)cap);
  return SWT_OK;
}

static void table_free(PairTable &T) {
  if (T.keys) (void)hipFree(T.keys);
  if (T.cnt) (void)hipFree(T.cnt);
  T.keys = nullptr;
  T.cnt = nullptr;
}

static int sync_result(swt_bpe_trainer *t) {
  SWT_HIP(hipMemcpy(&t->h_res, t->d_res, sizeof(TrainResult), hipMemcpyDeviceToHost));
  return SWT_OK;
}

// Rebuild the table at `bits` from its live entries (drops zero-count keys).
static int table_resize(swt_bpe_trainer *t, uint32_t bits) {
  PairTable nt{nullptr, nullptr, 0};
  int rc = table_alloc(nt, bits);
  if (rc) return rc;
  SWT_HIP(hipMemset(&t->d_res->n_used, 0, 8));
  const uint64_t cap = 1ull << t->T.bits;
  hipLaunchKernelGGL(table_rehash_kernel, dim3(grid_for(cap, 256, 4096)), dim3(256), 0, 0, t->T.keys, t->T.cnt, cap, nt, t->d_res);
  SWT_HIP(hipDeviceSynchronize());
  table_free(t->T);
  t->T = nt;
  return SWT_OK;
}

static int build_histogram(swt_bpe_trainer *t) {
  // size for the worst case first (every position a distinct pair), then shrink to what is used
  uint32_t bits = 10;
  while ((1ull << bits) < 2 * t->n_syms0 + 16 && bits < 31) bits++;
  int rc = table_alloc(t->T, bits);
  if (rc) return rc;
  if (t->n_words)
    hipLaunchKernelGGL(hist_build_kernel, dim3(grid_for(t->n_words, kTrainThreads)), dim3(kTrainThreads), 0, 0, t->d_sym, t->d_woff,
                       t->d_wlen, t->d_freq, t->n_words, t->T, t->d_res);
  SWT_HIP(hipDeviceSynchronize());
  if ((rc = sync_result(t))) return rc;
  uint32_t want = 10;
  while ((1ull << want) < 4 * t->h_res.n_used + 1024) want++;
  if (want < bits) {
    if ((rc = table_resize(t, want))) return rc;
  }
  t->hist_ready = true;
  return SWT_OK;
}

__global__ void wlen_kernel(const uint64_t *__restrict__ woff, uint32_t *__restrict__ wlen, uint64_t n_words) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n_words) wlen[w] = (uint32_t)(woff[w + 1] - woff[w]);
}

// take ownership of a unique-word stream that is already on the device (swt_words.hip)
static int trainer_adopt(swt_bpe_trainer *t, DeviceWords &dw) {
  t->n_words = dw.n_words;
  t->n_syms0 = dw.n_syms;
  t->d_sym = dw.d_sym;
  t->d_woff = dw.d_woff;
  t->d_freq = dw.d_freq;
  dw.d_sym = nullptr; dw.d_woff = nullptr; dw.d_freq = nullptr;
  SWT_HIP(hipMalloc((void **)&t->d_wlen, (t->n_words + 1) * 4));
  SWT_HIP(hipMalloc((void **)&t->d_res, sizeof(TrainResult)));
  SWT_HIP(hipMemset(t->d_res, 0, sizeof(TrainResult)));
  SWT_HIP(hipMalloc((void **)&t->d_cmd, sizeof(StepCmd)));
  SWT_HIP(hipMalloc((void **)&t->d_steplog, kMaxRunSteps * sizeof(StepLog)));
  SWT_HIP(hipMalloc((void **)&t->d_halt, 8));
  SWT_HIP(hipMalloc((void **)&t->d_parts, (kArgBlocks + 1) * sizeof(ArgPart)));
  SWT_HIP(hipMemset(t->d_parts, 0, (kArgBlocks + 1) * sizeof(ArgPart)));
  if (t->n_words)
    hipLaunchKernelGGL(wlen_kernel, dim3(grid_for(t->n_words, 256)), dim3(256), 0, 0, t->d_woff, t->d_wlen, t->n_words);
  unsigned long long ns = t->n_syms0;
  SWT_HIP(hipMemcpy(&t->d_res->n_syms, &ns, 8, hipMemcpyHostToDevice));
  t->base_syms = dw.base_syms;
  t->n_base = (uint32_t)t->base_syms.size();
  return build_histogram(t);
}

static int trainer_upload(swt_bpe_trainer *t, const uint32_t *syms, const uint64_t *word_off, const uint32_t *freq, uint64_t n_words) {
  int rc = ensure_device();
  if (rc) return rc;
  const uint64_t n_syms = word_off[n_words];
  t->n_words = n_words;
  t->n_syms0 = n_syms;
  std::vector<uint32_t> wlen(n_words + 1);
  for (uint64_t w = 0; w < n_words; w++) {
    if (word_off[w + 1] < word_off[w]) return fail(SWT_ERR_INVALID, "word offsets must be non-decreasing");
    if (word_off[w + 1] - word_off[w] > 0xFFFFFFFFull) return fail(SWT_ERR_UNSUPPORTED, "word too long");
    wlen[w] = (uint32_t)(word_off[w + 1] - word_off[w]);
  }
  SWT_HIP(hipMalloc((void **)&t->d_sym, (n_syms + 16) * 4));
  SWT_HIP(hipMalloc((void **)&t->d_woff, (n_words + 1) * 8));
  SWT_HIP(hipMalloc((void **)&t->d_wlen, (n_words + 1) * 4));
  SWT_HIP(hipMalloc((void **)&t->d_freq, (n_words + 1) * 4));
  SWT_HIP(hipMalloc((void **)&t->d_res, sizeof(TrainResult)));
  SWT_HIP(hipMemset(t->d_res, 0, sizeof(TrainResult)));
  SWT_HIP(hipMalloc((void **)&t->d_cmd, sizeof(StepCmd)));
  SWT_HIP(hipMalloc((void **)&t->d_steplog, kMaxRunSteps * sizeof(StepLog)));
  SWT_HIP(hipMalloc((void **)&t->d_halt, 8));
  SWT_HIP(hipMalloc((void **)&t->d_parts, (kArgBlocks + 1) * sizeof(ArgPart)));
  SWT_HIP(hipMemset(t->d_parts, 0, (kArgBlocks + 1) * sizeof(ArgPart)));
  if (n_syms) SWT_HIP(hipMemcpy(t->d_sym, syms, n_syms * 4, hipMemcpyHostToDevice));
  SWT_HIP(hipMemcpy(t->d_woff, word_off, (n_words + 1) * 8, hipMemcpyHostToDevice));
  if (n_words) {
    SWT_HIP(hipMemcpy(t->d_wlen, wlen.data(), n_words * 4, hipMemcpyHostToDevice));
    SWT_HIP(hipMemcpy(t->d_freq, freq, n_words * 4, hipMemcpyHostToDevice));
  }
  unsigned long long ns = n_syms;
  SWT_HIP(hipMemcpy(&t->d_res->n_syms, &ns, 8, hipMemcpyHostToDevice));
  // distinct code points (the initial vocab, bpe.py:75)
  {
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
  }
  return build_histogram(t);
}

extern "C" {

int swt_bpe_train_create_words(const uint32_t *syms, const uint64_t *word_off, const uint32_t *freq, uint64_t n_words,
                               swt_bpe_trainer **out) {
  if (!out || !word_off || (n_words && (!freq || (word_off[n_words] && !syms)))) return fail(SWT_ERR_INVALID, "null argument");
  if (word_off[0] != 0) return fail(SWT_ERR_INVALID, "word_off[0] must be 0");
  auto *t = new swt_bpe_trainer();
  int rc = trainer_upload(t, syms, word_off, freq, n_words);
  if (rc) { swt_bpe_train_destroy(t); return rc; }
  *out = t;
  return SWT_OK;
}

// bpe.py:70-81 on the device (swt_words.hip): split (utils.py:27), Counter(words) in first-occurrence order, symbolise.
int swt_bpe_train_create_text(const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent, swt_bpe_trainer **out) {
  if (!out || !sent_off || (n_sent && sent_off[n_sent] && !text)) return fail(SWT_ERR_INVALID, "null argument");
  if (sent_off[0] != 0) return fail(SWT_ERR_INVALID, "sent_off[0] must be 0");
  for (uint64_t s = 0; s < n_sent; s++)
    if (sent_off[s] > sent_off[s + 1]) return fail(SWT_ERR_INVALID, "sentence offsets must be non-decreasing");
  int rc = ensure_device();
  if (rc) return rc;
  const uint64_t n_bytes = sent_off[n_sent];
  DevBuf d_text, d_off;
  if ((rc = d_text.reserve(n_bytes + 64)) || (rc = d_off.reserve((n_sent + 1) * 8))) return rc;
  if (n_bytes) SWT_HIP(hipMemcpy(d_text.p, text, n_bytes, hipMemcpyHostToDevice));
  SWT_HIP(hipMemcpy(d_off.p, sent_off, (n_sent + 1) * 8, hipMemcpyHostToDevice));
  DeviceWords dw;
  rc = device_words_from_text(d_text.as<uint8_t>(), n_bytes, d_off.as<uint64_t>(), n_sent, &dw);
  d_text.release();
  d_off.release();
  if (rc) return rc;
  auto *t = new swt_bpe_trainer();
  rc = trainer_adopt(t, dw);
  if (rc) { swt_bpe_train_destroy(t); return rc; }
  *out = t;
  return SWT_OK;
}

// wordpiece.py:44-63 on the device: the same split and Counter, then [word[0]] + ["##" + c ...] and the symbol frequencies.
// The handle is used with the swt_bpe_train_* calls; `count` outputs carry the winning score's bit pattern.
int swt_wp_train_create_text(const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent, swt_bpe_trainer **out) {
  swt_bpe_trainer *t = nullptr;
  int rc = swt_bpe_train_create_text(text, sent_off, n_sent, &t);
  if (rc) return rc;
  // the pair histogram was built on plain code points: rebuild it on the WordPiece symbols
  table_free(t->T);
  if (t->n_words)
    hipLaunchKernelGGL(wp_symbolise_kernel, dim3(grid_for(t->n_words, kTrainThreads)), dim3(kTrainThreads), 0, 0, t->d_sym, t->d_woff,
                       t->n_words);
  SWT_HIP(hipMemset(&t->d_res->n_used, 0, 8));
  if ((rc = build_histogram(t))) { swt_bpe_train_destroy(t); return rc; }
  SWT_HIP(hipMalloc((void **)&t->d_sfreq, kWpSymCap * 8));
  SWT_HIP(hipMemset(t->d_sfreq, 0, kWpSymCap * 8));
  unsigned int *d_flag = reinterpret_cast<unsigned int *>(t->d_halt);
  SWT_HIP(hipMemset(d_flag, 0, 8));
  if (t->n_words)
    hipLaunchKernelGGL(sym_hist_kernel, dim3(grid_for(t->n_words, kTrainThreads)), dim3(kTrainThreads), 0, 0, t->d_sym, t->d_woff, t->d_wlen,
                       t->d_freq, t->n_words, t->d_sfreq, kWpSymCap, d_flag);
  // the initial vocabulary (wordpiece.py:62-63) = the symbols that occur
  DevBuf live;
  const uint32_t live_cap = 2 * kWpCont;
  if ((rc = live.reserve((size_t)live_cap * 4))) { swt_bpe_train_destroy(t); return rc; }
  hipLaunchKernelGGL(wp_live_symbols_kernel, dim3(1024), dim3(256), 0, 0, (const long long *)t->d_sfreq, (uint64_t)kWpMergedBase,
                     live.as<uint32_t>(), live_cap, d_flag + 1);
  unsigned int h[2] = {0, 0};
  SWT_HIP(hipMemcpy(h, d_flag, 8, hipMemcpyDeviceToHost));
  if (h[0]) { swt_bpe_train_destroy(t); return fail(SWT_ERR_UNSUPPORTED, "symbol id out of range for a WordPiece trainer"); }
  t->base_syms.resize(h[1]);
  if (h[1]) SWT_HIP(hipMemcpy(t->base_syms.data(), live.p, (size_t)h[1] * 4, hipMemcpyDeviceToHost));
  std::sort(t->base_syms.begin(), t->base_syms.end());
  t->n_base = h[1];
  SWT_HIP(hipMemset(d_flag, 0, 8));
  live.release();
  *out = t;
  return SWT_OK;
}

void swt_bpe_train_destroy(swt_bpe_trainer *t) {
  if (!t) return;
  for (void *p : {(void *)t->d_sym, (void *)t->d_woff, (void *)t->d_wlen, (void *)t->d_freq, (void *)t->d_res, (void *)t->d_parts, (void *)t->d_cmd, (void *)t->d_steplog, (void *)t->d_halt,
                  (void *)t->d_log_keys, (void *)t->d_log_vals, (void *)t->d_sfreq})
    if (p) (void)hipFree(p);
  table_free(t->T);
  t->tmp.release();
  delete t;
}

int swt_bpe_train_set_pos_base(swt_bpe_trainer *t, uint64_t pos_base) {
  if (!t) return fail(SWT_ERR_INVALID, "null trainer");
  t->pos_base = pos_base;
  return SWT_OK;
}

int swt_bpe_train_info(const swt_bpe_trainer *t, uint64_t *n_words, uint64_t *n_symbols, uint32_t *n_base_symbols, uint64_t *n_pairs) {
  if (!t) return fail(SWT_ERR_INVALID, "null trainer");
  TrainResult r;
  SWT_HIP(hipMemcpy(&r, t->d_res, sizeof r, hipMemcpyDeviceToHost));
  if (n_words) *n_words = t->n_words;
  if (n_symbols) *n_symbols = r.n_syms;
  if (n_base_symbols) *n_base_symbols = t->n_base;
  if (n_pairs) *n_pairs = r.n_used;
  return SWT_OK;
}

int swt_bpe_train_base_symbols(const swt_bpe_trainer *t, uint32_t *out, uint32_t cap) {
  if (!t) return fail(SWT_ERR_INVALID, "null trainer");
  if (cap < t->n_base) return fail(SWT_ERR_CAPACITY, "need room for %u symbols", t->n_base);
  std::copy(t->base_syms.begin(), t->base_syms.end(), out);
  return SWT_OK;
}

int swt_bpe_train_best(swt_bpe_trainer *t, uint32_t *left, uint32_t *right, uint64_t *count, uint64_t *n_tied,
                       uint64_t *first_pos) {
  if (!t || !left || !right || !count) return fail(SWT_ERR_INVALID, "null argument");
  int rc = ensure_device();
  if (rc) return rc;
  const uint64_t cap = 1ull << t->T.bits;
  const unsigned g = grid_for(cap, 256 * 8, kArgBlocks);  // 8 counts per thread: 4 double loads
  unsigned int *ticket = reinterpret_cast<unsigned int *>(t->d_parts + kArgBlocks);
  hipLaunchKernelGGL(argmax_kernel, dim3(g), dim3(256), 0, 0, t->T.keys, t->T.cnt, cap, t->d_parts, ticket, t->d_res,
                     (const long long *)t->d_sfreq);
  if ((rc = sync_result(t))) return rc;
  if (t->h_res.n_tied >= 2 && t->n_words) {
    // bpe.py:102: only a tied maximum needs the scan for the earliest (word, position)
    hipLaunchKernelGGL(first_pos_kernel, dim3(grid_for(t->n_words, kTrainThreads, 1024)), dim3(kTrainThreads), 0, 0, t->d_sym,
                       t->d_woff, t->d_wlen, t->n_words, t->T, t->d_res, (StepCmd *)nullptr, (StepLog *)nullptr, 0u, 0u,
                       (unsigned int *)nullptr, t->d_sfreq);
    hipLaunchKernelGGL(winner_kernel, dim3(1), dim3(1), 0, 0, t->d_sym, t->d_res);
    if ((rc = sync_result(t))) return rc;
  }
  const TrainResult &r = t->h_res;
  *count = r.max_count;
  if (n_tied) *n_tied = r.n_tied;
  if (r.max_count == 0) { *left = *right = 0; if (first_pos) *first_pos = kEmptyKey; return SWT_OK; }
  unsigned long long key = r.best_key;
  unsigned long long pos = kEmptyKey;
  if (r.n_tied >= 2) {
    if (r.best_pos != kEmptyKey) { key = r.win_key; pos = t->pos_base + r.best_pos; }
    else key = kEmptyKey;  // none of the tied pairs occurs in this shard
  }
  *left = (uint32_t)(key >> 32);
  *right = (uint32_t)key;
  if (first_pos) *first_pos = pos;
  return SWT_OK;
}

// New pairs one merge can create: two per occurrence (occurrences <= the pair's count, counts never grow), and never
// more than (x, m) / (m, y) over the distinct symbols x, y plus (m, m).
static uint64_t new_pairs_bound(const swt_bpe_trainer *t, uint64_t count_bound) {
  if (t->d_sfreq) count_bound = 0;  // WordPiece: max_count holds a score, not a count
  const uint64_t by_symbols = 2 * (t->n_base + t->n_applied + 1) + 1;
  uint64_t by_count = count_bound ? 2 * count_bound : by_symbols;
  if (t->h_res.n_syms && 2 * t->h_res.n_syms < by_count) by_count = 2 * t->h_res.n_syms;
  return by_symbols < by_count ? by_symbols : by_count;
}

static int ensure_room(swt_bpe_trainer *t, uint64_t extra) {
  const uint64_t cap = 1ull << t->T.bits;
  if (2 * (t->h_res.n_used + extra + 64) <= cap) return SWT_OK;
  uint32_t bits = t->T.bits;
  while ((1ull << bits) < 4 * (t->h_res.n_used + extra + 64)) bits++;
  if (bits > 32) return fail(SWT_ERR_UNSUPPORTED, "pair table would exceed 2^32 slots");
  int rc = table_resize(t, bits);
  if (rc) return rc;
  return sync_result(t);
}

int swt_bpe_train_apply(swt_bpe_trainer *t, uint32_t left, uint32_t right, uint32_t merged) {
  if (!t) return fail(SWT_ERR_INVALID, "null trainer");
  int rc = ensure_device();
  if (rc) return rc;
  // keep the load factor below 1/2 whatever this merge creates
  const uint64_t occ = new_pairs_bound(t, t->h_res.max_count);
  if ((rc = ensure_room(t, occ))) return rc;
  if (t->logging) SWT_HIP(hipMemsetAsync(&t->d_res->n_log, 0, 8, 0));
  if (t->d_sfreq) {
    if (left >= kWpSymCap || right >= kWpSymCap || merged >= kWpSymCap)
      return fail(SWT_ERR_UNSUPPORTED, "WordPiece symbol id beyond %llu", (unsigned long long)kWpSymCap);
    hipLaunchKernelGGL(wp_move_freq_kernel, dim3(1), dim3(1), 0, 0, t->T, left, right, merged, t->d_sfreq);
  }
  prof_begin(0);
  if (t->n_words)
    hipLaunchKernelGGL(apply_kernel, dim3(grid_for(t->n_words, kTrainThreads)), dim3(kTrainThreads), 0, 0, t->d_sym, t->d_woff,
                       t->d_wlen, t->d_freq, t->n_words, left, right, merged, t->T, t->d_res,
                       t->logging ? t->d_log_keys : nullptr, t->logging ? t->d_log_vals : nullptr, t->log_cap,
                       (const StepCmd *)nullptr, t->d_sfreq);
  prof_end(0);
  SWT_HIP(hipGetLastError());
  // n_used may have grown; the next best() refreshes h_res.  Be conservative until then.
  t->h_res.n_used += occ;
  t->n_applied++;
  return SWT_OK;
}

// Up to max_steps iterations of {argmax, tie-break, apply} enqueued back to back: the pair of step i stays on the device
// (decide_kernel -> apply_kernel), only the log comes back.  Step i merges into symbol first_merged + i.
int swt_bpe_train_run(swt_bpe_trainer *t, uint32_t max_steps, uint32_t first_merged, uint32_t *left, uint32_t *right,
                      uint64_t *count, uint32_t *n_done) {
  if (!t || !left || !right || !count || !n_done) return fail(SWT_ERR_INVALID, "null argument");
  if (t->logging) return fail(SWT_ERR_STATE, "swt_bpe_train_run is for unsharded training (deltas are exchanged per step)");
  if (t->d_sfreq && (uint64_t)first_merged + max_steps > kWpSymCap)
    return fail(SWT_ERR_UNSUPPORTED, "WordPiece symbol id beyond %llu", (unsigned long long)kWpSymCap);
  int rc = ensure_device();
  if (rc) return rc;
  *n_done = 0;
  std::vector<StepLog> hlog(kMaxRunSteps);
  uint32_t done = 0;
  unsigned int *ticket = reinterpret_cast<unsigned int *>(t->d_parts + kArgBlocks);
  unsigned int *ticket2 = reinterpret_cast<unsigned int *>(t->d_halt);
  bool exhausted = false;
  while (done < max_steps && !exhausted) {
    uint32_t k = max_steps - done;
    if (k > kRunBatch) k = kRunBatch;
    // room for the whole batch (the symbol count grows by one per step, counts never grow)
    const uint64_t by_sym = 2 * (t->n_base + t->n_applied + k + 1) + 1;
    uint64_t per = new_pairs_bound(t, t->h_res.max_count);
    if (t->h_res.max_count == 0 || by_sym < per) per = by_sym;
    if ((rc = ensure_room(t, per * k))) return rc;
    const uint64_t cap = 1ull << t->T.bits;
    const unsigned g = grid_for(cap, 256 * 8, kArgBlocks);  // 8 counts per thread: 4 double loads
    const unsigned gw = grid_for(t->n_words ? t->n_words : 1, kTrainThreads * 8, 256);
    SWT_HIP(hipMemsetAsync(t->d_halt, 0, 8, 0));
    prof_begin(0);  // one bracket around the whole batch of merge steps: bench.py divides by the merges done
    for (uint32_t i = 0; i < k; i++) {
      hipLaunchKernelGGL(argmax_kernel, dim3(g), dim3(256), 0, 0, t->T.keys, t->T.cnt, cap, t->d_parts, ticket, t->d_res,
                         (const long long *)t->d_sfreq);
      hipLaunchKernelGGL(first_pos_kernel, dim3(gw), dim3(kTrainThreads), 0, 0, t->d_sym, t->d_woff, t->d_wlen, t->n_words, t->T,
                         t->d_res, t->d_cmd, t->d_steplog, i, first_merged + done + i, ticket2, t->d_sfreq);
      if (t->n_words)
        hipLaunchKernelGGL(apply_kernel, dim3(grid_for(t->n_words, kTrainThreads)), dim3(kTrainThreads), 0, 0, t->d_sym, t->d_woff,
                           t->d_wlen, t->d_freq, t->n_words, 0u, 0u, 0u, t->T, t->d_res, (unsigned long long *)nullptr,
                           (long long *)nullptr, (uint64_t)0, (const StepCmd *)t->d_cmd, t->d_sfreq);
    }
    prof_end(0);
    SWT_HIP(hipGetLastError());
    SWT_HIP(hipMemcpy(hlog.data(), t->d_steplog, k * sizeof(StepLog), hipMemcpyDeviceToHost));
    if ((rc = sync_result(t))) return rc;
    uint32_t good = 0;
    while (good < k && hlog[good].flag == 0) {
      left[done] = hlog[good].l;
      right[done] = hlog[good].r;
      count[done] = hlog[good].count;
      done++;
      good++;
    }
    t->n_applied += good;
    if (good && !t->d_sfreq) t->h_res.max_count = hlog[good - 1].count;  // counts never grow: bound for the next batch
    if (good < k) exhausted = true;  // bpe.py:98-99: no pair left (later steps of the batch were no-ops)
  }
  *n_done = done;
  return SWT_OK;
}

int swt_bpe_train_export(swt_bpe_trainer *t, uint32_t *syms, uint64_t syms_cap, uint64_t *word_off, uint32_t *freq) {
  if (!t || !word_off) return fail(SWT_ERR_INVALID, "null argument");
  std::vector<uint32_t> wlen(t->n_words + 1), all(t->n_syms0 + 1);
  std::vector<uint64_t> woff(t->n_words + 1);
  if (t->n_words) SWT_HIP(hipMemcpy(wlen.data(), t->d_wlen, t->n_words * 4, hipMemcpyDeviceToHost));
  SWT_HIP(hipMemcpy(woff.data(), t->d_woff, (t->n_words + 1) * 8, hipMemcpyDeviceToHost));
  if (t->n_syms0) SWT_HIP(hipMemcpy(all.data(), t->d_sym, t->n_syms0 * 4, hipMemcpyDeviceToHost));
  uint64_t o = 0;
  for (uint64_t w = 0; w < t->n_words; w++) {
    word_off[w] = o;
    for (uint32_t i = 0; i < wlen[w]; i++) {
      if (o >= syms_cap) return fail(SWT_ERR_CAPACITY, "syms buffer too small");
      syms[o++] = all[woff[w] + i];
    }
  }
  word_off[t->n_words] = o;
  if (freq && t->n_words) SWT_HIP(hipMemcpy(freq, t->d_freq, t->n_words * 4, hipMemcpyDeviceToHost));
  return SWT_OK;
}

int swt_bpe_train_histogram(swt_bpe_trainer *t, uint64_t *keys, uint64_t *counts, uint64_t cap, uint64_t *n) {
  if (!t || !n) return fail(SWT_ERR_INVALID, "null argument");
  int rc;
  if ((rc = t->tmp.reserve(cap * 16 + 16))) return rc;
  unsigned long long *d_n = t->tmp.as<unsigned long long>();
  unsigned long long *d_k = d_n + 1;
  long long *d_v = reinterpret_cast<long long *>(d_k + cap);
  SWT_HIP(hipMemset(d_n, 0, 8));
  const uint64_t tcap = 1ull << t->T.bits;
  hipLaunchKernelGGL(table_export_kernel, dim3(grid_for(tcap, 256, 4096)), dim3(256), 0, 0, t->T.keys, t->T.cnt, tcap, d_k, d_v, cap, d_n);
  unsigned long long got = 0;
  SWT_HIP(hipMemcpy(&got, d_n, 8, hipMemcpyDeviceToHost));
  *n = got;
  if (got > cap) return fail(SWT_ERR_CAPACITY, "histogram holds %llu live pairs", got);
  if (got) {
    SWT_HIP(hipMemcpy(keys, d_k, got * 8, hipMemcpyDeviceToHost));
    SWT_HIP(hipMemcpy(counts, d_v, got * 8, hipMemcpyDeviceToHost));
  }
  return SWT_OK;
}

// ---- sharded training -----------------------------------------------------------------------------

int swt_bpe_train_take_deltas(swt_bpe_trainer *t, uint64_t *d_keys, int64_t *d_vals, uint64_t cap, uint64_t *n, void *stream) {
  if (!t || !n) return fail(SWT_ERR_INVALID, "null argument");
  hipStream_t st = (hipStream_t)stream;
  if (!t->logging) {
    // first call: switch logging on and hand out the whole local histogram as the initial "delta"
    t->log_cap = 4 * t->n_syms0 + 1024;
    SWT_HIP(hipMalloc((void **)&t->d_log_keys, t->log_cap * 8));
    SWT_HIP(hipMalloc((void **)&t->d_log_vals, t->log_cap * 8));
    t->logging = true;
    SWT_HIP(hipMemsetAsync(&t->d_res->n_log, 0, 8, st));
    const uint64_t tcap = 1ull << t->T.bits;
    hipLaunchKernelGGL(table_export_kernel, dim3(grid_for(tcap, 256, 4096)), dim3(256), 0, st, t->T.keys, t->T.cnt, tcap,
                       t->d_log_keys, t->d_log_vals, t->log_cap, &t->d_res->n_log);
  }
  unsigned long long got = 0;
  SWT_HIP(hipMemcpyAsync(&got, &t->d_res->n_log, 8, hipMemcpyDeviceToHost, st));
  SWT_HIP(hipStreamSynchronize(st));
  *n = got;
  if (got > t->log_cap) return fail(SWT_ERR_CAPACITY, "delta log overflow (%llu entries)", got);
  if (got > cap) return fail(SWT_ERR_CAPACITY, "delta buffer too small: need %llu entries", got);
  if (got) {
    SWT_HIP(hipMemcpyAsync(d_keys, t->d_log_keys, got * 8, hipMemcpyDeviceToDevice, st));
    SWT_HIP(hipMemcpyAsync(d_vals, t->d_log_vals, got * 8, hipMemcpyDeviceToDevice, st));
  }
  return SWT_OK;
}

int swt_bpe_train_add_remote(swt_bpe_trainer *t, const uint64_t *d_keys, const int64_t *d_vals, uint64_t n, void *stream) {
  if (!t || (n && (!d_keys || !d_vals))) return fail(SWT_ERR_INVALID, "null argument");
  if (!n) return SWT_OK;
  int rc = sync_result(t);
  if (rc) return rc;
  const uint64_t cap = 1ull << t->T.bits;
  if (2 * (t->h_res.n_used + n + 64) > cap) {
    uint32_t bits = t->T.bits;
    while ((1ull << bits) < 4 * (t->h_res.n_used + n + 64)) bits++;
    if (bits > 32) return fail(SWT_ERR_UNSUPPORTED, "pair table would exceed 2^32 slots");
    if ((rc = table_resize(t, bits))) return rc;
  }
  hipLaunchKernelGGL(add_remote_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const unsigned long long *>(d_keys), reinterpret_cast<const long long *>(d_vals), n, t->T, t->d_res);
  SWT_HIP(hipGetLastError());
  return SWT_OK;
}

}  // extern "C"
