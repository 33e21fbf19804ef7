// swt_wp.hip -- FastWP (end-to-end LinMaxMatch WordPiece) on gfx950.
//
// Replaces:
//   TrieNode / WPTrie_E2E.insert / precompute   /root/reference/source/utils.py:44-139   (host, then flattened)
//   FastWP.tokenize / matchloop / iswdbndry / ispunc   /root/reference/source/wordpiece.py:233-316
//   the NaiveWP.encode_word("##") corner               /root/reference/source/wordpiece.py:260-261, 132-159
//
// Flattened trie in HBM (L2-resident: ~1.6 MB at 20k vocab):
//   edges  open-addressing hash of packed 64-bit entries  node:21 | code point:21 | child:21  (one 8-byte load
//          per probe; key = the upper 42 bits)
//   nodes  16 bytes each {failure link, pop offset, pop count, flags}
//   pops   token ids of every failure_pops list, concatenated
// Node ids: 0 = root, 1 = root_p (detached, childless), 2.. in creation order; root_sharp is the node of "##".
#include <cstring>
#include <unordered_map>

#include "swt_dedup.h"
#include "swt_tile.h"
#include "swt_words.h"

namespace swt {

struct alignas(16) WpNode {
  int32_t link;      // failure_link, -1 = None
  uint32_t pop_off;  // into pops[]
  uint32_t pop_cnt;
  uint32_t flags;    // bit0 is_end
};

constexpr uint32_t kWpRoot = 0, kWpRootP = 1;
constexpr uint32_t kNodeBits = 21, kMaxNodes = (1u << kNodeBits) - 2;
constexpr uint64_t kEdgeEmpty = ~0ull;
constexpr uint8_t kPySpace = 1, kPyAlnum = 2;

__host__ __device__ inline uint64_t edge_key(uint32_t node, uint32_t cp) { return ((uint64_t)node << 21) | cp; }

struct WpDev {
  const uint64_t *edges;
  uint32_t edge_bits;
  const WpNode *nodes;
  const uint32_t *pops;
  uint32_t root_sharp;
  uint32_t unk_id;        // "['UNK']"
  uint32_t corner_id;     // single id emitted for the '##' corner (token, "[UNK]" or the marker)
  uint32_t corner_nonterm;  // 1: the reference never returns from NaiveWP.encode_word("##")
  uint32_t empty_status;    // status of an empty sentence: s = " " raises IndexError iff the root has a ' ' edge
};

__device__ __forceinline__ int32_t edge_lookup(const WpDev &T, uint32_t node, uint32_t cp) {
  const uint64_t key = edge_key(node, cp);
  const uint32_t mask = (1u << T.edge_bits) - 1u;
  uint32_t h = hash_slot(key, T.edge_bits);
  for (;;) {
    const uint64_t e = T.edges[h];
    if ((e >> 21) == key) return (int32_t)(e & 0x1FFFFFu);
    if (e == kEdgeEmpty) return -1;
    h = (h + 1) & mask;
  }
}

// Character source of the walkers: UTF-8 bytes (LDS-staged chunk or global memory), decoded on the fly; classes of
// U+0000..U+03FF from the LDS copy when there is one.  [b, e) is the sentence; positions are byte offsets.
struct TxtSrc {
  const uint8_t *txt;
  const uint8_t *cls_lo;   // may be null
  const uint8_t *cls_tab;
  __device__ __forceinline__ void load(uint64_t p, uint64_t e, uint32_t &cp, uint32_t &cc, uint32_t &len) const {
    const uint8_t b = txt[p];
    int n = utf8_len(b);
    if (p + n > e) n = (int)(e - p);
    cp = b;
    if (b >= 0x80 && n > 1) {
      cp = b & (0xFF >> (n + 1));
      for (int i = 1; i < n; i++) cp = (cp << 6) | (txt[p + i] & 0x3F);
    }
    const uint8_t c = (cls_lo && cp < 1024u) ? cls_lo[cp] : (cp < kNumCodePoints ? cls_tab[cp] : (uint8_t)0);
    cc = (c >> 2) & 3;
    uint64_t q = p + n;
    while (q < e && utf8_is_cont(txt[q])) q++;  // stray continuation bytes ride with the previous char
    len = (uint32_t)(q - p);
  }
};

// State of a walk through one sentence occupying bytes [b, e): i == e is the appended space (wordpiece.py:248),
// i == e + 1 is len(s).
template <class Src>
struct WpWalk {
  const Src &src;
  const WpDev &T;
  uint64_t e, i;
  uint32_t cp, cc, len;
  bool prev_punc;  // ispunc(s[i-1]); never across a sentence start
  __device__ __forceinline__ WpWalk(const Src &s, const WpDev &t, uint64_t start, uint64_t end, bool pp)
      : src(s), T(t), e(end), i(start), cp(' '), cc(kPySpace), len(1), prev_punc(pp) {
    if (i < e) src.load(i, e, cp, cc, len);
  }
  __device__ __forceinline__ void adv() {
    prev_punc = (cc & (kPySpace | kPyAlnum)) == 0;
    if (i < e) { i += len; if (i > e) i = e; }
    else i = e + 1;
    if (i < e) src.load(i, e, cp, cc, len);
    else { cp = ' '; cc = kPySpace; len = 1; }
  }
  __device__ __forceinline__ bool bndry() const {  // wordpiece.py:285
    return prev_punc || (cc & kPySpace) || (cc & (kPySpace | kPyAlnum)) == 0;
  }
  // One iteration of the loop at wordpiece.py:251-269: match a segment from i, emit its tokens to out[0..) (at most
  // `room` are stored), move i to the start of the next segment.  Returns the token count; status != OK aborts.
  template <class Out>
  __device__ __forceinline__ uint32_t segment(Out out, uint32_t room, int &status) {
    const uint64_t seg_i = i;
    uint32_t nt = 0;
    uint32_t node = kWpRoot;
    bool stop = false;
    while (i <= e) {  // matchloop, wordpiece.py:291-316
      int32_t child = edge_lookup(T, node, cp);
      while (child < 0) {
        const WpNode nd = T.nodes[node];
        if (nd.link < 0) { stop = true; break; }
        for (uint32_t k = 0; k < nd.pop_cnt; k++) {
          if (nt < room) out[nt] = T.pops[nd.pop_off + k];
          nt++;
        }
        node = (uint32_t)nd.link;
        child = edge_lookup(T, node, cp);
      }
      if (stop) break;
      node = (uint32_t)child;
      adv();
    }
    if (i > e) { status = SWT_WP_INDEXERROR; return 0; }  // iswdbndry indexes seq[len(seq)] (wordpiece.py:285)
    const bool root_like = node == kWpRoot || node == T.root_sharp || node == kWpRootP;
    if (!bndry() || !root_like) {  // wordpiece.py:255-257: the segment's tokens are replaced by the one literal
      for (uint32_t k = 1; k < nt && k < room; k++) out[k] = kInvalidTok;
      if (room) out[0] = T.unk_id;
      nt = 1;
    } else if (node == T.root_sharp && nt == 0) {  // wordpiece.py:260-261
      if (T.corner_nonterm) { status = SWT_WP_NONTERMINATING; return 0; }
      if (room) out[0] = T.corner_id;
      nt = 1;
    }
    while (i <= e && !bndry()) adv();          // wordpiece.py:265-266
    while (i <= e && (cc & kPySpace)) adv();   // wordpiece.py:268-269
    if (i == seg_i) { status = SWT_WP_NONTERMINATING; return 0; }  // same state again: the reference spins
    return nt;
  }
};

// FastWP.tokenize on one whole sentence, segment after segment (the sequential form: sentences the parallel form
// cannot certify, and sentences longer than a chunk).  out[k] receives the k-th id.  Returns the token count
// (0 when status != OK).  wordpiece.py:248-270.
template <class Src, class Out>
__device__ uint32_t wp_sentence(const Src &src, uint64_t b, uint64_t e, Out out, const WpDev &T, int &status) {
  WpWalk<Src> w(src, T, b, e, false);
  uint32_t nt = 0;
  status = SWT_WP_OK;
  while (w.i <= e) {  // wordpiece.py:251
    nt += w.segment(out + nt, 0xFFFFFFFFu, status);
    if (status != SWT_WP_OK) return 0;
  }
  return nt;
}

constexpr int kWpTile = 512;    // bytes of sentence starts per tile
constexpr int kWpCap = 1024;    // staged bytes per chunk
constexpr int kWpBlocks = kWpCap / 64;
constexpr int kWpClsLds = 1024;
constexpr uint64_t kWpDirectBytes = 2048, kWpDirectSents = 64;  // up to here one workgroup and one launch do the whole call
constexpr uint32_t kWpUTile = 256;       // smallest tile of the encode over the unique chunks (dedup path)
constexpr uint64_t kWpUMaxTiles = 8192;  // its fixed launch size

struct WpGiant { uint64_t end; uint32_t ntok; uint32_t nsent; };

struct WpLds {
  __attribute__((aligned(16))) uint8_t txt[kWpCap + 16];
  uint32_t tok[kWpCap];                      // per byte position: a token id or kInvalidTok
  uint16_t cand[kWpCap];                     // segment-start candidates, in position order
  unsigned long long sbits[kWpBlocks + 1];   // sentence-start bit per byte
  unsigned long long ppunc[kWpBlocks + 1];   // the char before this byte (same sentence) is punctuation-class
  unsigned long long irr[kWpBlocks + 1];     // per sentence-start position: needs the sequential walk
  unsigned long long vmask[kWpBlocks + 1];
  uint32_t blkpre[kWpBlocks + 1];
  __attribute__((aligned(16))) uint8_t cls_lo[kWpClsLds];
  WpGiant giant;
};

__device__ __forceinline__ bool wbit(const unsigned long long *m, uint32_t p) { return (m[p >> 6] >> (p & 63)) & 1ull; }

// One 64-lane wavefront per tile (workgroup = one wave).  Segments of a sentence depend on each other only through
// where the previous one ended (wordpiece.py:265-269), and that is almost always the next static boundary.  So:
//   B  64 bytes per step: classes (str.isspace / str.isalnum) -> ballot masks -> the positions where a segment CAN
//      start (sentence start; a non-space char after a space; either side of a punctuation-class char)
//   C  one lane per candidate walks the trie from there (matchloop + validity + skip), writing its tokens into its own
//      territory [candidate, next candidate); it certifies itself when it ended exactly at the next candidate
//   D  a sentence with an uncertified candidate (a vocabulary entry spanning a boundary, a non-terminating or raising
//      input) is redone by one lane with the sequential walker -- exactness never rests on the speculation
//   E/F  ballot compaction to the tile's output run, per-sentence offsets and statuses
__global__ __launch_bounds__(64) void wp_encode_kernel(
    const uint8_t *__restrict__ text, uint64_t n_bytes, const uint64_t *__restrict__ sent_off,
    const uint64_t *__restrict__ plan, const uint8_t *__restrict__ cls_tab, WpDev T, uint32_t *__restrict__ scratch,
    uint32_t *__restrict__ sent_local, uint32_t *__restrict__ tile_tok, uint8_t *__restrict__ status, DirectOut direct) {
  __shared__ WpLds L;
  const int lane = threadIdx.x;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const uint64_t t = blockIdx.x;
  const uint64_t s_lo = direct.off ? 0 : plan[t], s_hi = direct.off ? direct.n_sent : plan[t + 1];
  if (s_lo == s_hi) {
    if (lane == 0) tile_tok[t] = 0;
    return;
  }
  reinterpret_cast<uint4 *>(L.cls_lo)[lane] = reinterpret_cast<const uint4 *>(cls_tab)[lane];
  const uint64_t span_base = sent_off[s_lo], span_end = sent_off[s_hi];
  uint32_t *const tile_out = scratch + span_base;
  uint32_t run = 0;
  uint64_t s_next = s_lo;
  uint64_t cb = span_base;

  for (;;) {
    const uint64_t abase = cb & ~15ull;
    const uint32_t off0 = (uint32_t)(cb - abase);
    const uint64_t avail = span_end - abase;
    const bool last = avail <= (uint64_t)kWpCap;
    const uint32_t staged = last ? (uint32_t)avail : (uint32_t)kWpCap;
    const uint32_t nblk = (staged + 63) >> 6;

    // ---- A. stage
    for (uint32_t c = lane * 16; c < staged; c += 64 * 16) {
      const uint64_t g = abase + c;
      if (g + 16 <= n_bytes && ((reinterpret_cast<uintptr_t>(text + g) & 15) == 0)) {
        *reinterpret_cast<uint4 *>(&L.txt[c]) = *reinterpret_cast<const uint4 *>(text + g);
      } else {
        for (int i = 0; i < 16; i++) L.txt[c + i] = (g + i < n_bytes) ? text[g + i] : (uint8_t)' ';
      }
    }
    if (lane <= kWpBlocks) { L.sbits[lane] = 0ull; L.irr[lane] = 0ull; }
    __syncthreads();
    // sentence starts inside the staged bytes; the chunk ends at the last one that leaves its predecessor whole
    int cut = -1;
    uint32_t n_in = 0;  // sentences starting in [cb, abase + staged)
    for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
      const uint64_t o = sent_off[s];
      if (o >= abase + staged) break;
      atomicOr(&L.sbits[(o - abase) >> 6], 1ull << ((o - abase) & 63));
      if (o > cb) cut = (int)(o - abase);
      n_in++;
    }
    for (int d = 32; d >= 1; d >>= 1) {
      cut = max(cut, __shfl_xor(cut, d));
      n_in += __shfl_xor(n_in, d);
    }
    __syncthreads();

    uint32_t ce = staged;
    if (!last) {
      // does a sentence start exactly at the end of the staged bytes?  then everything staged is whole
      uint64_t s_after = s_next + n_in;
      const bool whole = s_after < s_hi && sent_off[s_after] == abase + staged;
      if (whole) ce = staged;
      else if (cut >= 0) ce = (uint32_t)cut;
      else {
        // one sentence longer than the LDS chunk: one lane walks it in global memory
        if (lane == 0) {
          uint64_t s = s_next;
          while (s + 1 < s_hi && sent_off[s + 1] <= cb) s++;  // the last sentence that starts at cb
          int stt;
          TxtSrc src{text, nullptr, cls_tab};
          const uint64_t e = sent_off[s + 1];
          const uint32_t n = wp_sentence(src, cb, e, tile_out + run, T, stt);
          // empty sentences that also start at cb come first and get no tokens
          for (uint64_t z = s_next; z <= s; z++) {
            if (direct.off) direct.off[z] = run; else sent_local[z] = run;
            status[z] = (uint8_t)T.empty_status;
          }
          status[s] = (uint8_t)stt;
          L.giant.end = e;
          L.giant.ntok = n;
          L.giant.nsent = (uint32_t)(s - s_next + 1);
        }
        __syncthreads();
        s_next += L.giant.nsent;
        run += L.giant.ntok;
        cb = L.giant.end;
        __syncthreads();
        if (cb >= span_end) {
          // trailing empty sentences at the very end of the span
          for (uint64_t z = s_next + lane; z < s_hi; z += 64) {
            if (direct.off) direct.off[z] = run; else sent_local[z] = run;
            status[z] = (uint8_t)T.empty_status;
          }
          break;
        }
        continue;
      }
    }

    // ---- B. classes -> masks -> candidates
    uint32_t nc = 0;
    bool prev_sp = true, prev_pu = false;  // class of the char owning the byte before this block
    for (uint32_t blk = 0; blk < nblk; blk++) {
      const uint32_t p = blk * 64 + lane;
      const bool inr = p >= off0 && p < ce;
      const uint8_t b = inr ? L.txt[p] : (uint8_t)' ';
      const bool lead = !utf8_is_cont(b);
      uint32_t cp = b;
      if (b >= 0xC0) {
        int len = utf8_len(b);
        if (p + len > ce) len = (int)(ce - p);
        if (len > 1) {
          cp = b & (0xFF >> (len + 1));
          for (int i = 1; i < len; i++) cp = (cp << 6) | (L.txt[p + i] & 0x3F);
        }
      }
      uint8_t c = SWT_CLS_PY_SPACE;  // bytes outside the chunk behave as spaces
      if (inr && lead) c = cp < (uint32_t)kWpClsLds ? L.cls_lo[cp] : (cp < kNumCodePoints ? cls_tab[cp] : (uint8_t)0);
      const unsigned long long INR = __ballot(inr);
      const unsigned long long LEAD = __ballot(lead);
      const unsigned long long SPm = __ballot(lead && (c & SWT_CLS_PY_SPACE));
      const unsigned long long ANm = __ballot(lead && (c & SWT_CLS_PY_ALNUM));
      const unsigned long long PUm = LEAD & ~SPm & ~ANm;
      const unsigned long long CONT = ~LEAD;
      unsigned long long SPb = SPm | ((prev_sp && (CONT & 1ull)) ? 1ull : 0ull);
      unsigned long long PUb = PUm | ((prev_pu && (CONT & 1ull)) ? 1ull : 0ull);
      SPb |= (SPb << 1) & CONT; SPb |= (SPb << 1) & CONT; SPb |= (SPb << 1) & CONT;
      PUb |= (PUb << 1) & CONT; PUb |= (PUb << 1) & CONT; PUb |= (PUb << 1) & CONT;
      const unsigned long long SS = L.sbits[blk] & INR;
      const unsigned long long before_sp = ((SPb << 1) | (prev_sp ? 1ull : 0ull)) & ~SS;  // no context across a sentence start
      const unsigned long long before_pu = ((PUb << 1) | (prev_pu ? 1ull : 0ull)) & ~SS;
      const unsigned long long CAND = INR & (SS | (LEAD & ~SPm & (before_sp | before_pu | PUm)));
      if (lane == 0) L.ppunc[blk] = before_pu;
      if ((CAND >> lane) & 1ull) L.cand[nc + __popcll(CAND & lt)] = (uint16_t)p;
      nc += __popcll(CAND);
      L.tok[p] = kInvalidTok;
      prev_sp = (SPb >> 63) & 1ull;
      prev_pu = (PUb >> 63) & 1ull;
    }
    __syncthreads();

    // ---- C. one lane per candidate
    for (uint32_t k = lane; k < nc; k += 64) {
      const uint32_t p0 = L.cand[k];
      // the sentence around p0: [s0, e)
      uint32_t s0, e;
      {
        int w = (int)(p0 >> 6);
        unsigned long long m = L.sbits[w] & ((2ull << (p0 & 63)) - 1ull);
        while (!m && w > 0) m = L.sbits[--w];
        s0 = m ? (uint32_t)(w * 64 + 63 - __builtin_clzll(m)) : off0;
        w = (int)(p0 >> 6);
        m = (p0 & 63) == 63 ? 0ull : (L.sbits[w] & ~((2ull << (p0 & 63)) - 1ull));
        while (!m && w + 1 < (int)nblk) m = L.sbits[++w];
        e = m ? (uint32_t)(w * 64 + __builtin_ctzll(m)) : ce;
        if (e > ce) e = ce;
      }
      const bool has_succ = k + 1 < nc && L.cand[k + 1] < e;
      const uint32_t terr_end = has_succ ? L.cand[k + 1] : e;
      const uint64_t want_next = has_succ ? L.cand[k + 1] : (uint64_t)e + 1;
      TxtSrc src{L.txt, L.cls_lo, cls_tab};
      WpWalk<TxtSrc> w(src, T, p0, e, p0 != s0 && wbit(L.ppunc, p0));
      int stt = SWT_WP_OK;
      const uint32_t n = w.segment(&L.tok[p0], terr_end - p0, stt);
      if (stt != SWT_WP_OK || w.i != want_next || n > terr_end - p0) atomicOr(&L.irr[s0 >> 6], 1ull << (s0 & 63));
    }
    __syncthreads();

    // ---- D. per sentence: status; the sequential walk where the speculation was not certified
    for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
      const uint64_t o = sent_off[s];
      const uint64_t rel = o - abase;
      if (rel > ce || (rel == ce && !last)) break;
      const uint64_t e = sent_off[s + 1] - abase;  // <= ce: chunks end at sentence starts
      int stt = T.empty_status;
      if (rel < e) {
        stt = SWT_WP_OK;
        if (wbit(L.irr, (uint32_t)rel)) {
          TxtSrc src{L.txt, L.cls_lo, cls_tab};
          // tokens are staged in the global output run first (the territory array still feeds no one, but the walker
          // must not overwrite text it has not read: it only reads L.txt, so L.tok is free to take them)
          const uint32_t n = wp_sentence(src, rel, e, &L.tok[rel], T, stt);
          for (uint64_t q = rel + n; q < e; q++) L.tok[q] = kInvalidTok;
        }
      }
      status[s] = (uint8_t)stt;
    }
    __syncthreads();

    // ---- E. compaction, F. sentence offsets
    uint32_t total = 0;
    for (uint32_t blk = 0; blk < nblk; blk++) {
      const uint32_t p = blk * 64 + lane;
      const uint32_t sv = (p >= off0 && p < ce) ? L.tok[p] : kInvalidTok;
      const unsigned long long m = __ballot(sv != kInvalidTok);
      if (lane == 0) { L.vmask[blk] = m; L.blkpre[blk] = total; }
      if (sv != kInvalidTok) tile_out[run + total + __popcll(m & lt)] = sv;
      total += __popcll(m);
    }
    __syncthreads();
    uint32_t mine = 0;
    for (uint64_t s = s_next + lane; s < s_hi; s += 64) {
      const uint64_t rel = sent_off[s] - abase;
      if (rel > ce || (rel == ce && !last)) break;
      uint32_t ex = total;
      if (rel < ce && (rel >> 6) < nblk) ex = L.blkpre[rel >> 6] + __popcll(L.vmask[rel >> 6] & ((1ull << (rel & 63)) - 1ull));
      if (direct.off) direct.off[s] = run + ex; else sent_local[s] = run + ex;
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
    if (direct.off) { direct.off[s_hi] = run; *direct.n_tokens = run; }
    else tile_tok[t] = run;
  }
}

// ---- host: trie build (utils.py:75-139) and flattening -------------------------------------------

struct HostTrie {
  std::vector<uint32_t> ch;
  std::vector<uint8_t> is_end;
  std::vector<int32_t> tok, link, parent;
  std::vector<std::vector<uint32_t>> pops, kids;
  std::unordered_map<uint64_t, uint32_t> edges;
  uint32_t root = 0, root_p = 1, root_sharp = 0;
  uint32_t n_vocab = 0;
  std::vector<uint32_t> corner;
  bool corner_nonterm = false;

  uint32_t new_node(uint32_t c, int32_t par) {
    ch.push_back(c); is_end.push_back(0); tok.push_back(-1); link.push_back(-1); parent.push_back(par);
    pops.emplace_back(); kids.emplace_back();
    return (uint32_t)ch.size() - 1;
  }
  int32_t child(uint32_t node, uint32_t c) const {
    auto it = edges.find(edge_key(node, c));
    return it == edges.end() ? -1 : (int32_t)it->second;
  }
  // utils.py:87-105
  uint32_t insert(const uint32_t *s, uint64_t n) {
    uint32_t node = root;
    for (uint64_t i = 0; i < n; i++) {
      int32_t c = child(node, s[i]);
      if (c < 0) {
        c = (int32_t)new_node(s[i], (int32_t)node);
        edges.emplace(edge_key(node, s[i]), (uint32_t)c);
        kids[node].push_back((uint32_t)c);
      }
      node = (uint32_t)c;
    }
    is_end[node] = 1;
    return node;
  }
};

static int build_trie(HostTrie &H, const uint32_t *blob, const uint64_t *off, uint32_t n_vocab) {
  const uint8_t *cls = host_class_table();
  H.n_vocab = n_vocab;
  H.root = H.new_node(0, -1);    // utils.py:77
  H.root_p = H.new_node(0, -1);  // utils.py:79
  const uint32_t sharp[2] = {'#', '#'};
  H.root_sharp = H.insert(sharp, 2);  // utils.py:81
  for (uint32_t v = 0; v < n_vocab; v++) {  // utils.py:83-84
    for (uint64_t i = off[v]; i < off[v + 1]; i++)
      if (blob[i] >= kNumCodePoints) return fail(SWT_ERR_INVALID, "vocab entry %u holds an invalid code point", v);
    const uint32_t node = H.insert(blob + off[v], off[v + 1] - off[v]);
    if (H.tok[node] < 0) H.tok[node] = (int32_t)v;
    if (H.ch.size() > kMaxNodes) return fail(SWT_ERR_UNSUPPORTED, "trie larger than %u nodes", kMaxNodes);
  }
  // utils.py:108-139: BFS from [root, root_sharp]
  std::vector<uint32_t> queue{H.root, H.root_sharp};
  for (size_t qh = 0; qh < queue.size(); qh++) {
    const uint32_t u = queue[qh];
    for (uint32_t c : H.kids[u]) {
      if (c == H.root_sharp) continue;  // :119-120
      const uint32_t chr = H.ch[c];
      if (H.is_end[c]) {  // :121-123
        H.link[c] = (int32_t)H.root_sharp;
        H.pops[c] = {(uint32_t)H.tok[c]};
      } else {  // :124-132
        int32_t f = H.link[u];
        std::vector<uint32_t> acc;
        while (f >= 0 && H.child((uint32_t)f, chr) < 0) {
          acc.insert(acc.end(), H.pops[f].begin(), H.pops[f].end());
          f = H.link[f];
        }
        if (f >= 0) {
          H.link[c] = H.child((uint32_t)f, chr);
          H.pops[c] = H.pops[u];
          H.pops[c].insert(H.pops[c].end(), acc.begin(), acc.end());
        }
      }
      if (!(cls[chr] & SWT_CLS_PY_ALNUM)) H.link[c] = (int32_t)H.root_p;  // :136-137
      queue.push_back(c);
    }
  }
  // NaiveWP.encode_word("##") (wordpiece.py:144-159): the word is a run of '#'; state = its length L
  {
    std::vector<uint32_t> chain{H.root};
    for (;;) {
      const int32_t c = H.child(chain.back(), '#');
      if (c < 0) break;
      chain.push_back((uint32_t)c);
    }
    const uint64_t D = chain.size() - 1;
    uint64_t L = 2, guard = 0;
    std::vector<uint8_t> visited(D + 8, 0);
    for (;;) {
      uint64_t i = L < D ? L : D;
      while (i > 0 && !(H.is_end[chain[i]] && H.tok[chain[i]] >= 0)) i--;
      if (i == 0) { H.corner = {n_vocab + 1}; break; }  // :148-149 ["[UNK]"]
      H.corner.push_back((uint32_t)H.tok[chain[i]]);
      L -= i;
      if (L == 0) break;
      L += 2;  // :155-156
      if (L < D + 8) {
        if (visited[L]) { H.corner_nonterm = true; break; }
        visited[L] = 1;
      }
      if (++guard > 1000000) { H.corner_nonterm = true; break; }
    }
    if (H.corner_nonterm) H.corner.clear();
  }
  return SWT_OK;
}

// After the encode over the unique chunks (every chunk a "sentence" of that launch): each chunk's token run -- its place in
// the launch's scratch and its length, or kRecFailed when the reference never returns on it -- goes to its table slot.
__global__ __launch_bounds__(64) void wp_urec_kernel(const uint64_t *__restrict__ uoff, const uint64_t *__restrict__ plan,
                                                     const uint32_t *__restrict__ sent_local, const uint32_t *__restrict__ tile_tok,
                                                     const uint8_t *__restrict__ status, const uint32_t *__restrict__ uslot,
                                                     unsigned long long *__restrict__ rec, unsigned long long *__restrict__ drec) {
  const uint64_t t = blockIdx.x;
  const uint64_t s_lo = plan[t], s_hi = plan[t + 1];
  if (s_lo == s_hi) return;
  const uint64_t span_base = uoff[s_lo];
  const uint32_t total = tile_tok[t];
  for (uint64_t s = s_lo + threadIdx.x; s < s_hi; s += 64) {
    const uint32_t a = sent_local[s], b = s + 1 < s_hi ? sent_local[s + 1] : total;
    const uint32_t cnt = status[s] != SWT_WP_OK ? kRecFailed : b - a;
    drec[s] = (span_base + a) | ((unsigned long long)cnt << 32);  // dense, for the last pass
    rec[uslot[s]] = s | ((unsigned long long)cnt << 32);           // the counting pass finds the unique index here
  }
}

}  // namespace swt

using namespace swt;

struct swt_wp_trie {
  HostTrie H;
  // flattened
  std::vector<uint64_t> h_edges;
  std::vector<WpNode> h_nodes;
  std::vector<uint32_t> h_pops;
  uint32_t edge_bits = 0;
  // device (uploaded on first encode)
  uint64_t *d_edges = nullptr;
  WpNode *d_nodes = nullptr;
  uint32_t *d_pops = nullptr;
  TileWorkspace ws;
  DevBuf in_text, in_off, out_ids, out_off, out_status, n_tok;
  // word-level dedup inside one call (swt_dedup.h): possible when no vocabulary token holds a str.isspace character
  bool dedup_ok = false;
  DedupEngine dd;
  TileWorkspace ws2;  // the encode over the unique chunks
  DevBuf u_status;
  PinnedBuf pin;  // small host calls: one copy each way
  DevBuf small_in, small_out;
};

static int wp_upload(swt_wp_trie *t) {
  if (t->d_edges) return SWT_OK;
  int rc = ensure_device();
  if (rc) return rc;
  SWT_HIP(hipMalloc((void **)&t->d_edges, t->h_edges.size() * 8));
  SWT_HIP(hipMemcpy(t->d_edges, t->h_edges.data(), t->h_edges.size() * 8, hipMemcpyHostToDevice));
  SWT_HIP(hipMalloc((void **)&t->d_nodes, t->h_nodes.size() * sizeof(WpNode)));
  SWT_HIP(hipMemcpy(t->d_nodes, t->h_nodes.data(), t->h_nodes.size() * sizeof(WpNode), hipMemcpyHostToDevice));
  SWT_HIP(hipMalloc((void **)&t->d_pops, (t->h_pops.size() + 1) * 4));
  if (!t->h_pops.empty())
    SWT_HIP(hipMemcpy(t->d_pops, t->h_pops.data(), t->h_pops.size() * 4, hipMemcpyHostToDevice));
  return SWT_OK;
}

extern "C" {

int swt_wp_trie_create(const uint32_t *vocab_cps, const uint64_t *vocab_off, uint32_t n_vocab, swt_wp_trie **out) try {
  if (!out || !vocab_off || (n_vocab && vocab_off[n_vocab] && !vocab_cps)) return fail(SWT_ERR_INVALID, "null argument");
  if (n_vocab > 0x7FFFFFF0u) return fail(SWT_ERR_INVALID, "vocabulary too large");
  auto *t = new swt_wp_trie();
  int rc = build_trie(t->H, vocab_cps, vocab_off, n_vocab);
  if (rc) { delete t; return rc; }
  const HostTrie &H = t->H;
  const size_t n_nodes = H.ch.size();
  // flatten: nodes + pops
  t->h_nodes.resize(n_nodes);
  for (size_t k = 0; k < n_nodes; k++) {
    WpNode &nd = t->h_nodes[k];
    nd.link = H.link[k];
    nd.pop_off = (uint32_t)t->h_pops.size();
    nd.pop_cnt = (uint32_t)H.pops[k].size();
    nd.flags = H.is_end[k];
    t->h_pops.insert(t->h_pops.end(), H.pops[k].begin(), H.pops[k].end());
  }
  // edges
  uint32_t bits = 4;
  while ((1ull << bits) < 2ull * H.edges.size() + 2) bits++;
  t->edge_bits = bits;
  t->h_edges.assign((size_t)1 << bits, kEdgeEmpty);
  const uint32_t mask = (1u << bits) - 1;
  for (const auto &kv : H.edges) {
    uint32_t h = hash_slot(kv.first, bits);
    while (t->h_edges[h] != kEdgeEmpty) h = (h + 1) & mask;
    t->h_edges[h] = (kv.first << 21) | kv.second;
  }
  // A segment of FastWP.tokenize never reads past the whitespace that ends its chunk unless the trie has an edge labelled
  // with a whitespace character (wordpiece.py:291-316 follows edges only): then, and only then, chunks are independent.
  t->dedup_ok = true;
  for (const auto &kv : H.edges) {
    const uint32_t cp = (uint32_t)(kv.first & ((1u << 21) - 1u));
    if (cp < kNumCodePoints && (host_class_table()[cp] & SWT_CLS_PY_SPACE)) t->dedup_ok = false;
  }
  *out = t;
  return SWT_OK;
} SWT_API_CATCH

int swt_wp_trie_set_option(swt_wp_trie *t, int option, int value) try {
  if (!t) return fail(SWT_ERR_INVALID, "null trie");
  switch (option) {
    case SWT_OPT_DEDUP:
      if (value < 0 || value > 2) return fail(SWT_ERR_INVALID, "SWT_OPT_DEDUP takes 0, 1 or 2");
      t->dd.opt_mode = value;
      return SWT_OK;
    case SWT_OPT_DEDUP_TABLE_BITS:
      if (value != 0 && (value < 4 || value > 24)) return fail(SWT_ERR_INVALID, "SWT_OPT_DEDUP_TABLE_BITS takes 0 or 4..24");
      t->dd.opt_table_bits = (uint32_t)value;
      return SWT_OK;
  }
  return fail(SWT_ERR_INVALID, "no such option");
} SWT_API_CATCH

void swt_wp_trie_destroy(swt_wp_trie *t) try {
  if (!t) return;
  if (t->d_edges) (void)hipFree(t->d_edges);
  if (t->d_nodes) (void)hipFree(t->d_nodes);
  if (t->d_pops) (void)hipFree(t->d_pops);
  t->ws.release();
  t->ws2.release();
  t->dd.release();
  t->pin.release();
  t->small_in.release();
  t->small_out.release();
  for (DevBuf *b : {&t->in_text, &t->in_off, &t->out_ids, &t->out_off, &t->out_status, &t->n_tok, &t->u_status}) b->release();
  delete t;
} SWT_API_CATCH_VOID

int swt_wp_trie_stats(const swt_wp_trie *t, uint32_t *n_nodes, uint32_t *n_edges, uint32_t *n_pops) try {
  if (!t) return fail(SWT_ERR_INVALID, "null trie");
  if (n_nodes) *n_nodes = (uint32_t)t->H.ch.size();
  if (n_edges) *n_edges = (uint32_t)t->H.edges.size();
  if (n_pops) *n_pops = (uint32_t)t->h_pops.size();
  return SWT_OK;
} SWT_API_CATCH

int64_t swt_wp_trie_corner(const swt_wp_trie *t, uint32_t *out, uint64_t cap) try {
  if (!t) return -2;
  if (t->H.corner_nonterm) return -1;
  for (size_t k = 0; k < t->H.corner.size() && k < cap; k++) out[k] = t->H.corner[k];
  return (int64_t)t->H.corner.size();
} SWT_API_CATCH

int swt_wp_trie_node(const swt_wp_trie *t, const uint32_t *path, uint64_t path_len, uint32_t *node_id, int32_t *link,
                     uint8_t *is_end, uint32_t *pops, uint32_t pops_cap, uint32_t *n_pops) try {
  if (!t) return fail(SWT_ERR_INVALID, "null trie");
  uint32_t node = t->H.root;
  for (uint64_t i = 0; i < path_len; i++) {
    const int32_t c = t->H.child(node, path[i]);
    if (c < 0) return fail(SWT_ERR_INVALID, "no such trie path");
    node = (uint32_t)c;
  }
  if (node_id) *node_id = node;
  if (link) *link = t->H.link[node];
  if (is_end) *is_end = t->H.is_end[node];
  if (n_pops) *n_pops = (uint32_t)t->H.pops[node].size();
  for (size_t k = 0; k < t->H.pops[node].size() && k < pops_cap; k++) pops[k] = t->H.pops[node][k];
  return SWT_OK;
} SWT_API_CATCH

int swt_wp_trie_node_path(const swt_wp_trie *t, uint32_t node_id, uint32_t *out, uint64_t cap, uint64_t *len) try {
  if (!t || node_id >= t->H.ch.size()) return fail(SWT_ERR_INVALID, "bad node id");
  std::vector<uint32_t> rev;
  for (int32_t n = (int32_t)node_id; n >= 0 && t->H.parent[n] >= 0; n = t->H.parent[n]) rev.push_back(t->H.ch[n]);
  if (len) *len = rev.size();
  for (size_t k = 0; k < rev.size() && k < cap; k++) out[k] = rev[rev.size() - 1 - k];
  return SWT_OK;
} SWT_API_CATCH

int swt_wp_encode_dev(swt_wp_trie *t, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off, uint64_t n_sent,
                      uint32_t *d_out_ids, uint64_t *d_out_off, uint8_t *d_status, uint64_t *d_n_tokens, void *stream) try {
  if (!t || !d_sent_off || !d_out_off || !d_n_tokens || (n_sent && !d_status) || (n_bytes && (!d_text || !d_out_ids)))
    return fail(SWT_ERR_INVALID, "null argument");
  int rc = wp_upload(t);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const uint8_t *d_cls = nullptr;
  if ((rc = device_class_table(&d_cls))) return rc;
  const uint64_t n_tiles = tile_count(n_bytes, kWpTile);
  if (n_tiles > 0x7FFFFFFFull)
    return fail(SWT_ERR_UNSUPPORTED, "text too large for one call (%llu bytes)", (unsigned long long)n_bytes);
  if ((rc = t->ws.reserve(n_bytes, n_sent, n_tiles))) return rc;
  if (n_sent == 0) {
    SWT_HIP(hipMemsetAsync(d_out_off, 0, 8, st));
    SWT_HIP(hipMemsetAsync(d_n_tokens, 0, 8, st));
    return SWT_OK;
  }
  WpDev T;
  T.edges = t->d_edges;
  T.edge_bits = t->edge_bits;
  T.nodes = t->d_nodes;
  T.pops = t->d_pops;
  T.root_sharp = t->H.root_sharp;
  T.unk_id = t->H.n_vocab;
  T.corner_nonterm = t->H.corner_nonterm ? 1u : 0u;
  T.empty_status = t->H.child(t->H.root, ' ') >= 0 ? SWT_WP_INDEXERROR : SWT_WP_OK;
  T.corner_id = t->H.corner.size() == 1 ? t->H.corner[0] : t->H.n_vocab + 2;
  if (n_bytes <= kWpDirectBytes && n_sent <= kWpDirectSents && t->dd.opt_mode != 2) {
    // a sentence or a few: one workgroup, one launch, the caller's arrays written by the kernel (DirectOut, swt_tile.h)
    hipLaunchKernelGGL(wp_encode_kernel, dim3(1), dim3(64), 0, st, d_text, n_bytes, d_sent_off, (const uint64_t *)nullptr, d_cls, T,
                       d_out_ids, t->ws.sent_local.as<uint32_t>(), t->ws.tile_tok.as<uint32_t>(), d_status,
                       DirectOut{d_out_off, d_n_tokens, n_sent});
    SWT_HIP(hipGetLastError());
    return SWT_OK;
  }
  // debug knob 1: bit 0 = never dedup, bit 1 = dedup whatever the batch size (tests)
  if (t->dedup_ok && T.empty_status == SWT_WP_OK && n_bytes <= kDedupMaxBytes && t->dd.opt_mode != 1 &&
      (n_bytes >= kDedupMinBytesWp || t->dd.opt_mode == 2)) {
    // Word-level dedup (swt_dedup.h): the chunks between whitespace are encoded once per call.  The encode over the unique
    // chunks is this same kernel with every chunk as a "sentence"; its launch size is fixed and the tile size follows on
    // the device (the number of unique chunks never reaches the host).
    const uint64_t max_uniq = n_bytes + 2;
    uint64_t n_tiles2 = tile_count(n_bytes, kWpUTile);
    if (n_tiles2 > kWpUMaxTiles) n_tiles2 = kWpUMaxTiles;
    if ((rc = t->ws2.reserve(n_bytes, max_uniq, n_tiles2)) || (rc = t->u_status.reserve(max_uniq + 2))) return rc;
    prof_begin(st, 2);
    if ((rc = dedup_front(t->dd, t->ws, d_text, n_bytes, d_sent_off, n_sent, d_cls, kDedupWp, st, t->ws2.plan.as<uint64_t>(), n_tiles2,
                          kWpUTile)))
      return rc;
    prof_begin(st);
    hipLaunchKernelGGL(wp_encode_kernel, dim3((unsigned)n_tiles2), dim3(64), 0, st, t->dd.utext.as<uint8_t>(), n_bytes,
                       t->dd.uoff.as<uint64_t>(), t->ws2.plan.as<uint64_t>(), d_cls, T, t->ws2.scratch.as<uint32_t>(),
                       t->ws2.sent_local.as<uint32_t>(), t->ws2.tile_tok.as<uint32_t>(), t->u_status.as<uint8_t>(),
                       DirectOut{nullptr, nullptr, 0});
    prof_end(st);
    hipLaunchKernelGGL(wp_urec_kernel, dim3((unsigned)n_tiles2), dim3(64), 0, st, t->dd.uoff.as<uint64_t>(), t->ws2.plan.as<uint64_t>(),
                       t->ws2.sent_local.as<uint32_t>(), t->ws2.tile_tok.as<uint32_t>(), t->u_status.as<uint8_t>(),
                       t->dd.uslot.as<uint32_t>(), t->dd.rec_ptr(), t->dd.drec_ptr());
    rc = dedup_back(t->dd, t->ws, d_sent_off, n_sent, n_bytes, t->ws2.scratch.as<uint32_t>(), kDedupWp, d_status, d_out_ids, d_out_off,
                    d_n_tokens, st);
    prof_end(st, 2);
    return rc;
  }
  prof_begin(st, 2);
  launch_plan(d_sent_off, n_sent, n_tiles, kWpTile, t->ws.plan.as<uint64_t>(), st);
  prof_begin(st);
  hipLaunchKernelGGL(wp_encode_kernel, dim3((unsigned)n_tiles), dim3(64), 0, st, d_text, n_bytes, d_sent_off,
                     t->ws.plan.as<uint64_t>(), d_cls, T, t->ws.scratch.as<uint32_t>(), t->ws.sent_local.as<uint32_t>(),
                     t->ws.tile_tok.as<uint32_t>(), d_status, DirectOut{nullptr, nullptr, 0});
  prof_end(st);
  launch_scan_gather(d_sent_off, n_sent, n_tiles, t->ws, d_out_ids, d_out_off, d_n_tokens, st);
  prof_end(st, 2);
  SWT_HIP(hipGetLastError());
  return SWT_OK;
} SWT_API_CATCH

// text and offsets on the device -> ids, offsets, statuses and the count in the caller's host arrays
static int wp_encode_to_host(swt_wp_trie *t, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_off, uint64_t n_sent,
                             uint32_t *out_ids, uint64_t out_cap, uint64_t *out_off, uint8_t *status, uint64_t *n_tokens) {
  int rc;
  if ((rc = t->out_ids.reserve((n_bytes + 64) * 4))) return rc;
  if ((rc = t->out_off.reserve((n_sent + 1) * 8))) return rc;
  if ((rc = t->out_status.reserve(n_sent + 8))) return rc;
  if ((rc = t->n_tok.reserve(8))) return rc;
  rc = swt_wp_encode_dev(t, d_text, n_bytes, d_off, n_sent, t->out_ids.as<uint32_t>(), t->out_off.as<uint64_t>(), t->out_status.as<uint8_t>(),
                         t->n_tok.as<uint64_t>(), nullptr);
  if (rc) return rc;
  uint64_t nt = 0;
  SWT_HIP(hipMemcpy(&nt, t->n_tok.p, 8, hipMemcpyDeviceToHost));
  *n_tokens = nt;
  SWT_HIP(hipMemcpy(out_off, t->out_off.p, (n_sent + 1) * 8, hipMemcpyDeviceToHost));
  if (n_sent) SWT_HIP(hipMemcpy(status, t->out_status.p, n_sent, hipMemcpyDeviceToHost));
  if (nt > out_cap)
    return fail(SWT_ERR_CAPACITY, "out_ids too small: need %llu ids, have %llu", (unsigned long long)nt, (unsigned long long)out_cap);
  if (nt) SWT_HIP(hipMemcpy(out_ids, t->out_ids.p, nt * 4, hipMemcpyDeviceToHost));
  return SWT_OK;
}

int swt_wp_encode(swt_wp_trie *t, const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent, uint32_t *out_ids,
                  uint64_t out_cap, uint64_t *out_off, uint8_t *status, uint64_t *n_tokens) try {
  if (!t || !sent_off || !out_off || !n_tokens || (n_sent && !status)) return fail(SWT_ERR_INVALID, "null argument");
  int rc = wp_upload(t);
  if (rc) return rc;
  const uint64_t n_bytes = sent_off[n_sent];
  if (sent_off[0] != 0) return fail(SWT_ERR_INVALID, "sent_off[0] must be 0");
  for (uint64_t s = 0; s < n_sent; s++)
    if (sent_off[s] > sent_off[s + 1])
      return fail(SWT_ERR_INVALID, "sentence offsets must be non-decreasing (at %llu)", (unsigned long long)s);
  if (n_bytes && !text) return fail(SWT_ERR_INVALID, "null text");
  if (n_bytes <= kWpDirectBytes && n_sent <= kWpDirectSents && n_sent > 0) {
    // tokenize(text) on one sentence (wordpiece.py:233): the kernels read the text and the offsets from pinned host memory and
    // write the count, the offsets, the statuses and the ids there -- no copy call at all (see swt_bpe_encode)
    const size_t off_bytes = ((n_sent + 1) * 8 + 15) & ~(size_t)15, st_bytes = (n_sent + 15) & ~(size_t)15;
    const size_t text_bytes = (n_bytes + 64 + 15) & ~(size_t)15, out_at = off_bytes + text_bytes;
    if ((rc = t->pin.reserve(out_at + 16 + off_bytes + st_bytes + (n_bytes + 64) * 4))) return rc;
    uint8_t *h = t->pin.as<uint8_t>();
    memcpy(h, sent_off, (n_sent + 1) * 8);
    if (n_bytes) memcpy(h + off_bytes, text, n_bytes);
    memset(h + off_bytes + n_bytes, ' ', 64);
    uint8_t *o = h + out_at;
    rc = swt_wp_encode_dev(t, h + off_bytes, n_bytes, reinterpret_cast<const uint64_t *>(h), n_sent,
                           reinterpret_cast<uint32_t *>(o + 16 + off_bytes + st_bytes), reinterpret_cast<uint64_t *>(o + 16), o + 16 + off_bytes,
                           reinterpret_cast<uint64_t *>(o), nullptr);
    if (rc) return rc;
    SWT_HIP(hipStreamSynchronize(0));
    const uint64_t nt = *reinterpret_cast<const volatile uint64_t *>(o);
    *n_tokens = nt;
    memcpy(out_off, o + 16, (n_sent + 1) * 8);
    if (n_sent) memcpy(status, o + 16 + off_bytes, n_sent);
    if (nt > out_cap)
      return fail(SWT_ERR_CAPACITY, "out_ids too small: need %llu ids, have %llu", (unsigned long long)nt, (unsigned long long)out_cap);
    if (nt) memcpy(out_ids, o + 16 + off_bytes + st_bytes, nt * 4);
    return SWT_OK;
  }
  if (n_bytes <= kSmallCallBytes && n_sent <= kSmallCallSents) {
    // the reference-style call (one sentence, or a few): one copy up (offsets + text), one copy down (count, offsets, statuses, ids)
    const size_t off_bytes = ((n_sent + 1) * 8 + 15) & ~(size_t)15, st_bytes = (n_sent + 15) & ~(size_t)15;
    const size_t in_bytes = off_bytes + n_bytes + 64;
    const size_t out_bytes = 16 + off_bytes + st_bytes + (n_bytes + 64) * 4;
    if ((rc = t->pin.reserve(in_bytes > out_bytes ? in_bytes : out_bytes)) || (rc = t->small_in.reserve(in_bytes)) ||
        (rc = t->small_out.reserve(out_bytes)))
      return rc;
    uint8_t *h = t->pin.as<uint8_t>();
    memcpy(h, sent_off, (n_sent + 1) * 8);
    if (n_bytes) memcpy(h + off_bytes, text, n_bytes);
    SWT_HIP(hipMemcpyAsync(t->small_in.p, h, off_bytes + n_bytes, hipMemcpyHostToDevice, 0));
    uint8_t *d_in = t->small_in.as<uint8_t>(), *d_out = t->small_out.as<uint8_t>();
    rc = swt_wp_encode_dev(t, d_in + off_bytes, n_bytes, reinterpret_cast<const uint64_t *>(d_in), n_sent,
                           reinterpret_cast<uint32_t *>(d_out + 16 + off_bytes + st_bytes), reinterpret_cast<uint64_t *>(d_out + 16),
                           d_out + 16 + off_bytes, reinterpret_cast<uint64_t *>(d_out), nullptr);
    if (rc) return rc;
    SWT_HIP(hipMemcpyAsync(h, d_out, 16 + off_bytes + st_bytes + (n_bytes + 64) * 4, hipMemcpyDeviceToHost, 0));
    SWT_HIP(hipStreamSynchronize(0));
    const uint64_t nt = *reinterpret_cast<const uint64_t *>(h);
    *n_tokens = nt;
    memcpy(out_off, h + 16, (n_sent + 1) * 8);
    if (n_sent) memcpy(status, h + 16 + off_bytes, n_sent);
    if (nt > out_cap)
      return fail(SWT_ERR_CAPACITY, "out_ids too small: need %llu ids, have %llu", (unsigned long long)nt, (unsigned long long)out_cap);
    if (nt) memcpy(out_ids, h + 16 + off_bytes + st_bytes, nt * 4);
    return SWT_OK;
  }
  if ((rc = t->in_text.reserve(n_bytes + 64))) return rc;
  if ((rc = t->in_off.reserve((n_sent + 1) * 8))) return rc;
  if (n_bytes) SWT_HIP(hipMemcpyAsync(t->in_text.p, text, n_bytes, hipMemcpyHostToDevice, 0));
  SWT_HIP(hipMemcpyAsync(t->in_off.p, sent_off, (n_sent + 1) * 8, hipMemcpyHostToDevice, 0));
  return wp_encode_to_host(t, t->in_text.as<uint8_t>(), n_bytes, t->in_off.as<uint64_t>(), n_sent, out_ids, out_cap, out_off, status, n_tokens);
} SWT_API_CATCH

// list[str] joined with U+0000 -> ids, the prepared text staying on the device (see swt_bpe_encode_joined).
// *n_tokens = UINT64_MAX on return: a sentence needs the host's str.lower() and nothing was encoded.
int swt_wp_encode_joined(swt_wp_trie *t, const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint32_t *out_ids, uint64_t out_cap,
                         uint64_t *out_off, uint8_t *status, uint64_t *n_tokens, uint8_t *need_host) try {
  if (!t || !out_off || !n_tokens || (n_sent && (!need_host || !status)) || (n_joined && !joined)) return fail(SWT_ERR_INVALID, "null argument");
  int rc = wp_upload(t);
  if (rc) return rc;
  *n_tokens = UINT64_MAX;
  struct Ctx { swt_wp_trie *t; uint64_t n_sent; uint32_t *out_ids; uint64_t out_cap; uint64_t *out_off; uint8_t *status; uint64_t *n_tokens; };
  Ctx c{t, n_sent, out_ids, out_cap, out_off, status, n_tokens};
  bool consumed = false;
  return with_prepared_joined(joined, n_joined, n_sent, need_host, &consumed,
      [](void *p, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_off) {
        Ctx *c = static_cast<Ctx *>(p);
        return wp_encode_to_host(c->t, d_text, n_bytes, d_off, c->n_sent, c->out_ids, c->out_cap, c->out_off, c->status, c->n_tokens);
      }, &c);
} SWT_API_CATCH

}  // extern "C"
