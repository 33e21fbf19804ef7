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
#include <unordered_map>

#include "swt_tile.h"

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
constexpr uint8_t kPySpace = 1, kPyAlnum = 2, kWpCont = 0x80;

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

// Character sources for the sentence walker: LDS-staged chunk, or global memory (sentences longer than a chunk).
struct LdsSrc {
  const TileLds *L;
  __device__ __forceinline__ void load(uint64_t p, uint64_t e, uint32_t &cp, uint32_t &cc, uint32_t &len) const {
    cp = L->sym[p];
    cc = L->cls[p];
    uint32_t q = (uint32_t)p + 1;
    while (q < e && (L->cls[q] & kWpCont)) q++;
    len = q - (uint32_t)p;
  }
};
struct GlobalSrc {
  const uint8_t *text;
  const uint8_t *cls_tab;
  __device__ __forceinline__ void load(uint64_t p, uint64_t e, uint32_t &cp, uint32_t &cc, uint32_t &len) const {
    const uint8_t b = text[p];
    int n = utf8_len(b);
    if (p + n > e) n = (int)(e - p);
    cp = b;
    if (b >= 0x80 && n > 1) {
      cp = b & (0xFF >> (n + 1));
      for (int i = 1; i < n; i++) cp = (cp << 6) | (text[p + i] & 0x3F);
    }
    const uint8_t c = cp < kNumCodePoints ? cls_tab[cp] : (uint8_t)0;
    cc = (c >> 2) & 3;
    uint64_t q = p + n;
    while (q < e && utf8_is_cont(text[q])) q++;  // stray continuation bytes ride with the previous char
    len = (uint32_t)(q - p);
  }
};

// FastWP.tokenize on one sentence occupying bytes [b, e); `out` receives the ids (out[k] for the k-th).
// Returns the token count (0 when status != OK).  wordpiece.py:248-270 with s = text + " ".
template <class Src, class Out>
__device__ __forceinline__ uint32_t wp_sentence(const Src &src, uint64_t b, uint64_t e, Out out, const WpDev &T, int &status) {
  uint64_t i = b;  // i == e: the appended space (wordpiece.py:248); i > e: i == len(s)
  uint32_t cp = ' ', cc = kPySpace, len = 1;
  if (i < e) src.load(i, e, cp, cc, len);
  bool prev_punc = false;  // ispunc(s[i-1]) -- never across a sentence start
  uint32_t nt = 0;
  status = SWT_WP_OK;
#define WP_ADV()                                   \
  do {                                             \
    prev_punc = (cc & (kPySpace | kPyAlnum)) == 0; \
    if (i < e) { i += len; if (i > e) i = e; }     \
    else i = e + 1;                                \
    if (i < e) src.load(i, e, cp, cc, len);        \
    else { cp = ' '; cc = kPySpace; len = 1; }     \
  } while (0)
#define WP_BNDRY() (prev_punc || (cc & kPySpace) || (cc & (kPySpace | kPyAlnum)) == 0) /* wordpiece.py:285 */
  while (i <= e) {  // wordpiece.py:251
    const uint64_t seg_i = i;
    const uint32_t seg_nt = nt;
    uint32_t node = kWpRoot;
    // matchloop, wordpiece.py:291-316
    bool stop = false;
    while (i <= e) {
      int32_t child = edge_lookup(T, node, cp);
      while (child < 0) {
        const WpNode nd = T.nodes[node];
        if (nd.link < 0) { stop = true; break; }
        for (uint32_t k = 0; k < nd.pop_cnt; k++) out[nt++] = T.pops[nd.pop_off + k];
        node = (uint32_t)nd.link;
        child = edge_lookup(T, node, cp);
      }
      if (stop) break;
      node = (uint32_t)child;
      WP_ADV();
    }
    if (i > e) { status = SWT_WP_INDEXERROR; return 0; }  // iswdbndry indexes seq[len(seq)] (wordpiece.py:285)
    const bool root_like = node == kWpRoot || node == T.root_sharp || node == kWpRootP;
    if (!WP_BNDRY() || !root_like) {  // wordpiece.py:255-257
      nt = seg_nt;
      out[nt++] = T.unk_id;
    } else if (node == T.root_sharp && nt == seg_nt) {  // wordpiece.py:260-261
      if (T.corner_nonterm) { status = SWT_WP_NONTERMINATING; return 0; }
      out[nt++] = T.corner_id;
    }
    while (i <= e && !WP_BNDRY()) WP_ADV();        // wordpiece.py:265-266
    while (i <= e && (cc & kPySpace)) WP_ADV();    // wordpiece.py:268-269
    if (i == seg_i) { status = SWT_WP_NONTERMINATING; return 0; }  // same state again: the reference spins
  }
#undef WP_ADV
#undef WP_BNDRY
  return nt;
}

struct WpGiant { uint64_t end; uint32_t ntok; };

__global__ __launch_bounds__(kThreads) void wp_encode_kernel(
    const uint8_t *__restrict__ text, uint64_t n_bytes, const uint64_t *__restrict__ sent_off,
    const uint64_t *__restrict__ plan, const uint8_t *__restrict__ cls_tab, WpDev T, uint32_t *__restrict__ scratch,
    uint32_t *__restrict__ sent_local, uint32_t *__restrict__ tile_tok, uint8_t *__restrict__ status) {
  __shared__ TileLds L;
  __shared__ WpGiant s_giant;
  __shared__ uint32_t s_nsent;

  const int tid = threadIdx.x;
  const uint64_t t = blockIdx.x;
  const uint64_t s_lo = plan[t], s_hi = plan[t + 1];
  if (s_lo == s_hi) {
    if (tid == 0) tile_tok[t] = 0;
    return;
  }
  const uint64_t span_base = sent_off[s_lo], span_end = sent_off[s_hi];
  uint32_t *const tile_out = scratch + span_base;
  uint32_t run = 0;
  uint64_t s_next = s_lo;
  uint64_t cb = span_base;

  for (;;) {
    const uint64_t abase = cb & ~15ull;
    const uint32_t off0 = (uint32_t)(cb - abase);
    const uint64_t avail = span_end - abase;
    const bool last = avail <= (uint64_t)kCap;
    const uint32_t staged = last ? (uint32_t)avail : (uint32_t)kCap;

    tile_stage(L, text, n_bytes, abase, staged);
    if (tid == 0) s_nsent = 0;

    // ---- B. per byte: code point + str.isspace / str.isalnum bits at every lead byte
    for (uint32_t p = tid; p < staged; p += kThreads) {
      const uint8_t b = L.txt[p];
      uint32_t sv = kInvalidTok;
      uint8_t cv = kWpCont;
      if (!utf8_is_cont(b) && p >= off0) {
        int len = utf8_len(b);
        if (p + len > staged) len = (int)(staged - p);
        uint32_t cp = b;
        if (b >= 0x80 && len > 1) {
          cp = b & (0xFF >> (len + 1));
          for (int i = 1; i < len; i++) cp = (cp << 6) | (L.txt[p + i] & 0x3F);
        }
        const uint8_t c = cp < kNumCodePoints ? cls_tab[cp] : (uint8_t)0;
        cv = (c >> 2) & 3;
        sv = cp;
      }
      L.sym[p] = sv;
      L.cls[p] = cv;
    }
    // chunk end: whole sentences only -- the last sentence start that fits (or the end of the span)
    uint32_t ce = staged;
    if (!last) {
      int best = -1;
      for (uint64_t s = s_next + tid; s < s_hi; s += kThreads) {
        const uint64_t o = sent_off[s];
        if (o > abase + staged) break;
        if (o > cb) best = (int)(o - abase);
      }
      if (best >= 0) atomicMax(&L.cut, best);
    }
    __syncthreads();
    if (!last) {
      if (L.cut < 0) {
        // one sentence longer than the LDS chunk: one lane walks it in global memory
        if (tid == 0) {
          uint64_t s = s_next;
          while (s + 1 < s_hi && sent_off[s + 1] <= cb) s++;  // the last sentence that starts at cb
          int stt;
          GlobalSrc src{text, cls_tab};
          const uint64_t e = sent_off[s + 1];
          const uint32_t n = wp_sentence(src, cb, e, tile_out + run, T, stt);
          // empty sentences that also start at cb come first and get no tokens
          for (uint64_t z = s_next; z <= s; z++) { sent_local[z] = run; status[z] = (uint8_t)T.empty_status; }
          status[s] = (uint8_t)stt;
          s_giant.end = e;
          s_giant.ntok = n;
          s_nsent = (uint32_t)(s - s_next + 1);
        }
        __syncthreads();
        s_next += s_nsent;
        run += s_giant.ntok;
        cb = s_giant.end;
        __syncthreads();
        if (cb >= span_end) {
          // trailing empty sentences at the very end of the span
          for (uint64_t z = s_next + tid; z < s_hi; z += kThreads) { sent_local[z] = run; status[z] = (uint8_t)T.empty_status; }
          break;
        }
        continue;
      }
      ce = (uint32_t)L.cut;
    }

    // ---- D (v1). one lane per sentence: the reference's state machine over the staged chunk; tokens
    // overwrite the sentence's own bytes in sym[] (a sentence never yields more tokens than bytes)
    for (uint64_t s = s_next + tid; s < s_hi; s += kThreads) {
      const uint64_t o = sent_off[s];
      const uint64_t rel = o - abase;
      if (rel > ce || (rel == ce && !last)) break;
      const uint64_t e = sent_off[s + 1] - abase;  // <= ce: chunks end at sentence starts
      int stt = SWT_WP_OK;
      // an empty sentence still runs: s = " " (wordpiece.py:248); it yields no token but may raise (a vocabulary
      // with a " " edge at the root) -- it never writes to out
      LdsSrc src{&L};
      const uint32_t n = wp_sentence(src, rel, e, &L.sym[rel < kCap ? rel : 0], T, stt);
      for (uint64_t q = rel + n; q < e; q++) L.sym[q] = kInvalidTok;
      status[s] = (uint8_t)stt;
    }
    __syncthreads();

    const uint32_t total = tile_compact(L, off0, ce, tile_out + run);
    s_next += tile_record(L, sent_off, sent_local, s_next, s_hi, abase, ce, last, run, total);
    run += total;
    if (last) break;
    cb = abase + ce;
  }
  if (tid == 0) tile_tok[t] = run;
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

int swt_wp_trie_create(const uint32_t *vocab_cps, const uint64_t *vocab_off, uint32_t n_vocab, swt_wp_trie **out) {
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
  *out = t;
  return SWT_OK;
}

void swt_wp_trie_destroy(swt_wp_trie *t) {
  if (!t) return;
  if (t->d_edges) (void)hipFree(t->d_edges);
  if (t->d_nodes) (void)hipFree(t->d_nodes);
  if (t->d_pops) (void)hipFree(t->d_pops);
  t->ws.release();
  for (DevBuf *b : {&t->in_text, &t->in_off, &t->out_ids, &t->out_off, &t->out_status, &t->n_tok}) b->release();
  delete t;
}

int swt_wp_trie_stats(const swt_wp_trie *t, uint32_t *n_nodes, uint32_t *n_edges, uint32_t *n_pops) {
  if (!t) return fail(SWT_ERR_INVALID, "null trie");
  if (n_nodes) *n_nodes = (uint32_t)t->H.ch.size();
  if (n_edges) *n_edges = (uint32_t)t->H.edges.size();
  if (n_pops) *n_pops = (uint32_t)t->h_pops.size();
  return SWT_OK;
}

int64_t swt_wp_trie_corner(const swt_wp_trie *t, uint32_t *out, uint64_t cap) {
  if (!t) return -2;
  if (t->H.corner_nonterm) return -1;
  for (size_t k = 0; k < t->H.corner.size() && k < cap; k++) out[k] = t->H.corner[k];
  return (int64_t)t->H.corner.size();
}

int swt_wp_trie_node(const swt_wp_trie *t, const uint32_t *path, uint64_t path_len, uint32_t *node_id, int32_t *link,
                     uint8_t *is_end, uint32_t *pops, uint32_t pops_cap, uint32_t *n_pops) {
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
}

int swt_wp_trie_node_path(const swt_wp_trie *t, uint32_t node_id, uint32_t *out, uint64_t cap, uint64_t *len) {
  if (!t || node_id >= t->H.ch.size()) return fail(SWT_ERR_INVALID, "bad node id");
  std::vector<uint32_t> rev;
  for (int32_t n = (int32_t)node_id; n >= 0 && t->H.parent[n] >= 0; n = t->H.parent[n]) rev.push_back(t->H.ch[n]);
  if (len) *len = rev.size();
  for (size_t k = 0; k < rev.size() && k < cap; k++) out[k] = rev[rev.size() - 1 - k];
  return SWT_OK;
}

int swt_wp_encode_dev(swt_wp_trie *t, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off, uint64_t n_sent,
                      uint32_t *d_out_ids, uint64_t *d_out_off, uint8_t *d_status, uint64_t *d_n_tokens, void *stream) {
  if (!t || !d_sent_off || !d_out_off || !d_n_tokens || (n_sent && !d_status) || (n_bytes && (!d_text || !d_out_ids)))
    return fail(SWT_ERR_INVALID, "null argument");
  int rc = wp_upload(t);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const uint8_t *d_cls = nullptr;
  if ((rc = device_class_table(&d_cls))) return rc;
  const uint64_t n_tiles = tile_count(n_bytes);
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
  launch_plan(d_sent_off, n_sent, n_tiles, kTile, t->ws.plan.as<uint64_t>(), st);
  prof_begin(st);
  hipLaunchKernelGGL(wp_encode_kernel, dim3((unsigned)n_tiles), dim3(kThreads), 0, st, d_text, n_bytes, d_sent_off,
                     t->ws.plan.as<uint64_t>(), d_cls, T, t->ws.scratch.as<uint32_t>(), t->ws.sent_local.as<uint32_t>(),
                     t->ws.tile_tok.as<uint32_t>(), d_status);
  prof_end(st);
  launch_scan_gather(d_sent_off, n_sent, n_tiles, t->ws, d_out_ids, d_out_off, d_n_tokens, st);
  SWT_HIP(hipGetLastError());
  return SWT_OK;
}

int swt_wp_encode(swt_wp_trie *t, const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent, uint32_t *out_ids,
                  uint64_t out_cap, uint64_t *out_off, uint8_t *status, uint64_t *n_tokens) {
  if (!t || !sent_off || !out_off || !n_tokens || (n_sent && !status)) return fail(SWT_ERR_INVALID, "null argument");
  int rc = wp_upload(t);
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
  if ((rc = t->out_status.reserve(n_sent + 8))) return rc;
  if ((rc = t->n_tok.reserve(8))) return rc;
  if (n_bytes) SWT_HIP(hipMemcpyAsync(t->in_text.p, text, n_bytes, hipMemcpyHostToDevice, 0));
  SWT_HIP(hipMemcpyAsync(t->in_off.p, sent_off, (n_sent + 1) * 8, hipMemcpyHostToDevice, 0));
  rc = swt_wp_encode_dev(t, t->in_text.as<uint8_t>(), n_bytes, t->in_off.as<uint64_t>(), n_sent, t->out_ids.as<uint32_t>(),
                         t->out_off.as<uint64_t>(), t->out_status.as<uint8_t>(), t->n_tok.as<uint64_t>(), nullptr);
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

}  // extern "C"
