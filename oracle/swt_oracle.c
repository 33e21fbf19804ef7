/*
 * oracle/swt_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * See swt_oracle.h for who may load this and for the id conventions.  Plain C11, no dependencies.
 * Every block cites the lines of /root/reference it restates.  The algorithms are kept in the
 * reference's own formulation (full pair recount and full rewrite per merge; set-of-pairs min-rank
 * loop per word; pointer trie walked char by char) so that this file is a readable second statement
 * of the same semantics, not an optimised tokenizer.
 */
#include "swt_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "unicode_classes.inc"

/* ------------------------------------------------------------------------------------------ */
/* code-point classes (fixture data probed from the wheel / interpreter, tools/gen_unicode_tables.py) */

static uint8_t *g_cls;
static pthread_once_t g_cls_once = PTHREAD_ONCE_INIT;

static void cls_fill(const unsigned int (*r)[2], unsigned int n, uint8_t bit) {
  for (unsigned int i = 0; i < n; i++)
    for (unsigned int c = r[i][0]; c <= r[i][1]; c++) g_cls[c] |= bit;
}

static void cls_init(void) {
  g_cls = (uint8_t *)calloc(0x110000, 1);
  cls_fill(SWT_BERT_WS_RANGES, SWT_BERT_WS_NRANGES, ORC_BERT_WS);
  cls_fill(SWT_BERT_PUNCT_RANGES, SWT_BERT_PUNCT_NRANGES, ORC_BERT_PUNCT);
  cls_fill(SWT_PY_SPACE_RANGES, SWT_PY_SPACE_NRANGES, ORC_PY_SPACE);
  cls_fill(SWT_PY_ALNUM_RANGES, SWT_PY_ALNUM_NRANGES, ORC_PY_ALNUM);
}

unsigned orc_class(uint32_t cp) {
  pthread_once(&g_cls_once, cls_init);
  return cp < 0x110000u ? g_cls[cp] : 0u;
}

static inline int py_isspace(uint32_t c) { return (orc_class(c) & ORC_PY_SPACE) != 0; }
static inline int py_isalnum(uint32_t c) { return (orc_class(c) & ORC_PY_ALNUM) != 0; }
/* source/wordpiece.py:287-288 */
static inline int wp_ispunc(uint32_t c) { return !py_isalnum(c) && !py_isspace(c); }

/* ------------------------------------------------------------------------------------------ */
/* small containers */

typedef struct {
  uint64_t *keys, *vals;
  uint64_t cap, n; /* cap is a power of two */
} map64;
#define MAP_EMPTY (~0ull)

static uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
static void map_init(map64 *m, uint64_t cap) {
  uint64_t c = 16; while (c < cap) c <<= 1;
  m->cap = c; m->n = 0;
  m->keys = (uint64_t *)malloc(c * 8); m->vals = (uint64_t *)malloc(c * 8);
  memset(m->keys, 0xff, c * 8);
}
static void map_free(map64 *m) { free(m->keys); free(m->vals); m->keys = m->vals = NULL; }
static void map_clear(map64 *m) { memset(m->keys, 0xff, m->cap * 8); m->n = 0; }
static uint64_t *map_find(const map64 *m, uint64_t k) {
  uint64_t i = mix64(k) & (m->cap - 1);
  while (m->keys[i] != MAP_EMPTY) {
    if (m->keys[i] == k) return &m->vals[i];
    i = (i + 1) & (m->cap - 1);
  }
  return NULL;
}
static void map_put(map64 *m, uint64_t k, uint64_t v);
static void map_grow(map64 *m) {
  map64 o = *m;
  map_init(m, o.cap * 2);
  for (uint64_t i = 0; i < o.cap; i++) if (o.keys[i] != MAP_EMPTY) map_put(m, o.keys[i], o.vals[i]);
  map_free(&o);
}
static void map_put(map64 *m, uint64_t k, uint64_t v) {
  if ((m->n + 1) * 2 > m->cap) map_grow(m);
  uint64_t i = mix64(k) & (m->cap - 1);
  while (m->keys[i] != MAP_EMPTY) {
    if (m->keys[i] == k) { m->vals[i] = v; return; }
    i = (i + 1) & (m->cap - 1);
  }
  m->keys[i] = k; m->vals[i] = v; m->n++;
}

typedef struct { uint32_t *p; uint64_t n, cap; } vec32;
typedef struct { uint64_t *p; uint64_t n, cap; } vec64;
static void v32_push(vec32 *v, uint32_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->p = (uint32_t *)realloc(v->p, v->cap * 4); }
  v->p[v->n++] = x;
}
static void v64_push(vec64 *v, uint64_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->p = (uint64_t *)realloc(v->p, v->cap * 8); }
  v->p[v->n++] = x;
}

/* string table: interns UTF-32 strings, index = order of first appearance */
typedef struct {
  vec32 blob;
  vec64 off; /* n+1 offsets */
  uint64_t *slots; /* index+1, 0 = empty */
  uint64_t cap;
} strtab;

static void st_init(strtab *t) {
  memset(t, 0, sizeof *t);
  v64_push(&t->off, 0);
  t->cap = 1024;
  t->slots = (uint64_t *)calloc(t->cap, 8);
}
static void st_free(strtab *t) { free(t->blob.p); free(t->off.p); free(t->slots); }
static uint64_t st_count(const strtab *t) { return t->off.n - 1; }
static uint64_t st_hash(const uint32_t *s, uint64_t n) {
  uint64_t h = 0xcbf29ce484222325ULL ^ n;
  for (uint64_t i = 0; i < n; i++) { h ^= s[i]; h *= 0x100000001b3ULL; }
  return mix64(h);
}
static void st_rehash(strtab *t) {
  uint64_t nc = t->cap * 2;
  uint64_t *ns = (uint64_t *)calloc(nc, 8);
  for (uint64_t k = 0; k < st_count(t); k++) {
    uint64_t h = st_hash(t->blob.p + t->off.p[k], t->off.p[k + 1] - t->off.p[k]) & (nc - 1);
    while (ns[h]) h = (h + 1) & (nc - 1);
    ns[h] = k + 1;
  }
  free(t->slots); t->slots = ns; t->cap = nc;
}
/* returns index; *is_new set when the string was added */
static uint64_t st_intern(strtab *t, const uint32_t *s, uint64_t n, int *is_new) {
  uint64_t h = st_hash(s, n) & (t->cap - 1);
  while (t->slots[h]) {
    uint64_t k = t->slots[h] - 1;
    if (t->off.p[k + 1] - t->off.p[k] == n && (n == 0 || memcmp(t->blob.p + t->off.p[k], s, n * 4) == 0)) {
      if (is_new) *is_new = 0;
      return k;
    }
    h = (h + 1) & (t->cap - 1);
  }
  uint64_t k = st_count(t);
  for (uint64_t i = 0; i < n; i++) v32_push(&t->blob, s[i]);
  v64_push(&t->off, t->blob.n);
  t->slots[h] = k + 1;
  if (is_new) *is_new = 1;
  if ((k + 1) * 2 > t->cap) st_rehash(t);
  return k;
}

/* canonical symbol ids (see header) */
static uint32_t sym_intern(strtab *t, const uint32_t *s, uint64_t n, int *is_new) {
  if (n == 1) { if (is_new) *is_new = 0; return s[0]; }
  return ORC_SYM_BASE + (uint32_t)st_intern(t, s, n, is_new);
}
static uint64_t sym_string(const strtab *t, uint32_t id, uint32_t *out, uint64_t cap) {
  if (id < ORC_SYM_BASE) { if (cap) out[0] = id; return 1; }
  uint64_t k = id - ORC_SYM_BASE;
  if (k >= st_count(t)) return 0;
  uint64_t n = t->off.p[k + 1] - t->off.p[k];
  for (uint64_t i = 0; i < n && i < cap; i++) out[i] = t->blob.p[t->off.p[k] + i];
  return n;
}
static uint32_t sym_concat(strtab *t, uint32_t l, uint32_t r, int *is_new) {
  uint32_t lb[1], rb[1];
  const uint32_t *lp, *rp; uint64_t ln, rn;
  if (l < ORC_SYM_BASE) { lb[0] = l; lp = lb; ln = 1; }
  else { uint64_t k = l - ORC_SYM_BASE; lp = t->blob.p + t->off.p[k]; ln = t->off.p[k + 1] - t->off.p[k]; }
  if (r < ORC_SYM_BASE) { rb[0] = r; rp = rb; rn = 1; }
  else { uint64_t k = r - ORC_SYM_BASE; rp = t->blob.p + t->off.p[k]; rn = t->off.p[k + 1] - t->off.p[k]; }
  uint32_t *tmp = (uint32_t *)malloc((ln + rn + 1) * 4);
  memcpy(tmp, lp, ln * 4); memcpy(tmp + ln, rp, rn * 4);
  uint32_t id = sym_intern(t, tmp, ln + rn, is_new);
  free(tmp);
  return id;
}

#define PAIR_KEY(l, r) (((uint64_t)(l) << 32) | (uint64_t)(r))

/* ------------------------------------------------------------------------------------------ */
/* a1: pre-tokenization -- source/utils.py:15-29.  The reference lowercases (host, Python) and then
 * calls HF BertPreTokenizer: split on White_Space (removed), then every punctuation code point is
 * isolated as its own word (pinned by the probed class table). */

uint64_t orc_pretokenize(const uint32_t *s, uint64_t n, uint64_t *starts, uint64_t *ends, uint64_t cap) {
  uint64_t nw = 0, i = 0;
  while (i < n) {
    unsigned c = orc_class(s[i]);
    if (c & ORC_BERT_WS) { i++; continue; }
    uint64_t j = i + 1;
    if (!(c & ORC_BERT_PUNCT))
      while (j < n && !(orc_class(s[j]) & (ORC_BERT_WS | ORC_BERT_PUNCT))) j++;
    if (nw < cap) { starts[nw] = i; ends[nw] = j; }
    nw++;
    i = j;
  }
  return nw;
}

/* ------------------------------------------------------------------------------------------ */
/* a6-a8: FastBPE model + encode */

struct orc_bpe {
  strtab st;
  map64 ranks; /* PAIR_KEY -> rank (last duplicate wins, source/bpe.py:200,257) */
  uint32_t n;
  uint32_t *l, *r, *mg;
};

orc_bpe *orc_bpe_new(const uint32_t *blob, const uint64_t *off, uint32_t n_merges) {
  orc_bpe *m = (orc_bpe *)calloc(1, sizeof *m);
  st_init(&m->st);
  map_init(&m->ranks, (uint64_t)n_merges * 2 + 16);
  m->n = n_merges;
  m->l = (uint32_t *)malloc((n_merges + 1) * 4);
  m->r = (uint32_t *)malloc((n_merges + 1) * 4);
  m->mg = (uint32_t *)malloc((n_merges + 1) * 4);
  for (uint32_t i = 0; i < n_merges; i++) {
    const uint32_t *lp = blob + off[2 * i], *rp = blob + off[2 * i + 1];
    uint64_t ln = off[2 * i + 1] - off[2 * i], rn = off[2 * i + 2] - off[2 * i + 1];
    m->l[i] = sym_intern(&m->st, lp, ln, NULL);
    m->r[i] = sym_intern(&m->st, rp, rn, NULL);
    uint32_t *tmp = (uint32_t *)malloc((ln + rn + 1) * 4);
    memcpy(tmp, lp, ln * 4); memcpy(tmp + ln, rp, rn * 4);
    m->mg[i] = sym_intern(&m->st, tmp, ln + rn, NULL);
    free(tmp);
    map_put(&m->ranks, PAIR_KEY(m->l[i], m->r[i]), i); /* bpe.py:257 {pair: i}: later i overwrites */
  }
  return m;
}

void orc_bpe_free(orc_bpe *m) {
  if (!m) return;
  st_free(&m->st); map_free(&m->ranks); free(m->l); free(m->r); free(m->mg); free(m);
}
uint32_t orc_bpe_n_symbols(const orc_bpe *m) { return (uint32_t)st_count(&m->st); }
uint64_t orc_bpe_symbol(const orc_bpe *m, uint32_t id, uint32_t *out, uint64_t cap) {
  return sym_string(&m->st, id & ~ORC_CONT_FLAG, out, cap);
}
void orc_bpe_merge_ids(const orc_bpe *m, uint32_t i, uint32_t *l, uint32_t *r, uint32_t *mg) {
  *l = m->l[i]; *r = m->r[i]; *mg = m->mg[i];
}

/* source/bpe.py:205-243 */
uint64_t orc_bpe_encode_word(const orc_bpe *m, const uint32_t *w, uint64_t n, uint32_t *out) {
  if (n < 2) { /* bpe.py:207-208 */
    if (n == 1) out[0] = w[0];
    return n;
  }
  uint32_t *sym = (uint32_t *)malloc(n * 4), *tmp = (uint32_t *)malloc(n * 4);
  memcpy(sym, w, n * 4); /* bpe.py:206 list(word) */
  for (;;) {
    /* bpe.py:211-219: min rank over the set of adjacent pairs */
    uint64_t best = ~0ull; uint32_t bl = 0, br = 0;
    for (uint64_t i = 0; i + 1 < n; i++) {
      const uint64_t *r = map_find(&m->ranks, PAIR_KEY(sym[i], sym[i + 1]));
      if (r && *r < best) { best = *r; bl = sym[i]; br = sym[i + 1]; }
    }
    if (best == ~0ull) break;
    uint32_t merged = m->mg[best];
    /* bpe.py:221-235: L->R non-overlapping replacement of every occurrence */
    uint64_t j = 0, i = 0;
    while (i < n) {
      if (i + 1 < n && sym[i] == bl && sym[i + 1] == br) { tmp[j++] = merged; i += 2; }
      else tmp[j++] = sym[i++];
    }
    uint32_t *sw = sym; sym = tmp; tmp = sw; n = j;
    if (n == 1) break; /* bpe.py:236-237 */
  }
  out[0] = sym[0];
  for (uint64_t i = 1; i < n; i++) out[i] = sym[i] | ORC_CONT_FLAG; /* bpe.py:240-241 */
  free(sym); free(tmp);
  return n;
}

/* source/bpe.py:245-249 */
uint64_t orc_bpe_tokenize(const orc_bpe *m, const uint32_t *s, uint64_t n, uint32_t *out) {
  uint64_t nt = 0, i = 0;
  while (i < n) { /* same walk as orc_pretokenize, fused so no word list is materialised */
    unsigned c = orc_class(s[i]);
    if (c & ORC_BERT_WS) { i++; continue; }
    uint64_t j = i + 1;
    if (!(c & ORC_BERT_PUNCT))
      while (j < n && !(orc_class(s[j]) & (ORC_BERT_WS | ORC_BERT_PUNCT))) j++;
    nt += orc_bpe_encode_word(m, s + i, j - i, out + nt);
    i = j;
  }
  return nt;
}

uint64_t orc_bpe_tokenize_batch(const orc_bpe *m, const uint32_t *text, const uint64_t *sent_off,
                                uint64_t n_sent, uint32_t *out, uint64_t *out_off) {
  uint64_t nt = 0;
  for (uint64_t s = 0; s < n_sent; s++) {
    out_off[s] = nt;
    nt += orc_bpe_tokenize(m, text + sent_off[s], sent_off[s + 1] - sent_off[s], out + nt);
  }
  out_off[n_sent] = nt;
  return nt;
}

/* ------------------------------------------------------------------------------------------ */
/* a2-a5: BPE training -- source/bpe.py:50-112 */

struct orc_train {
  strtab st;     /* multi-char symbols */
  vec32 syms;    /* corpus_as_symbols, concatenated */
  vec64 woff;    /* n_words + 1 */
  vec32 freq;
  uint32_t vocab_size;
  vec32 ml, mr, mm; vec64 mc; /* merges_list (+ the winning count, for diagnostics) */
  map64 pairs;   /* scratch: PAIR_KEY -> index into pk/pc */
  vec64 pk, pc;
  int wp;        /* 1: NaiveWP.train (wordpiece.py:29-103): '##' symbols, likelihood score */
  map64 sfreq;   /* wp scratch: symbol id -> frequency */
};

static orc_train *train_new(const uint32_t *text, const uint64_t *sent_off, uint64_t n_sent, int wp) {
  orc_train *t = (orc_train *)calloc(1, sizeof *t);
  st_init(&t->st);
  map_init(&t->pairs, 1 << 16);
  map_init(&t->sfreq, 1 << 12);
  t->wp = wp;
  /* bpe.py:70-77: preprocessing, then Counter(new_words) -- insertion order = first occurrence */
  strtab words; st_init(&words);
  vec32 wfreq = {0};
  uint8_t *seen = (uint8_t *)calloc(0x110000, 1);
  for (uint64_t s = 0; s < n_sent; s++) {
    const uint32_t *p = text + sent_off[s];
    uint64_t n = sent_off[s + 1] - sent_off[s], i = 0;
    while (i < n) {
      unsigned c = orc_class(p[i]);
      if (c & ORC_BERT_WS) { i++; continue; }
      uint64_t j = i + 1;
      if (!(c & ORC_BERT_PUNCT))
        while (j < n && !(orc_class(p[j]) & (ORC_BERT_WS | ORC_BERT_PUNCT))) j++;
      int is_new;
      uint64_t k = st_intern(&words, p + i, j - i, &is_new);
      if (is_new) v32_push(&wfreq, 0);
      wfreq.p[k]++;
      /* bpe.py:75: vocab.update({ch for w in new_words for ch in w}); the WordPiece initial symbols are counted below */
      if (!wp)
        for (uint64_t q = i; q < j; q++)
          if (p[q] < 0x110000u && !seen[p[q]]) { seen[p[q]] = 1; t->vocab_size++; }
      i = j;
    }
  }
  /* bpe.py:79-81: symbols = [s for s in word] */
  uint64_t nw = st_count(&words);
  v64_push(&t->woff, 0);
  for (uint64_t k = 0; k < nw; k++) {
    for (uint64_t q = words.off.p[k]; q < words.off.p[k + 1]; q++) {
      uint32_t c = words.blob.p[q];
      if (wp) {
        /* wordpiece.py:54-57: [word[0]] + ["##" + c for c in word[1:]]; :62-63 vocab |= initial symbols */
        int is_new = 0;
        if (q == words.off.p[k]) {
          if (c < 0x110000u && !seen[c]) { seen[c] = 1; t->vocab_size++; }
        } else {
          uint32_t tmp[3] = {'#', '#', c};
          c = sym_intern(&t->st, tmp, 3, &is_new);
          if (is_new) t->vocab_size++;
        }
      }
      v32_push(&t->syms, c);
    }
    v64_push(&t->woff, t->syms.n);
    v32_push(&t->freq, wfreq.p[k]);
  }
  free(seen);
  st_free(&words); free(wfreq.p);
  return t;
}
orc_train *orc_train_new(const uint32_t *text, const uint64_t *sent_off, uint64_t n_sent) { return train_new(text, sent_off, n_sent, 0); }
/* the state after bpe.py:73-81 given directly: unique words as code points (CSR) with their frequencies */
orc_train *orc_train_new_words(const uint32_t *syms, const uint64_t *word_off, const uint32_t *freq, uint64_t n_words) {
  orc_train *t = (orc_train *)calloc(1, sizeof *t);
  st_init(&t->st);
  map_init(&t->pairs, 1 << 16);
  map_init(&t->sfreq, 1 << 12);
  uint8_t *seen = (uint8_t *)calloc(0x110000, 1);
  v64_push(&t->woff, 0);
  for (uint64_t w = 0; w < n_words; w++) {
    for (uint64_t q = word_off[w]; q < word_off[w + 1]; q++) {
      v32_push(&t->syms, syms[q]);
      if (syms[q] < 0x110000u && !seen[syms[q]]) { seen[syms[q]] = 1; t->vocab_size++; } /* bpe.py:75 */
    }
    v64_push(&t->woff, t->syms.n);
    v32_push(&t->freq, freq[w]);
  }
  free(seen);
  return t;
}
orc_train *orc_wptrain_new(const uint32_t *text, const uint64_t *sent_off, uint64_t n_sent) { return train_new(text, sent_off, n_sent, 1); }

/* Python's int / int (wordpiece.py:86): the quotient of two exact integers, correctly rounded (half to even) to a
 * double.  Returned as the bit pattern (positive doubles order like their patterns).  d = fl * fr can exceed 2^53, where
 * (double)cnt / (double)d would round twice; then the mantissa comes from a bitwise long division. */
uint64_t orc_wp_score_bits(uint64_t cnt, uint64_t fl, uint64_t fr) {
  unsigned __int128 d = (unsigned __int128)fl * fr;
  double s;
  if (cnt == 0 || d == 0) return 0;
  if (d < ((unsigned __int128)1 << 53) && cnt < (1ull << 53)) {
    s = (double)cnt / (double)(uint64_t)d; /* both exact: IEEE division rounds once */
  } else {
    unsigned __int128 r = cnt;
    int e = 0;
    while (r < d) { r <<= 1; e--; }            /* r / d in [1, 2) ... */
    while (r >= 2 * d) { d <<= 1; e++; }       /* ... also when cnt >= 2 d (not reached by a score, kept for the test) */
    uint64_t mant = 1; r -= d;
    for (int i = 0; i < 52; i++) { r <<= 1; mant <<= 1; if (r >= d) { r -= d; mant |= 1; } }
    r <<= 1;
    int rb = r >= d; if (rb) r -= d;
    if (rb && (r != 0 || (mant & 1))) mant++;
    if (mant == (1ull << 53)) { mant >>= 1; e++; }
    uint64_t bits = ((uint64_t)(e + 1023) << 52) | (mant & ((1ull << 52) - 1));
    memcpy(&s, &bits, 8);
  }
  uint64_t b; memcpy(&b, &s, 8);
  return b;
}

void orc_train_free(orc_train *t) {
  if (!t) return;
  st_free(&t->st); free(t->syms.p); free(t->woff.p); free(t->freq.p);
  free(t->ml.p); free(t->mr.p); free(t->mm.p); free(t->mc.p);
  map_free(&t->pairs); map_free(&t->sfreq); free(t->pk.p); free(t->pc.p); free(t);
}
uint64_t orc_train_n_words(const orc_train *t) { return t->woff.n - 1; }
uint64_t orc_train_n_symbols(const orc_train *t) { return t->syms.n; }
uint32_t orc_train_vocab_size(const orc_train *t) { return t->vocab_size; }
uint32_t orc_train_n_merges(const orc_train *t) { return (uint32_t)t->ml.n; }
void orc_train_merge_ids(const orc_train *t, uint32_t i, uint32_t *l, uint32_t *r, uint32_t *mg, uint64_t *count) {
  *l = t->ml.p[i]; *r = t->mr.p[i]; *mg = t->mm.p[i]; if (count) *count = t->mc.p[i];
}
uint64_t orc_train_symbol(const orc_train *t, uint32_t id, uint32_t *out, uint64_t cap) {
  return sym_string(&t->st, id, out, cap);
}
void orc_train_export(const orc_train *t, uint32_t *syms, uint64_t *word_off, uint32_t *freq) {
  memcpy(syms, t->syms.p, t->syms.n * 4);
  memcpy(word_off, t->woff.p, t->woff.n * 8);
  memcpy(freq, t->freq.p, t->freq.n * 4);
}

uint32_t orc_train_run(orc_train *t, uint32_t max_vocab, uint32_t max_steps) {
  uint32_t done = 0;
  uint64_t nw = t->woff.n - 1;
  while (t->vocab_size < max_vocab) { /* bpe.py:88 */
    if (max_steps && done >= max_steps) break;
    /* bpe.py:90-95: Counter over every adjacent pair of every unique word, `freq` times each.
     * Insertion order of the Counter = first occurrence in (word, position) order. */
    map_clear(&t->pairs); t->pk.n = 0; t->pc.n = 0;
    for (uint64_t w = 0; w < nw; w++) {
      uint64_t a = t->woff.p[w], b = t->woff.p[w + 1];
      uint32_t f = t->freq.p[w];
      for (uint64_t i = a; i + 1 < b; i++) {
        uint64_t key = PAIR_KEY(t->syms.p[i], t->syms.p[i + 1]);
        uint64_t *slot = map_find(&t->pairs, key);
        if (slot) t->pc.p[*slot] += f;
        else { map_put(&t->pairs, key, t->pk.n); v64_push(&t->pk, key); v64_push(&t->pc, f); }
      }
    }
    if (t->pk.n == 0) break; /* bpe.py:98-99, wordpiece.py:75-76 */
    uint64_t best = 0, best_val;
    if (!t->wp) {
      /* bpe.py:102: most_common(1) -> max(): the FIRST maximum in insertion order */
      for (uint64_t k = 1; k < t->pk.n; k++) if (t->pc.p[k] > t->pc.p[best]) best = k;
      best_val = t->pc.p[best];
    } else {
      /* wordpiece.py:78-92: symbol frequencies, score = freq / (f_left * f_right), max() = first maximum */
      map_clear(&t->sfreq);
      for (uint64_t w = 0; w < nw; w++) {
        uint32_t f = t->freq.p[w];
        for (uint64_t i = t->woff.p[w]; i < t->woff.p[w + 1]; i++) {
          uint64_t *slot = map_find(&t->sfreq, t->syms.p[i]);
          if (slot) *slot += f; else map_put(&t->sfreq, t->syms.p[i], f);
        }
      }
      best_val = 0;
      for (uint64_t k = 0; k < t->pk.n; k++) {
        uint64_t fl = *map_find(&t->sfreq, t->pk.p[k] >> 32), fr = *map_find(&t->sfreq, (uint32_t)t->pk.p[k]);
        uint64_t sc = orc_wp_score_bits(t->pc.p[k], fl, fr);
        if (k == 0 || sc > best_val) { best = k; best_val = sc; }
      }
    }
    uint32_t l = (uint32_t)(t->pk.p[best] >> 32), r = (uint32_t)t->pk.p[best];
    int is_new;
    uint32_t mg;
    if (!t->wp) {
      mg = sym_concat(&t->st, l, r, &is_new);
    } else {
      /* wordpiece.py:95: merged = left + right[2:] (the right symbol of a pair is never word-initial: it starts with ##) */
      uint32_t lb[1], *tmp; const uint32_t *lp, *rp; uint64_t ln, rn;
      if (l < ORC_SYM_BASE) { lb[0] = l; lp = lb; ln = 1; }
      else { uint64_t k = l - ORC_SYM_BASE; lp = t->st.blob.p + t->st.off.p[k]; ln = t->st.off.p[k + 1] - t->st.off.p[k]; }
      { uint64_t k = r - ORC_SYM_BASE; rp = t->st.blob.p + t->st.off.p[k] + 2; rn = t->st.off.p[k + 1] - t->st.off.p[k] - 2; }
      tmp = (uint32_t *)malloc((ln + rn + 1) * 4);
      memcpy(tmp, lp, ln * 4); memcpy(tmp + ln, rp, rn * 4);
      if (ln + rn == 1) { mg = tmp[0]; is_new = 0; }  /* cannot happen: both sides hold at least one character */
      else mg = sym_intern(&t->st, tmp, ln + rn, &is_new);
      free(tmp);
    }
    if (is_new) t->vocab_size++; /* bpe.py:103 / wordpiece.py:96: set.add grows only for an unseen string */
    v32_push(&t->ml, l); v32_push(&t->mr, r); v32_push(&t->mm, mg); v64_push(&t->mc, best_val); /* :104 */
    /* bpe.py:108-111 with _replace_pair (bpe.py:25-48): rebuild every word */
    uint64_t o = 0;
    for (uint64_t w = 0; w < nw; w++) {
      uint64_t a = t->woff.p[w], b = t->woff.p[w + 1], i = a;
      t->woff.p[w] = o;
      while (i < b) {
        if (i + 1 < b && t->syms.p[i] == l && t->syms.p[i + 1] == r) { t->syms.p[o++] = mg; i += 2; }
        else t->syms.p[o++] = t->syms.p[i++];
      }
    }
    t->woff.p[nw] = o; t->syms.n = o;
    done++;
  }
  return done;
}

/* ------------------------------------------------------------------------------------------ */
/* a9-a12: WordPiece trie + FastWP encode */

typedef struct { uint32_t *p; uint32_t n; } poplist;

struct orc_wp {
  uint32_t n_vocab;
  uint32_t n_nodes, cap_nodes;
  uint32_t *ch;       /* utils.py:54 char */
  uint8_t *is_end;    /* utils.py:56 */
  int32_t *tok;       /* vocab index of chars_seen when the node ends a vocab token, else -1 */
  int32_t *link;      /* utils.py:61 failure_link, -1 = None */
  poplist *pops;      /* utils.py:60 failure_pops (as token ids) */
  uint32_t *first_child, *last_child, *next_sib; /* children in insertion order (dict order) */
  map64 edges;        /* (node << 32 | cp) -> child */
  uint32_t root, root_p, root_sharp;
  /* NaiveWP.encode_word("##") evaluated once (wordpiece.py:132-159, reached from :260-261) */
  poplist corner; int corner_nonterminating;
};

static uint32_t wp_new_node(orc_wp *w, uint32_t ch) {
  if (w->n_nodes == w->cap_nodes) {
    uint32_t c = w->cap_nodes ? w->cap_nodes * 2 : 1024;
    w->ch = (uint32_t *)realloc(w->ch, c * 4); w->is_end = (uint8_t *)realloc(w->is_end, c);
    w->tok = (int32_t *)realloc(w->tok, c * 4); w->link = (int32_t *)realloc(w->link, c * 4);
    w->pops = (poplist *)realloc(w->pops, c * sizeof(poplist));
    w->first_child = (uint32_t *)realloc(w->first_child, c * 4);
    w->last_child = (uint32_t *)realloc(w->last_child, c * 4);
    w->next_sib = (uint32_t *)realloc(w->next_sib, c * 4);
    w->cap_nodes = c;
  }
  uint32_t k = w->n_nodes++;
  w->ch[k] = ch; w->is_end[k] = 0; w->tok[k] = -1; w->link[k] = -1;
  w->pops[k].p = NULL; w->pops[k].n = 0;
  w->first_child[k] = w->last_child[k] = w->next_sib[k] = ~0u;
  return k;
}
static int64_t wp_child(const orc_wp *w, uint32_t node, uint32_t cp) {
  const uint64_t *v = map_find(&w->edges, PAIR_KEY(node, cp));
  return v ? (int64_t)*v : -1;
}
/* utils.py:87-105 */
static uint32_t wp_insert(orc_wp *w, const uint32_t *s, uint64_t n) {
  uint32_t node = w->root;
  for (uint64_t i = 0; i < n; i++) {
    int64_t c = wp_child(w, node, s[i]);
    if (c < 0) {
      uint32_t k = wp_new_node(w, s[i]);
      map_put(&w->edges, PAIR_KEY(node, s[i]), k);
      if (w->first_child[node] == ~0u) w->first_child[node] = k; else w->next_sib[w->last_child[node]] = k;
      w->last_child[node] = k;
      c = k;
    }
    node = (uint32_t)c;
  }
  w->is_end[node] = 1;
  return node;
}
static void pops_set(poplist *d, const uint32_t *a, uint32_t na, const uint32_t *b, uint32_t nb) {
  d->n = na + nb;
  d->p = (uint32_t *)malloc((d->n + 1) * 4);
  if (na) memcpy(d->p, a, na * 4);
  if (nb) memcpy(d->p + na, b, nb * 4);
}

orc_wp *orc_wp_new(const uint32_t *blob, const uint64_t *off, uint32_t n_vocab) {
  orc_wp *w = (orc_wp *)calloc(1, sizeof *w);
  w->n_vocab = n_vocab;
  map_init(&w->edges, 1 << 16);
  w->root = wp_new_node(w, 0);   /* utils.py:77 */
  w->root_p = wp_new_node(w, 0); /* utils.py:79 detached, childless */
  static const uint32_t sharp[2] = {'#', '#'};
  w->root_sharp = wp_insert(w, sharp, 2); /* utils.py:81 */
  for (uint32_t v = 0; v < n_vocab; v++) { /* utils.py:83-84 */
    uint32_t node = wp_insert(w, blob + off[v], off[v + 1] - off[v]);
    if (w->tok[node] < 0) w->tok[node] = (int32_t)v;
  }
  /* utils.py:108-139 precompute */
  vec32 queue = {0};
  v32_push(&queue, w->root); v32_push(&queue, w->root_sharp);
  for (uint64_t qh = 0; qh < queue.n; qh++) {
    uint32_t u = queue.p[qh];
    for (uint32_t c = w->first_child[u]; c != ~0u; c = w->next_sib[c]) {
      if (c == w->root_sharp) continue; /* utils.py:119-120 */
      uint32_t chr = w->ch[c];
      if (w->is_end[c]) { /* utils.py:121-123 */
        w->link[c] = (int32_t)w->root_sharp;
        uint32_t self = (uint32_t)w->tok[c];
        pops_set(&w->pops[c], &self, 1, NULL, 0);
      } else { /* utils.py:124-132 */
        int32_t f = w->link[u];
        vec32 acc = {0};
        while (f >= 0 && wp_child(w, (uint32_t)f, chr) < 0) {
          for (uint32_t k = 0; k < w->pops[f].n; k++) v32_push(&acc, w->pops[f].p[k]);
          f = w->link[f];
        }
        if (f >= 0) {
          w->link[c] = (int32_t)wp_child(w, (uint32_t)f, chr);
          pops_set(&w->pops[c], w->pops[u].p, w->pops[u].n, acc.p, (uint32_t)acc.n);
        }
        free(acc.p);
      }
      if (!py_isalnum(chr)) w->link[c] = (int32_t)w->root_p; /* utils.py:136-137 */
      v32_push(&queue, c);
    }
  }
  free(queue.p);

  /* NaiveWP.encode_word("##"), wordpiece.py:144-159: the word is always a run of '#'.  State = its
   * length L; each round takes the longest '#'*i in vocab (i <= L), then L = L - i (+2 if > 0). */
  {
    vec32 chain = {0}; /* chain.p[d] = node of '#'*d */
    v32_push(&chain, w->root);
    for (;;) {
      int64_t c = wp_child(w, chain.p[chain.n - 1], '#');
      if (c < 0) break;
      v32_push(&chain, (uint32_t)c);
    }
    uint64_t D = chain.n - 1, L = 2, guard = 0;
    vec32 toks = {0};
    uint8_t *visited = (uint8_t *)calloc(D + 8, 1);
    for (;;) {
      uint64_t i = L < D ? L : D;
      while (i > 0 && !(w->is_end[chain.p[i]] && w->tok[chain.p[i]] >= 0)) i--;
      if (i == 0) { toks.n = 0; v32_push(&toks, n_vocab + 1); break; } /* :148-149 ["[UNK]"] */
      v32_push(&toks, (uint32_t)w->tok[chain.p[i]]);
      L -= i;
      if (L == 0) break;
      L += 2; /* :155-156 */
      if (L < D + 8) { if (visited[L]) { w->corner_nonterminating = 1; break; } visited[L] = 1; }
      if (++guard > 1000000) { w->corner_nonterminating = 1; break; }
    }
    free(visited); free(chain.p);
    if (!w->corner_nonterminating) pops_set(&w->corner, toks.p, (uint32_t)toks.n, NULL, 0);
    free(toks.p);
  }
  return w;
}

void orc_wp_free(orc_wp *w) {
  if (!w) return;
  for (uint32_t k = 0; k < w->n_nodes; k++) free(w->pops[k].p);
  free(w->corner.p);
  free(w->ch); free(w->is_end); free(w->tok); free(w->link); free(w->pops);
  free(w->first_child); free(w->last_child); free(w->next_sib);
  map_free(&w->edges); free(w);
}
uint32_t orc_wp_n_nodes(const orc_wp *w) { return w->n_nodes; }

/* source/wordpiece.py:233-316.  sp(i) reads `text.lower() + " "` without materialising it. */
uint64_t orc_wp_tokenize(const orc_wp *w, const uint32_t *s, uint64_t n, uint32_t *out, uint64_t cap, int *status) {
  const uint64_t N = n + 1; /* wordpiece.py:248 */
#define SP(i) ((i) < n ? s[(i)] : (uint32_t)' ')
#define BNDRY(i) (((i) > 0 && wp_ispunc(SP((i) - 1))) || py_isspace(SP(i)) || wp_ispunc(SP(i))) /* :285 */
#define EMIT(x) do { if (nt < cap) out[nt] = (x); nt++; } while (0)
  uint64_t nt = 0, i = 0;
  *status = ORC_WP_OK;
  while (i < N) { /* :251 */
    const uint64_t seg_i = i, seg_nt = nt;
    /* matchloop, :291-316 */
    uint32_t node = w->root;
    int stop = 0;
    while (i < N && !stop) {
      int64_t c;
      while ((c = wp_child(w, node, SP(i))) < 0) {
        if (w->link[node] < 0) { stop = 1; break; }
        for (uint32_t k = 0; k < w->pops[node].n; k++) EMIT(w->pops[node].p[k]);
        node = (uint32_t)w->link[node];
      }
      if (stop) break;
      node = (uint32_t)c; i++;
    }
    /* :255 -- iswdbndry(s, i) indexes seq[i]; with i == len(seq) that raises unless seq[i-1] is punct */
    if (i == N && !wp_ispunc(SP(i - 1))) { *status = ORC_WP_INDEXERROR; return seg_nt; }
    int bnd = (i == N) ? 1 : BNDRY(i);
    if (!bnd || !(node == w->root || node == w->root_sharp || node == w->root_p)) {
      nt = seg_nt; EMIT(w->n_vocab); /* :257 "['UNK']" */
    } else if (node == w->root_sharp && nt == seg_nt) { /* :260-261 */
      if (w->corner_nonterminating) { *status = ORC_WP_NONTERMINATING; return seg_nt; }
      for (uint32_t k = 0; k < w->corner.n; k++) EMIT(w->corner.p[k]);
    }
    while (i < N && !BNDRY(i)) i++;        /* :265-266 */
    while (i < N && py_isspace(SP(i))) i++; /* :268-269 */
    if (i == seg_i) { *status = ORC_WP_NONTERMINATING; return nt; } /* same state again: loops forever */
  }
#undef SP
#undef BNDRY
#undef EMIT
  return nt;
}

/* A sentence whose status is not OK contributes zero tokens (the reference returns nothing for it).
 * Tokens past out_cap are counted but not stored. */
uint64_t orc_wp_tokenize_batch(const orc_wp *w, const uint32_t *text, const uint64_t *sent_off, uint64_t n_sent,
                               uint32_t *out, uint64_t out_cap, uint64_t *out_off, uint8_t *status) {
  uint64_t nt = 0;
  for (uint64_t s = 0; s < n_sent; s++) {
    int st;
    out_off[s] = nt;
    uint64_t n = sent_off[s + 1] - sent_off[s];
    uint64_t room = nt < out_cap ? out_cap - nt : 0;
    uint64_t k = orc_wp_tokenize(w, text + sent_off[s], n, room ? out + nt : NULL, room, &st);
    if (status) status[s] = (uint8_t)st;
    if (st == ORC_WP_OK) nt += k;
  }
  out_off[n_sent] = nt;
  return nt;
}

int64_t orc_wp_corner(const orc_wp *w, uint32_t *out, uint64_t cap) {
  if (w->corner_nonterminating) return -1;
  for (uint32_t k = 0; k < w->corner.n && k < cap; k++) out[k] = w->corner.p[k];
  return (int64_t)w->corner.n;
}

/* ------------------------------------------------------------------------------------------ */
/* All-host-cores form of the two batch encoders, for bench.py's cpu_baseline ("1 thread and all host cores", SURVEY.md
 * section 8d).  Sentences are independent (source/bpe.py:245-249, source/wordpiece.py:233-270 keep no state across calls):
 * thread k encodes a contiguous range into a buffer of its own, then the ranges are concatenated in order -- the result
 * is the single-thread result, byte for byte. */
typedef struct {
  const orc_bpe *bpe; const orc_wp *wp;
  const uint32_t *text; const uint64_t *sent_off;
  uint64_t lo, hi;
  uint32_t *buf; uint64_t cap, n_tok;
  uint64_t *cnt;     /* tokens per sentence, global array */
  uint8_t *status;   /* global array or NULL */
} mt_job;

static void *mt_worker(void *arg) {
  mt_job *j = (mt_job *)arg;
  uint64_t nt = 0;
  for (uint64_t s = j->lo; s < j->hi; s++) {
    const uint32_t *p = j->text + j->sent_off[s];
    uint64_t n = j->sent_off[s + 1] - j->sent_off[s], k;
    if (j->bpe) {
      k = orc_bpe_tokenize(j->bpe, p, n, j->buf + nt);
    } else {
      int st;
      k = orc_wp_tokenize(j->wp, p, n, j->buf + nt, j->cap - nt, &st);
      if (j->status) j->status[s] = (uint8_t)st;
      if (st != ORC_WP_OK) k = 0;
    }
    j->cnt[s] = k;
    nt += k;
  }
  j->n_tok = nt;
  return NULL;
}

static uint64_t mt_run(const orc_bpe *bpe, const orc_wp *wp, const uint32_t *text, const uint64_t *sent_off, uint64_t n_sent,
                       uint32_t *out, uint64_t out_cap, uint64_t *out_off, uint8_t *status, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if ((uint64_t)n_threads > n_sent) n_threads = n_sent ? (int)n_sent : 1;
  mt_job *jobs = (mt_job *)calloc((size_t)n_threads, sizeof *jobs);
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof *th);
  uint64_t *cnt = (uint64_t *)calloc(n_sent + 1, 8);
  /* ranges of about equal text, so the threads finish together */
  uint64_t total = sent_off[n_sent] - sent_off[0], s = 0;
  for (int k = 0; k < n_threads; k++) {
    mt_job *j = &jobs[k];
    j->bpe = bpe; j->wp = wp; j->text = text; j->sent_off = sent_off; j->cnt = cnt; j->status = status;
    j->lo = s;
    uint64_t want = sent_off[0] + total * (uint64_t)(k + 1) / (uint64_t)n_threads;
    while (s < n_sent && (sent_off[s + 1] <= want || k == n_threads - 1)) s++;
    j->hi = s;
    j->cap = 4 * (sent_off[j->hi] - sent_off[j->lo]) + 64 * (j->hi - j->lo) + 64; /* the bound oracle.py's single-thread wrapper uses */
    j->buf = (uint32_t *)malloc(j->cap * 4);
  }
  for (int k = 0; k < n_threads; k++) pthread_create(&th[k], NULL, mt_worker, &jobs[k]);
  uint64_t nt = 0;
  for (int k = 0; k < n_threads; k++) {
    pthread_join(th[k], NULL);
    uint64_t room = nt < out_cap ? out_cap - nt : 0, n = jobs[k].n_tok < room ? jobs[k].n_tok : room;
    if (n) memcpy(out + nt, jobs[k].buf, n * 4);
    nt += jobs[k].n_tok;
    free(jobs[k].buf);
  }
  uint64_t o = 0;
  for (uint64_t q = 0; q < n_sent; q++) { out_off[q] = o; o += cnt[q]; }
  out_off[n_sent] = o;
  free(cnt); free(th); free(jobs);
  return nt;
}

uint64_t orc_bpe_tokenize_batch_mt(const orc_bpe *m, const uint32_t *text, const uint64_t *sent_off, uint64_t n_sent,
                                   uint32_t *out, uint64_t *out_off, int n_threads) {
  return mt_run(m, NULL, text, sent_off, n_sent, out, ~0ull, out_off, NULL, n_threads);
}

uint64_t orc_wp_tokenize_batch_mt(const orc_wp *w, const uint32_t *text, const uint64_t *sent_off, uint64_t n_sent,
                                  uint32_t *out, uint64_t out_cap, uint64_t *out_off, uint8_t *status, int n_threads) {
  return mt_run(NULL, w, text, sent_off, n_sent, out, out_cap, out_off, status, n_threads);
}
