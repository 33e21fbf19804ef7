// swt_pyhost.c -- the host side ABOVE the C ABI for a Python caller: list[str] -> the UTF-8 of "\0".join(texts) in one pass
// over the strings' internal representation (PEP 393), into a buffer the caller owns.  "\0".join(texts).encode("utf-8",
// "surrogatepass") + the isinstance check + bytes.count cost ~18 ms for the 85,000 sentences of S85k in CPython; this is ~2 ms.
// Loaded with ctypes.PyDLL (the GIL is held, nothing here calls back into the interpreter).  Not part of libswt_hip.so:
// that library knows nothing of Python.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>
#include <emmintrin.h>  // SSE2: part of every x86-64
#include <tmmintrin.h>  // SSSE3: used behind __builtin_cpu_supports only

// Upper bound of the joined size in bytes (separators included).  -1: not a list; -2 - i: item i is not a str.
long long swt_py_join_bound(PyObject *list) {
  if (!PyList_Check(list)) return -1;
  const Py_ssize_t n = PyList_GET_SIZE(list);
  long long total = n;
  for (Py_ssize_t i = 0; i < n; i++) {
    PyObject *o = PyList_GET_ITEM(list, i);
    if (!PyUnicode_Check(o)) return -2 - (long long)i;
    if (PyUnicode_READY(o) < 0) { PyErr_Clear(); return -2 - (long long)i; }
    const long long len = PyUnicode_GET_LENGTH(o);
    if (PyUnicode_IS_ASCII(o)) total += len;
    else {
      const int kind = PyUnicode_KIND(o);
      total += len * (kind == PyUnicode_1BYTE_KIND ? 2 : kind == PyUnicode_2BYTE_KIND ? 3 : 4);
    }
  }
  return total;
}

static inline uint8_t *put_cp(uint8_t *d, uint32_t c) {
  if (c < 0x80) { *d++ = (uint8_t)c; }
  else if (c < 0x800) { *d++ = (uint8_t)(0xC0 | (c >> 6)); *d++ = (uint8_t)(0x80 | (c & 0x3F)); }
  else if (c < 0x10000) {  // lone surrogates included: errors="surrogatepass"
    *d++ = (uint8_t)(0xE0 | (c >> 12)); *d++ = (uint8_t)(0x80 | ((c >> 6) & 0x3F)); *d++ = (uint8_t)(0x80 | (c & 0x3F));
  } else {
    *d++ = (uint8_t)(0xF0 | (c >> 18)); *d++ = (uint8_t)(0x80 | ((c >> 12) & 0x3F)); *d++ = (uint8_t)(0x80 | ((c >> 6) & 0x3F));
    *d++ = (uint8_t)(0x80 | (c & 0x3F));
  }
  return d;
}

// Two-byte strings (PEP 393 kind 2: one 'l-stroke' makes the whole sentence two-byte), mostly ASCII in practice.
// SSE2 form: eight code points at a time, the ASCII runs between the others copied as words.
static uint8_t *ucs2_sse2(const uint16_t *p, Py_ssize_t len, uint8_t *d) {
  const __m128i hi = _mm_set1_epi16((short)0xFF80), zero = _mm_setzero_si128();
  Py_ssize_t k = 0;
  for (; k + 8 <= len; k += 8) {
    const __m128i x = _mm_loadu_si128((const __m128i *)(p + k));
    // bit j: code point j of the eight is not ASCII
    unsigned m = (unsigned)_mm_movemask_epi8(_mm_packs_epi16(_mm_cmpeq_epi16(_mm_and_si128(x, hi), zero), zero)) ^ 0xFFu;
    uint64_t asc;  // the eight low bytes (right for the ASCII ones)
    _mm_storel_epi64((__m128i *)&asc, _mm_packus_epi16(_mm_and_si128(x, _mm_set1_epi16(0x00FF)), zero));
    unsigned pos = 0;
    while (m & 0xFFu) {  // ASCII run up to the next other code point (eight bytes stored, the run's length kept), then that one
      const unsigned j = (unsigned)__builtin_ctz(m);
      memcpy(d, &asc, 8);
      d += j - pos;
      d = put_cp(d, p[k + j]);
      asc = j + 1 - pos >= 8 ? 0 : asc >> (8 * (j + 1 - pos));
      pos = j + 1;
      m &= m - 1;
    }
    memcpy(d, &asc, 8);
    d += 8 - pos;
  }
  for (; k < len; k++) d = put_cp(d, p[k]);
  return d;
}

// SSSE3 form (every x86-64 server since 2006; chosen at run time): no branch on the data.  Each of eight code points below
// U+0800 becomes a 16-bit lane holding its one or two UTF-8 bytes, and a byte shuffle picked by the 8-bit "has two bytes" mask
// closes the gaps; a group with a code point from U+0800 up goes one by one.
static uint8_t g_shuf[256][16], g_len[256];  // g_len[m] = 8 + the bits set in m (no popcnt instruction in baseline x86-64)
static int g_shuf_ready;
static void shuf_init(void) {
  for (int m = 0; m < 256; m++) {
    int o = 0;
    for (int j = 0; j < 8; j++) {
      g_shuf[m][o++] = (uint8_t)(2 * j);
      if ((m >> j) & 1) g_shuf[m][o++] = (uint8_t)(2 * j + 1);
    }
    g_len[m] = (uint8_t)o;
    while (o < 16) g_shuf[m][o++] = 0x80;
  }
  g_shuf_ready = 1;
}

__attribute__((target("ssse3"))) static uint8_t *ucs2_ssse3(const uint16_t *p, Py_ssize_t len, uint8_t *d) {
  const __m128i zero = _mm_setzero_si128();
  Py_ssize_t k = 0;
  for (; k + 8 <= len; k += 8) {
    const __m128i x = _mm_loadu_si128((const __m128i *)(p + k));
    if (_mm_movemask_epi8(_mm_cmpeq_epi16(_mm_and_si128(x, _mm_set1_epi16((short)0xF800)), zero)) != 0xFFFF) {
      for (int j = 0; j < 8; j++) d = put_cp(d, p[k + j]);
      continue;
    }
    const __m128i ascii = _mm_cmpeq_epi16(_mm_and_si128(x, _mm_set1_epi16((short)0xFF80)), zero);
    const unsigned m = (unsigned)_mm_movemask_epi8(_mm_packs_epi16(ascii, zero)) ^ 0xFFu;
    // low byte 0xC0 | c >> 6, high byte 0x80 | (c & 0x3F)
    const __m128i two = _mm_or_si128(_mm_or_si128(_mm_srli_epi16(x, 6), _mm_set1_epi16(0x00C0)),
                                     _mm_slli_epi16(_mm_or_si128(_mm_and_si128(x, _mm_set1_epi16(0x003F)), _mm_set1_epi16(0x0080)), 8));
    const __m128i v = _mm_or_si128(_mm_and_si128(ascii, x), _mm_andnot_si128(ascii, two));
    _mm_storeu_si128((__m128i *)d, _mm_shuffle_epi8(v, _mm_loadu_si128((const __m128i *)g_shuf[m])));
    d += g_len[m];
  }
  for (; k < len; k++) d = put_cp(d, p[k]);
  return d;
}

// ---- one string: its exact UTF-8 length, and its bytes ---------------------------------------------------------------------
static inline long long utf8_size(PyObject *o) {
  const Py_ssize_t len = PyUnicode_GET_LENGTH(o);
  if (PyUnicode_IS_ASCII(o)) return (long long)len;
  const int kind = PyUnicode_KIND(o);
  const void *data = PyUnicode_DATA(o);
  long long n = (long long)len;
  if (kind == PyUnicode_1BYTE_KIND) {
    const uint8_t *p = (const uint8_t *)data;
    for (Py_ssize_t k = 0; k < len; k++) n += p[k] >> 7;
  } else if (kind == PyUnicode_2BYTE_KIND) {
    const uint16_t *p = (const uint16_t *)data;
    Py_ssize_t k = 0;
    const __m128i zero = _mm_setzero_si128();
    __m128i acc = zero;  // per 16-bit lane: (c >= 0x80) + (c >= 0x800), summed
    for (; k + 8 <= len; k += 8) {
      const __m128i x = _mm_loadu_si128((const __m128i *)(p + k));
      const __m128i ge80 = _mm_cmpeq_epi16(_mm_and_si128(x, _mm_set1_epi16((short)0xFF80)), zero);   // 0xFFFF where c < 0x80
      const __m128i ge800 = _mm_cmpeq_epi16(_mm_and_si128(x, _mm_set1_epi16((short)0xF800)), zero);  // 0xFFFF where c < 0x800
      // lanes hold -1 where the test FAILS to add: count the zeros instead: 2 + ge80 + ge800 (each -1 or 0)
      acc = _mm_add_epi16(acc, _mm_add_epi16(_mm_set1_epi16(2), _mm_add_epi16(ge80, ge800)));
      if ((k & 0x3FF8) == 0x3FF8) {  // flush before a 16-bit lane can overflow (<= 2 per step)
        uint16_t l[8];
        _mm_storeu_si128((__m128i *)l, acc);
        for (int j = 0; j < 8; j++) n += l[j];
        acc = zero;
      }
    }
    uint16_t l[8];
    _mm_storeu_si128((__m128i *)l, acc);
    for (int j = 0; j < 8; j++) n += l[j];
    for (; k < len; k++) n += (p[k] >= 0x80) + (p[k] >= 0x800);
  } else {
    const uint32_t *p = (const uint32_t *)data;
    for (Py_ssize_t k = 0; k < len; k++) n += (p[k] >= 0x80) + (p[k] >= 0x800) + (p[k] >= 0x10000);
  }
  return n;
}

// `exact`: nothing is stored behind the string's last byte (the vector forms store whole 16-byte lanes and advance by what was
// valid: harmless while the next string of the SAME thread overwrites the excess, fatal at the end of a thread's range, where
// the bytes behind belong to another thread -- the first GPU-box run of the threaded join came back with one stray zero byte per
// range boundary, which the device rightly refused: "the joined text holds 5006 separators, not n_sent - 1")
static inline uint8_t *put_str(PyObject *o, uint8_t *d, int ssse3, int exact) {
  const Py_ssize_t len = PyUnicode_GET_LENGTH(o);
  const int kind = PyUnicode_KIND(o);
  const void *data = PyUnicode_DATA(o);
  if (PyUnicode_IS_ASCII(o)) {
    memcpy(d, data, (size_t)len);
    return d + len;
  }
  if (kind == PyUnicode_1BYTE_KIND) {
    const uint8_t *p = (const uint8_t *)data;
    for (Py_ssize_t k = 0; k < len; k++) d = put_cp(d, p[k]);
  } else if (kind == PyUnicode_2BYTE_KIND) {
    if (exact) {
      const uint16_t *p = (const uint16_t *)data;
      for (Py_ssize_t k = 0; k < len; k++) d = put_cp(d, p[k]);
    } else {
      d = ssse3 ? ucs2_ssse3((const uint16_t *)data, len, d) : ucs2_sse2((const uint16_t *)data, len, d);
    }
  } else {
    const uint32_t *p = (const uint32_t *)data;
    for (Py_ssize_t k = 0; k < len; k++) d = put_cp(d, p[k]);
  }
  return d;
}

static long long count_zero_bytes(const uint8_t *q, const uint8_t *end) {
  const __m128i zero = _mm_setzero_si128(), one = _mm_set1_epi8(1);
  __m128i acc = zero;  // two 64-bit sums of the bytes that are zero
  for (; q + 16 <= end; q += 16)
    acc = _mm_add_epi64(acc, _mm_sad_epu8(_mm_and_si128(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)q), zero), one), zero));
  uint64_t part[2];
  _mm_storeu_si128((__m128i *)part, acc);
  long long n = (long long)(part[0] + part[1]);
  for (; q < end; q++) n += *q == 0;
  return n;
}

// ---- the join over several host threads -----------------------------------------------------------------------------------
// One core converts ~2 GB/s; the GPU side of an encode call takes the 8.5 MB of S85k in ~0.5 ms, so the join WAS the call
// (4 ms of 6).  The strings are immutable and the caller's thread holds the GIL for the whole call (nothing else in the
// interpreter runs), so worker threads may read the strings' buffers: two passes over contiguous ranges of the list -- exact
// UTF-8 sizes, a prefix sum over the ranges, then every range written straight to its final place.
#include <pthread.h>
#include <unistd.h>

typedef struct {
  PyObject **items;
  Py_ssize_t lo, hi;     // strings [lo, hi)
  uint8_t *dst;          // pass 2: where this range starts
  long long bytes;       // pass 1: out -- the range's size (separators in front of every string but the list's first)
  long long nul;         // pass 2: out -- zero bytes written
  int ssse3, pass;
} JoinJob;

static void *join_worker(void *arg) {
  JoinJob *j = (JoinJob *)arg;
  if (j->pass == 1) {
    long long n = 0;
    for (Py_ssize_t i = j->lo; i < j->hi; i++) n += utf8_size(j->items[i]) + (i ? 1 : 0);
    j->bytes = n;
  } else {
    uint8_t *d = j->dst;
    for (Py_ssize_t i = j->lo; i < j->hi; i++) {
      if (i) *d++ = 0;
      // every string but the list's first takes at least its separator byte, so the last 17 strings of a range cover the 16
      // bytes in front of the next range
      d = put_str(j->items[i], d, j->ssse3, j->hi - i <= 17);
    }
    j->nul = count_zero_bytes(j->dst, d);
  }
  return NULL;
}

static int join_threads(Py_ssize_t n, long long bound) {
  if (n < 4096 || bound < (1 << 20)) return 1;
  const char *e = getenv("SWT_JOIN_THREADS");
  long t = e && *e ? strtol(e, NULL, 10) : 0;
  if (t <= 0) {
    t = sysconf(_SC_NPROCESSORS_ONLN);
    if (t > 8) t = 8;  // the copy is memory-bound well before that many cores
  }
  if (t > 32) t = 32;
  return t < 1 ? 1 : (int)t;
}

// Writes the texts' UTF-8 with ONE zero byte between neighbours to dst (cap >= swt_py_join_bound + 32).  Returns the bytes written;
// *n_nul = zero code points INSIDE the texts (the separator form needs 0).  -1: the list changed under us / cap too small.
long long swt_py_join_fill(PyObject *list, uint8_t *dst, long long cap, long long *n_nul) {
  if (!PyList_Check(list) || !dst || !n_nul) return -1;
  const Py_ssize_t n = PyList_GET_SIZE(list);
  const int ssse3 = __builtin_cpu_supports("ssse3");
  if (ssse3 && !g_shuf_ready) shuf_init();
  PyObject **items = ((PyListObject *)list)->ob_item;
  long long bound = n;
  for (Py_ssize_t i = 0; i < n; i++) {  // (swt_py_join_bound has made every string ready; this is the check that the list is still that list)
    PyObject *o = items[i];
    if (!PyUnicode_Check(o) || PyUnicode_READY(o) < 0) { PyErr_Clear(); return -1; }
    const long long len = PyUnicode_GET_LENGTH(o);
    bound += PyUnicode_IS_ASCII(o) ? len : len * (PyUnicode_KIND(o) == PyUnicode_1BYTE_KIND ? 2 : PyUnicode_KIND(o) == PyUnicode_2BYTE_KIND ? 3 : 4);
  }
  if (bound + 17 > cap) return -1;  // 16 bytes of slack: the vector stores
  int T = join_threads(n, bound);
  JoinJob job[32];
  pthread_t th[32];
  for (int pass = 1; pass <= 2; pass++) {
    long long at = 0;
    for (int t = 0; t < T; t++) {
      job[t].items = items;
      job[t].lo = n * t / T;
      job[t].hi = n * (t + 1) / T;
      job[t].ssse3 = ssse3;
      job[t].pass = pass;
      if (pass == 2) { job[t].dst = dst + at; at += job[t].bytes; }
    }
    if (pass == 2 && at + 17 > cap) return -1;
    int started = 0;
    for (int t = 1; t < T; t++) {
      if (pthread_create(&th[t], NULL, join_worker, &job[t]) != 0) break;
      started = t;
    }
    join_worker(&job[0]);
    for (int t = started + 1; t < T; t++) join_worker(&job[t]);  // threads that could not be had: their ranges on this one
    for (int t = 1; t <= started; t++) pthread_join(th[t], NULL);
  }
  long long total = 0, nul = 0;
  for (int t = 0; t < T; t++) { total += job[t].bytes; nul += job[t].nul; }
  // a zero byte in UTF-8 is U+0000 and nothing else: what is not a separator came from inside a text
  *n_nul = nul - (n ? (long long)(n - 1) : 0);
  return total;
}

// ids -> the reference's output shape, List[List[str]]: sentence s gets [table[inv[k]] for k in off[s]..off[s+1]) (new lists, the
// strings shared).  Returns a new reference, or NULL with an exception set (index out of range, offsets not ascending).
static PyObject *nested_body(PyObject *table, const int32_t *inv, long long n_inv, const uint64_t *off, long long n_sent) {
  if (!PyList_Check(table) || n_sent < 0 || n_inv < 0 || (n_sent && !off) || (n_inv && !inv)) {
    PyErr_SetString(PyExc_ValueError, "swt_py_nested: bad argument");
    return NULL;
  }
  const Py_ssize_t n_tab = PyList_GET_SIZE(table);
  PyObject *outer = PyList_New((Py_ssize_t)n_sent);
  if (!outer) return NULL;
  for (long long s = 0; s < n_sent; s++) {
    const uint64_t a = off[s], b = off[s + 1];
    if (a > b || b > (uint64_t)n_inv) {
      Py_DECREF(outer);
      PyErr_SetString(PyExc_ValueError, "swt_py_nested: offsets must ascend within the ids");
      return NULL;
    }
    PyObject *inner = PyList_New((Py_ssize_t)(b - a));
    if (!inner) { Py_DECREF(outer); return NULL; }
    PyList_SET_ITEM(outer, (Py_ssize_t)s, inner);
    for (uint64_t k = a; k < b; k++) {
      const int32_t t = inv[k];
      if (t < 0 || t >= n_tab) {
        Py_DECREF(outer);  // the unset slots of `inner` are NULL: list_dealloc copes with them
        PyErr_SetString(PyExc_IndexError, "swt_py_nested: token index outside the table");
        return NULL;
      }
      PyObject *o = PyList_GET_ITEM(table, t);
      Py_INCREF(o);
      PyList_SET_ITEM(inner, (Py_ssize_t)(k - a), o);
    }
  }
  return outer;
}

// The distinct values of key[0..n) in order of first appearance: pos[key] (all -1 on entry, cap entries) becomes the value's
// rank, uniq[rank] the value.  Returns how many, -1 for a key outside [0, cap).
long long swt_py_distinct(const uint32_t *key, long long n, int32_t *pos, long long cap, uint32_t *uniq) {
  long long cnt = 0;
  for (long long k = 0; k < n; k++) {
    const uint32_t v = key[k];
    if ((long long)v >= cap) return -1;
    if (pos[v] < 0) {
      pos[v] = (int32_t)cnt;
      uniq[cnt++] = v;
    }
  }
  return cnt;
}

// swt_py_nested with the indices looked up on the way: sentence s gets [table[pos[key[k]]] for k in off[s]..off[s+1])
static PyObject *nested_via_body(PyObject *table, const uint32_t *key, long long n_key, const int32_t *pos, long long cap,
                                 const uint64_t *off, long long n_sent) {
  if (!PyList_Check(table) || n_sent < 0 || n_key < 0 || (n_sent && !off) || (n_key && (!key || !pos))) {
    PyErr_SetString(PyExc_ValueError, "swt_py_nested_via: bad argument");
    return NULL;
  }
  const Py_ssize_t n_tab = PyList_GET_SIZE(table);
  PyObject *outer = PyList_New((Py_ssize_t)n_sent);
  if (!outer) return NULL;
  for (long long s = 0; s < n_sent; s++) {
    const uint64_t a = off[s], b = off[s + 1];
    if (a > b || b > (uint64_t)n_key) {
      Py_DECREF(outer);
      PyErr_SetString(PyExc_ValueError, "swt_py_nested_via: offsets must ascend within the ids");
      return NULL;
    }
    PyObject *inner = PyList_New((Py_ssize_t)(b - a));
    if (!inner) { Py_DECREF(outer); return NULL; }
    PyList_SET_ITEM(outer, (Py_ssize_t)s, inner);
    for (uint64_t k = a; k < b; k++) {
      const uint32_t v = key[k];
      const int32_t t = (long long)v < cap ? pos[v] : -1;
      if (t < 0 || t >= n_tab) {
        Py_DECREF(outer);
        PyErr_SetString(PyExc_IndexError, "swt_py_nested_via: token outside the table");
        return NULL;
      }
      PyObject *o = PyList_GET_ITEM(table, t);
      Py_INCREF(o);
      PyList_SET_ITEM(inner, (Py_ssize_t)(k - a), o);
    }
  }
  return outer;
}

// The cyclic collector runs every few hundred container allocations and finds nothing to do among lists of strings; with
// 85,000 new lists it was half of the time.  It is switched off while the lists are built (they stay tracked).
#if PY_VERSION_HEX >= 0x030A0000
#define SWT_GC_OFF() const int gc_was_on = PyGC_Disable()
#define SWT_GC_BACK() do { if (gc_was_on) PyGC_Enable(); } while (0)
#else
#define SWT_GC_OFF() do { } while (0)
#define SWT_GC_BACK() do { } while (0)
#endif

PyObject *swt_py_nested(PyObject *table, const int32_t *inv, long long n_inv, const uint64_t *off, long long n_sent) {
  SWT_GC_OFF();
  PyObject *r = nested_body(table, inv, n_inv, off, n_sent);
  SWT_GC_BACK();
  return r;
}

PyObject *swt_py_nested_via(PyObject *table, const uint32_t *key, long long n_key, const int32_t *pos, long long cap,
                            const uint64_t *off, long long n_sent) {
  SWT_GC_OFF();
  PyObject *r = nested_via_body(table, key, n_key, pos, cap, off, n_sent);
  SWT_GC_BACK();
  return r;
}

// The per-merge bookkeeping of NaiveBPE.train (bpe.py:102-104) for a whole device run: for merge i of n,
//   ls, rs = spelling of left[i], right[i] (a code point below `base`, else strings[id - base]); joined = ls + rs;
//   merged = _SymbolTable.intern(joined)   (index: str -> k, strings: k -> str, id = base + k; a one-code-point string is its ord);
//   vocab.add(joined); merges.append((ls, rs));
// and it stops behind the first merge whose id is not first + i (two merges spelled the same string: the caller replays).
// Returns that merge's index (its id in *collided) or -1 when every id was the expected one; -2 with an exception set on error.
long long swt_py_bpe_merge_strings(const uint32_t *left, const uint32_t *right, long long n, uint32_t base, uint32_t first,
                                   PyObject *strings, PyObject *index, PyObject *vocab, PyObject *merges, uint32_t *collided) {
  if (!PyList_Check(strings) || !PyDict_Check(index) || !PySet_Check(vocab) || !PyList_Check(merges) || n < 0 || (n && (!left || !right))) {
    PyErr_SetString(PyExc_TypeError, "swt_py_bpe_merge_strings: bad argument");
    return -2;
  }
  for (long long i = 0; i < n; i++) {
    PyObject *sp[2] = {NULL, NULL};
    const uint32_t id[2] = {left[i], right[i]};
    for (int s = 0; s < 2; s++) {
      if (id[s] < base) sp[s] = PyUnicode_FromOrdinal((int)id[s]);
      else if ((Py_ssize_t)(id[s] - base) < PyList_GET_SIZE(strings)) { sp[s] = PyList_GET_ITEM(strings, id[s] - base); Py_INCREF(sp[s]); }
      else PyErr_SetString(PyExc_IndexError, "swt_py_bpe_merge_strings: symbol id without a string");
      if (!sp[s]) { Py_XDECREF(sp[0]); return -2; }
    }
    PyObject *joined = PyUnicode_Concat(sp[0], sp[1]);
    PyObject *pair = joined ? PyTuple_Pack(2, sp[0], sp[1]) : NULL;
    Py_DECREF(sp[0]);
    Py_DECREF(sp[1]);
    if (!joined || !pair) { Py_XDECREF(joined); Py_XDECREF(pair); return -2; }
    long long merged;
    if (PyUnicode_GET_LENGTH(joined) == 1) merged = (long long)PyUnicode_READ_CHAR(joined, 0);
    else {
      PyObject *k = PyDict_GetItemWithError(index, joined);  // borrowed
      if (k) merged = (long long)base + PyLong_AsLongLong(k);
      else {
        if (PyErr_Occurred()) { Py_DECREF(joined); Py_DECREF(pair); return -2; }
        const Py_ssize_t at = PyList_GET_SIZE(strings);
        PyObject *v = PyLong_FromSsize_t(at);
        const int bad = !v || PyDict_SetItem(index, joined, v) < 0 || PyList_Append(strings, joined) < 0;
        Py_XDECREF(v);
        if (bad) { Py_DECREF(joined); Py_DECREF(pair); return -2; }
        merged = (long long)base + at;
      }
    }
    const int bad = PySet_Add(vocab, joined) < 0 || PyList_Append(merges, pair) < 0;
    Py_DECREF(joined);
    Py_DECREF(pair);
    if (bad) return -2;
    if (merged != (long long)first + i) {
      if (collided) *collided = (uint32_t)merged;
      return i;
    }
  }
  return -1;
}
