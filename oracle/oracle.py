"""ctypes binding of the CPU oracle (oracle/swt_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package never does.  The classes mirror the reference's surface just enough for parity tests to read like
the reference's own usage (`tokenize(text) -> List[str]`, `train(corpus, max_vocab)`, `merges_list`).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

SYM_BASE = 0x110000
CONT_FLAG = 0x80000000
WP_OK, WP_NONTERMINATING, WP_INDEXERROR = 0, 1, 2

_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    """Compile liboracle.so with gcc (idempotent)."""
    src = [os.path.join(_HERE, f) for f in ("swt_oracle.c", "swt_oracle.h", "unicode_classes.inc")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src if os.path.exists(s))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "_build/liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        sig = {
            "orc_class": (C.c_uint, [C.c_uint32]),
            "orc_pretokenize": (C.c_uint64, [_u32p, C.c_uint64, _u64p, _u64p, C.c_uint64]),
            "orc_bpe_new": (C.c_void_p, [_u32p, _u64p, C.c_uint32]),
            "orc_bpe_free": (None, [C.c_void_p]),
            "orc_bpe_n_symbols": (C.c_uint32, [C.c_void_p]),
            "orc_bpe_symbol": (C.c_uint64, [C.c_void_p, C.c_uint32, _u32p, C.c_uint64]),
            "orc_bpe_merge_ids": (None, [C.c_void_p, C.c_uint32, _u32p, _u32p, _u32p]),
            "orc_bpe_encode_word": (C.c_uint64, [C.c_void_p, _u32p, C.c_uint64, _u32p]),
            "orc_bpe_tokenize": (C.c_uint64, [C.c_void_p, _u32p, C.c_uint64, _u32p]),
            "orc_bpe_tokenize_batch": (C.c_uint64, [C.c_void_p, _u32p, _u64p, C.c_uint64, _u32p, _u64p]),
            "orc_bpe_tokenize_batch_mt": (C.c_uint64, [C.c_void_p, _u32p, _u64p, C.c_uint64, _u32p, _u64p, C.c_int]),
            "orc_wp_tokenize_batch_mt": (C.c_uint64, [C.c_void_p, _u32p, _u64p, C.c_uint64, _u32p, C.c_uint64, _u64p, _u8p, C.c_int]),
            "orc_train_new": (C.c_void_p, [_u32p, _u64p, C.c_uint64]),
            "orc_wptrain_new": (C.c_void_p, [_u32p, _u64p, C.c_uint64]),
            "orc_train_new_words": (C.c_void_p, [_u32p, _u64p, _u32p, C.c_uint64]),
            "orc_wp_score_bits": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64]),
            "orc_train_free": (None, [C.c_void_p]),
            "orc_train_n_words": (C.c_uint64, [C.c_void_p]),
            "orc_train_n_symbols": (C.c_uint64, [C.c_void_p]),
            "orc_train_vocab_size": (C.c_uint32, [C.c_void_p]),
            "orc_train_run": (C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32]),
            "orc_train_n_merges": (C.c_uint32, [C.c_void_p]),
            "orc_train_merge_ids": (None, [C.c_void_p, C.c_uint32, _u32p, _u32p, _u32p, _u64p]),
            "orc_train_symbol": (C.c_uint64, [C.c_void_p, C.c_uint32, _u32p, C.c_uint64]),
            "orc_train_export": (None, [C.c_void_p, _u32p, _u64p, _u32p]),
            "orc_wp_new": (C.c_void_p, [_u32p, _u64p, C.c_uint32]),
            "orc_wp_free": (None, [C.c_void_p]),
            "orc_wp_n_nodes": (C.c_uint32, [C.c_void_p]),
            "orc_wp_tokenize": (C.c_uint64, [C.c_void_p, _u32p, C.c_uint64, _u32p, C.c_uint64, C.POINTER(C.c_int)]),
            "orc_wp_tokenize_batch": (C.c_uint64, [C.c_void_p, _u32p, _u64p, C.c_uint64, _u32p, C.c_uint64, _u64p, _u8p]),
            "orc_wp_corner": (C.c_int64, [C.c_void_p, _u32p, C.c_uint64]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


# --------------------------------------------------------------------------------------------
# packing helpers (UTF-32 code points; lowercasing is Python's, as in the reference)

def cps(s):
    """str -> np.uint32 code points (lone surrogates pass through)."""
    if not s:
        return np.zeros(0, dtype=np.uint32)
    return np.frombuffer(s.encode("utf-32-le", "surrogatepass"), dtype=np.uint32).copy()


def uncps(a):
    return bytes(np.ascontiguousarray(a, dtype=np.uint32)).decode("utf-32-le", "surrogatepass")


def pack(strings):
    """list[str] -> (blob uint32, offsets uint64[n+1])"""
    lens = np.fromiter((len(s) for s in strings), dtype=np.uint64, count=len(strings))
    off = np.zeros(len(strings) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    blob = cps("".join(strings))
    assert blob.size == int(off[-1])
    return blob, off


def _p32(a):
    return a.ctypes.data_as(_u32p)


def _p64(a):
    return a.ctypes.data_as(_u64p)


def pretokenize(lowered):
    """a1 on one lowercased string -> list of words (source/utils.py:27)."""
    a = cps(lowered)
    n = a.size
    st = np.zeros(n + 1, dtype=np.uint64)
    en = np.zeros(n + 1, dtype=np.uint64)
    k = lib().orc_pretokenize(_p32(a), n, _p64(st), _p64(en), n + 1)
    return [lowered[int(st[i]):int(en[i])] for i in range(k)]


class OracleBPE:
    """FastBPE restated (source/bpe.py:192-263): load a merges list, tokenize."""

    def __init__(self, merges_list):
        self.merges_list = [tuple(p) for p in merges_list]
        flat = [s for p in self.merges_list for s in p]
        blob, off = pack(flat)
        self._h = lib().orc_bpe_new(_p32(blob), _p64(off), len(self.merges_list))
        self._strings = {}

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_bpe_free(self._h)
            self._h = None

    def merge_ids(self):
        out = np.zeros((len(self.merges_list), 3), dtype=np.uint32)
        l, r, m = C.c_uint32(), C.c_uint32(), C.c_uint32()
        for i in range(len(self.merges_list)):
            lib().orc_bpe_merge_ids(self._h, i, C.byref(l), C.byref(r), C.byref(m))
            out[i] = (l.value, r.value, m.value)
        return out

    def symbol(self, sid):
        sid &= ~CONT_FLAG & 0xFFFFFFFF
        s = self._strings.get(sid)
        if s is None:
            buf = np.zeros(256, dtype=np.uint32)
            n = lib().orc_bpe_symbol(self._h, sid, _p32(buf), buf.size)
            if n > buf.size:
                buf = np.zeros(n, dtype=np.uint32)
                lib().orc_bpe_symbol(self._h, sid, _p32(buf), buf.size)
            s = self._strings[sid] = uncps(buf[:n])
        return s

    def decode(self, ids):
        return [("##" if int(t) & CONT_FLAG else "") + self.symbol(int(t)) for t in ids]

    def encode_word_ids(self, word):
        a = cps(word)
        out = np.zeros(max(a.size, 1), dtype=np.uint32)
        k = lib().orc_bpe_encode_word(self._h, _p32(a), a.size, _p32(out))
        return out[:k]

    def encode_word(self, word):
        if word == "":
            return [""]  # source/bpe.py:208
        return self.decode(self.encode_word_ids(word))

    def tokenize_ids(self, text):
        a = cps(text.lower())
        out = np.zeros(max(a.size, 1), dtype=np.uint32)
        k = lib().orc_bpe_tokenize(self._h, _p32(a), a.size, _p32(out))
        return out[:k]

    def tokenize(self, text):
        return self.decode(self.tokenize_ids(text))

    def tokenize_batch_ids(self, texts):
        blob, off = pack([t.lower() for t in texts])
        out = np.zeros(max(blob.size, 1), dtype=np.uint32)
        out_off = np.zeros(len(texts) + 1, dtype=np.uint64)
        lib().orc_bpe_tokenize_batch(self._h, _p32(blob), _p64(off), len(texts), _p32(out), _p64(out_off))
        return out[:int(out_off[-1])], out_off

    def tokenize_packed_mt(self, blob, off, n_threads):
        """already lowercased + packed (pack()) -> (ids, offsets) over n_threads host threads (cpu_baseline, all cores)"""
        n = int(off.size - 1)
        out = np.zeros(max(blob.size, 1), dtype=np.uint32)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        lib().orc_bpe_tokenize_batch_mt(self._h, _p32(blob), _p64(off), n, _p32(out), _p64(out_off), int(n_threads))
        return out[:int(out_off[-1])], out_off


class OracleBPETrainer:
    """NaiveBPE.train restated (source/bpe.py:50-112)."""

    def __init__(self, corpus):
        blob, off = pack([t.lower() for t in corpus])
        self._h = lib().orc_train_new(_p32(blob), _p64(off), len(corpus))

    @classmethod
    def from_words(cls, syms, word_off, freq):
        """the state after bpe.py:73-81 given directly: unique words (code points, CSR) with their frequencies"""
        self = cls.__new__(cls)
        syms = np.ascontiguousarray(syms, dtype=np.uint32)
        word_off = np.ascontiguousarray(word_off, dtype=np.uint64)
        freq = np.ascontiguousarray(freq, dtype=np.uint32)
        self._h = lib().orc_train_new_words(_p32(syms), _p64(word_off), _p32(freq), int(word_off.size - 1))
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_train_free(self._h)
            self._h = None

    n_words = property(lambda self: lib().orc_train_n_words(self._h))
    n_symbols = property(lambda self: lib().orc_train_n_symbols(self._h))
    vocab_size = property(lambda self: lib().orc_train_vocab_size(self._h))
    n_merges = property(lambda self: lib().orc_train_n_merges(self._h))

    def run(self, max_vocab, max_steps=0):
        return lib().orc_train_run(self._h, max_vocab, max_steps)

    def symbol(self, sid):
        buf = np.zeros(256, dtype=np.uint32)
        n = lib().orc_train_symbol(self._h, sid, _p32(buf), buf.size)
        if n > buf.size:
            buf = np.zeros(n, dtype=np.uint32)
            lib().orc_train_symbol(self._h, sid, _p32(buf), buf.size)
        return uncps(buf[:n])

    def merge_ids(self):
        n = self.n_merges
        out = np.zeros((n, 3), dtype=np.uint32)
        cnt = np.zeros(n, dtype=np.uint64)
        l, r, m, c = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        for i in range(n):
            lib().orc_train_merge_ids(self._h, i, C.byref(l), C.byref(r), C.byref(m), C.byref(c))
            out[i] = (l.value, r.value, m.value)
            cnt[i] = c.value
        return out, cnt

    @property
    def merges_list(self):
        ids, _ = self.merge_ids()
        return [(self.symbol(int(l)), self.symbol(int(r))) for l, r, _m in ids]

    def export(self):
        """Current unique-word stream: (symbols uint32, word_off uint64[W+1], freq uint32[W])."""
        W = self.n_words
        syms = np.zeros(max(self.n_symbols, 1), dtype=np.uint32)
        woff = np.zeros(W + 1, dtype=np.uint64)
        freq = np.zeros(max(W, 1), dtype=np.uint32)
        lib().orc_train_export(self._h, _p32(syms), _p64(woff), _p32(freq))
        return syms[:self.n_symbols], woff, freq[:W]


class OracleWPTrainer(OracleBPETrainer):
    """NaiveWP.train restated (source/wordpiece.py:29-103): same accessors; merge_ids()' counts are score bit patterns."""

    def __init__(self, corpus):
        blob, off = pack([t.lower() for t in corpus])
        self._h = lib().orc_wptrain_new(_p32(blob), _p64(off), len(corpus))

    @property
    def merged_tokens(self):
        ids, _ = self.merge_ids()
        return [self.symbol(int(m)) for _l, _r, m in ids]


def wp_score_bits(cnt, fl, fr):
    return int(lib().orc_wp_score_bits(cnt, fl, fr))


class OracleWP:
    """WPTrie_E2E + FastWP.tokenize restated (source/utils.py:66-139, source/wordpiece.py:233-316)."""

    UNK_E2E = "['UNK']"  # source/wordpiece.py:257
    UNK_NAIVE = "[UNK]"  # source/wordpiece.py:149

    def __init__(self, vocab_list):
        self.vocab_list = list(vocab_list)
        blob, off = pack(self.vocab_list)
        self._h = lib().orc_wp_new(_p32(blob), _p64(off), len(self.vocab_list))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_wp_free(self._h)
            self._h = None

    n_nodes = property(lambda self: lib().orc_wp_n_nodes(self._h))

    def decode(self, ids):
        n = len(self.vocab_list)
        return [self.vocab_list[t] if t < n else (self.UNK_E2E if t == n else self.UNK_NAIVE) for t in map(int, ids)]

    def corner(self):
        buf = np.zeros(64, dtype=np.uint32)
        n = lib().orc_wp_corner(self._h, _p32(buf), buf.size)
        return None if n < 0 else buf[:n].copy()

    def tokenize_ids(self, text):
        a = cps(text.lower())
        cap = 4 * a.size + 64
        out = np.zeros(cap, dtype=np.uint32)
        st = C.c_int()
        k = lib().orc_wp_tokenize(self._h, _p32(a), a.size, _p32(out), cap, C.byref(st))
        assert k <= cap
        return out[:k], st.value

    def tokenize(self, text):
        ids, st = self.tokenize_ids(text)
        if st == WP_NONTERMINATING:
            raise RuntimeError("reference FastWP.tokenize does not terminate on this input")
        if st == WP_INDEXERROR:
            raise IndexError("string index out of range")
        return self.decode(ids)

    def tokenize_batch_ids(self, texts):
        blob, off = pack([t.lower() for t in texts])
        cap = 4 * blob.size + 64 * len(texts) + 64
        out = np.zeros(cap, dtype=np.uint32)
        out_off = np.zeros(len(texts) + 1, dtype=np.uint64)
        status = np.zeros(max(len(texts), 1), dtype=np.uint8)
        tot = lib().orc_wp_tokenize_batch(self._h, _p32(blob), _p64(off), len(texts), _p32(out), cap,
                                          _p64(out_off), status.ctypes.data_as(_u8p))
        assert tot <= cap
        return out[:tot], out_off, status[:len(texts)]

    def tokenize_packed_mt(self, blob, off, n_threads):
        """already lowercased + packed -> (ids, offsets, status) over n_threads host threads"""
        n = int(off.size - 1)
        cap = 4 * blob.size + 64 * n + 64
        out = np.zeros(cap, dtype=np.uint32)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        status = np.zeros(max(n, 1), dtype=np.uint8)
        tot = lib().orc_wp_tokenize_batch_mt(self._h, _p32(blob), _p64(off), n, _p32(out), cap, _p64(out_off),
                                             status.ctypes.data_as(_u8p), int(n_threads))
        assert tot <= cap
        return out[:tot], out_off, status[:n]
