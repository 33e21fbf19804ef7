"""ctypes binding of libswt_hip.so (include/swt.h).

The library is the product: there is no Python or CPU implementation of the hot path behind it.  If the
shared object has not been built, or a call needs a GPU that is not there, this module raises -- loudly.
"""
import ctypes as C
import os

import numpy as np

from . import _build

SYM_BASE = 0x110000
WP_CONT = 0x110000  # '##c' = WP_CONT + ord(c) in a WordPiece trainer
WP_MERGED_BASE = 0x220000
BPE_CONT = 0x80000000
BPE_RAW_WORDS = 1
BPE_NO_DEDUP = 2
WP_OK, WP_NONTERMINATING, WP_INDEXERROR = 0, 1, 2
ERR_NO_DEVICE, ERR_INVALID, ERR_CAPACITY, ERR_HIP, ERR_UNSUPPORTED, ERR_STATE, ERR_NOMEM, ERR_INTERNAL = -1, -2, -3, -4, -5, -6, -7, -8
CLS_BERT_WS, CLS_BERT_PUNCT, CLS_PY_SPACE, CLS_PY_ALNUM = 1, 2, 4, 8
NO_POS = 0xFFFFFFFFFFFFFFFF

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
vpp = C.POINTER(C.c_void_p)

# every symbol include/swt.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "swt_last_error": (C.c_char_p, []),
    "swt_version": (C.c_int, []),
    "swt_abi_selftest": (C.c_int, [C.c_int]),
    "swt_init": (C.c_int, [C.c_int]),
    "swt_device_count": (C.c_int, []),
    "swt_device_info": (C.c_int, [C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "swt_profile_enable": (C.c_int, [C.c_int]),
    "swt_profile_read": (C.c_int, [C.POINTER(C.c_double), u64p]),
    "swt_class_of": (C.c_uint, [C.c_uint32]),
    "swt_bpe_table_create": (C.c_int, [u32p, u32p, u32p, C.c_uint32, vpp]),
    "swt_bpe_table_destroy": (None, [C.c_void_p]),
    "swt_bpe_table_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "swt_wp_trie_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "swt_bpe_encode": (C.c_int, [C.c_void_p, u8p, u64p, C.c_uint64, u32p, C.c_uint64, u64p, u64p, C.c_uint32]),
    "swt_bpe_encode_joined": (C.c_int, [C.c_void_p, u8p, C.c_uint64, C.c_uint64, u32p, C.c_uint64, u64p, u64p, u8p, C.c_uint32]),
    "swt_bpe_encode_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_uint32, C.c_void_p]),
    "swt_wp_trie_create": (C.c_int, [u32p, u64p, C.c_uint32, vpp]),
    "swt_wp_trie_destroy": (None, [C.c_void_p]),
    "swt_wp_trie_stats": (C.c_int, [C.c_void_p, u32p, u32p, u32p]),
    "swt_wp_trie_corner": (C.c_int64, [C.c_void_p, u32p, C.c_uint64]),
    "swt_wp_trie_node": (C.c_int, [C.c_void_p, u32p, C.c_uint64, u32p, i32p, u8p, u32p, C.c_uint32, u32p]),
    "swt_wp_trie_node_path": (C.c_int, [C.c_void_p, C.c_uint32, u32p, C.c_uint64, u64p]),
    "swt_wp_encode": (C.c_int, [C.c_void_p, u8p, u64p, C.c_uint64, u32p, C.c_uint64, u64p, u8p, u64p]),
    "swt_wp_encode_joined": (C.c_int, [C.c_void_p, u8p, C.c_uint64, C.c_uint64, u32p, C.c_uint64, u64p, u8p, u64p, u8p]),
    "swt_wp_encode_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "swt_lower_of": (C.c_uint32, [C.c_uint32]),
    "swt_unidata_version": (C.c_char_p, []),
    "swt_utf8_lower": (C.c_int, [u8p, u64p, C.c_uint64, u8p]),
    "swt_utf8_prepare": (C.c_int, [u8p, C.c_uint64, u64p, C.c_uint64, u64p, u8p]),
    "swt_utf8_prepare_joined": (C.c_int, [u8p, C.c_uint64, C.c_uint64, u8p, u64p, u8p]),
    "swt_utf8_lower_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "swt_token_histogram": (C.c_int, [u32p, C.c_uint64, C.c_uint32, u64p, u64p]),
    "swt_token_histogram_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "swt_bpe_train_create_text": (C.c_int, [u8p, u64p, C.c_uint64, vpp]),
    "swt_bpe_train_create_joined": (C.c_int, [u8p, C.c_uint64, C.c_uint64, u8p, vpp]),
    "swt_wp_train_create_text": (C.c_int, [u8p, u64p, C.c_uint64, vpp]),
    "swt_wp_train_create_joined": (C.c_int, [u8p, C.c_uint64, C.c_uint64, u8p, vpp]),
    "swt_bpe_train_create_words": (C.c_int, [u32p, u64p, u32p, C.c_uint64, vpp]),
    "swt_bpe_train_destroy": (None, [C.c_void_p]),
    "swt_bpe_train_set_pos_base": (C.c_int, [C.c_void_p, C.c_uint64]),
    "swt_bpe_train_info": (C.c_int, [C.c_void_p, u64p, u64p, u32p, u64p]),
    "swt_bpe_train_base_symbols": (C.c_int, [C.c_void_p, u32p, C.c_uint32]),
    "swt_bpe_train_best": (C.c_int, [C.c_void_p, u32p, u32p, u64p, u64p, u64p]),
    "swt_bpe_train_apply": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "swt_bpe_train_run": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, u32p, u32p, u64p, u32p]),
    "swt_bpe_train_export": (C.c_int, [C.c_void_p, u32p, C.c_uint64, u64p, u32p]),
    "swt_bpe_train_histogram": (C.c_int, [C.c_void_p, u64p, u64p, C.c_uint64, u64p]),
    "swt_bpe_train_stats": (C.c_int, [C.c_void_p, u64p, C.c_uint32]),
    "swt_bpe_train_trace": (C.c_int, [C.c_void_p, u64p, C.c_uint64, u64p]),
    "swt_dist_unique_id": (C.c_int, [u8p]),
    "swt_dist_init": (C.c_int, [C.c_int, C.c_int, u8p, vpp]),
    "swt_dist_init_local": (C.c_int, [C.c_int, vpp]),
    "swt_dist_destroy": (None, [C.c_void_p]),
    "swt_dist_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "swt_bpe_train_shard_begin": (C.c_int, [vpp, C.c_uint32, C.c_void_p, u32p, C.c_uint32, u32p]),
    "swt_bpe_train_run_sharded": (C.c_int, [vpp, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, u32p, u32p, u64p, u32p]),
}


class SwtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libswt_hip error %d: %s" % (code, message))
        self.code = code


class NoDeviceError(SwtError):
    """Raised when a compute call is made without a HIP device: there is no CPU path."""


_lib = None


def lib():
    """Load libswt_hip.so (built by `_build.build()`); never falls back to anything else."""
    global _lib
    if _lib is None:
        path = _build.LIB_PATH
        if not os.path.exists(path):
            raise ImportError(
                "libswt_hip.so is not built (%s). Run `python -c \"import __graft_entry__ as g; g.build()\"` "
                "or `python subword-tokenizers_amd/_build.py`; there is no Python/CPU fallback." % path)
        # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  When
        # torch is importable, load it FIRST so that libswt_hip.so binds to the runtime torch already holds;
        # the other order leaves torch unable to see the GPU ("No HIP GPUs are available").
        if not os.environ.get("SWT_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = the .so is stale against include/swt.h
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc == 0:
        return
    msg = lib().swt_last_error().decode("utf-8", "replace")
    if rc == ERR_NO_DEVICE:
        raise NoDeviceError(rc, msg)
    raise SwtError(rc, msg)


def ptr(a, typ):
    return a.ctypes.data_as(typ)


def class_of(cp):
    return lib().swt_class_of(cp)


def device_count():
    return lib().swt_device_count()


def init(device=0):
    check(lib().swt_init(device))


OPT_DEDUP, OPT_DEDUP_TABLE_BITS, OPT_UNIQUE_TILE = 1, 2, 3
DEDUP_AUTO, DEDUP_NEVER, DEDUP_ALWAYS = 0, 1, 2


def profile_enable(on=True):
    """True/1 = time the dominant kernel of each path, 2 = time all kernels of a call, False/0 = off"""
    check(lib().swt_profile_enable(int(on)))


def profile_read():
    """-> (summed milliseconds of the dominant kernel, launches) since the last read"""
    ms, n = C.c_double(), C.c_uint64()
    check(lib().swt_profile_read(C.byref(ms), C.byref(n)))
    return ms.value, n.value


def device_info():
    n = C.c_int()
    name = C.create_string_buffer(256)
    check(lib().swt_device_info(C.byref(n), name, 256))
    return n.value, name.value.decode()


# --------------------------------------------------------------------------------------------------
# packing: what the reference does per sentence in Python (str.lower(), utils.py:27 / wordpiece.py:248)
# stays in Python; the device consumes the lowercased UTF-8 bytes of the whole batch.

def lower_of(cp):
    """the device lowercase table: lowercase code point, or 0xFFFFFFFF for the code points left to the host"""
    return int(lib().swt_lower_of(cp))


_lower_ok = None


def device_lower_ok():
    """The device's lowercase table speaks for this interpreter's str.lower(): both follow the same Unicode database.  If not,
    every batch is lowercased on the host, so that a text tokenizes the same whatever the size of the batch it comes in."""
    global _lower_ok
    if _lower_ok is None:
        import unicodedata
        _lower_ok = lib().swt_unidata_version().decode() == unicodedata.unidata_version
    return _lower_ok


_pyhost = None


def pyhost():
    """csrc/swt_pyhost.c through ctypes.PyDLL (the GIL stays held), or None where it could not be built."""
    global _pyhost
    if _pyhost is None:
        path = _build.build_pyhost()
        dll = False
        if path:
            try:
                dll = C.PyDLL(path)
                dll.swt_py_join_bound.argtypes = [C.py_object]
                dll.swt_py_join_bound.restype = C.c_longlong
                dll.swt_py_join_fill.argtypes = [C.py_object, C.c_void_p, C.c_longlong, C.POINTER(C.c_longlong)]
                dll.swt_py_join_fill.restype = C.c_longlong
                dll.swt_py_nested.argtypes = [C.py_object, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong]
                dll.swt_py_nested.restype = C.py_object
                dll.swt_py_bpe_merge_strings.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_uint32, C.c_uint32, C.py_object, C.py_object,
                                                         C.py_object, C.py_object, C.POINTER(C.c_uint32)]
                dll.swt_py_bpe_merge_strings.restype = C.c_longlong
                dll.swt_py_distinct.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p]
                dll.swt_py_distinct.restype = C.c_longlong
                dll.swt_py_nested_via.argtypes = [C.py_object, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong]
                dll.swt_py_nested_via.restype = C.py_object
            except OSError:
                dll = False
        _pyhost = dll
    return _pyhost or None


def join_texts(texts, error="Text must be a string."):
    """list[str] -> (uint8 array: the UTF-8 (surrogatepass) of the texts with ONE zero byte between neighbours, the number of
    U+0000 inside the texts).  TypeError(error) when an item is not a str.  One pass over the strings in C where
    csrc/swt_pyhost.c could be built, str.join + str.encode otherwise."""
    h = pyhost()
    if h is not None:
        bound = h.swt_py_join_bound(texts)
        if bound < 0:
            raise TypeError(error)
        buf = np.empty(bound + 32, dtype=np.uint8)
        n_nul = C.c_longlong()
        w = h.swt_py_join_fill(texts, buf.ctypes.data, bound + 32, C.byref(n_nul))
        if w < 0:
            raise RuntimeError("swt_py_join_fill: the list changed during the call")
        return buf[:w], int(n_nul.value)
    if not all(isinstance(t, str) for t in texts):
        raise TypeError(error)
    data = "\x00".join(texts).encode("utf-8", "surrogatepass")
    return np.frombuffer(data, dtype=np.uint8), data.count(0) - max(len(texts) - 1, 0)


def nested_lists(table, inv, off):
    """[[table[inv[k]] for k in range(off[s], off[s + 1])] for s in range(len(off) - 1)]: token strings per sentence, the
    reference's output shape, from a list of strings, int32 indices into it and uint64 sentence offsets"""
    inv = np.ascontiguousarray(inv, dtype=np.int32)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n_sent = int(off.size - 1)
    h = pyhost()
    if h is not None:
        return h.swt_py_nested(table, inv.ctypes.data, int(inv.size), off.ctypes.data, n_sent)
    if inv.size and (int(inv.min()) < 0 or int(inv.max()) >= len(table)):
        raise IndexError("token index outside the table")
    arr = np.empty(len(table), dtype=object)
    arr[:] = table
    toks = arr[inv].tolist()
    return [toks[int(off[i]):int(off[i + 1])] for i in range(n_sent)]


def nested_lists_by_key(key, cap, spell, off):
    """The same for tokens named by sparse keys in [0, cap): spell(k) is called once per DISTINCT key (in order of first
    appearance) and every sentence gets references to those strings.  key: uint32 array."""
    key = np.ascontiguousarray(key, dtype=np.uint32)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    h = pyhost()
    if h is None:
        if key.size and int(key.max()) >= cap:
            raise IndexError("token outside the table")
        uniq, inv = np.unique(key, return_inverse=True)
        return nested_lists([spell(int(k)) for k in uniq.tolist()], inv, off)
    pos = np.full(cap, -1, dtype=np.int32)
    uniq = np.empty(min(cap, max(int(key.size), 1)), dtype=np.uint32)
    n = h.swt_py_distinct(key.ctypes.data, int(key.size), pos.ctypes.data, cap, uniq.ctypes.data)
    if n < 0:
        raise IndexError("token outside the table")
    table = [spell(k) for k in uniq[:n].tolist()]
    return h.swt_py_nested_via(table, key.ctypes.data, int(key.size), pos.ctypes.data, cap, off.ctypes.data, int(off.size - 1))


def pack_and_lower(texts):
    """list[str] -> (uint8 bytes of the LOWERCASED texts, uint64 byte offsets[n+1]).  The host joins (with U+0000 between the
    texts) and encodes once; the sentence offsets and str.lower() come from the device (swt_utf8_prepare_joined; texts that
    hold U+0000 themselves go by their code-point lengths, swt_utf8_prepare); the few sentences the device flags are lowercased
    here and spliced in."""
    n = len(texts)
    if (n <= 64 and sum(map(len, texts)) <= 16384) or not device_lower_ok():
        # the reference-style call (one sentence, or a few): a device round trip costs more than str.lower() here
        return pack_utf8([t.lower() for t in texts])
    off = np.zeros(n + 1, dtype=np.uint64)
    if n == 0:
        return np.zeros(0, dtype=np.uint8), off
    need = np.zeros(n, dtype=np.uint8)
    # one join with U+0000 between the texts; the device finds the separators (as long as the texts hold no U+0000 of their
    # own), closes the gaps, lowercases
    joined, n_nul = join_texts(texts)
    if joined.size + 1 == n:  # nothing but separators: every text is empty
        return np.zeros(0, dtype=np.uint8), off
    if n_nul == 0:
        buf = np.empty(joined.size - (n - 1), dtype=np.uint8)
        check(lib().swt_utf8_prepare_joined(ptr(joined, u8p), int(joined.size), n, ptr(buf, u8p), ptr(off, u64p), ptr(need, u8p)))
    else:
        # a text with U+0000 in it: code-point lengths tell the sentences apart (len(str) is free, byte lengths are not)
        cp_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.fromiter(map(len, texts), dtype=np.uint64, count=n), out=cp_off[1:])
        data = "".join(texts).encode("utf-8", "surrogatepass")
        buf = np.frombuffer(data, dtype=np.uint8).copy()
        check(lib().swt_utf8_prepare(ptr(buf, u8p), int(buf.size), ptr(cp_off, u64p), n, ptr(off, u64p), ptr(need, u8p)))
    if need.any():
        data = buf.tobytes()
        parts = [texts[i].lower().encode("utf-8", "surrogatepass") if need[i] else data[int(off[i]):int(off[i + 1])] for i in range(n)]
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.fromiter(map(len, parts), dtype=np.uint64, count=n), out=off[1:])
        buf = np.frombuffer(b"".join(parts), dtype=np.uint8)
    return buf, off


def pack_utf8(lowered):
    """list[str] (already lowercased) -> (uint8 bytes, uint64 offsets[n+1])."""
    enc = [s.encode("utf-8", "surrogatepass") for s in lowered]
    off = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        np.cumsum(np.fromiter(map(len, enc), dtype=np.uint64, count=len(enc)), out=off[1:])
    buf = np.frombuffer(b"".join(enc), dtype=np.uint8)
    if buf.size == 0:
        buf = np.zeros(1, dtype=np.uint8)[:0]
    return buf, off


def pack_utf32(strings):
    """list[str] -> (uint32 code points, uint64 offsets[n+1])."""
    off = np.zeros(len(strings) + 1, dtype=np.uint64)
    if strings:
        np.cumsum(np.fromiter(map(len, strings), dtype=np.uint64, count=len(strings)), out=off[1:])
    blob = np.frombuffer("".join(strings).encode("utf-32-le", "surrogatepass"), dtype=np.uint32)
    return blob, off


def token_histogram(ids, id_cap):
    """uint32 token ids -> uint64 counts[2 * id_cap] (second half: ids carrying BPE_CONT), counted on the device"""
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    counts = np.zeros(2 * int(id_cap), dtype=np.uint64)
    oor = C.c_uint64()
    check(lib().swt_token_histogram(ptr(ids, u32p) if ids.size else None, int(ids.size), int(id_cap), ptr(counts, u64p), C.byref(oor)))
    if oor.value:
        raise ValueError("%d token ids are outside [0, %d)" % (oor.value, id_cap))
    return counts


class BpeTable:
    """Device merge-rank table (swt_bpe_table)."""

    def __init__(self, left, right, merged):
        left = np.ascontiguousarray(left, dtype=np.uint32)
        right = np.ascontiguousarray(right, dtype=np.uint32)
        merged = np.ascontiguousarray(merged, dtype=np.uint32)
        self.n_merges = int(left.size)
        h = C.c_void_p()
        check(lib().swt_bpe_table_create(ptr(left, u32p), ptr(right, u32p), ptr(merged, u32p), self.n_merges, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.swt_bpe_table_destroy(self._h)
        self._h = None

    __del__ = close

    def set_option(self, option, value):
        check(lib().swt_bpe_table_set_option(self._h, option, value))

    def encode(self, text_u8, sent_off, flags=0):
        """host buffers in -> (ids uint32, offsets uint64[n+1])"""
        n_sent = int(sent_off.size - 1)
        n_bytes = int(sent_off[-1])
        out = np.empty(max(n_bytes, 1), dtype=np.uint32)
        out_off = np.zeros(n_sent + 1, dtype=np.uint64)
        nt = C.c_uint64()
        check(lib().swt_bpe_encode(self._h, ptr(text_u8, u8p), ptr(sent_off, u64p), n_sent, ptr(out, u32p), out.size,
                                   ptr(out_off, u64p), C.byref(nt), flags))
        return out[:nt.value], out_off

    def encode_joined(self, joined, n_sent, flags=0):
        """join_texts' bytes in (not lowercased) -> (ids, offsets), the prepared text staying on the device; None when a sentence
        needs the host's str.lower() (the caller goes pack_and_lower -> encode)"""
        out = np.empty(max(int(joined.size), 1), dtype=np.uint32)
        out_off = np.zeros(n_sent + 1, dtype=np.uint64)
        need = np.zeros(max(n_sent, 1), dtype=np.uint8)
        nt = C.c_uint64()
        check(lib().swt_bpe_encode_joined(self._h, ptr(joined, u8p), int(joined.size), n_sent, ptr(out, u32p), out.size,
                                          ptr(out_off, u64p), C.byref(nt), ptr(need, u8p), flags))
        if nt.value == 0xFFFFFFFFFFFFFFFF:
            return None
        return out[:nt.value], out_off

    def encode_dev(self, d_text, n_bytes, d_off, n_sent, d_out, d_out_off, d_ntok, flags=0, stream=0):
        """device pointers (ints) in; enqueues on `stream` and returns"""
        check(lib().swt_bpe_encode_dev(self._h, d_text, n_bytes, d_off, n_sent, d_out, d_out_off, d_ntok, flags, stream))


class WpTrie:
    """Flattened failure-link trie (swt_wp_trie)."""

    def __init__(self, vocab_list):
        self.n_vocab = len(vocab_list)
        blob, off = pack_utf32(list(vocab_list))
        h = C.c_void_p()
        check(lib().swt_wp_trie_create(ptr(blob, u32p) if blob.size else None, ptr(off, u64p), self.n_vocab, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.swt_wp_trie_destroy(self._h)
        self._h = None

    __del__ = close

    def set_option(self, option, value):
        check(lib().swt_wp_trie_set_option(self._h, option, value))

    def stats(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().swt_wp_trie_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"nodes": a.value, "edges": b.value, "pops": c.value}

    def corner(self):
        """ids of NaiveWP.encode_word("##"), or None when the reference never returns from it"""
        buf = np.zeros(64, dtype=np.uint32)
        n = lib().swt_wp_trie_corner(self._h, ptr(buf, u32p), buf.size)
        if n > buf.size:
            buf = np.zeros(n, dtype=np.uint32)
            n = lib().swt_wp_trie_corner(self._h, ptr(buf, u32p), buf.size)
        return None if n < 0 else buf[:n].copy()

    def node(self, path):
        """(node_id, link, is_end, pops) of the node spelled by `path`, or None"""
        a, _ = pack_utf32([path])
        nid, link, end, npops = C.c_uint32(), C.c_int32(), C.c_uint8(), C.c_uint32()
        pops = np.zeros(256, dtype=np.uint32)
        rc = lib().swt_wp_trie_node(self._h, ptr(a, u32p) if a.size else None, a.size, C.byref(nid), C.byref(link),
                                    C.byref(end), ptr(pops, u32p), pops.size, C.byref(npops))
        if rc == ERR_INVALID:
            return None
        check(rc)
        return nid.value, link.value, bool(end.value), pops[:npops.value].copy()

    def node_path(self, node_id):
        buf = np.zeros(512, dtype=np.uint32)
        n = C.c_uint64()
        check(lib().swt_wp_trie_node_path(self._h, node_id, ptr(buf, u32p), buf.size, C.byref(n)))
        return bytes(buf[:n.value]).decode("utf-32-le", "surrogatepass")

    def encode(self, text_u8, sent_off):
        n_sent = int(sent_off.size - 1)
        n_bytes = int(sent_off[-1])
        out = np.empty(max(n_bytes, 1), dtype=np.uint32)
        out_off = np.zeros(n_sent + 1, dtype=np.uint64)
        status = np.zeros(max(n_sent, 1), dtype=np.uint8)
        nt = C.c_uint64()
        check(lib().swt_wp_encode(self._h, ptr(text_u8, u8p), ptr(sent_off, u64p), n_sent, ptr(out, u32p), out.size,
                                  ptr(out_off, u64p), ptr(status, u8p), C.byref(nt)))
        return out[:nt.value], out_off, status[:n_sent]

    def encode_joined(self, joined, n_sent):
        """join_texts' bytes in (not lowercased) -> (ids, offsets, status), or None when a sentence needs the host's str.lower()"""
        out = np.empty(max(int(joined.size), 1), dtype=np.uint32)
        out_off = np.zeros(n_sent + 1, dtype=np.uint64)
        status = np.zeros(max(n_sent, 1), dtype=np.uint8)
        need = np.zeros(max(n_sent, 1), dtype=np.uint8)
        nt = C.c_uint64()
        check(lib().swt_wp_encode_joined(self._h, ptr(joined, u8p), int(joined.size), n_sent, ptr(out, u32p), out.size,
                                         ptr(out_off, u64p), ptr(status, u8p), C.byref(nt), ptr(need, u8p)))
        if nt.value == 0xFFFFFFFFFFFFFFFF:
            return None
        return out[:nt.value], out_off, status[:n_sent]

    def encode_dev(self, d_text, n_bytes, d_off, n_sent, d_out, d_out_off, d_status, d_ntok, stream=0):
        check(lib().swt_wp_encode_dev(self._h, d_text, n_bytes, d_off, n_sent, d_out, d_out_off, d_status, d_ntok, stream))


class BpeTrainer:
    """Device BPE trainer (swt_bpe_trainer): histogram + argmax + merge-apply."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_text(cls, text_u8, sent_off):
        h = C.c_void_p()
        n_sent = int(sent_off.size - 1)
        check(lib().swt_bpe_train_create_text(ptr(text_u8, u8p), ptr(sent_off, u64p), n_sent, C.byref(h)))
        return cls(h)

    @classmethod
    def from_texts(cls, texts, joined=None, wordpiece=False):
        """list[str] -> trainer, the prepared text never leaving the device (swt_bpe_train_create_joined); None when the texts do
        not lend themselves to it (few of them, a U+0000 inside, a code point only the host lowercases): the caller then goes
        pack_and_lower -> from_text.  joined: join_texts(texts), if the caller has it already."""
        n = len(texts)
        if n <= 64 or not device_lower_ok():
            return None
        joined, n_nul = joined if joined is not None else join_texts(texts, "Corpus must be a list of strings.")
        if joined.size + 1 == n or n_nul:
            return None
        need = np.zeros(n, dtype=np.uint8)
        h = C.c_void_p()
        create = lib().swt_wp_train_create_joined if wordpiece else lib().swt_bpe_train_create_joined
        check(create(ptr(joined, u8p), int(joined.size), n, ptr(need, u8p), C.byref(h)))
        return cls(h) if h.value else None

    @classmethod
    def from_text_wordpiece(cls, text_u8, sent_off):
        """NaiveWP.train's state (wordpiece.py:44-63): '##' symbols, likelihood score; `count` outputs = score bit patterns"""
        h = C.c_void_p()
        n_sent = int(sent_off.size - 1)
        check(lib().swt_wp_train_create_text(ptr(text_u8, u8p), ptr(sent_off, u64p), n_sent, C.byref(h)))
        return cls(h)

    @classmethod
    def from_words(cls, syms, word_off, freq):
        syms = np.ascontiguousarray(syms, dtype=np.uint32)
        word_off = np.ascontiguousarray(word_off, dtype=np.uint64)
        freq = np.ascontiguousarray(freq, dtype=np.uint32)
        h = C.c_void_p()
        check(lib().swt_bpe_train_create_words(ptr(syms, u32p), ptr(word_off, u64p), ptr(freq, u32p), int(word_off.size - 1),
                                               C.byref(h)))
        return cls(h)

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.swt_bpe_train_destroy(self._h)
        self._h = None

    __del__ = close

    def set_pos_base(self, base):
        check(lib().swt_bpe_train_set_pos_base(self._h, base))

    def info(self):
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint64()
        check(lib().swt_bpe_train_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"n_words": a.value, "n_symbols": b.value, "n_base_symbols": c.value, "n_pairs": d.value}

    def base_symbols(self):
        n = self.info()["n_base_symbols"]
        out = np.zeros(max(n, 1), dtype=np.uint32)
        check(lib().swt_bpe_train_base_symbols(self._h, ptr(out, u32p), out.size))
        return out[:n]

    def best(self):
        """-> (left, right, count, n_tied, first_pos)"""
        l, r = C.c_uint32(), C.c_uint32()
        cnt, tied, pos = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib().swt_bpe_train_best(self._h, C.byref(l), C.byref(r), C.byref(cnt), C.byref(tied), C.byref(pos)))
        return l.value, r.value, cnt.value, tied.value, pos.value

    def apply(self, left, right, merged):
        check(lib().swt_bpe_train_apply(self._h, left, right, merged))

    def run(self, max_steps, first_merged):
        """device-driven steps -> (left uint32[n], right uint32[n], count uint64[n]); n < max_steps: no pair was left"""
        left = np.zeros(max(max_steps, 1), dtype=np.uint32)
        right = np.zeros(max(max_steps, 1), dtype=np.uint32)
        count = np.zeros(max(max_steps, 1), dtype=np.uint64)
        n = C.c_uint32()
        check(lib().swt_bpe_train_run(self._h, max_steps, first_merged, ptr(left, u32p), ptr(right, u32p), ptr(count, u64p),
                                      C.byref(n)))
        return left[:n.value], right[:n.value], count[:n.value]

    def export(self):
        inf = self.info()
        syms = np.zeros(max(inf["n_symbols"], 1), dtype=np.uint32)
        woff = np.zeros(inf["n_words"] + 1, dtype=np.uint64)
        freq = np.zeros(max(inf["n_words"], 1), dtype=np.uint32)
        check(lib().swt_bpe_train_export(self._h, ptr(syms, u32p), syms.size, ptr(woff, u64p), ptr(freq, u32p)))
        return syms[:int(woff[-1])], woff, freq[:inf["n_words"]]

    def histogram(self):
        cap = max(self.info()["n_pairs"], 1) + 16
        keys = np.zeros(cap, dtype=np.uint64)
        cnts = np.zeros(cap, dtype=np.uint64)
        n = C.c_uint64()
        check(lib().swt_bpe_train_histogram(self._h, ptr(keys, u64p), ptr(cnts, u64p), cap, C.byref(n)))
        return keys[:n.value], cnts[:n.value]

    def step_trace(self):
        """uint64[n_merges, 4]: winning count, pairs tied at it, candidate list length, live symbols before the merge"""
        n = C.c_uint64()
        check(lib().swt_bpe_train_trace(self._h, None, 0, C.byref(n)))
        rows = np.zeros((max(n.value, 1), 4), dtype=np.uint64)
        check(lib().swt_bpe_train_trace(self._h, ptr(rows, u64p), n.value, C.byref(n)))
        return rows[:n.value]

    def stats(self):
        out = np.zeros(10, dtype=np.uint64)
        check(lib().swt_bpe_train_stats(self._h, ptr(out, u64p), 10))
        names = ("replans", "theta", "candidates", "index_entries", "table_slots", "flags", "steps", "keys", "entries_scanned", "tie_words")
        return dict(zip(names, map(int, out)))


class Dist:
    """Communicator of corpus-sharded training (swt_dist): RCCL between processes (one per GPU), or a loop-back whose ranks
    are all trainers of this process (one-GPU tests)."""

    def __init__(self, handle, rank, world, local):
        self._h, self.rank, self.world, self.local = handle, rank, world, local

    @staticmethod
    def unique_id():
        """128 bytes made by rank 0 (ncclGetUniqueId); hand them to every rank"""
        buf = np.zeros(128, dtype=np.uint8)
        check(lib().swt_dist_unique_id(ptr(buf, u8p)))
        return buf

    @classmethod
    def rccl(cls, rank, world, unique_id):
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        h = C.c_void_p()
        check(lib().swt_dist_init(rank, world, ptr(uid, u8p), C.byref(h)))
        return cls(h, rank, world, False)

    @classmethod
    def loopback(cls, world):
        h = C.c_void_p()
        check(lib().swt_dist_init_local(world, C.byref(h)))
        return cls(h, 0, world, True)

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.swt_dist_destroy(self._h)
        self._h = None

    __del__ = close

    def _handles(self, trainers):
        arr = (C.c_void_p * len(trainers))(*[t._h for t in trainers])
        return C.cast(arr, vpp), arr

    def shard_begin(self, trainers):
        """-> the distinct initial symbols of the whole corpus (uint32, ascending)"""
        hp, keep = self._handles(trainers)
        out = np.zeros(0x110000 + 4096, dtype=np.uint32)
        n = C.c_uint32()
        check(lib().swt_bpe_train_shard_begin(hp, len(trainers), self._h, ptr(out, u32p), out.size, C.byref(n)))
        return out[:n.value].copy()

    def run(self, trainers, max_steps, first_merged):
        """-> (left, right, count) of the merges done (fewer than max_steps: no pair was left)"""
        hp, keep = self._handles(trainers)
        left = np.zeros(max(max_steps, 1), dtype=np.uint32)
        right = np.zeros(max(max_steps, 1), dtype=np.uint32)
        count = np.zeros(max(max_steps, 1), dtype=np.uint64)
        n = C.c_uint32()
        check(lib().swt_bpe_train_run_sharded(hp, len(trainers), self._h, max_steps, first_merged, ptr(left, u32p), ptr(right, u32p),
                                              ptr(count, u64p), C.byref(n)))
        return left[:n.value], right[:n.value], count[:n.value]
