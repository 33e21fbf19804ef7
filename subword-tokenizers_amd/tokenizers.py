"""The reference's tokenizer classes with the inner loops on the MI355X.

Same names, constructor, methods, attributes and error behaviour as phtryll/subword-tokenizers
(`source/bpe.py`, `source/wordpiece.py`, `source/utils.py`; citations below are relative to /root/reference):

    SubwordTokenizer   utils.py:5-41      preprocessing(), vocab_length()
    NaiveBPE           bpe.py:9-189       train() on the device; encode_word()/tokenize() stay the slow didactic loop
    FastBPE            bpe.py:192-263     train() + tokenize()/encode_word() on the device
    NaiveWP            wordpiece.py:8-208 CPU Python (out of the GPU scope: SURVEY.md section 2, row 10)
    FastWP             wordpiece.py:211-330  trie build in C++, tokenize() on the device

plus batch entry points the reference lacks (`tokenize_batch`, `encode_ids_batch`), because one sentence per
call cannot feed a GPU.  Python keeps exactly what the reference does in Python: `str.lower()`, JSON I/O, the
id <-> string maps and the stop test of the training loop.  Everything else goes through libswt_hip.so; there
is no CPU fallback -- without a GPU the compute calls raise `NoDeviceError`.
"""
import json
import os
from collections import Counter
from typing import Dict, List, Optional, Tuple

import ctypes as C

import numpy as np

from . import _native as N

__all__ = ["SubwordTokenizer", "NaiveBPE", "FastBPE", "NaiveWP", "FastWP", "TrieView"]


def _is_bert_pretokenizer(tokenizer) -> bool:
    try:
        pt = tokenizer.backend_tokenizer.pre_tokenizer
    except AttributeError:
        return False
    return type(pt).__name__ == "BertPreTokenizer"


class SubwordTokenizer:
    """Parent class (utils.py:5-41).

    `tokenizer` is the HF tokenizer object the reference takes (cli.py:163,194); it is only ever used for its
    BertPreTokenizer (utils.py:27).  It may be None: the split is restated from the probed class table that is
    compiled into libswt_hip.so (White_Space removed, each punctuation code point isolated).
    """

    def __init__(self, tokenizer=None) -> None:
        if tokenizer is not None and not _is_bert_pretokenizer(tokenizer):
            raise ValueError("only a BertPreTokenizer-backed tokenizer (the reference's default, cli.py:85) is supported")
        self.tokenizer = tokenizer

    # -- utils.py:15-29
    def preprocessing(self, corpus: List[str]) -> List[List[Tuple[str, Tuple[int, int]]]]:
        return [self._split(example.lower()) for example in corpus]

    @staticmethod
    def _split(text: str) -> List[Tuple[str, Tuple[int, int]]]:
        out = []
        i, n = 0, len(text)
        cls = _class_of
        while i < n:
            c = cls(text[i])
            if c & N.CLS_BERT_WS:
                i += 1
                continue
            j = i + 1
            if not c & N.CLS_BERT_PUNCT:
                while j < n and not cls(text[j]) & (N.CLS_BERT_WS | N.CLS_BERT_PUNCT):
                    j += 1
            out.append((text[i:j], (i, j)))
            i = j
        return out

    # -- utils.py:31-41
    def vocab_length(self, corpus: List[str]) -> int:
        return len({symbol for example in corpus for symbol in example})


_CLS_CACHE: Dict[str, int] = {}


def _class_of(ch: str) -> int:
    c = _CLS_CACHE.get(ch)
    if c is None:
        c = _CLS_CACHE[ch] = N.class_of(ord(ch))
    return c


# ------------------------------------------------------------------------------------------------------
# BPE

class _SymbolTable:
    """Symbol string <-> id, by STRING identity (bpe.py:41,103,227): one code point -> its ordinal, anything
    else -> 0x110000 + k in order of first appearance."""

    def __init__(self):
        self.strings: List[str] = []
        self.index: Dict[str, int] = {}

    def intern(self, s: str) -> int:
        if len(s) == 1:
            return ord(s)
        k = self.index.get(s)
        if k is None:
            k = self.index[s] = len(self.strings)
            self.strings.append(s)
        return N.SYM_BASE + k

    def string(self, sid: int) -> str:
        sid &= 0x7FFFFFFF
        return chr(sid) if sid < N.SYM_BASE else self.strings[sid - N.SYM_BASE]


_PLAIN_INTERN = _SymbolTable.intern  # (a test swaps intern() for another to force the collision replay: then the Python loop runs)


class NaiveBPE(SubwordTokenizer):
    """Byte-Pair Encoding (bpe.py:9-189).  `train` runs on the device; `encode_word`/`tokenize` keep the
    reference's didactic O(merges x length) loop in Python (out of the GPU scope)."""

    def __init__(self, tokenizer=None) -> None:
        super().__init__(tokenizer)
        self.merges_list: List[Tuple[str, str]] = []
        self.vocab: set = set()
        self._trainer: Optional[N.BpeTrainer] = None
        self._train_syms: Optional[_SymbolTable] = None
        self._train_ids = None  # (left, right, merged) ids of merges_list as the device named them, when no replay was needed
        self._corpus_cache = None

    # -- bpe.py:25-48
    def _replace_pair(self, pair: Tuple[str, str], word: List[str]) -> List[str]:
        left, right = pair
        joined = left + right
        out: List[str] = []
        k, n = 0, len(word)
        while k < n:
            if k + 1 < n and word[k] == left and word[k + 1] == right:
                out.append(joined)
                k += 2
            else:
                out.append(word[k])
                k += 1
        return out

    # -- bpe.py:50-112: the merge loop, on the device
    def train(self, corpus: List[str], max_vocab: int = 30_000) -> None:
        if not isinstance(corpus, list):
            raise TypeError("Corpus must be a list of strings.")
        joined = None
        if len(corpus) > 64:
            joined = N.join_texts(corpus, "Corpus must be a list of strings.")  # checks the items on its way through them
        elif not all(isinstance(example, str) for example in corpus):
            raise TypeError("Corpus must be a list of strings.")
        if not isinstance(max_vocab, int):
            raise TypeError("Maximum vocabulary size must be an integer.")
        self.reset()
        text = off = None
        trainer = N.BpeTrainer.from_texts(corpus, joined)  # bpe.py:70-81 (lower, split, Counter, symbolise), the text staying on the device
        if trainer is None:
            text, off = N.pack_and_lower(corpus)
            trainer = N.BpeTrainer.from_text(text, off)
        syms = _SymbolTable()
        self.vocab.update(chr(int(c)) for c in trainer.base_symbols())  # bpe.py:75
        applied_l: List[int] = []  # (left, right, merged) ids of every merge so far, for the collision replay
        applied_r: List[int] = []
        applied_m: List[int] = []
        done = set()  # left << 32 | right of every merge so far
        exhausted = False
        runs, clean = [], True  # the id arrays of every device run, as long as no string collision forced a replay
        while len(self.vocab) < max_vocab and not exhausted:  # bpe.py:88
            # The device runs `want` iterations of bpe.py:90-111 back to back; it names the merged symbol of step i
            # SYM_BASE + (strings so far) + i, which is right as long as every merged string is new (bpe.py:103).
            want = max_vocab - len(self.vocab)
            first = N.SYM_BASE + len(syms.strings)
            lefts, rights, counts = trainer.run(want, first)
            if len(lefts) < want:
                exhausted = True  # bpe.py:98-99: no pair left
            runs.append((lefts, rights, first))
            # (plain Python ints and local names: this loop runs once per merge, beside a device that needs ~10 us for one;
            # what can be done for the whole run at once -- the replay record, the "never twice" check -- is done after it)
            strings, intern, base = syms.strings, syms.intern, N.SYM_BASE
            vocab_add, merges_append = self.vocab.add, self.merges_list.append
            ll, rl = lefts.tolist(), rights.tolist()
            taken, collided = len(ll), None
            host = N.pyhost() if _SymbolTable.intern is _PLAIN_INTERN and "intern" not in vars(syms) else None
            if host is not None:  # the same loop in C (csrc/swt_pyhost.c), unless someone has put another intern() in place
                la, ra = np.ascontiguousarray(lefts, dtype=np.uint32), np.ascontiguousarray(rights, dtype=np.uint32)
                hit = C.c_uint32()
                at = host.swt_py_bpe_merge_strings(la.ctypes.data, ra.ctypes.data, len(ll), base, first, strings, syms.index, self.vocab,
                                                   self.merges_list, C.byref(hit))
                if at >= 0:
                    taken, collided = at + 1, int(hit.value)
            for i, (left, right) in enumerate(zip(ll, rl) if host is None else ()):
                ls = chr(left) if left < base else strings[left - base]   # _SymbolTable.string
                rs = chr(right) if right < base else strings[right - base]
                joined = ls + rs
                merged = intern(joined)
                vocab_add(joined)  # bpe.py:103
                merges_append((ls, rs))  # bpe.py:104
                if merged != first + i:
                    taken, collided = i + 1, merged
                    break
            keys = ((lefts[:taken].astype(np.uint64) << np.uint64(32)) | rights[:taken].astype(np.uint64)).tolist()
            fresh = set(keys)
            if len(fresh) != len(keys) or not done.isdisjoint(fresh):  # symbols only ever merge: a merged pair cannot come back
                raise RuntimeError("pair histogram inconsistent: a pair was selected twice")
            done |= fresh
            applied_l.extend(ll[:taken])
            applied_r.extend(rl[:taken])
            applied_m.extend(range(first, first + taken))
            if collided is not None:
                # two different merges spelled the same string (never observed).  The device
                # continued with a fresh id; rebuild the state with the right one and carry on from here.
                applied_m[-1] = collided
                clean = False
                trainer.close()
                if text is None:
                    text, off = N.pack_and_lower(corpus)
                trainer = N.BpeTrainer.from_text(text, off)
                for l_, r_, m_ in zip(applied_l, applied_r, applied_m):
                    trainer.apply(l_, r_, m_)
                exhausted = False
        self._trainer, self._train_syms, self._corpus_cache = trainer, syms, None
        if clean:
            self._train_ids = (np.concatenate([l for l, _, _ in runs]) if runs else np.zeros(0, np.uint32),
                               np.concatenate([r for _, r, _ in runs]) if runs else np.zeros(0, np.uint32),
                               np.concatenate([f + np.arange(len(l), dtype=np.uint32) for l, _, f in runs]) if runs else np.zeros(0, np.uint32))

    @property
    def corpus_as_symbols(self) -> List[Tuple[List[str], int]]:
        """bpe.py:23,108-111: the unique words as symbol lists with their frequency (read back on demand)."""
        if self._trainer is None:
            return []
        if self._corpus_cache is None:
            ids, woff, freq = self._trainer.export()
            st = self._train_syms
            self._corpus_cache = ([st.string(int(x)) for x in ids], woff, freq)
        strs, woff, freq = self._corpus_cache
        return [(strs[int(woff[w]):int(woff[w + 1])], int(freq[w])) for w in range(len(woff) - 1)]

    # -- bpe.py:114-134
    def encode_word(self, word: str) -> List[str]:
        pieces = list(word)
        for pair in self.merges_list:
            pieces = self._replace_pair(pair, pieces)
        if len(pieces) > 1:
            pieces[1:] = ["##" + p for p in pieces[1:]]
        return pieces

    # -- bpe.py:136-158
    def tokenize(self, text: str) -> List[str]:
        if not isinstance(text, str):
            raise TypeError("Text to tokenize must be a string.")
        words = [w for w, _ in self.preprocessing([text])[0]]
        out: List[str] = []
        for w in words:
            out += self.encode_word(w)
        return out

    # -- bpe.py:160-164
    def reset(self) -> None:
        self.merges_list.clear()
        self.vocab.clear()
        if self._trainer is not None:
            self._trainer.close()
        self._trainer, self._train_syms, self._corpus_cache = None, None, None
        self._train_ids = None

    # -- bpe.py:167-189
    def save_resources(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "merges.json"), "w", encoding="utf-8") as f:
            json.dump(self.merges_list, f, ensure_ascii=False)

    def load_resources(self, path: str) -> None:
        merges_file = os.path.join(path, "merges.json")
        if os.path.isfile(merges_file):  # a missing file is silently ignored (bpe.py:187)
            with open(merges_file, "r", encoding="utf-8") as f:
                self.merges_list = [tuple(pair) for pair in json.load(f)]


class FastBPE(NaiveBPE):
    """bpe.py:192-263 with the rank table and the encode loop on the device."""

    def __init__(self, tokenizer=None):
        super().__init__(tokenizer)
        self._bpe_ranks: Dict[Tuple[str, str], int] = {}
        self._table: Optional[N.BpeTable] = None
        self._syms = _SymbolTable()

    def _build_table(self) -> None:
        # bpe.py:200 / :257
        self._bpe_ranks = {pair: i for i, pair in enumerate(self.merges_list)}
        syms = _SymbolTable()
        intern = syms.intern
        ids = [x for l, r in self.merges_list for x in (intern(l), intern(r), intern(l + r))]
        ids = np.array(ids, dtype=np.uint32).reshape(-1, 3)
        self._set_table(syms, ids[:, 0], ids[:, 1], ids[:, 2])

    def _set_table(self, syms, left, right, merged) -> None:
        if self._table is not None:
            self._table.close()
        self._syms = syms
        self._table = N.BpeTable(left, right, merged)

    def _ensure_table(self) -> N.BpeTable:
        # like _bpe_ranks, the table only changes in train()/load_resources() (bpe.py:200,257)
        if self._table is None:
            self._table = N.BpeTable([], [], [])
        return self._table

    def train(self, corpus: List[str], max_vocab: int = 30_000) -> None:
        super().train(corpus, max_vocab)
        if self._train_ids is None or len(self._train_ids[0]) != len(self.merges_list):
            self._build_table()
            return
        # the device named the symbols exactly as _build_table would (a merged string gets the next id the first time it is
        # spelled, and in training every merge spells a new one): its ids go to the rank table as they are
        self._bpe_ranks = {pair: i for i, pair in enumerate(self.merges_list)}
        syms = _SymbolTable()
        syms.strings = list(self._train_syms.strings)
        syms.index = dict(self._train_syms.index)
        self._set_table(syms, *self._train_ids)

    def _pairs(self, seq: List[str]) -> set:
        return {(seq[i], seq[i + 1]) for i in range(len(seq) - 1)}

    def decode_ids(self, ids) -> List[str]:
        st = self._syms
        one = lambda t: ("##" + st.string(t)) if t & N.BPE_CONT else st.string(t)
        ids = np.asarray(ids, dtype=np.uint32)
        if ids.size < 4096:
            return [one(t) for t in map(int, ids)]
        # large outputs: spell each DISTINCT id once, then hand out references (the per-token Python loop was 1.5 s per 2 M tokens)
        key = (ids << np.uint32(1)) | (ids >> np.uint32(31))  # symbol * 2 + the BPE_CONT bit
        spell = lambda k: ("##" + st.string(k >> 1)) if k & 1 else st.string(k >> 1)
        return N.nested_lists_by_key(key, 2 * (N.SYM_BASE + len(st.strings)), spell, np.array([0, ids.size], dtype=np.uint64))[0]

    # -- batch entry points (not in the reference)
    def encode_ids_batch(self, texts: List[str]) -> Tuple[np.ndarray, np.ndarray]:
        """texts -> (token ids uint32, sentence offsets uint64[n+1]); ids as defined in include/swt.h."""
        if not isinstance(texts, list):
            raise TypeError("Text must be a string.")
        table = self._ensure_table()
        if len(texts) > 64:
            # strings -> joined bytes -> ids: the device splits, lowercases (utils.py:27) and encodes, the
            # prepared text never comes back.  Texts with U+0000 inside, or that only str.lower() lowercases, go the long way.
            joined, n_nul = N.join_texts(texts)  # TypeError for an item that is no str
            if n_nul == 0 and joined.size + 1 != len(texts) and N.device_lower_ok():
                got = table.encode_joined(joined, len(texts))
                if got is not None:
                    return got
        elif not all(isinstance(t, str) for t in texts):
            raise TypeError("Text must be a string.")
        text, off = N.pack_and_lower(texts)
        return table.encode(text, off)

    def tokenize_batch(self, texts: List[str]) -> List[List[str]]:
        ids, off = self.encode_ids_batch(texts)
        if ids.size < 4096:
            toks = self.decode_ids(ids)
            return [toks[int(off[i]):int(off[i + 1])] for i in range(len(texts))]
        # spell each DISTINCT id once (found by a presence map over the id space, not a sort), then hand out references
        st = self._syms
        key = (ids << np.uint32(1)) | (ids >> np.uint32(31))  # symbol * 2 + the BPE_CONT bit
        spell = lambda k: ("##" + st.string(k >> 1)) if k & 1 else st.string(k >> 1)
        return N.nested_lists_by_key(key, 2 * (N.SYM_BASE + len(st.strings)), spell, off)

    # -- bpe.py:205-243, one word = one device "sentence" with the pre-tokenizer split switched off
    def encode_word(self, word: str) -> List[str]:
        if word == "":
            return [""]  # bpe.py:207-208
        table = self._ensure_table()
        text, off = N.pack_utf8([word])
        ids, _ = table.encode(text, off, flags=N.BPE_RAW_WORDS)
        return self.decode_ids(ids)

    # -- bpe.py:245-249
    def tokenize(self, text: str) -> List[str]:
        if not isinstance(text, str):
            raise TypeError("Text must be a string.")
        return self.tokenize_batch([text])[0]

    # -- bpe.py:251-263
    def load_resources(self, path: str) -> None:
        super().load_resources(path)
        self._build_table()

    def save_resources(self, path: str) -> None:
        super().save_resources(path)


# ------------------------------------------------------------------------------------------------------
# WordPiece

class _WpSymbols:
    """WordPiece trainer symbol id <-> string: c -> ord(c); '##c' -> WP_CONT + ord(c); a merged string -> WP_MERGED_BASE + k
    in order of first appearance (symbols are identified by their STRING, wordpiece.py:95-96,121)."""

    def __init__(self):
        self.strings: List[str] = []
        self.index: Dict[str, int] = {}

    def intern_merged(self, s: str) -> int:
        k = self.index.get(s)
        if k is None:
            k = self.index[s] = len(self.strings)
            self.strings.append(s)
        return N.WP_MERGED_BASE + k

    def string(self, sid: int) -> str:
        if sid < N.WP_CONT:
            return chr(sid)
        if sid < N.WP_MERGED_BASE:
            return "##" + chr(sid - N.WP_CONT)
        return self.strings[sid - N.WP_MERGED_BASE]


class NaiveWP(SubwordTokenizer):
    """WordPiece by the likelihood score (wordpiece.py:8-208).  `train` runs on the device: the
    BPE trainer's stream, histogram and merge-apply with exact symbol frequencies and the score as the argmax key;
    `encode_word`/`tokenize` keep the reference's longest-prefix loop in Python (FastWP is the device encoder)."""

    def __init__(self, tokenizer=None):
        super().__init__(tokenizer)
        self.vocab: set = set()
        self._trainer: Optional[N.BpeTrainer] = None
        self._train_syms: Optional[_WpSymbols] = None
        self._corpus_cache = None

    # -- wordpiece.py:29-103: the merge loop, on the device
    def train(self, corpus, max_vocab: int = 30_000):
        if not isinstance(corpus, list):
            raise TypeError("corpus must be a list of strings.")
        joined = None
        if len(corpus) > 64:
            joined = N.join_texts(corpus, "corpus must be a list of strings.")  # checks the items on its way
        elif not all(isinstance(example, str) for example in corpus):
            raise TypeError("corpus must be a list of strings.")
        if not isinstance(max_vocab, int):
            raise TypeError("max_vocab must be an int.")
        self.reset()
        text = off = None
        # wordpiece.py:44-58 (lower, split, Counter, '##' symbols), the text staying on the device where it can
        trainer = N.BpeTrainer.from_texts(corpus, joined, wordpiece=True)
        if trainer is None:
            text, off = N.pack_and_lower(corpus)
            trainer = N.BpeTrainer.from_text_wordpiece(text, off)
        syms = _WpSymbols()
        self.vocab |= {syms.string(int(c)) for c in trainer.base_symbols()}  # wordpiece.py:62-63
        applied: List[Tuple[int, int, int]] = []
        done = set()
        exhausted = False
        while len(self.vocab) < max_vocab and not exhausted:  # wordpiece.py:69
            # `want` iterations of wordpiece.py:70-102 back to back on the device; step i merges into WP_MERGED_BASE +
            # (strings so far) + i, which is right as long as every merged string is new (wordpiece.py:96)
            want = max_vocab - len(self.vocab)
            first = N.WP_MERGED_BASE + len(syms.strings)
            lefts, rights, _scores = trainer.run(want, first)
            if len(lefts) < want:
                exhausted = True  # wordpiece.py:75-76: no pair left
            for i in range(len(lefts)):
                left, right = int(lefts[i]), int(rights[i])
                if (left, right) in done:
                    raise RuntimeError("pair histogram inconsistent: %r selected twice" % ((left, right),))
                done.add((left, right))
                token = syms.string(left) + syms.string(right)[2:]  # wordpiece.py:95
                merged = syms.intern_merged(token)
                self.vocab.add(token)  # wordpiece.py:96
                applied.append((left, right, merged))
                if merged != first + i:
                    # two different merges spelled the same string: the device went on with a fresh id; rebuild the state
                    # with the shared one and carry on from here
                    trainer.close()
                    if text is None:
                        text, off = N.pack_and_lower(corpus)
                    trainer = N.BpeTrainer.from_text_wordpiece(text, off)
                    for l_, r_, m_ in applied:
                        trainer.apply(l_, r_, m_)
                    exhausted = False
                    break
        self._trainer, self._train_syms, self._corpus_cache = trainer, syms, None
        self._merge_order = [(syms.string(l), syms.string(r)) for l, r, _ in applied]

    @property
    def corpus_as_symbols(self) -> List[Tuple[List[str], int]]:
        """wordpiece.py:27,99-102: the unique words as symbol lists with their frequency (read back on demand)."""
        if self._trainer is None:
            return []
        if self._corpus_cache is None:
            ids, woff, freq = self._trainer.export()
            st = self._train_syms
            self._corpus_cache = ([st.string(int(x)) for x in ids], woff, freq)
        strs, woff, freq = self._corpus_cache
        return [(strs[int(woff[w]):int(woff[w + 1])], int(freq[w])) for w in range(len(woff) - 1)]

    # -- wordpiece.py:105-130
    def _replace_pair(self, pair, word):
        left, right = pair
        joined = left + right[2:]
        out = []
        k, n = 0, len(word)
        while k < n:
            if k + 1 < n and word[k] == left and word[k + 1] == right:
                out.append(joined)
                k += 2
            else:
                out.append(word[k])
                k += 1
        return out

    # -- wordpiece.py:132-159 (longest prefix in vocab, '##' on the remainder)
    def encode_word(self, word):
        pieces = []
        while word:
            cut = len(word)
            while cut > 0 and word[:cut] not in self.vocab:
                cut -= 1
            if cut == 0:
                return ["[UNK]"]
            pieces.append(word[:cut])
            word = word[cut:]
            if word:
                word = "##" + word
        return pieces

    # -- wordpiece.py:161-181
    def tokenize(self, text):
        if not isinstance(text, str):
            raise TypeError("Text to tokenize must be a string.")
        out = []
        for w, _ in self.preprocessing([text])[0]:
            out.extend(self.encode_word(w))
        return out

    # -- wordpiece.py:183-208
    def reset(self) -> None:
        self.vocab.clear()
        if self._trainer is not None:
            self._trainer.close()
        self._trainer, self._train_syms, self._corpus_cache = None, None, None

    def save_resources(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "vocab.json"), "w", encoding="utf-8") as f:
            json.dump(list(self.vocab), f, ensure_ascii=False)

    def load_resources(self, path: str) -> None:
        vocab_file = os.path.join(path, "vocab.json")
        if os.path.isfile(vocab_file):
            with open(vocab_file, "r", encoding="utf-8") as f:
                self.vocab = set(json.load(f))


class TrieNodeView:
    """Read-only view of one node of the flattened trie (utils.py:44-63 attribute names)."""

    def __init__(self, trie: "TrieView", node_id: int, path: Optional[str]):
        self._trie, self.node_id, self.chars_seen = trie, node_id, path

    def _info(self):
        return self._trie._native.node(self.chars_seen)

    @property
    def char(self):
        return self.chars_seen[-1:] if self.chars_seen else ""

    @property
    def is_end(self):
        return False if self.node_id == 1 else self._info()[2]

    @property
    def failure_link(self):
        if self.node_id == 1:
            return None
        link = self._info()[1]
        return None if link < 0 else self._trie.node_by_id(link)

    @property
    def failure_pops(self):
        if self.node_id == 1:
            return []
        return self._trie._decode(self._info()[3])

    def child(self, ch: str):
        if self.node_id == 1:
            return None
        info = self._trie._native.node(self.chars_seen + ch)
        return None if info is None else TrieNodeView(self._trie, info[0], self.chars_seen + ch)

    def __eq__(self, other):
        return isinstance(other, TrieNodeView) and other.node_id == self.node_id and other._trie is self._trie

    def __hash__(self):
        return hash(self.node_id)


class TrieView:
    """`vocab_trie` as the reference exposes it (utils.py:66-85): .root, .root_sharp, .root_p."""

    def __init__(self, native: N.WpTrie, tokens: List[str]):
        self._native, self._tokens = native, tokens
        self.root = TrieNodeView(self, 0, "")
        self.root_p = TrieNodeView(self, 1, "")
        self.root_sharp = TrieNodeView(self, native.node("##")[0], "##")

    def _decode(self, ids):
        return [self._tokens[int(t)] for t in ids]

    def node_by_id(self, node_id: int) -> TrieNodeView:
        if node_id == 0:
            return self.root
        if node_id == 1:
            return self.root_p
        return TrieNodeView(self, node_id, self._native.node_path(node_id))


class FastWP(NaiveWP):
    """wordpiece.py:211-330: LinMaxMatch end-to-end WordPiece; trie built in C++, matched on the device."""

    UNK = "['UNK']"  # wordpiece.py:257 (sic)

    def __init__(self, tokenizer=None):
        super().__init__(tokenizer)
        self._trie: Optional[N.WpTrie] = None
        self._tokens: List[str] = []
        self._corner: Optional[List[str]] = None
        self.vocab_trie: Optional[TrieView] = None

    def _build_trie(self) -> None:
        # utils.py:75-85; ids = position in the sorted vocabulary (vocab.json order is arbitrary, wordpiece.py:196)
        self._tokens = sorted(self.vocab)
        self._decode_table = self._decode_list = None
        if self._trie is not None:
            self._trie.close()
        self._trie = N.WpTrie(self._tokens)
        c = self._trie.corner()
        self._corner = None if c is None else self._decode(c)
        self.vocab_trie = TrieView(self._trie, self._tokens)

    def _decode(self, ids) -> List[str]:
        toks, n = self._tokens, len(self._tokens)
        ids = np.asarray(ids)
        if ids.size >= 4096 and int(ids.max()) <= n + 1:  # large outputs without the multi-token corner: one gather
            if getattr(self, "_decode_table", None) is None or self._decode_table.size != n + 2:
                self._decode_table = np.empty(n + 2, dtype=object)
                self._decode_table[:] = list(toks) + [self.UNK, "[UNK]"]
            return self._decode_table[ids.astype(np.int64)].tolist()
        out = []
        for t in map(int, ids):
            if t < n:
                out.append(toks[t])
            elif t == n:
                out.append(self.UNK)
            elif t == n + 1:
                out.append("[UNK]")  # wordpiece.py:149 via :261
            else:
                out.extend(self._corner)  # multi-token NaiveWP.encode_word("##")
        return out

    # -- wordpiece.py:227-231
    def train(self, corpus, max_vocab=30_000):
        super().train(corpus, max_vocab)
        self._build_trie()

    # -- batch entry points (not in the reference)
    def encode_ids_batch(self, texts: List[str]):
        """texts -> (ids uint32, offsets uint64[n+1], status uint8[n]); see include/swt.h for ids and status."""
        if not isinstance(texts, list):
            raise TypeError("Text to tokenize must be a string.")
        if len(texts) <= 64 and not all(isinstance(t, str) for t in texts):
            raise TypeError("Text to tokenize must be a string.")
        if self._trie is None:
            if not all(isinstance(t, str) for t in texts):
                raise TypeError("Text to tokenize must be a string.")
            raise AttributeError("'FastWP' object has no attribute 'vocab_trie'")  # as the reference before load/train
        if len(texts) > 64:
            joined, n_nul = N.join_texts(texts, "Text to tokenize must be a string.")  # as FastBPE.encode_ids_batch
            if n_nul == 0 and joined.size + 1 != len(texts) and N.device_lower_ok():
                got = self._trie.encode_joined(joined, len(texts))
                if got is not None:
                    return got
        text, off = N.pack_and_lower(texts)  # wordpiece.py:248 lower(), on the device; the " " is implicit
        return self._trie.encode(text, off)

    def tokenize_batch(self, texts: List[str]) -> List[List[str]]:
        ids, off, status = self.encode_ids_batch(texts)
        for i in np.flatnonzero(status):
            self._raise_for_status(int(status[i]), texts[int(i)])
        if ids.size and int(ids.max()) <= len(self._tokens) + 1:  # no multi-token corner: one table, references handed out
            if ids.size < 4096:
                toks = self._decode(ids)
                return [toks[int(off[i]):int(off[i + 1])] for i in range(len(texts))]
            if getattr(self, "_decode_list", None) is None or len(self._decode_list) != len(self._tokens) + 2:
                self._decode_list = list(self._tokens) + [self.UNK, "[UNK]"]
            return N.nested_lists(self._decode_list, ids, off)
        return [self._decode(ids[int(off[i]):int(off[i + 1])]) for i in range(len(texts))]

    @staticmethod
    def _raise_for_status(st: int, text: str) -> None:
        if st == N.WP_NONTERMINATING:
            # documented deviation: the reference spins forever here; we refuse instead
            raise RuntimeError("FastWP.tokenize: the reference does not terminate on this input: %r" % text[:80])
        if st == N.WP_INDEXERROR:
            raise IndexError("string index out of range")  # wordpiece.py:285 with i == len(seq)

    # -- wordpiece.py:233-270
    def tokenize(self, text):
        if not isinstance(text, str):
            raise TypeError("Text to tokenize must be a string.")
        return self.tokenize_batch([text])[0]

    # -- wordpiece.py:318-330
    def load_resources(self, path: str) -> None:
        super().load_resources(path)
        self._build_trie()

    def save_resources(self, path: str) -> None:
        super().save_resources(path)
