/*
 * include/swt.h -- C ABI of libswt_hip.so: the MI355X (gfx950) subword-tokenizer hot path.
 *
 * The reference (phtryll/subword-tokenizers) is pure Python and has NO FFI/plugin boundary of its own; the
 * drop-in boundary is its Python class surface.  This header is the C ABI a
 * maintainer binds underneath those classes (ctypes stub in INTEGRATION.md).  Each entry point names
 * the reference lines whose inner loop it replaces; citations are relative to /root/reference.
 *
 * Conventions
 *   - every function returns an int status: 0 = SWT_OK, negative = error (swt_last_error() has text);
 *     nothing throws, aborts or falls back to a CPU path: if the HIP runtime or the device is missing
 *     the call fails with SWT_ERR_NO_DEVICE.
 *   - handles are opaque; buffers are caller-owned; sizes are in elements of the pointed-to type.
 *   - text is UTF-8 of the ALREADY LOWERCASED string (Python str.lower() stays with the caller:
 *     source/utils.py:27, source/wordpiece.py:248), sentences concatenated, with n_sent+1 byte offsets.
 *     Lone surrogates encoded "surrogatepass"-style are accepted and decoded as their code point.
 *   - `_dev` entry points take DEVICE pointers (e.g. torch tensors' data_ptr()) and a hipStream_t passed
 *     as void* (NULL = default stream); they enqueue work and return without synchronising.
 *   - a handle may be used by one host thread at a time.
 *
 * Token ids
 *   BPE:  symbol id = the code point for a one-code-point symbol, else 0x110000 + k, where k is the
 *         caller's interning index of the merged string (symbols are identified by their STRING,
 *         source/bpe.py:41,103,227-228).  Token id = symbol id | SWT_BPE_CONT for every token after the
 *         first of its word (the '##' prefix of source/bpe.py:240-241).
 *   WP:   index into the vocabulary list given to swt_wp_trie_create; n_vocab = "['UNK']"
 *         (source/wordpiece.py:257); n_vocab+1 = "[UNK]" (source/wordpiece.py:149); n_vocab+2 =
 *         SWT_WP corner marker, emitted when NaiveWP.encode_word("##") (source/wordpiece.py:260-261)
 *         yields more than one token -- the caller substitutes the list it got from swt_wp_trie_corner.
 */
#ifndef SWT_H
#define SWT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWT_OK 0
#define SWT_ERR_NO_DEVICE (-1)    /* no HIP device / runtime */
#define SWT_ERR_INVALID (-2)      /* bad argument */
#define SWT_ERR_CAPACITY (-3)     /* caller buffer too small; the needed size is reported */
#define SWT_ERR_HIP (-4)          /* a HIP call failed */
#define SWT_ERR_UNSUPPORTED (-5)
#define SWT_ERR_STATE (-6)        /* call out of order, or an internal capacity check failed (the handle is unusable) */
#define SWT_ERR_NOMEM (-7)        /* host memory exhausted (std::bad_alloc / std::length_error inside the library) */
#define SWT_ERR_INTERNAL (-8)     /* any other C++ exception stopped at the ABI boundary */

#define SWT_SYM_BASE 0x110000u
#define SWT_BPE_CONT 0x80000000u

/* per-sentence status of the WordPiece encoder */
#define SWT_WP_OK 0
#define SWT_WP_NONTERMINATING 1   /* the reference loops forever on this input */
#define SWT_WP_INDEXERROR 2       /* the reference raises IndexError (source/wordpiece.py:285, i == len) */

const char *swt_last_error(void);
int swt_version(void);
/* The contract above, testable without a GPU: raises a C++ exception of the given kind inside the library (1 std::bad_alloc,
 * 2 std::length_error, 3 std::runtime_error, 4 a non-std object) and returns what the ABI's exception barrier made of it
 * (SWT_ERR_NOMEM, SWT_ERR_NOMEM, SWT_ERR_INTERNAL, SWT_ERR_INTERNAL; swt_last_error() names the exception).  0: SWT_OK. */
int swt_abi_selftest(int kind);
/* Selects the HIP device for this process (one process per GPU).  Fails loudly without a GPU. */
int swt_init(int device_ordinal);
int swt_device_count(void);
/* multiprocessor count / name of the selected device (for launch sizing and reports) */
int swt_device_info(int *n_cu, char *name, size_t name_cap);

/* Kernel timing for bench.py's roofline line (state of the CALLING THREAD, like the error text).  on = 1: every launch of a path's DOMINANT kernel (bpe_lane_kernel,
 * wp_encode_kernel, the training apply/argmax pair) is bracketed by two HIP events on the stream it is launched on;
 * on = 2: the bracket spans all kernels of one call instead (first launch .. last launch); 0 = off.
 * swt_profile_read waits for the events, returns the summed elapsed milliseconds and the number of brackets since
 * the last read, and clears the list. */
int swt_profile_enable(int on);
int swt_profile_read(double *ms_total, uint64_t *n_launches);

/* str.lower() on the device for the code points it maps one-to-one at equal UTF-8 length (the
 * reference lowercases on the host: source/utils.py:27 via preprocessing, source/wordpiece.py:248).  The text is rewritten
 * in place; need_host[s] = 1 marks a sentence that holds one of the 26 code points the device leaves alone (a lowercase of
 * another UTF-8 length or of several code points, or U+03A3 whose lowercase depends on its neighbours): the caller
 * lowercases THAT sentence on the host.  swt_lower_of: the table itself (0xFFFFFFFF = host only), for tests. */
uint32_t swt_lower_of(uint32_t cp);
/* The Unicode database the lowercase and class tables were generated from ("13.0.0"): a host whose own str.lower() follows
 * another version must lowercase every batch itself (swt_*_encode on its own text), or small and large batches disagree. */
const char *swt_unidata_version(void);
int swt_utf8_lower(uint8_t *text, const uint64_t *sent_off, uint64_t n_sent, uint8_t *need_host);
/* The same for a host that only knows its sentences' lengths in CODE POINTS (Python: "".join(texts).encode() and len(str)
 * are cheap, every string's byte length is not): text = well-formed UTF-8 of all sentences joined, cp_off[n_sent + 1] =
 * running code-point counts.  Writes byte_off[n_sent + 1] (a code point starts at every byte that is not a continuation
 * byte), lowercases in place and flags as above. */
int swt_utf8_prepare(uint8_t *text, uint64_t n_bytes, const uint64_t *cp_off, uint64_t n_sent, uint64_t *byte_off, uint8_t *need_host);
/* And for a host that knows nothing but the strings (Python: "\0".join(texts).encode(), and one bytes.count to make sure the
 * texts hold no U+0000 of their own): joined = the sentences with ONE zero byte between neighbours (n_joined bytes, exactly
 * n_sent - 1 of them zero).  Writes the text without the separators (n_joined - (n_sent - 1) bytes, lowercased as above) to
 * text_out, byte_off[n_sent + 1] and the flags.  SWT_ERR_INVALID when the separators do not add up. */
int swt_utf8_prepare_joined(const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint8_t *text_out, uint64_t *byte_off,
                            uint8_t *need_host);
int swt_utf8_lower_dev(uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off, uint64_t n_sent, uint8_t *d_need_host,
                       void *stream);

/* Per-handle options of the two encoders (swt_bpe_table_set_option / swt_wp_trie_set_option).  Every choice gives the same
 * output; they pick the path, for tests and measurements.
 *   SWT_OPT_DEDUP             0 (default): batches above a size threshold encode each DISTINCT word once (word-level dedup);
 *                             1: never; 2: always
 *   SWT_OPT_DEDUP_TABLE_BITS  log2 of the dedup word table's slots, 4..24; 0 (default): sized from the batch.  A tiny table
 *                             makes words overflow into their own slots; results stay exact
 *   SWT_OPT_UNIQUE_TILE       (FastBPE) tile size of the pass over the unique words: 0 (default), 64, 128 or 256 */
#define SWT_OPT_DEDUP 1
#define SWT_OPT_DEDUP_TABLE_BITS 2
#define SWT_OPT_UNIQUE_TILE 3

/* Code-point classes compiled into the library (fixture data probed from the wheel/interpreter the
 * reference runs on; tools/gen_unicode_tables.py).  bit0 pre-tokenizer whitespace, bit1 pre-tokenizer
 * punctuation (source/utils.py:27), bit2 str.isspace, bit3 str.isalnum (source/wordpiece.py:285-288). */
#define SWT_CLS_BERT_WS 1u
#define SWT_CLS_BERT_PUNCT 2u
#define SWT_CLS_PY_SPACE 4u
#define SWT_CLS_PY_ALNUM 8u
unsigned swt_class_of(uint32_t code_point);

/* ------------------------------------------------------------------------------------------------
 * FastBPE encode: replaces FastBPE.load_resources' rank dict (source/bpe.py:251-257, :200) and the
 * per-sentence loop FastBPE.tokenize -> encode_word (source/bpe.py:245-249, 205-243) including the
 * pre-tokenizer split of SubwordTokenizer.preprocessing (source/utils.py:26-29).
 */
typedef struct swt_bpe_table swt_bpe_table;

/* merges in list order as symbol-id triples; a later duplicate (left,right) overrides an earlier one
 * (dict semantics of source/bpe.py:257). */
int swt_bpe_table_create(const uint32_t *left, const uint32_t *right, const uint32_t *merged,
                         uint32_t n_merges, swt_bpe_table **out);
void swt_bpe_table_destroy(swt_bpe_table *t);
int swt_bpe_table_set_option(swt_bpe_table *t, int option, int value);

/* Host-buffer form: copies in, encodes on the device, copies out.
 *   text[n_bytes], sent_off[n_sent+1] -> out_ids[<= out_cap], out_off[n_sent+1], *n_tokens
 * out_cap >= n_bytes is always sufficient (every token covers at least one byte). */
#define SWT_BPE_RAW_WORDS 1u   /* flags: every "sentence" is ONE word, no pre-tokenizer split -- this is
                                  FastBPE.encode_word(word) (source/bpe.py:205), batched */
#define SWT_BPE_NO_DEDUP 2u    /* flags: encode every word occurrence (by default batches of 1.75 MiB and more encode each
                                  DISTINCT word once and copy its tokens to every occurrence; same output either way) */
int swt_bpe_encode(swt_bpe_table *t, const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent,
                   uint32_t *out_ids, uint64_t out_cap, uint64_t *out_off, uint64_t *n_tokens, uint32_t flags);

/* Host-buffer form for a caller that holds strings, not offsets (Python's FastBPE.tokenize over a list: bpe.py:245-249 per
 * text): joined = the sentences with ONE zero byte between neighbours, as for swt_utf8_prepare_joined, NOT yet lowercased.
 * The device finds the separators, lowercases (utils.py:27) and encodes; the prepared text never travels back to the host.
 * out_cap >= n_joined is always sufficient.  need_host[n_sent] as from swt_utf8_lower: when it flags a sentence nothing is
 * encoded, *n_tokens = UINT64_MAX (SWT_OK), and the caller lowercases on the host and uses swt_bpe_encode. */
int swt_bpe_encode_joined(swt_bpe_table *t, const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint32_t *out_ids,
                          uint64_t out_cap, uint64_t *out_off, uint64_t *n_tokens, uint8_t *need_host, uint32_t flags);

/* Device-buffer form.  d_out_ids needs room for n_bytes ids (worst case); d_out_off[n_sent+1].
 * d_n_tokens (device, 1 element) receives the total. */
int swt_bpe_encode_dev(swt_bpe_table *t, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off,
                       uint64_t n_sent, uint32_t *d_out_ids, uint64_t *d_out_off, uint64_t *d_n_tokens,
                       uint32_t flags, void *stream);

/* ------------------------------------------------------------------------------------------------
 * FastWP encode: replaces WPTrie_E2E (source/utils.py:66-139: insert + precompute, built on the host
 * and flattened into device arrays) and FastWP.tokenize/matchloop/iswdbndry/ispunc
 * (source/wordpiece.py:233-316).
 */
typedef struct swt_wp_trie swt_wp_trie;

/* vocabulary as UTF-32 code points + n_vocab+1 offsets; token id = index (first wins on duplicates) */
int swt_wp_trie_create(const uint32_t *vocab_cps, const uint64_t *vocab_off, uint32_t n_vocab, swt_wp_trie **out);
void swt_wp_trie_destroy(swt_wp_trie *t);
int swt_wp_trie_set_option(swt_wp_trie *t, int option, int value);
int swt_wp_trie_stats(const swt_wp_trie *t, uint32_t *n_nodes, uint32_t *n_edges, uint32_t *n_pops);
/* NaiveWP.encode_word("##") evaluated once at build: returns its length in ids (copied to out up to
 * cap), or -1 when the reference never returns from it. */
int64_t swt_wp_trie_corner(const swt_wp_trie *t, uint32_t *out, uint64_t cap);
/* Debug/parity view of one node of the flattened trie, addressed by its path string (with the '##'
 * prefix where the token has one): failure link as a node id, pops copied out.  Node ids: 0 root,
 * 1 root_p, 2.. in creation order.  Returns SWT_ERR_INVALID when the path does not exist. */
int swt_wp_trie_node(const swt_wp_trie *t, const uint32_t *path, uint64_t path_len, uint32_t *node_id,
                     int32_t *link, uint8_t *is_end, uint32_t *pops, uint32_t pops_cap, uint32_t *n_pops);
int swt_wp_trie_node_path(const swt_wp_trie *t, uint32_t node_id, uint32_t *out, uint64_t cap, uint64_t *len);

int swt_wp_encode(swt_wp_trie *t, const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent,
                  uint32_t *out_ids, uint64_t out_cap, uint64_t *out_off, uint8_t *status, uint64_t *n_tokens);
/* From the sentences joined with ONE zero byte between neighbours, not yet lowercased (wordpiece.py:248 lower() happens on the
 * device): see swt_bpe_encode_joined, including the meaning of need_host and *n_tokens = UINT64_MAX. */
int swt_wp_encode_joined(swt_wp_trie *t, const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint32_t *out_ids,
                         uint64_t out_cap, uint64_t *out_off, uint8_t *status, uint64_t *n_tokens, uint8_t *need_host);
int swt_wp_encode_dev(swt_wp_trie *t, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off,
                      uint64_t n_sent, uint32_t *d_out_ids, uint64_t *d_out_off, uint8_t *d_status,
                      uint64_t *d_n_tokens, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Token-id histogram on the device: the Counter over token strings of the reference's
 * zipf_distribution (source/benchmarks.py:240-253), over ids -- a '##' token and the same token without the prefix are
 * different strings and different ids.  counts has 2 * id_cap entries: counts[id & 0x7FFFFFFF] for ids without SWT_BPE_CONT,
 * counts[id_cap + (id & 0x7FFFFFFF)] for ids with it; ids whose low 31 bits reach id_cap are tallied in *out_of_range.
 * FastWP ids carry no flag: only the first half is used.  The `_dev` form takes device pointers and zeroes the outputs itself.
 */
int swt_token_histogram(const uint32_t *ids, uint64_t n, uint32_t id_cap, uint64_t *counts, uint64_t *out_of_range);
int swt_token_histogram_dev(const uint32_t *d_ids, uint64_t n, uint32_t id_cap, uint64_t *d_counts, uint64_t *d_out_of_range,
                            void *stream);

/* ------------------------------------------------------------------------------------------------
 * BPE training: replaces the merge loop of NaiveBPE.train (source/bpe.py:88-111; FastBPE.train
 * inherits it, source/bpe.py:198-200) and its word dedup (source/bpe.py:73-81).
 *
 * The caller keeps the string set and the stop test (`len(vocab) < max_vocab`, source/bpe.py:88,103):
 *   swt_bpe_train_create*  ->  loop { swt_bpe_train_best; intern(left+right); swt_bpe_train_apply }.
 * Device state: the unique-word symbol stream (uint32; a merge leaves a hole, addresses are stable), word offsets and
 * frequencies, the pair histogram (hash table of 64-bit pair keys with 64-bit counts) which is built once and then updated
 * incrementally by every merge -- the same counts the reference recomputes from scratch each round -- an inverted index
 * pair -> words so that a merge visits only the words that hold it, and the list of high-count candidates the argmax scans.
 * All of a trainer's work is enqueued on one stream (the legacy default stream).
 */
typedef struct swt_bpe_trainer swt_bpe_trainer;

/* From lowercased UTF-8 text: pre-tokenize, dedup words in first-occurrence order, symbolise
 * (source/bpe.py:70-81).  n_base_symbols = number of distinct code points (the initial len(vocab)). */
int swt_bpe_train_create_text(const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent,
                              swt_bpe_trainer **out);
/* The same from the sentences joined with ONE zero byte between neighbours (exactly n_sent - 1 zero bytes: the input of
 * swt_utf8_prepare_joined, which also says what "lowercased" means here): the device finds the separators, lowercases and
 * takes the census without the prepared text travelling to the host and back.  need_host[n_sent] comes back as from
 * swt_utf8_lower; when it flags a sentence, *out stays NULL (SWT_OK) and the caller lowercases on the host and uses
 * swt_bpe_train_create_text. */
int swt_bpe_train_create_joined(const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint8_t *need_host, swt_bpe_trainer **out);
/* NaiveWP.train's state instead (source/wordpiece.py:44-63): the same split and word dedup, symbols
 * = the word's first code point c and SWT_WP_CONT + c ("##c") for the others, plus exact symbol frequencies.  The handle
 * works with every swt_bpe_train_* call below; the step maximises the likelihood score freq / (f_left * f_right)
 * (source/wordpiece.py:84-92: Python's int / int, the correctly rounded quotient; first maximum in first-occurrence
 * order) and the `count` outputs carry that score's IEEE-754 bit pattern.  Merged symbols start at SWT_WP_MERGED_BASE;
 * the caller names them left + right[2:] (source/wordpiece.py:95). */
#define SWT_WP_CONT 0x110000u
#define SWT_WP_MERGED_BASE 0x220000u
int swt_wp_train_create_text(const uint8_t *text, const uint64_t *sent_off, uint64_t n_sent,
                             swt_bpe_trainer **out);
/* ... and from the joined texts, as swt_bpe_train_create_joined (NULL handle + SWT_OK: a sentence needs the host's lower()). */
int swt_wp_train_create_joined(const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint8_t *need_host, swt_bpe_trainer **out);
/* From an already deduplicated word list (symbol ids, CSR offsets, frequencies). */
int swt_bpe_train_create_words(const uint32_t *syms, const uint64_t *word_off, const uint32_t *freq,
                               uint64_t n_words, swt_bpe_trainer **out);
void swt_bpe_train_destroy(swt_bpe_trainer *t);
/* Rank-sharded training: pos_base orders this shard's words after those of lower ranks in the
 * first-occurrence tie-break (source/bpe.py:102).  Default 0. */
int swt_bpe_train_set_pos_base(swt_bpe_trainer *t, uint64_t pos_base);
int swt_bpe_train_info(const swt_bpe_trainer *t, uint64_t *n_words, uint64_t *n_symbols, uint32_t *n_base_symbols,
                       uint64_t *n_pairs);
/* distinct code points of the corpus, ascending (the initial vocab, source/bpe.py:75) */
int swt_bpe_train_base_symbols(const swt_bpe_trainer *t, uint32_t *out, uint32_t cap);
/* Most frequent pair (source/bpe.py:90-102).  *count == 0 means no pair is left (source/bpe.py:98-99).
 * *n_tied = number of pairs holding the maximum.  When it is 1, (left,right) is that pair and *first_pos is
 * ~0.  When it is > 1 the tie is broken by the earliest (word, position) in THIS handle's stream:
 * (left,right) is the pair found there and *first_pos = pos_base + its stream position; if none of the tied
 * pairs occurs in this handle's stream, *first_pos = ~0 and left = right = 0xFFFFFFFF. */
int swt_bpe_train_best(swt_bpe_trainer *t, uint32_t *left, uint32_t *right, uint64_t *count, uint64_t *n_tied,
                       uint64_t *first_pos);
/* Replace every L->R non-overlapping occurrence of (left,right) by merged (source/bpe.py:25-48,
 * 108-111) and update the histogram. */
int swt_bpe_train_apply(swt_bpe_trainer *t, uint32_t left, uint32_t right, uint32_t merged);
/* Device-driven training: up to max_steps iterations of {best, apply} enqueued back to back with no host round trip
 * per merge.  Step i merges into symbol id first_merged + i -- valid while every merged STRING is new, which the caller
 * checks afterwards from (left, right) (symbols are identified by their string; on the never-observed collision the
 * caller replays with swt_bpe_train_apply).  Stops early when no pair is left; *n_done = merges performed. */
int swt_bpe_train_run(swt_bpe_trainer *t, uint32_t max_steps, uint32_t first_merged, uint32_t *left, uint32_t *right,
                      uint64_t *count, uint32_t *n_done);
/* Copies the current stream back (parity checks, corpus_as_symbols): syms[n_symbols], word_off[n_words+1],
 * freq[n_words] (freq may be NULL). */
int swt_bpe_train_export(swt_bpe_trainer *t, uint32_t *syms, uint64_t syms_cap, uint64_t *word_off, uint32_t *freq);
/* Copies the histogram back: up to cap (key = left<<32|right, count) entries with count > 0. */
int swt_bpe_train_histogram(swt_bpe_trainer *t, uint64_t *keys, uint64_t *counts, uint64_t cap, uint64_t *n);

/* Diagnostics of the trainer's machinery (tests, bench.py): out[0..n) = re-plans of the candidate threshold so far, the
 * threshold theta (0: full-table argmax), candidate list length, inverted-index log entries, pair-table slots, state flags
 * (bit 0: the index was abandoned for whole-stream applies), steps enqueued, distinct keys in the table, index entries the
 * apply launches have gone through, words the tie scans have covered (the last two: bench.py's bytes-per-merge model). */
int swt_bpe_train_stats(const swt_bpe_trainer *t, uint64_t *out, uint32_t n);

/* One row per merge performed by swt_bpe_train_run(_sharded) so far: rows[4 i ..] = the winning count (score bits for
 * WordPiece), the number of pairs that held that maximum (> 1: the first-occurrence tie-break ran), the candidate list
 * length, and the live symbols N_t before the merge (what the reference would have rescanned).
 * rows may be NULL to query *n_rows. */
int swt_bpe_train_trace(const swt_bpe_trainer *t, uint64_t *rows, uint64_t cap_rows, uint64_t *n_rows);

/* ------------------------------------------------------------------------------------------------
 * Corpus-sharded BPE training, one process per GPU over RCCL: the multi-GPU form of the merge loop of
 * source/bpe.py:88-111 (the reference itself has no parallelism).  Every rank builds a trainer from ITS contiguous range of
 * sentences (swt_bpe_train_create_text) and keeps the pair histogram of the whole corpus; per merge the runner issues ONE
 * fixed-size ncclAllGather of packed (pair, delta) records and ONE 16-byte all-gather for the first-occurrence tie-break
 * (source/bpe.py:102; ranks are ordered by their sentence ranges), enqueued on the training stream with no host round trip,
 * up to 256 merges per call into the device.
 *
 *   rank 0: swt_dist_unique_id(id) -> the caller hands the 128 bytes to every rank (torch.distributed, MPI, a file ...)
 *   every rank: swt_init(local device); swt_dist_init(rank, world, id, &comm);
 *               swt_bpe_train_create_text(shard) -> t;  swt_bpe_train_shard_begin(&t, 1, comm, base, cap, &n_base);
 *               loop { swt_bpe_train_run_sharded(&t, 1, comm, k, first_merged, left, right, count, &n_done); intern strings }
 * The outputs are identical on every rank.  swt_dist_init_local makes a loop-back communicator whose `world` ranks are all
 * trainers of the calling process (trainers[r] = rank r; device copies instead of RCCL): the same runner on one GPU. */
typedef struct swt_dist swt_dist;
int swt_dist_unique_id(uint8_t *out128);
int swt_dist_init(int rank, int world, const uint8_t *unique_id128, swt_dist **out);
int swt_dist_init_local(int world, swt_dist **out);
void swt_dist_destroy(swt_dist *d);
int swt_dist_info(const swt_dist *d, int *rank, int *world, int *is_local);
/* Enters sharded mode: gathers the distinct initial symbols of ALL shards (the initial vocab, source/bpe.py:75; written to
 * base_out[0..*n_base) ascending when base_out is not NULL) and reduces the local histograms into every replica. */
int swt_bpe_train_shard_begin(swt_bpe_trainer **trainers, uint32_t n_local, swt_dist *d, uint32_t *base_out, uint32_t base_cap,
                              uint32_t *n_base);
/* As swt_bpe_train_run, over all shards. */
int swt_bpe_train_run_sharded(swt_bpe_trainer **trainers, uint32_t n_local, swt_dist *d, uint32_t max_steps, uint32_t first_merged,
                              uint32_t *left, uint32_t *right, uint64_t *count, uint32_t *n_done);
#ifdef __cplusplus
}
#endif
#endif /* SWT_H */
