// swt_dedup.h -- word-level dedup inside one encode call, shared by the FastBPE and the FastWP encoder.
//
// Natural text repeats its words (S85k: 1.5 M words, 79 k distinct).  Both encoders are pure functions of a WORD of the
// text -- FastBPE.encode_word of a pre-tokenizer word (bpe.py:205-249); FastWP's segment loop of a whitespace-delimited
// chunk, as long as no vocabulary token contains whitespace (wordpiece.py:251-269: a segment never reads past the
// whitespace that ends its chunk) -- so each DISTINCT word of a call is encoded once:
//   front   plan, wordref (one wave per 1-KiB tile: the split, then one lane per word -- hash, table lookup with an EXACT
//           byte compare against the slot's representative occurrence, or insert), scan, ureg (the words a tile inserted
//           are copied to the unique-word text in tile order)
//   (the caller encodes the unique words: unique word u leaves count:32 | place:32 of its token run in drec[u] -- a dense
//   array that stays in L2 -- and count:32 | u:32 in rec[its table slot])
//   back    refcount (tokens per tile from the word records; the records are rewritten from table slots to unique indices on
//           the way, so that the last pass gathers from the dense array only), scan, refwrite (tokens + sentence offsets)
// Slots carry an 8-bit epoch, so the table is never cleared between calls; nothing here returns to the host.
#pragma once
#include "swt_common.h"
#include "swt_tile.h"

namespace swt {

// below these sizes the extra launches cost more than the work they save (measured crossovers: FastBPE 1.7 MB, FastWP 2.8 MB;
// tools/gpu_scale_sweep.py, tools/gpu_wp_crossover.py)
constexpr uint64_t kDedupMinBytes = 7u << 18;     // FastBPE: 1.75 MiB
constexpr uint64_t kDedupMinBytesWp = 11u << 18;  // FastWP: 2.75 MiB
constexpr uint64_t kDedupMaxBytes = 1ull << 30;  // 32-bit fields of the records
constexpr uint32_t kDedupRetry = 256;            // direct calls between two looks at a text that repeats few of its words
constexpr uint32_t kRecFailed = 0xFFFFFFFFu;     // count field of rec[]: the word cannot be encoded (FastWP non-termination)

enum DedupMode {
  kDedupBpe = 0,  // words of the BertPreTokenizer split (utils.py:27); a one-symbol word is its own token
  kDedupWp = 1,   // chunks between str.isspace characters (wordpiece.py:268); a sentence with a failed chunk yields no tokens
};

struct DedupEngine {
  DevBuf slot, rec, drec, uslot, utext, uoff, misc, newlist, tile_new, new_local, new_blk, tile_words;
  uint32_t bits = 0, epoch = 0;
  int opt_mode = 0;            // SWT_OPT_DEDUP of the owning handle: 0 by batch size, 1 never, 2 always
  uint32_t opt_table_bits = 0; // SWT_OPT_DEDUP_TABLE_BITS: log2 of the word table's slots (0: sized from the batch)
  // What the last deduplicated call found, copied to pinned memory behind its kernels (never waited for): the owner skips
  // the dedup while the text repeats too few of its words for it to pay, and looks again every kDedupRetry calls.
  PinnedBuf seen;              // [0] unique words:32 | their bytes:32 of the last dedup call, [1] that call's text bytes
  uint32_t skipped = 0;        // calls since the dedup last ran
  // the dedup pays while unique bytes x pay_ratio <= text bytes.  Measured with the word-lane kernel (S85k-lex, 7 % unique bytes:
  // 170 us deduplicated against 174 us direct; S85k-open, 34 %: ~400 against 183): the front and back halves cost about what the
  // direct kernel costs on a tenth of the text
  uint32_t pay_ratio = 12;
  bool pays(uint64_t n_bytes);
  void note(uint64_t n_bytes, hipStream_t st);
  void release();
  unsigned long long *rec_ptr() const { return rec.as<unsigned long long>(); }    // per table slot: count:32 | unique index:32
  unsigned long long *drec_ptr() const { return drec.as<unsigned long long>(); }  // per unique word: count:32 | place:32
  const unsigned long long *total_ptr() const { return misc.as<unsigned long long>(); }  // unique words:32 | their bytes:32
};

// Front half.  On return (stream order): ws.scratch = dense word records per tile, ws.sent_local = words of the tile before
// each sentence, E.tile_words, E.utext / E.uoff / E.uslot = the unique words, E.misc[0] = their number and bytes.
// d_plan2[0 .. n_tiles2]: the tile plan of the caller's launch over the unique words (first unique word at or after byte
// tt * tile2, with tile2 = the smallest size >= tile2_min that covers the unique bytes with n_tiles2 tiles).
// Returns 1 when the batch does not fit this path (the caller encodes directly).
int dedup_front(DedupEngine &E, TileWorkspace &ws, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off, uint64_t n_sent,
                const uint8_t *d_cls, DedupMode mode, hipStream_t st, uint64_t *d_plan2, uint64_t n_tiles2, uint32_t tile2_min);

// Back half.  d_unique_tokens = the buffer the rec[] places point into.  kDedupWp also writes d_status per sentence.
int dedup_back(DedupEngine &E, TileWorkspace &ws, const uint64_t *d_sent_off, uint64_t n_sent, uint64_t n_bytes,
               const uint32_t *d_unique_tokens, DedupMode mode, uint8_t *d_status, uint32_t *d_out_ids, uint64_t *d_out_off,
               uint64_t *d_n_tokens, hipStream_t st);

}  // namespace swt
