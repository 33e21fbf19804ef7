// swt_words.h -- unique words of a corpus, built on the device (swt_words.hip).
#pragma once

#include "swt_common.h"

namespace swt {

struct DeviceWords {
  uint32_t *d_sym = nullptr;   // list(word) of every unique word, concatenated (code points)   [n_syms + 16]
  uint64_t *d_woff = nullptr;  // n_words + 1 offsets into d_sym
  uint32_t *d_freq = nullptr;  // occurrences of each unique word
  uint64_t n_words = 0, n_syms = 0;
  std::vector<uint32_t> base_syms;  // distinct code points, ascending
};

// utils.py:26-29 split + bpe.py:73-81 Counter/symbolise over lowercased UTF-8 resident in HBM.  The arrays handed
// back are owned by the caller (hipFree).
int device_words_from_text(const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off, uint64_t n_sent, DeviceWords *out);
// swt_lower.hip: U+0000-joined host text -> lowercased text + sentence offsets in the calling thread's device workspace
int prepare_joined_dev(const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint8_t *need_host, const uint8_t **d_text,
                       const uint64_t **d_off, uint64_t *n_bytes_out);
// The same under the workspace's guard, for an encoder: `consume(ctx, d_text, n_bytes, d_off)` runs on the prepared text unless a
// sentence needs the host's str.lower() (*consumed says which); the workspace is released if anything fails.
typedef int (*PreparedConsumer)(void *ctx, const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_off);
int with_prepared_joined(const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint8_t *need_host, bool *consumed,
                         PreparedConsumer consume, void *ctx);

}  // namespace swt
