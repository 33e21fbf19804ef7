// swt_tile.h -- the tile/chunk skeleton shared by the two encode kernels (BPE and WordPiece).
//
//   tile   = the sentences whose first byte lies in one window of the text (256 B .. 1 KiB, per kernel).  A tile owns
//            whole sentences: no sentence is seen by two workgroups, there are no data-path atomics, and a tile's
//            tokens are contiguous in the final output.
//   chunk  = the part of the tile's span staged in LDS at a time (16-byte aligned base so the staging loads are
//            dwordx4).  One chunk is the common case; longer spans are walked chunk by chunk.
//   per chunk: A stage bytes -> B decode/classify per byte (encoder specific) -> C/D encoder body leaves one
//            token id or kInvalidTok in sym[] at byte granularity -> E order-preserving ballot compaction to
//            the tile's output run -> F per-sentence local offsets.
//   after the encode kernel: scan of tile totals, gather into the caller's CSR (ids + sentence offsets).
#pragma once

#include "swt_common.h"

namespace swt {

constexpr int kThreads = 256;  // gather_kernel

// Direct mode of the encode kernels, for inputs of a few hundred bytes (the reference-style call: one sentence): ONE workgroup
// takes all sentences as its tile and writes the caller's arrays itself -- 64-bit sentence offsets, the closing offset and the
// token count -- so the call is one launch instead of plan + encode + scan + gather.  off == nullptr: the normal mode.
struct DirectOut {
  uint64_t *off;
  uint64_t *n_tokens;
  uint64_t n_sent;
};

// Per-call workspaces shared by both encoders (grow-only).
struct TileWorkspace {
  DevBuf plan, scratch, sent_local, tile_tok, tile_base, blk;
  DevBuf state;        // look-back words of the fused form (tile_lookback): one per tile
  PinnedBuf lb_err;    // [0] != 0: a look-back gave up (never observed); the owner leaves the fused form for good
  uint32_t epoch = 0;  // tag of the current call in the look-back words (they are never cleared between calls)
  int reserve(uint64_t n_bytes, uint64_t n_sent, uint64_t n_tiles);
  int reserve_state(uint64_t n_tiles);  // also advances the epoch
  void release();
};

// ---- single-pass output placement ("decoupled look-back") ------------------------------------------------------------
// A tile that knows its token count publishes it, finds the number of tokens of all tiles before it by looking back over
// its predecessors' words, and then writes its tokens and sentence offsets straight to their final place: no scratch run, no
// scan launch, no gather launch.  One 64-bit word per tile: epoch:30 | flag:2 | value:32, flag 1 = value is the tile's own
// count, 2 = value is the count of all tiles up to and including it.  The word carries everything that is handed over, so
// relaxed device-scope loads and stores are enough (no fence: an acquire fence invalidates the XCD's L2).  Workgroups are
// dispatched in index order, so every predecessor is resident or finished when a tile waits for it; the wait is bounded
// all the same (err is set and the host leaves this form) -- a wrong assumption must not hang the GPU.
constexpr unsigned long long kLbAgg = 1ull << 32, kLbIncl = 2ull << 32;
constexpr uint32_t kLbSpinLimit = 1u << 22;

__device__ __forceinline__ uint32_t tile_lookback(unsigned long long *__restrict__ state, uint64_t t, uint32_t epoch, uint32_t count,
                                                  int lane, uint32_t *__restrict__ err) {
  const unsigned long long tag = (unsigned long long)epoch << 34;
  if (t == 0) {
    if (lane == 0) __hip_atomic_store(&state[0], tag | kLbIncl | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
  }
  if (lane == 0) __hip_atomic_store(&state[t], tag | kLbAgg | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t before = 0;
  int64_t j = (int64_t)t - 1;  // nearest tile not yet accounted for
  uint32_t spins = 0;
  for (;;) {
    const int64_t idx = j - lane;
    unsigned long long v = tag | kLbIncl;  // "tile -1": nothing before the first tile
    if (idx >= 0) v = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t flag = (v >> 34) == (unsigned long long)epoch ? (uint32_t)(v >> 32) & 3u : 0u;
    const unsigned long long READY = __ballot(flag != 0u), INCL = __ballot(flag == 2u);
    const uint32_t run = ~READY ? (uint32_t)__builtin_ctzll(~READY) : 64u;  // lanes 0 .. run-1 have published
    if (run == 0) {
      if (++spins > kLbSpinLimit) { if (lane == 0) *err = 1u; return 0; }
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    const unsigned long long in_run = run == 64 ? ~0ull : (1ull << run) - 1ull;
    const bool closed = (INCL & in_run) != 0ull;
    const uint32_t last = closed ? (uint32_t)__builtin_ctzll(INCL & in_run) : run - 1;  // sum lanes 0 .. last
    uint32_t x = (uint32_t)lane <= last ? (uint32_t)v : 0u;
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    before += x;
    if (closed) break;
    j -= run;
  }
  if (lane == 0) __hip_atomic_store(&state[t], tag | kLbIncl | (unsigned long long)(before + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return before;
}

inline uint64_t tile_count(uint64_t n_bytes, uint32_t tile) { return n_bytes ? (n_bytes + tile - 1) / tile : 1; }

// Host launchers of the skeleton's own kernels (defined in swt_tile.hip).
void launch_plan(const uint64_t *d_sent_off, uint64_t n_sent, uint64_t n_tiles, uint32_t tile, uint64_t *d_plan, hipStream_t st);
void launch_scan_only(uint64_t n_tiles, const TileWorkspace &ws, uint64_t *d_n_tokens, hipStream_t st);
// the same scan over 64-bit values: d_local[i] = exclusive sum inside i's group of 1024, blk = [ticket (zero), totals[nb],
// bases[nb]] with nb = ceil(n / 1024); global exclusive sum of i = blk[1 + nb + (i >> 10)] + d_local[i]
void launch_scan_u64(uint64_t n, const unsigned long long *d_in, unsigned long long *d_local, unsigned long long *blk,
                     uint64_t *d_total, hipStream_t st);
void launch_scan_gather(const uint64_t *d_sent_off, uint64_t n_sent, uint64_t n_tiles, const TileWorkspace &ws,
                        uint32_t *d_out_ids, uint64_t *d_out_off, uint64_t *d_n_tokens, hipStream_t st);

}  // namespace swt
