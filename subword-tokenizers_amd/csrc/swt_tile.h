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
  int reserve(uint64_t n_bytes, uint64_t n_sent, uint64_t n_tiles);
  void release();
};

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
