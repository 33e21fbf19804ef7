// swt_tile.h -- the tile/chunk skeleton shared by the two encode kernels (BPE and WordPiece).
//
//   tile   = the sentences whose first byte lies in one kTile-byte window of the text.  A tile owns whole
//            sentences: no sentence is seen by two workgroups, there are no data-path atomics, and a tile's
//            tokens are contiguous in the final output.
//   chunk  = up to kCap bytes of the tile's span staged in LDS (16-byte aligned base so the staging loads are
//            dwordx4).  One chunk is the common case; longer spans are walked chunk by chunk.
//   per chunk: A stage bytes -> B decode/classify per byte (encoder specific) -> C/D encoder body leaves one
//            token id or kInvalidTok in sym[] at byte granularity -> E order-preserving ballot compaction to
//            the tile's output run -> F per-sentence local offsets.
//   after the encode kernel: scan of tile totals, gather into the caller's CSR (ids + sentence offsets).
#pragma once

#include "swt_common.h"

namespace swt {

constexpr int kTile = 2048;
constexpr int kCap = 4096;
constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kQuarter = kCap / kWaves;  // byte positions per wave
constexpr int kBlocks64 = kCap / 64;

struct TileLds {
  __attribute__((aligned(16))) uint8_t txt[kCap + 16];  // staged bytes
  uint8_t cls[kCap + 16];                                // per byte: encoder-specific class bits
  uint32_t sym[kCap];                                    // per byte: code point -> token id / kInvalidTok
  unsigned long long sbits[kBlocks64 + 1];               // sentence-start bit per byte
  unsigned long long vmask[kBlocks64 + 1];               // valid-token bit per byte (phase E)
  uint32_t blkpre[kBlocks64 + 1];                        // tokens before each 64-byte block (phase E)
  uint32_t wtot[kWaves];
  int cut;
  uint32_t cnt;
};

__device__ __forceinline__ bool tile_sbit(const TileLds &L, uint32_t p) { return (L.sbits[p >> 6] >> (p & 63)) & 1ull; }

// A: stage [abase, abase+staged) and reset the per-chunk words.  Ends with a barrier.
__device__ __forceinline__ void tile_stage(TileLds &L, const uint8_t *__restrict__ text, uint64_t n_bytes, uint64_t abase,
                                           uint32_t staged) {
  const int tid = threadIdx.x;
  for (uint32_t c = tid * 16; c < staged; c += kThreads * 16) {
    const uint64_t g = abase + c;
    if (g + 16 <= n_bytes && ((reinterpret_cast<uintptr_t>(text + g) & 15) == 0)) {
      *reinterpret_cast<uint4 *>(&L.txt[c]) = *reinterpret_cast<const uint4 *>(text + g);
    } else {
      for (int i = 0; i < 16; i++) L.txt[c + i] = (g + i < n_bytes) ? text[g + i] : (uint8_t)' ';
    }
  }
  for (int i = tid; i <= kBlocks64; i += kThreads) L.sbits[i] = 0ull;
  if (tid == 0) { L.cut = -1; L.cnt = 0; }
  __syncthreads();
}

// Sentence-start bits for the sentences that begin inside the staged bytes.  No barrier.
__device__ __forceinline__ void tile_mark_sentences(TileLds &L, const uint64_t *__restrict__ sent_off, uint64_t s_next,
                                                    uint64_t s_hi, uint64_t cb, uint64_t abase, uint32_t staged) {
  for (uint64_t s = s_next + threadIdx.x; s < s_hi; s += kThreads) {
    const uint64_t o = sent_off[s];
    if (o >= abase + staged) break;
    if (o >= cb) atomicOr(&L.sbits[(o - abase) >> 6], 1ull << ((o - abase) & 63));
  }
}

// E: compaction of sym[off0, ce) (everything != kInvalidTok, in order) to out[0..total).  Returns total.
// Must be entered after a barrier that makes sym[] final; ends with a barrier.
__device__ __forceinline__ uint32_t tile_compact(TileLds &L, uint32_t off0, uint32_t ce, uint32_t *__restrict__ out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t mytot = 0;
  for (int r = 0; r < kQuarter / 64; r++) {
    const uint32_t p = wave * kQuarter + r * 64 + lane;
    const bool v = p >= off0 && p < ce && L.sym[p] != kInvalidTok;
    const unsigned long long m = __ballot(v);
    if (lane == 0) L.vmask[wave * (kQuarter / 64) + r] = m;
    mytot += __popcll(m);
  }
  if (lane == 0) L.wtot[wave] = mytot;
  __syncthreads();
  uint32_t wbase = 0, total = 0;
  for (int w = 0; w < kWaves; w++) {
    if (w < wave) wbase += L.wtot[w];
    total += L.wtot[w];
  }
  uint32_t pre = wbase;
  for (int r = 0; r < kQuarter / 64; r++) {
    const int blk = wave * (kQuarter / 64) + r;
    const uint32_t p = blk * 64 + lane;
    const unsigned long long m = L.vmask[blk];
    if (lane == 0) L.blkpre[blk] = pre;
    if ((m >> lane) & 1ull) out[pre + __popcll(m & lt)] = L.sym[p];
    pre += __popcll(m);
  }
  __syncthreads();
  return total;
}

// F: tile-local token offset of every sentence starting in [cb, ce) (and == ce on the last chunk).
// Counts the recorded sentences in L.cnt; ends with a barrier and returns that count.
__device__ __forceinline__ uint32_t tile_record(TileLds &L, const uint64_t *__restrict__ sent_off,
                                                uint32_t *__restrict__ sent_local, uint64_t s_next, uint64_t s_hi,
                                                uint64_t abase, uint32_t ce, bool last, uint32_t run, uint32_t total) {
  for (uint64_t s = s_next + threadIdx.x; s < s_hi; s += kThreads) {
    const uint64_t rel = sent_off[s] - abase;
    if (rel > ce || (rel == ce && !last)) break;
    uint32_t e = total;
    if (rel < ce) e = L.blkpre[rel >> 6] + __popcll(L.vmask[rel >> 6] & ((1ull << (rel & 63)) - 1ull));
    sent_local[s] = run + e;
    atomicAdd(&L.cnt, 1u);
  }
  __syncthreads();
  const uint32_t c = L.cnt;
  __syncthreads();  // L.cnt is reset by the next tile_stage
  return c;
}

// Per-call workspaces shared by both encoders (grow-only).
struct TileWorkspace {
  DevBuf plan, scratch, sent_local, tile_tok, tile_base, blk;
  int reserve(uint64_t n_bytes, uint64_t n_sent, uint64_t n_tiles);
  void release();
};

inline uint64_t tile_count(uint64_t n_bytes, uint32_t tile = kTile) { return n_bytes ? (n_bytes + tile - 1) / tile : 1; }

// Host launchers of the skeleton's own kernels (defined in swt_tile.hip).
void launch_plan(const uint64_t *d_sent_off, uint64_t n_sent, uint64_t n_tiles, uint32_t tile, uint64_t *d_plan, hipStream_t st);
// sizes on the device: *d_total = sentences:32 | bytes:32; plan[0 .. n_tiles_max] is written for the smallest tile size
// >= tile_min that needs at most n_tiles_max tiles; the tiles behind the real ones are empty
void launch_plan_dev(const uint64_t *d_sent_off, const unsigned long long *d_total, uint64_t n_tiles_max, uint32_t tile_min,
                     uint64_t *d_plan, hipStream_t st);
void launch_scan_only(uint64_t n_tiles, const TileWorkspace &ws, uint64_t *d_n_tokens, hipStream_t st);
// the same scan over 64-bit values: d_local[i] = exclusive sum inside i's group of 1024, blk = [ticket (zero), totals[nb],
// bases[nb]] with nb = ceil(n / 1024); global exclusive sum of i = blk[1 + nb + (i >> 10)] + d_local[i]
void launch_scan_u64(uint64_t n, const unsigned long long *d_in, unsigned long long *d_local, unsigned long long *blk,
                     uint64_t *d_total, hipStream_t st);
void launch_scan_gather(const uint64_t *d_sent_off, uint64_t n_sent, uint64_t n_tiles, const TileWorkspace &ws,
                        uint32_t *d_out_ids, uint64_t *d_out_off, uint64_t *d_n_tokens, hipStream_t st);

}  // namespace swt
