// swt_train.h -- the trainer's device state and the pieces swt_dist.hip (sharded training) drives.
#pragma once

#include <vector>

#include "swt_common.h"

namespace swt {

constexpr uint32_t kMaxRunSteps = 4096;  // rows of the step log = merges per host round trip
constexpr uint32_t kMaxBatch = 16;       // merges ONE step of the fast path may carry: tied pairs that cannot affect each other
constexpr uint32_t kRunBatch = 256;     // merges enqueued per host round trip
constexpr uint32_t kArgParts = 256;     // workgroup partials of an argmax launch (combined by every consumer)
constexpr uint32_t kCandBlocks = 32;    // cand_argmax_kernel
constexpr uint32_t kCandTarget = 1024;  // pairs the candidate threshold theta lets through at a re-plan: every workgroup of the
                                        // fast path scans the whole list itself, so it is kept short
constexpr uint32_t kCandHigh = 2048;    // re-plan at a batch boundary once pushes have grown the list beyond this
constexpr uint32_t kCandCap = 8192;     // room for the pairs that cross theta afterwards
constexpr uint32_t kTieSet = 256;       // tied pairs a workgroup keeps in its LDS set (more: membership by table probe)
#ifndef SWT_TIE_BLOCKS
#define SWT_TIE_BLOCKS 256
#endif
constexpr uint32_t kTieBlocks = SWT_TIE_BLOCKS;
constexpr uint32_t kApplyBlocks = 512;
constexpr uint32_t kWpStepBlocks = 32;     // wp_step_kernel: every workgroup takes the argmax of the list itself
constexpr uint32_t kWpStepList = 16384;    // ... which pays while the list of live pairs is at most this long
#ifndef SWT_APPLY_BLOCKS
#define SWT_APPLY_BLOCKS 128
#endif
constexpr uint32_t kFastApplyBlocks = SWT_APPLY_BLOCKS;
constexpr uint32_t kPackBlocks = 16;
constexpr uint32_t kSegBase = 0xFFFFFFFFu;  // seg_of[]: the id names a symbol of the initial stream

constexpr uint32_t kFlagIndexBroken = 1u;  // a merged id was reused (or the index log overflowed): applies scan every word
constexpr uint32_t kFlagBrokenPending = 4u; // fast path: kFlagIndexBroken from the next step on
constexpr uint32_t kFlagReplan = 2u;       // the candidate list ran dry or overflowed: steps are no-ops until the host re-plans
constexpr uint32_t kFlagTableFull = 8u;    // an insert found no free slot in a whole turn of the pair table: the host's head-room
                                           // bound was wrong.  The counts are no longer exact; the host returns SWT_ERR_STATE

struct TrainState {
  unsigned long long max_count;  // result of the last decide: the maximum (count, or WordPiece score bits)
  unsigned long long n_tied;     //   pairs holding it
  unsigned long long best_pos;   // tie scan: word << 32 | offset (kEmptyKey: none found); reset by decide
  unsigned long long best_key;   //   the winning pair
  unsigned long long n_used;     // distinct keys in the table
  unsigned long long res_pos;    //   position of the tie winner in this shard (kEmptyKey: not tied / not here)
  unsigned long long n_syms;     // live symbols
  unsigned long long win_key;
  unsigned long long n_cand;     // candidate list length
  unsigned long long idx_cursor; // entries in the index log
  unsigned long long plateau;    // count level the tie cursor holds for
  unsigned long long n_touched;  // sharded: slots with pending deltas
  unsigned long long scratch;
  unsigned long long best2[2];   // fast path: tie scan result of the step, by step parity (the other one is reset meanwhile)
  unsigned long long n_synced;   // candidates [0, n_synced) have their count and key mirrored in ccnt[] / ckey[]
  unsigned long long n_synced_next;
  unsigned int cursor_w;         // the tie scan may start at this word
  unsigned int flags;
  // fast path, reset by the host before every round trip
  unsigned long long run_done[2];  // merges logged so far in this round trip, by step parity
  unsigned long long halt;         // 0, 2 (no pair left), 3 (candidate list dry or lost: re-plan)
  unsigned long long win_end;      // tie positions below this (word << 32) lie in the window EVERY workgroup scanned
  unsigned long long n_list[2];    // length of tied_idx[] / tied_key[], by step parity
  unsigned long long step_syms;    // live symbols when the step began (the tie launch notes it: no apply is in flight then)
  unsigned long long run_active;   // steps of this round trip that carried merges
  // what the merge steps have looked at so far (bench.py's bytes-per-merge model; one lane adds, launches are serial)
  unsigned long long ent_scanned;  // index entries the apply launches went through
  unsigned long long tie_words;    // words the tie scans' first trips covered
  unsigned long long last_open;    // the last step whose index segment has been opened (seg_start[] is written up to here): a
                                   // segment at or after it ends where the log stands, not at seg_start[seg + 1]
  unsigned long long ticket;       // wp_step_kernel: workgroups that have finished (the last one decides the step)
};

struct StepCmd {
  uint32_t l, r, m, valid;
};

struct StepLog {
  uint32_t l, r;
  unsigned long long count;
  unsigned long long flag;    // 0 merged, 2 no pair left, 3 candidate list dry (re-plan), 4 exchange block overflow (sharded)
  unsigned long long n_syms;  // live symbols before this merge
  unsigned long long n_tied;  // pairs that held the maximum
  unsigned long long n_cand;  // candidate list length
};

struct ArgPart {
  unsigned long long mx, cnt, key;
};

struct DeltaRec {
  unsigned long long key;
  long long delta;
};

struct PairTable {
  unsigned long long *keys;
  long long *cnt;
  uint32_t bits;
};

// the words of every pair of the initial stream, grouped by key (static)
struct K0Index {
  unsigned long long *keys;
  uint32_t *start, *len, *fill;
  uint32_t *words;
  uint32_t bits;
};

// where the words of a pair are listed: kind 0 nowhere (an empty list), 1 the static index of the initial pairs
// (K.words + start), 2 a segment of the log (idx_word / idx_tag + start, entries of other pairs told apart by `want`)
struct TiedPlan {
  unsigned long long start, n_ent;
  uint32_t kind, want;
};

// Sharded fast path (swt_dist.hip): what one rank tells the others about the tie scan of a step -- its (at most kMaxBatch)
// earliest tied pairs INSIDE the window every one of its workgroups scanned, with the neighbour evidence that can clear a
// "dangerous" pair, and the earliest tied occurrence anywhere in its shard.  One fixed-size all-gather per step.
struct TieEntry {
  unsigned long long pos, key;  // word << 32 | offset in this shard; the pair
  uint32_t nb_lo[2], nb_hi[2];  // smallest / largest left [0] and right [1] neighbour seen of the pair's occurrences
  uint32_t danger, pad;         // dangerous by the symbol sets of the tied pairs (the same answer on every rank)
};
struct TieMsg {
  unsigned long long win_end;   // the window's end (word << 32)
  unsigned long long min_pos;   // earliest tied occurrence in this shard (kEmptyKey: none; the scan goes on past the window
  unsigned long long min_key;   //   until it has one) and its pair.  Untied step: min_pos = kEmptyKey, min_key = THE pair
  uint32_t n;                   // entries
  uint32_t exhausted;           // the window reached the end of the shard, or the shard holds no tied pair at all: what this
                                // rank reports is everything it has
  TieEntry e[kMaxBatch];
};

// everything a training kernel needs, by value
struct TrainCtx {
  PairTable T;
  K0Index K;
  TrainState *st;
  long long *sfreq;        // WordPiece: symbol frequencies
  unsigned long long theta;
  uint32_t *cand;
  uint64_t cand_cap;
  long long *ccnt;           // compact mirror of the candidates' counts (cidx[slot] = place in the list), so that an argmax
  unsigned long long *ckey;  // reads the list as a coalesced stream instead of a gather from the pair table
  uint32_t *cidx;
  uint32_t *idx_tag, *idx_word;
  uint64_t idx_cap;
  unsigned long long *seg_start;
  uint32_t *seg_of;
  uint32_t seg_cap, id_base;
  uint32_t *wstamp;
  unsigned long long *wkey;  // fast path: the tied pair a tie scan found in word w
  uint32_t *tied_idx;        // fast path, [2][kTieSet]: candidate index of every tied pair | danger << 31 (see fast_tie_kernel)
  unsigned long long *tied_key;  // their keys
  TiedPlan *tied_plan;       // fast path, [2][kTieSet]: where the words of each tied pair are listed (fast_tie_kernel's planner)
  uint32_t *gnb_min, *gnb_max;  // fast path, [2][cand_cap][2]: smallest / largest left (0) and right (1) neighbour seen of a tied pair's
                                // occurrences (fast_tie_kernel: what clears a "dangerous" pair); none seen: min > max
  unsigned long long *gpos;  // fast path, [2][cand_cap]: first position of a tied pair within the scanned words, by candidate index
  uint32_t step;
  long long *pend;         // sharded: per-slot pending deltas (nullptr: deltas go straight into cnt)
  uint32_t *tstamp, *touched;
  uint64_t touched_cap;
  // sharded fast path
  const unsigned int *halt_ext;  // an exchange block overflowed somewhere: steps are no-ops until the host has repeated it
  TieMsg *tie_msg;               // this rank's message of the step
  const TieMsg *tie_all;         // every rank's, in rank order (after the all-gather)
  uint32_t world, rank;
};

}  // namespace swt

struct swt_bpe_trainer {
  uint64_t n_words = 0, n_syms0 = 0;
  uint32_t n_base = 0;
  uint64_t n_base_global = 0;  // sharded: distinct initial symbols over all ranks
  std::vector<uint32_t> base_syms;
  hipStream_t stream = 0;
  uint32_t *d_sym = nullptr;
  uint64_t *d_woff = nullptr;
  uint64_t extent = 0;            // stream slots behind d_woff[n_words] (n_syms0 until the first squeeze)
  uint32_t *d_sym_alt = nullptr;  // the other half of the squeeze's ping-pong (swt_bpe_train.hip: squeeze_stream)
  uint64_t *d_woff_alt = nullptr;
  uint64_t n_squeezes = 0;
  uint32_t *d_freq = nullptr;
  swt::PairTable T{nullptr, nullptr, 0};
  swt::K0Index K{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  swt::TrainState *d_st = nullptr;
  swt::ArgPart *d_parts = nullptr;
  uint32_t n_parts = 0;
  swt::StepCmd *d_cmd = nullptr;
  swt::StepLog *d_steplog = nullptr;
  uint64_t n_applied = 0;  // merges applied so far (bounds the number of distinct symbols)
  uint32_t step_no = 0;    // steps enqueued so far (index segments, word stamps)
  swt::TrainState h_st{};
  uint64_t pos_base = 0;
  bool hist_ready = false;
  long long *d_sfreq = nullptr;  // WordPiece mode: symbol frequencies, dense by symbol id
  uint32_t id_base = SWT_SYM_BASE;
  // candidates
  unsigned long long theta = 0;
  uint32_t *d_cand = nullptr;
  uint64_t cand_cap = 0;
  long long *d_ccnt = nullptr;
  unsigned long long *d_ckey = nullptr;
  uint32_t *d_cidx = nullptr;
  uint32_t cidx_bits = 0;
  unsigned long long *d_buckets = nullptr;
  bool cand_valid = false;
  uint64_t n_replans = 0;
  // when will the list run dry?  A list of n pairs has so far been good for about dry_ratio * n merges
  uint64_t cand_built = 0;    // pairs the last re-plan listed
  uint64_t since_replan = 0;  // merges since then
  double dry_ratio = 0.6;
  // index
  uint32_t *d_idx_tag = nullptr, *d_idx_word = nullptr, *d_wstamp = nullptr;
  unsigned long long *d_wkey = nullptr;
  uint32_t *d_tied_idx = nullptr;
  unsigned long long *d_tied_key = nullptr, *d_gpos = nullptr;
  swt::TiedPlan *d_tied_plan = nullptr;
  uint32_t *d_gnb_min = nullptr, *d_gnb_max = nullptr;
  uint64_t idx_cap = 0;
  unsigned long long *d_seg_start = nullptr;
  uint64_t seg_start_cap = 0;
  uint32_t *d_seg_of = nullptr;
  uint32_t seg_cap = 0;
  // sharded
  bool sharded = false;
  uint32_t world = 1;
  long long *d_pend = nullptr;
  uint32_t *d_tstamp = nullptr, *d_touched = nullptr;
  uint64_t touched_cap = 0, block_cap = 0;
  swt::DeltaRec *d_block = nullptr, *d_blocks_all = nullptr;
  unsigned long long *d_tie_line = nullptr, *d_tie_all = nullptr;
  unsigned int *d_halt = nullptr;
  swt::TieMsg *d_tie_msg = nullptr, *d_tie_msgs = nullptr;  // fast path: this rank's message, all ranks' messages
  uint32_t rank = 0;
  swt::DevBuf tmp;
  std::vector<swt::StepLog> trace;  // every merge so far (swt_bpe_train_trace)

  swt::TrainCtx ctx() const;
  int sync_state();
  int replan();
  void enqueue_argmax();
  void enqueue_apply();
  void enqueue_fast_step(uint32_t first_merged, uint32_t limit);
  // SWT_OK when the handle went through finish_create (histogram, index) and no capacity check has failed since:
  // every entry point that steps or reads the trainer asks first (a half-built handle used to mean a null pair table
  // under a kernel: a GPU memory fault, i.e. a process abort)
  int ready() const;
  // after a synchronisation: the invariants the device code relies on, from h_st
  int check_state();
  bool broken = false;  // check_state failed once: the handle refuses further steps
};

namespace swt {
int trainer_enter_sharded(swt_bpe_trainer *t, uint32_t world, uint64_t block_cap);
int trainer_set_block_cap(swt_bpe_trainer *t, uint64_t block_cap);
int trainer_export_records(swt_bpe_trainer *t, DeltaRec *d_out, uint64_t cap, uint64_t *n);
int trainer_add_records(swt_bpe_trainer *t, const DeltaRec *d_recs, uint64_t n);
int trainer_prepare_batch(swt_bpe_trainer *t, uint32_t k, uint32_t max_merged);
void trainer_enqueue_tie_send(swt_bpe_trainer *t);
void trainer_enqueue_decide_apply(swt_bpe_trainer *t, uint32_t rank, uint32_t log_i, uint32_t merged);
void trainer_enqueue_pack(swt_bpe_trainer *t);
void trainer_enqueue_add_blocks(swt_bpe_trainer *t);
// the fast two-launch step, sharded (several tied merges per step; see fast_apply_sharded_kernel)
struct ShardTrip { uint32_t steps, cap; bool replan_first, fast; };
int trainer_fast_room(swt_bpe_trainer *t, uint32_t remaining, double per_step, ShardTrip *trip);  // head room, wants a re-plan?
int trainer_fast_plan(swt_bpe_trainer *t, uint32_t remaining, double per_step, bool replan, uint32_t first_id, ShardTrip *trip);
int trainer_fast_begin(swt_bpe_trainer *t);
void trainer_enqueue_fast_tie(swt_bpe_trainer *t, uint32_t limit);
void trainer_enqueue_fast_apply(swt_bpe_trainer *t, uint32_t first_merged, uint32_t limit);
}  // namespace swt
