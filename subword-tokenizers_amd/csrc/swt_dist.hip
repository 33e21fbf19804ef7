// swt_dist.hip -- corpus-sharded BPE training over RCCL: one process per GPU, no Python in the merge loop.
//
// The multi-GPU form of the merge loop of /root/reference/source/bpe.py:88-111 (the reference has no parallelism at all).
// Every rank owns a contiguous range of sentences, pre-tokenizes and dedups it locally, and keeps
//     * its own symbol stream and inverted index (only local words), and
//     * the pair histogram of the WHOLE corpus, replicated.
// The replicas are hash tables with rank-local slots, so they are reduced sparsely: per merge every rank packs the
// (pair, delta) records its own apply produced into a fixed-size block, ONE ncclAllGather moves the blocks, and every rank
// adds every block (its own too) to its replica.  All replicas then hold the same counts, so every rank derives the same
// maximum and the same tie set on its own; a tie is settled by ONE 16-byte all-gather of (first position, pair) per rank:
// ranks are ordered by their sentence ranges, so the first rank that holds a tied pair holds the earliest occurrence
// (bpe.py:102).  Both collectives are enqueued on the training stream behind the kernels that fill their buffers; up to 256
// merges go out per host round trip (as in swt_bpe_train_run).  A record block that overflows anywhere halts the following
// steps on every rank (the header travels with the block); the host grows the blocks and repeats that one exchange.
//
// RCCL is loaded with dlopen when the first communicator is made, so a process that never shards never touches it.  The
// loop-back communicator (swt_dist_init_local) runs all ranks as trainers of ONE process on one stream and "gathers" with
// device copies: the same runner, the same kernels, no RCCL -- it is how one GPU tests the exchange.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>

#include "swt_common.h"
#include "swt_train.h"

using namespace swt;

namespace {

struct RcclApi {
  void *dl = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int load_rccl() {
  if (g_rccl.dl) return SWT_OK;
  // the copy this process already holds (PyTorch ships its own librccl.so.1) wins: two RCCLs in one process do not mix
  void *dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!dl) dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!dl) dl = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!dl) return fail(SWT_ERR_UNSUPPORTED, "librccl.so.1 not found: %s", dlerror());
  RcclApi a;
  a.dl = dl;
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(dl, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(dl, "ncclCommInitRank"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(dl, "ncclCommDestroy"));
  a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(dl, "ncclAllGather"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(dl, "ncclGetErrorString"));
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString)
    return fail(SWT_ERR_UNSUPPORTED, "librccl.so.1 lacks an expected symbol");
  g_rccl = a;
  return SWT_OK;
}

#define SWT_NCCL(expr)                                                                                        \
  do {                                                                                                        \
    ncclResult_t _r = (expr);                                                                                 \
    if (_r != ncclSuccess) return fail(SWT_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));       \
  } while (0)

}  // namespace

struct swt_dist {
  int rank = 0, world = 1;
  bool local = false;        // loop-back: all ranks are trainers of this process
  ncclComm_t comm = nullptr;
  DevBuf stage_send, stage_recv;
};

namespace {

// One all-gather of `bytes` per rank on the trainers' stream.  RCCL: trainers[0] is this process's rank.  Loop-back:
// trainers[r] is rank r, and every rank's receive buffer gets every rank's send buffer by device copies.
template <class SendOf, class RecvOf>
int gather_dev(swt_dist *d, swt_bpe_trainer **tr, uint32_t n_local, size_t bytes, SendOf send_of, RecvOf recv_of) {
  if (!d->local) {
    SWT_NCCL(g_rccl.AllGather(send_of(tr[0]), recv_of(tr[0]), bytes, ncclChar, d->comm, tr[0]->stream));
    return SWT_OK;
  }
  for (uint32_t dst = 0; dst < n_local; dst++)
    for (uint32_t src = 0; src < n_local; src++)
      SWT_HIP(hipMemcpyAsync(static_cast<uint8_t *>(recv_of(tr[dst])) + (size_t)src * bytes, send_of(tr[src]), bytes,
                             hipMemcpyDeviceToDevice, tr[dst]->stream));
  return SWT_OK;
}

// small host buffers, one per local rank, `bytes` each -> all ranks' buffers in rank order (synchronous)
int gather_host(swt_dist *d, const std::vector<const void *> &send, size_t bytes, std::vector<uint8_t> &out, hipStream_t st) {
  out.assign((size_t)d->world * bytes, 0);
  if (d->local) {
    for (int r = 0; r < d->world; r++) memcpy(out.data() + (size_t)r * bytes, send[r], bytes);
    return SWT_OK;
  }
  int rc;
  if ((rc = d->stage_send.reserve(bytes + 16)) || (rc = d->stage_recv.reserve((size_t)d->world * bytes + 16))) return rc;
  SWT_HIP(hipMemcpyAsync(d->stage_send.p, send[0], bytes, hipMemcpyHostToDevice, st));
  SWT_NCCL(g_rccl.AllGather(d->stage_send.p, d->stage_recv.p, bytes, ncclChar, d->comm, st));
  SWT_HIP(hipMemcpyAsync(out.data(), d->stage_recv.p, (size_t)d->world * bytes, hipMemcpyDeviceToHost, st));
  SWT_HIP(hipStreamSynchronize(st));
  return SWT_OK;
}

int check_group(swt_dist *d, swt_bpe_trainer **tr, uint32_t n_local) {
  if (!d || !tr || !n_local) return fail(SWT_ERR_INVALID, "null argument");
  if (d->local ? n_local != (uint32_t)d->world : n_local != 1)
    return fail(SWT_ERR_INVALID, "%s communicator of %d ranks needs %d local trainer(s), got %u", d->local ? "a loop-back" : "an RCCL", d->world,
                d->local ? d->world : 1, n_local);
  for (uint32_t i = 0; i < n_local; i++) {
    if (!tr[i]) return fail(SWT_ERR_INVALID, "null trainer");
    int rc = tr[i]->ready();  // a handle without histogram / index never reaches a kernel
    if (rc) return rc;
  }
  return SWT_OK;
}

int rank_of(const swt_dist *d, uint32_t i) { return d->local ? (int)i : d->rank; }

}  // namespace

extern "C" {

int swt_dist_unique_id(uint8_t *out128) try {
  if (!out128) return fail(SWT_ERR_INVALID, "null argument");
  int rc = ensure_device();
  if (rc) return rc;
  if ((rc = load_rccl())) return rc;
  ncclUniqueId id;
  SWT_NCCL(g_rccl.GetUniqueId(&id));
  static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, 128);
  return SWT_OK;
} SWT_API_CATCH

int swt_dist_init(int rank, int world, const uint8_t *unique_id128, swt_dist **out) try {
  if (!out || !unique_id128 || world < 1 || rank < 0 || rank >= world) return fail(SWT_ERR_INVALID, "bad rank / world / id");
  int rc = ensure_device();
  if (rc) return rc;
  if ((rc = load_rccl())) return rc;
  auto *d = new swt_dist();
  d->rank = rank;
  d->world = world;
  ncclUniqueId id;
  memcpy(&id, unique_id128, 128);
  ncclResult_t r = g_rccl.CommInitRank(&d->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete d;
    return fail(SWT_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
  }
  *out = d;
  return SWT_OK;
} SWT_API_CATCH

int swt_dist_init_local(int world, swt_dist **out) try {
  if (!out || world < 1 || world > 64) return fail(SWT_ERR_INVALID, "bad world size");
  int rc = ensure_device();
  if (rc) return rc;
  auto *d = new swt_dist();
  d->world = world;
  d->local = true;
  *out = d;
  return SWT_OK;
} SWT_API_CATCH

void swt_dist_destroy(swt_dist *d) try {
  if (!d) return;
  if (d->comm) (void)g_rccl.CommDestroy(d->comm);
  d->stage_send.release();
  d->stage_recv.release();
  delete d;
} SWT_API_CATCH_VOID

int swt_dist_info(const swt_dist *d, int *rank, int *world, int *is_local) try {
  if (!d) return fail(SWT_ERR_INVALID, "null argument");
  if (rank) *rank = d->rank;
  if (world) *world = d->world;
  if (is_local) *is_local = d->local ? 1 : 0;
  return SWT_OK;
} SWT_API_CATCH

// Enter sharded mode: the distinct initial symbols of the whole corpus (bpe.py:75 over all shards) and the one-off
// reduction of the local histograms (every rank adds every other rank's (pair, count) list to its replica).
int swt_bpe_train_shard_begin(swt_bpe_trainer **tr, uint32_t n_local, swt_dist *d, uint32_t *base_out, uint32_t base_cap, uint32_t *n_base) try {
  int rc = check_group(d, tr, n_local);
  if (rc) return rc;
  if (!n_base) return fail(SWT_ERR_INVALID, "null argument");
  hipStream_t st = tr[0]->stream;
  // records per exchange block to start with (it grows on demand); SWT_DIST_BLOCK_RECORDS lets a test start small enough to
  // meet the overflow path
  uint64_t block0 = 4096;
  if (const char *e = getenv("SWT_DIST_BLOCK_RECORDS")) block0 = strtoull(e, nullptr, 10);
  // -- initial symbols: sizes, then the padded lists
  std::vector<uint64_t> sizes(n_local);
  std::vector<const void *> ptrs(n_local);
  for (uint32_t i = 0; i < n_local; i++) { sizes[i] = tr[i]->base_syms.size(); ptrs[i] = &sizes[i]; }
  std::vector<uint8_t> all;
  if ((rc = gather_host(d, ptrs, 8, all, st))) return rc;
  uint64_t mx = 1;
  for (int r = 0; r < d->world; r++) mx = std::max<uint64_t>(mx, reinterpret_cast<uint64_t *>(all.data())[r]);
  std::vector<uint64_t> all_sizes(reinterpret_cast<uint64_t *>(all.data()), reinterpret_cast<uint64_t *>(all.data()) + d->world);
  std::vector<std::vector<uint32_t>> padded(n_local, std::vector<uint32_t>(mx, 0xFFFFFFFFu));
  for (uint32_t i = 0; i < n_local; i++) {
    std::copy(tr[i]->base_syms.begin(), tr[i]->base_syms.end(), padded[i].begin());
    ptrs[i] = padded[i].data();
  }
  if ((rc = gather_host(d, ptrs, mx * 4, all, st))) return rc;
  std::vector<uint32_t> uni;
  for (int r = 0; r < d->world; r++) {
    const uint32_t *p = reinterpret_cast<uint32_t *>(all.data()) + (size_t)r * mx;
    uni.insert(uni.end(), p, p + all_sizes[r]);
  }
  std::sort(uni.begin(), uni.end());
  uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
  *n_base = (uint32_t)uni.size();
  if (base_out) {
    if (base_cap < uni.size()) return fail(SWT_ERR_CAPACITY, "need room for %zu symbols", uni.size());
    std::copy(uni.begin(), uni.end(), base_out);
  }
  // -- local histograms: sizes, then the padded record lists; every rank adds the others'
  for (uint32_t i = 0; i < n_local; i++) {
    tr[i]->n_base_global = uni.size();
    if ((rc = tr[i]->sync_state())) return rc;
    sizes[i] = tr[i]->h_st.n_used;
    ptrs[i] = &sizes[i];
  }
  if ((rc = gather_host(d, ptrs, 8, all, st))) return rc;
  for (int r = 0; r < d->world; r++) all_sizes[r] = reinterpret_cast<uint64_t *>(all.data())[r];
  uint64_t cap = 16;
  for (uint64_t s : all_sizes) cap = std::max(cap, s);
  std::vector<DevBuf> send(n_local), recv(n_local);
  auto release = [&]() { for (auto &b : send) b.release(); for (auto &b : recv) b.release(); };
  for (uint32_t i = 0; i < n_local; i++) {
    if ((rc = send[i].reserve(cap * sizeof(DeltaRec))) || (rc = recv[i].reserve((size_t)d->world * cap * sizeof(DeltaRec)))) { release(); return rc; }
    uint64_t got = 0;
    if ((rc = trainer_export_records(tr[i], send[i].as<DeltaRec>(), cap, &got))) { release(); return rc; }
    if (got > cap) { release(); return fail(SWT_ERR_STATE, "histogram export: %llu live pairs exceed the %llu keys in use", (unsigned long long)got, (unsigned long long)cap); }
    sizes[i] = got;
    ptrs[i] = &sizes[i];
  }
  if ((rc = gather_host(d, ptrs, 8, all, st))) { release(); return rc; }
  for (int r = 0; r < d->world; r++) all_sizes[r] = reinterpret_cast<uint64_t *>(all.data())[r];
  {
    // plain pointer tables for the gather helper
    std::vector<void *> sp(n_local), rp(n_local);
    for (uint32_t i = 0; i < n_local; i++) { sp[i] = send[i].p; rp[i] = recv[i].p; }
    auto idx = [&](swt_bpe_trainer *t) { for (uint32_t i = 0; i < n_local; i++) if (tr[i] == t) return i; return 0u; };
    rc = gather_dev(d, tr, n_local, cap * sizeof(DeltaRec), [&](swt_bpe_trainer *t) { return (const void *)sp[idx(t)]; },
                    [&](swt_bpe_trainer *t) { return rp[idx(t)]; });
    if (rc) { release(); return rc; }
  }
  for (uint32_t i = 0; i < n_local; i++) {
    for (int r = 0; r < d->world; r++) {
      if (r == rank_of(d, i)) continue;
      if ((rc = trainer_add_records(tr[i], recv[i].as<DeltaRec>() + (size_t)r * cap, all_sizes[r]))) { release(); return rc; }
    }
    if ((rc = trainer_enter_sharded(tr[i], (uint32_t)d->world, block0))) { release(); return rc; }
    SWT_HIP(hipStreamSynchronize(tr[i]->stream));
  }
  release();
  return SWT_OK;
} SWT_API_CATCH

// One failed exchange: the largest block any rank wanted, from the headers every rank received.
static int exchange_again(swt_dist *d, swt_bpe_trainer **tr, uint32_t n_local) {
  int rc;
  uint64_t want = 0;
  for (uint32_t i = 0; i < n_local; i++) {
    SWT_HIP(hipStreamSynchronize(tr[i]->stream));
    std::vector<DeltaRec> heads(d->world);
    for (int r = 0; r < d->world; r++)
      SWT_HIP(hipMemcpy(&heads[r], tr[i]->d_blocks_all + (size_t)r * tr[i]->block_cap, sizeof(DeltaRec), hipMemcpyDeviceToHost));
    for (auto &h : heads) want = std::max<uint64_t>(want, h.key);
  }
  const uint64_t cap = 2 * (want + 2);
  for (uint32_t i = 0; i < n_local; i++) {
    if ((rc = trainer_set_block_cap(tr[i], cap))) return rc;  // also re-lists the pending slots from pend[]
    SWT_HIP(hipMemsetAsync(tr[i]->d_halt, 0, 8, tr[i]->stream));
    trainer_enqueue_pack(tr[i]);
  }
  if ((rc = gather_dev(d, tr, n_local, tr[0]->block_cap * sizeof(DeltaRec), [](swt_bpe_trainer *t) { return (const void *)t->d_block; },
                       [](swt_bpe_trainer *t) { return (void *)t->d_blocks_all; })))
    return rc;
  for (uint32_t i = 0; i < n_local; i++) trainer_enqueue_add_blocks(tr[i]);
  for (uint32_t i = 0; i < n_local; i++) {
    unsigned int halt = 0;
    SWT_HIP(hipMemcpyAsync(&halt, tr[i]->d_halt, 4, hipMemcpyDeviceToHost, tr[i]->stream));
    SWT_HIP(hipStreamSynchronize(tr[i]->stream));
    if (halt) return fail(SWT_ERR_STATE, "the delta exchange overflowed again after growing the blocks to %llu records", (unsigned long long)cap);
  }
  return SWT_OK;
}

// One round trip of the generic step (argmax -> tie scan -> 16-byte all-gather -> decide + apply -> pack -> block all-gather ->
// add): one merge per step, k steps.  The fast path falls back to it while a plateau is wider than the candidate list.
static int enqueue_generic_steps(swt_dist *d, swt_bpe_trainer **tr, uint32_t n_local, uint32_t k, uint32_t first_id) {
  int rc;
  for (uint32_t s = 0; s < k; s++) {
    for (uint32_t i = 0; i < n_local; i++) {
      tr[i]->step_no++;
      trainer_enqueue_tie_send(tr[i]);
    }
    if ((rc = gather_dev(d, tr, n_local, 16, [](swt_bpe_trainer *t) { return (const void *)t->d_tie_line; },
                         [](swt_bpe_trainer *t) { return (void *)t->d_tie_all; })))
      return rc;
    for (uint32_t i = 0; i < n_local; i++) trainer_enqueue_decide_apply(tr[i], (uint32_t)rank_of(d, i), s, first_id + s);
    if ((rc = gather_dev(d, tr, n_local, tr[0]->block_cap * sizeof(DeltaRec), [](swt_bpe_trainer *t) { return (const void *)t->d_block; },
                         [](swt_bpe_trainer *t) { return (void *)t->d_blocks_all; })))
      return rc;
    for (uint32_t i = 0; i < n_local; i++) trainer_enqueue_add_blocks(tr[i]);
  }
  return SWT_OK;
}

// The fast path, sharded: per STEP fast_tie + tie_pack -> ONE all-gather of the ranks' tie messages (sizeof(TieMsg) bytes each)
// -> fast_apply_sharded (every rank derives the same batch of up to 16 tied merges) + pack -> ONE all-gather of the record
// blocks -> add + finish.  Up to 256 steps per host round trip, nothing synchronises in between.
static int run_sharded_fast(swt_dist *d, swt_bpe_trainer **tr, uint32_t n_local, uint32_t max_steps, uint32_t first_merged, uint32_t *left,
                            uint32_t *right, uint64_t *count, uint32_t *n_done) {
  int rc;
  std::vector<StepLog> hlog(kMaxRunSteps);
  uint32_t done = 0;
  bool exhausted = false;
  int dry_runs = 0;
  double per_step = 1.0;
  int deferred_rc = 0;  // an error found after a trip: it leaves through the gather below, so that EVERY rank leaves
  while (done < max_steps && !exhausted) {
    const uint32_t remaining = max_steps - done;
    // head room on every local trainer; does any rank need a re-plan (its table grew, its list is long)?  All or none.
    // An error on one rank (a capacity check, an allocation) must not leave the others waiting in a collective: the ranks
    // exchange a status byte with the re-plan bit, and all of them return when one of them failed.
    uint8_t msg[2] = {0, 0};  // [0] wants a re-plan, [1] failed
    std::vector<ShardTrip> trips(n_local);
    int local_rc = deferred_rc;
    for (uint32_t i = 0; i < n_local && !local_rc; i++) {
      if ((local_rc = tr[i]->sync_state()) || (local_rc = tr[i]->check_state())) break;
      if ((local_rc = trainer_fast_room(tr[i], remaining, per_step, &trips[i]))) break;
      msg[0] |= trips[i].replan_first ? 1 : 0;
    }
    msg[1] = local_rc ? 1 : 0;
    if (!d->local) {
      const std::string kept = swt_last_error();  // (the gather may overwrite the text of this rank's own error)
      std::vector<const void *> ptrs(1, msg);
      std::vector<uint8_t> all;
      if ((rc = gather_host(d, ptrs, 2, all, tr[0]->stream))) return local_rc ? local_rc : rc;
      for (int r = 0; r < d->world; r++) {
        msg[0] |= all[2 * r];
        if (all[2 * r + 1] && !local_rc) return fail(SWT_ERR_STATE, "rank %d reported an error in its training state; this rank stops with it", r);
      }
      if (local_rc) return fail(local_rc, "%s", kept.c_str());
    } else if (local_rc) {
      return local_rc;
    }
    const uint8_t want = msg[0];
    for (uint32_t i = 0; i < n_local; i++)
      if ((rc = trainer_fast_plan(tr[i], remaining, per_step, want != 0, first_merged + done, &trips[i]))) return rc;
    const ShardTrip trip = trips[0];
    for (uint32_t i = 1; i < n_local; i++)
      if (trips[i].steps != trip.steps || trips[i].cap != trip.cap || trips[i].fast != trip.fast)
        return fail(SWT_ERR_STATE, "the shards sized a round trip differently (%u/%u steps, %u/%u merges): their replicas have diverged",
                    trips[i].steps, trip.steps, trips[i].cap, trip.cap);
    if (trip.cap > kMaxRunSteps) return fail(SWT_ERR_STATE, "a round trip was sized beyond the step log");
    if (!d->local) {
      // Every rank is about to enqueue `steps` pairs of collectives without looking up again: ranks that sized the trip
      // differently would wait for each other for ever.  They cannot differ while the replicas agree -- so check that they do,
      // from what every rank sees of every rank (all of them take the same way out): one small host gather per round trip.
      unsigned long long mine[4] = {((unsigned long long)trip.steps << 32) | trip.cap, trip.fast ? 1ull : 0ull, tr[0]->theta, tr[0]->h_st.n_cand};
      std::vector<const void *> ptrs(1, mine);
      std::vector<uint8_t> all;
      if ((rc = gather_host(d, ptrs, sizeof mine, all, tr[0]->stream))) return rc;
      for (int r = 0; r < d->world; r++)
        if (memcmp(all.data() + (size_t)r * sizeof mine, all.data(), sizeof mine) != 0) {
          const unsigned long long *o = reinterpret_cast<const unsigned long long *>(all.data() + (size_t)r * sizeof mine);
          const unsigned long long *z = reinterpret_cast<const unsigned long long *>(all.data());
          return fail(SWT_ERR_STATE, "rank %d sized the round trip differently from rank 0 (steps|cap %llx / %llx, fast %llu / %llu, theta %llu / %llu, "
                                     "candidates %llu / %llu): the replicas have diverged", r, o[0], z[0], o[1], z[1], o[2], z[2], o[3], z[3]);
        }
    }
    prof_begin(tr[0]->stream);
    if (trip.fast) {
      for (uint32_t i = 0; i < n_local; i++)
        if ((rc = trainer_fast_begin(tr[i]))) return rc;
      for (uint32_t s = 0; s < trip.steps; s++) {
        for (uint32_t i = 0; i < n_local; i++) {
          tr[i]->step_no++;
          tr[i]->rank = (uint32_t)rank_of(d, i);
          trainer_enqueue_fast_tie(tr[i], trip.cap);
        }
        if ((rc = gather_dev(d, tr, n_local, sizeof(TieMsg), [](swt_bpe_trainer *t) { return (const void *)t->d_tie_msg; },
                             [](swt_bpe_trainer *t) { return (void *)t->d_tie_msgs; })))
          return rc;
        for (uint32_t i = 0; i < n_local; i++) trainer_enqueue_fast_apply(tr[i], first_merged + done, trip.cap);
        if ((rc = gather_dev(d, tr, n_local, tr[0]->block_cap * sizeof(DeltaRec), [](swt_bpe_trainer *t) { return (const void *)t->d_block; },
                             [](swt_bpe_trainer *t) { return (void *)t->d_blocks_all; })))
          return rc;
        for (uint32_t i = 0; i < n_local; i++) trainer_enqueue_add_blocks(tr[i]);
      }
    } else if ((rc = enqueue_generic_steps(d, tr, n_local, trip.steps, first_merged + done))) {
      return rc;
    }
    prof_end(tr[0]->stream);
    SWT_HIP(hipGetLastError());
    unsigned int halt = 0;
    SWT_HIP(hipMemcpyAsync(hlog.data(), tr[0]->d_steplog, trip.cap * sizeof(StepLog), hipMemcpyDeviceToHost, tr[0]->stream));
    SWT_HIP(hipMemcpyAsync(&halt, tr[0]->d_halt, 4, hipMemcpyDeviceToHost, tr[0]->stream));
    for (uint32_t i = 0; i < n_local && !deferred_rc; i++)
      if ((deferred_rc = tr[i]->sync_state()) || (deferred_rc = tr[i]->check_state())) break;
    if (deferred_rc) continue;  // (through the status gather at the top)
    uint32_t good = 0;
    unsigned long long stop = 0;  // 0, 2 no pair left, 3 re-plan
    if (trip.fast) {
      const unsigned long long logged = tr[0]->h_st.run_done[(tr[0]->step_no + 1) & 1u];
      if (logged > trip.cap) { deferred_rc = fail(SWT_ERR_STATE, "the step log overran its round trip"); continue; }
      good = (uint32_t)logged;
      stop = tr[0]->h_st.halt;
      for (uint32_t i = 0; i < good && !deferred_rc; i++)
        if (hlog[i].flag != 0) deferred_rc = fail(SWT_ERR_STATE, "the step log has a hole");
      for (uint32_t i = 1; i < n_local && !deferred_rc; i++)
        if (tr[i]->h_st.run_done[(tr[i]->step_no + 1) & 1u] != logged)
          deferred_rc = fail(SWT_ERR_STATE, "the shards logged different numbers of merges: their replicas have diverged");
      if (deferred_rc) continue;
    } else {
      while (good < trip.cap && hlog[good].flag == 0) good++;
      if (good < trip.cap && hlog[good].flag != 4) stop = hlog[good].flag;
    }
    for (uint32_t g = 0; g < good; g++) {
      left[done] = hlog[g].l;
      right[done] = hlog[g].r;
      count[done] = hlog[g].count;
      for (uint32_t i = 0; i < n_local; i++) tr[i]->trace.push_back(hlog[g]);
      done++;
    }
    for (uint32_t i = 0; i < n_local; i++) {
      swt_bpe_trainer *t = tr[i];
      t->n_applied += good;
      t->since_replan += good;
      if (good) t->h_st.max_count = hlog[good - 1].count;
      if (stop == 3) {
        if (trip.fast && t->theta > 1 && t->cand_built && t->since_replan) {
          t->dry_ratio = 0.9 * (double)t->since_replan / (double)t->cand_built;
          t->dry_ratio = t->dry_ratio < 0.3 ? 0.3 : (t->dry_ratio > 2.0 ? 2.0 : t->dry_ratio);
        }
        t->cand_valid = false;
      }
    }
    if (halt) {  // the last step was applied everywhere but its deltas did not fit: bigger blocks, that exchange again
      if ((rc = exchange_again(d, tr, n_local))) return rc;
    } else if (stop == 3) {
      if (!good && ++dry_runs > 64) return fail(SWT_ERR_STATE, "the candidate list cannot be rebuilt");
    } else if (stop) {
      exhausted = true;  // bpe.py:98-99: no pair left anywhere
    }
    if (trip.fast && good && tr[0]->h_st.run_active) {
      per_step = (double)good / (double)tr[0]->h_st.run_active;
      if (per_step < 1.0) per_step = 1.0;
      if (per_step > (double)kMaxBatch) per_step = (double)kMaxBatch;
    }
    if (good) dry_runs = 0;
  }
  *n_done = done;
  return SWT_OK;
}

// Up to max_steps merges over all shards.  Outputs as swt_bpe_train_run (identical on every rank).
int swt_bpe_train_run_sharded(swt_bpe_trainer **tr, uint32_t n_local, swt_dist *d, uint32_t max_steps, uint32_t first_merged, uint32_t *left,
                              uint32_t *right, uint64_t *count, uint32_t *n_done) try {
  int rc = check_group(d, tr, n_local);
  if (rc) return rc;
  if (!left || !right || !count || !n_done) return fail(SWT_ERR_INVALID, "null argument");
  for (uint32_t i = 0; i < n_local; i++)
    if (!tr[i]->sharded) return fail(SWT_ERR_STATE, "call swt_bpe_train_shard_begin first");
  if ((uint64_t)first_merged + max_steps >= 0xFFFFFFFFull) return fail(SWT_ERR_UNSUPPORTED, "merged symbol ids would reach the reserved id");
  *n_done = 0;
  // SWT_DIST_GENERIC=1: the one-merge-per-step runner of round 2 (kept for comparison and as the fallback form)
  const char *generic = getenv("SWT_DIST_GENERIC");
  if (!generic || !*generic || *generic == '0') return run_sharded_fast(d, tr, n_local, max_steps, first_merged, left, right, count, n_done);
  std::vector<StepLog> hlog(kMaxRunSteps);
  uint32_t done = 0;
  bool exhausted = false;
  int dry_runs = 0;
  while (done < max_steps && !exhausted) {
    uint32_t k = max_steps - done;
    if (k > kRunBatch) k = kRunBatch;
    for (uint32_t i = 0; i < n_local; i++)
      if ((rc = trainer_prepare_batch(tr[i], k, first_merged + done + k))) return rc;
    prof_begin(tr[0]->stream);
    for (uint32_t s = 0; s < k; s++) {
      for (uint32_t i = 0; i < n_local; i++) {
        tr[i]->step_no++;
        trainer_enqueue_tie_send(tr[i]);
      }
      if ((rc = gather_dev(d, tr, n_local, 16, [](swt_bpe_trainer *t) { return (const void *)t->d_tie_line; },
                           [](swt_bpe_trainer *t) { return (void *)t->d_tie_all; })))
        return rc;
      for (uint32_t i = 0; i < n_local; i++) trainer_enqueue_decide_apply(tr[i], (uint32_t)rank_of(d, i), s, first_merged + done + s);
      if ((rc = gather_dev(d, tr, n_local, tr[0]->block_cap * sizeof(DeltaRec), [](swt_bpe_trainer *t) { return (const void *)t->d_block; },
                           [](swt_bpe_trainer *t) { return (void *)t->d_blocks_all; })))
        return rc;
      for (uint32_t i = 0; i < n_local; i++) trainer_enqueue_add_blocks(tr[i]);
    }
    prof_end(tr[0]->stream);
    SWT_HIP(hipGetLastError());
    unsigned int halt = 0;
    SWT_HIP(hipMemcpyAsync(hlog.data(), tr[0]->d_steplog, k * sizeof(StepLog), hipMemcpyDeviceToHost, tr[0]->stream));
    SWT_HIP(hipMemcpyAsync(&halt, tr[0]->d_halt, 4, hipMemcpyDeviceToHost, tr[0]->stream));
    for (uint32_t i = 0; i < n_local; i++)
      if ((rc = tr[i]->sync_state()) || (rc = tr[i]->check_state())) return rc;
    uint32_t good = 0;
    while (good < k && hlog[good].flag == 0) {
      left[done] = hlog[good].l;
      right[done] = hlog[good].r;
      count[done] = hlog[good].count;
      for (uint32_t i = 0; i < n_local; i++) tr[i]->trace.push_back(hlog[good]);
      done++;
      good++;
    }
    for (uint32_t i = 0; i < n_local; i++) {
      tr[i]->n_applied += good;
      if (good) tr[i]->h_st.max_count = hlog[good - 1].count;
    }
    if (halt) {  // the last good merge was applied everywhere but its deltas did not fit: bigger blocks, that exchange again
      if ((rc = exchange_again(d, tr, n_local))) return rc;
    } else if (good < k) {
      if (hlog[good].flag == 3) {
        for (uint32_t i = 0; i < n_local; i++) tr[i]->cand_valid = false;
        if (!good && ++dry_runs > 64) return fail(SWT_ERR_STATE, "the candidate list cannot be rebuilt");
      } else {
        exhausted = true;  // bpe.py:98-99: no pair left anywhere
      }
    }
    if (good) dry_runs = 0;
  }
  *n_done = done;
  return SWT_OK;
} SWT_API_CATCH

}  // extern "C"
