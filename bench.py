#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X subword-tokenizer hot path, one JSON line on stdout.

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--workload headline|bpe_encode|wp_encode|bpe_train|wp_train|bpe_train_1g|mixed_encode]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one process per GPU; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment).  Encode shards the corpus
by sentence: every rank encodes its own S85k-shaped shard with a replicated table and there is NO data-path
collective (weak scaling); torch.distributed/RCCL is only the barrier and the MAX over ranks of the time.

Default workload `headline` = BOTH halves of BASELINE.json's metric in one line:

  value / unit   FastBPE encode (configs[1]) MB/s of input text (UTF-8, 1 MB = 1e6 B), inputs resident in HBM, first 8,000
                 pretrained merges, measured on TWO seeded stand-ins for the absent data/train-85k.json -- S85k-lex (closed
                 lexicon: train-5K's types + 60 k novel ones) and S85k-open (SURVEY.md section 8d-2 to the letter: every
                 word trigram-sampled, an open vocabulary) -- and reported for the SLOWER of the two; the other one sits in
                 "other_corpus".  A "step" is one pass of the whole encode path over the batch.
  "train"        FastBPE.train S85k-open -> vocab 8,000 (the north-star training target): s_per_1k_merges, the per-merge
                 device time, a roofline against the reference formulation's bytes (12 N_t + 8 W per merge)
                 divided by the WHOLE merge time, and the C oracle beside it.
  "encode_detail" per corpus: distinct-word ratio (what the word-level dedup feeds on), the time with the dedup switched
                 off, and the end-to-end rate from list[str] (host lower/pack + PCIe included; never `value`).
  roofline       HIP events on the launch stream inside the library (swt_profile_*); achieved = algorithmic bytes / time;
                 algorithmic bytes per call = input bytes + 4 B per output token + 8 B per sentence offset.
  cpu_baseline   the C oracle (oracle/, a port of the reference's algorithm) on this box's host: 1 thread, and all host
                 cores (core count stated).  The oracle is only the checker/baseline here, never the thing measured.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def dist_setup(n_gpus):
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if n_gpus > 1 and world != n_gpus:
        raise SystemExit("--gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run)" % (n_gpus, n_gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    return torch, dist, rank, world, local


def barrier_sync(torch, dist):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(torch, dist, seconds):
    from subword_tokenizers_amd.distributed import reduce_scalar

    return reduce_scalar(dist, seconds, "max", "cuda")


def sum_over_ranks(torch, dist, value):
    from subword_tokenizers_amd.distributed import reduce_scalar

    return reduce_scalar(dist, value, "sum", "cuda")


def traffic_from_profile(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary (or None)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None


def to_dev(torch, arr):
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def check_frac(roof):
    """a roofline fraction above 1 means the bytes were not moved in that time: refuse to print it"""
    if roof is not None and roof.get("frac") is not None and roof["frac"] > 1.0:
        raise SystemExit("INVALID roofline: frac %.4f > 1 (%s)" % (roof["frac"], roof.get("kernel")))
    return roof


def encode_corpus_bench(args, torch, dist, rank, N, bpe, sents, merges, name, cpu_leg):
    """K timed steps of the FastBPE device path over one corpus resident in HBM, then (outside the timed region) the whole-call
    event timing, the same call with the word-level dedup off, the end-to-end rate from list[str], and the parity check."""
    text, off = N.pack_utf8([s.lower() for s in sents])
    n_bytes, n_sent = int(text.size), len(sents)
    d_text, d_off = to_dev(torch, text), to_dev(torch, off.view(np.int64))
    d_out = torch.empty(n_bytes + 64, dtype=torch.int32, device="cuda")
    d_out_off = torch.empty(n_sent + 1, dtype=torch.int64, device="cuda")
    d_ntok = torch.zeros(1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step(flags=0):
        bpe._table.encode_dev(d_text.data_ptr(), n_bytes, d_off.data_ptr(), n_sent, d_out.data_ptr(), d_out_off.data_ptr(),
                              d_ntok.data_ptr(), flags, stream)

    for _ in range(args.warmup):
        step()
    barrier_sync(torch, dist)
    n_tok = int(d_ntok.item())
    N.profile_enable(True)
    N.profile_read()
    barrier_sync(torch, dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier_sync(torch, dist)
    elapsed = max_over_ranks(torch, dist, time.perf_counter() - t0)
    kernel_ms, launches = N.profile_read()
    total_bytes = sum_over_ranks(torch, dist, float(n_bytes))
    res = {"corpus": name, "mb_s": total_bytes * args.steps / 1e6 / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
           "n_bytes": n_bytes, "n_sent": n_sent, "n_tok": n_tok, "roofline": None, "cpu": None, "detail": None}
    if args.lean:  # profiling runs (tools/gpu_profile_r02.sh): only the launches of the timed path reach the counters
        N.profile_enable(False)
        return res
    # ---- outside the timed region ----
    extra = min(max(args.steps, 1), 20)
    N.profile_enable(2)  # one event pair around ALL kernels of a call
    for _ in range(extra):
        step()
    torch.cuda.synchronize()
    call_ms, calls = N.profile_read()
    for _ in range(3):
        step(N.BPE_NO_DEDUP)
    torch.cuda.synchronize()
    N.profile_read()
    for _ in range(extra):
        step(N.BPE_NO_DEDUP)
    torch.cuda.synchronize()
    nd_ms, nd_calls = N.profile_read()
    N.profile_enable(False)
    step()
    torch.cuda.synchronize()
    if rank != 0:
        return res
    from oracle import oracle as O

    orc = O.OracleBPE(merges)
    ids = d_out[:n_tok].cpu().numpy().view(np.uint32)
    offs = d_out_off.cpu().numpy().view(np.uint64)
    blob, boff = O.pack([s.lower() for s in sents])
    out = np.zeros(max(blob.size, 1), dtype=np.uint32)
    oo = np.zeros(n_sent + 1, dtype=np.uint64)
    # the C call alone (lower() + packing excluded, as on the GPU side): 1 thread, then all host cores
    t1 = time.perf_counter()
    O.lib().orc_bpe_tokenize_batch(orc._h, O._p32(blob), O._p64(boff), n_sent, O._p32(out), O._p64(oo))
    cpu1_s = time.perf_counter() - t1
    if not (np.array_equal(ids, out[:int(oo[-1])]) and np.array_equal(offs, oo)):
        raise SystemExit("PARITY FAILURE: device ids differ from the oracle on the benchmark batch (%s)" % name)
    if cpu_leg:
        cores = host_cores()
        t1 = time.perf_counter()
        O.lib().orc_bpe_tokenize_batch_mt(orc._h, O._p32(blob), O._p64(boff), n_sent, O._p32(out), O._p64(oo), cores)
        cpun_s = time.perf_counter() - t1
        if not (np.array_equal(ids, out[:int(oo[-1])]) and np.array_equal(offs, oo)):
            raise SystemExit("PARITY FAILURE: the threaded oracle differs (%s)" % name)
        res["cpu"] = {"value": round(n_bytes / 1e6 / cpu1_s, 3), "unit": "MB/s", "cores": 1, "kind": "port",
                      "sample": "the whole %s batch (%.1f MB), one pass of oracle/swt_oracle.c orc_bpe_tokenize_batch" % (name, n_bytes / 1e6),
                      "all_cores": {"value": round(n_bytes / 1e6 / cpun_s, 3), "unit": "MB/s", "cores": cores,
                                    "sample": "the same batch, orc_bpe_tokenize_batch_mt over %d host threads" % cores}}
    # the call is a pipeline of short kernels and none of them touches all of the algorithmic bytes, so the roofline line is
    # that of the whole call: algorithmic bytes of the batch over first kernel start .. last kernel end (HIP events on the
    # launch stream); the longest single kernel (bpe_lane_kernel: over the text on the direct path, over the unique words behind
    # the dedup) is reported beside it
    algo = n_bytes + 4.0 * n_tok + 8.0 * (n_sent + 1)
    per_call_s = call_ms / 1e3 / max(calls, 1)
    achieved = algo / per_call_s / 1e9
    res["roofline"] = check_frac({
        "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": traffic_from_profile("bpe_encode_" + name) or traffic_from_profile("bpe_encode"),
        "kernel": "whole call: the word-dedup pipeline, or the direct path while the text repeats too few of its words (the longest kernel beside it)", "kernel_us": round(per_call_s * 1e6, 2),
        "algorithmic_bytes_per_launch": int(algo), "launches_timed": int(calls),
        "dominant_kernel": {"name": "bpe_lane_kernel", "us": round(kernel_ms * 1e3 / max(launches, 1), 2), "launches_timed": int(launches)}})
    # what the dedup feeds on, exactly: the device's own split + Counter (bpe.py:73-77) of this batch
    tr = N.BpeTrainer.from_text(text, off)
    _ids, _woff, freq = tr.export()
    tr.close()
    n_words, n_distinct = int(freq.astype(np.int64).sum()), int(freq.size)
    # end to end from list[str]: lower + pack on the host/device, PCIe both ways, ids out (never `value`)
    bpe.encode_ids_batch(sents)  # the first call of a size grows the workspaces
    t1 = time.perf_counter()
    e_ids, e_off = bpe.encode_ids_batch(sents)
    e2e_s = time.perf_counter() - t1
    if not (np.array_equal(e_ids, ids) and np.array_equal(e_off, offs)):
        raise SystemExit("PARITY FAILURE: encode_ids_batch differs from the device-resident call (%s)" % name)
    # ... and all the way to the reference's output shape, List[List[str]] (bpe.py:245-249 per text)
    t1 = time.perf_counter()
    toks = bpe.tokenize_batch(sents)
    tb_s = time.perf_counter() - t1
    probe = len(sents) // 2
    if len(toks) != len(sents) or toks[probe] != bpe.decode_ids(e_ids[int(e_off[probe]):int(e_off[probe + 1])]):
        raise SystemExit("PARITY FAILURE: tokenize_batch differs from encode_ids_batch + decode_ids (%s)" % name)
    del toks
    res["detail"] = {"words": n_words, "distinct_words": n_distinct, "distinct_word_ratio": round(n_distinct / max(n_words, 1), 4),
                     "dedup_call_us": round(per_call_s * 1e6, 2), "no_dedup_ms": round(nd_ms / max(nd_calls, 1), 4),
                     "end_to_end_mb_s": round(n_bytes / 1e6 / e2e_s, 1), "tokenize_batch_mb_s": round(n_bytes / 1e6 / tb_s, 1),
                     "end_to_end_note": "FastBPE.encode_ids_batch(list[str]) -> ids, second call of this size: strings -> joined UTF-8 (csrc/swt_pyhost.c), H2D, device split/lower + encode (swt_bpe_encode_joined), ids D2H"}
    return res


def train_roofline(trace, stats, n0, w0, n_final, n_merges, per_merge_s, launches, rescan, world=1, whole_table=False, traffic=None):
    """Bytes THIS implementation must move per merge, from the run's own counters:
      the argmax's input once per step: the candidate list (count + key, 16 B each) -- or the whole pair table for WordPiece,
        whose score has no monotone bound to list candidates by;
      the words of the tie scan's window (4 B per stream slot, holes included, + 8 B of bounds per word), on tied steps;
      the index entries an apply goes through (word id + tag, 8 B);
      per merged occurrence ~360 B: its word (entry, bounds, frequency, claim stamp, 24 staged slots, 2 slot writes = 140 B)
        and five histogram deltas (key probe, count read-modify-write, mirror index + mirror count = 44 B each).
    The reference formulation (a full recount per merge, `rescan` bytes) is what the design avoids moving: the
    merge time against THOSE bytes is an effective rate (it passes the HBM peak on configs[3]) and is reported as such, never
    as the fraction."""
    occurrences = n0 - n_final
    slot_bytes = 4.0 * n0 / max(w0, 1) + 8.0
    if trace is not None and len(trace) and stats is not None:
        step_rows = np.unique(trace[:, 3], return_index=True)[1]  # the merges of one step log the same live-symbol count
        n_steps = int(step_rows.size)
        cand_bytes = 16.0 * (float(stats["table_slots"]) * n_steps if whole_table else float(trace[step_rows, 2].sum()))
        tied_steps = float((trace[step_rows, 1] > 1).sum())
        # the fast path counts the words its windows covered; the other paths scan from the plateau cursor to the first hit
        # (bounded here by the whole stream)
        tie_bytes = slot_bytes * float(stats["tie_words"]) if stats["tie_words"] else (4.0 * n0 + 8.0 * w0) * tied_steps
        idx_bytes = 8.0 * float(stats["entries_scanned"])
    else:  # sharded: rank 0's upper bound (every step tied, its whole shard scanned)
        n_steps = n_merges
        cand_bytes = 16.0 * 2048 * n_merges
        tie_bytes = (4.0 * n0 + 8.0 * w0) / world * n_merges
        idx_bytes = 8.0 * float((stats or {}).get("entries_scanned", 0))
    algo = (cand_bytes + tie_bytes + idx_bytes + 360.0 * occurrences) / max(n_merges, 1)
    achieved = algo / per_merge_s / 1e9 if per_merge_s else 0.0
    return check_frac({
        "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
        "traffic": traffic, "kernel": "one merge, all of its kernels (the launches of a step, shared by the merges it carries)",
        "kernel_us": round(per_merge_s * 1e6, 2), "algorithmic_bytes_per_launch": int(algo), "launches_timed": int(launches),
        "steps": n_steps, "merges_per_step": round(n_merges / max(n_steps, 1), 2),
        "bytes_model": {"argmax_input": int(cand_bytes / max(n_merges, 1)), "tie_window": int(tie_bytes / max(n_merges, 1)),
                        "index_entries": int(idx_bytes / max(n_merges, 1)), "occurrences": int(360.0 * occurrences / max(n_merges, 1))},
        "rescan_formulation": {"bytes_per_merge": int(rescan), "effective_gbs": round(rescan / per_merge_s / 1e9, 1) if per_merge_s else 0.0},
        "note": "bytes = what the incremental design must move per merge (bench.py train_roofline); the path is bound by dependent "
                "round trips (~0.7 us each, about a dozen per launch), not by bandwidth.  rescan_formulation = the reference's "
                "full recount per merge: an effective rate, not traffic"})


def train_bench(args, torch, dist, rank, world, N, sents, max_vocab, name, repeats=2, cpu_sample=200):
    """FastBPE.train (bpe.py:50-112) of `sents` to max_vocab: wall seconds per 1,000 merges over the whole call (host lower +
    pack, H2D, device split/Counter/histogram, the merge loop), the merge loop's device time per merge (HIP events around every
    batch of device-driven merges), the roofline against the reference formulation's bytes and the C oracle's first merges."""
    from subword_tokenizers_amd import tokenizers

    times, loops = [], []
    merges, info, trace, stats = [], {}, None, None
    for it in range(1 + repeats):  # first pass = warm-up (allocations, code objects)
        N.profile_enable(True)
        N.profile_read()
        barrier_sync(torch, dist)
        t0 = time.perf_counter()
        if world == 1 and not os.environ.get("SWT_BENCH_FORCE_SHARDED"):
            tok = tokenizers.FastBPE()
            tok.train(sents, max_vocab)
            t_train = time.perf_counter() - t0  # the reference's call ends here: reading the diagnostics back and freeing are not it
            merges = list(tok.merges_list)
            info = tok._trainer.info()
            trace = tok._trainer.step_trace()
            stats = tok._trainer.stats()
            tok.reset()
        else:
            t_train = None
            from subword_tokenizers_amd.distributed import train_sharded

            merges, info = train_sharded(sents, max_vocab, rank, world, dist)  # C++ runner over RCCL (csrc/swt_dist.hip)
        barrier_sync(torch, dist)
        dt = max_over_ranks(torch, dist, t_train if t_train is not None else time.perf_counter() - t0)
        ms, _n = N.profile_read()
        if it:
            times.append(dt)
            loops.append(ms / 1e3)
    N.profile_enable(False)
    if rank != 0:
        return None
    from oracle import oracle as O

    n_merges = max(len(merges), 1)
    orc = O.OracleBPETrainer(sents)
    n0, w0 = orc.n_symbols, orc.n_words
    t1 = time.perf_counter()
    orc.run(max_vocab, cpu_sample)
    cpu_s = time.perf_counter() - t1
    if orc.merges_list != merges[:cpu_sample]:
        raise SystemExit("PARITY FAILURE: device merges differ from the oracle on the first %d merges (%s)" % (cpu_sample, name))
    n_final = int(info["n_symbols"])
    loop_s = sum(loops) / len(loops)
    per_merge_s = loop_s / n_merges
    wall = sum(times) / len(times)
    roof = train_roofline(trace, stats if stats is not None else info, n0, w0, n_final, n_merges, per_merge_s, n_merges * len(loops),
                          12.0 * (n0 + n_final) / 2.0 + 8.0 * w0, world=world, traffic=traffic_from_profile("bpe_train"))
    return {"metric": "BPE train seconds per 1k merges", "s_per_1k_merges": round(wall / n_merges * 1000, 5), "train_wall_s": round(wall, 4),
            "merge_loop_s": round(loop_s, 4), "us_per_merge_device": round(per_merge_s * 1e6, 2), "n_merges": len(merges),
            "workload": "FastBPE.train on %s to max_vocab=%d: %d merges, %d unique words, %d -> %d symbols" % (name, max_vocab, len(merges), w0, n0, n_final),
            "parallelism": ("single GPU" if world == 1 and not os.environ.get("SWT_BENCH_FORCE_SHARDED") else
                            "corpus-sharded x%d over RCCL: per step one tie-message all-gather + one record-block all-gather (%s form)"
                            % (world, "generic one-merge-per-step" if os.environ.get("SWT_DIST_GENERIC", "0") not in ("", "0") else "fast")),
            "roofline": roof,
            "cpu_baseline": {"value": round(cpu_s / cpu_sample * 1000, 3), "unit": "s/1k-merges", "cores": 1, "kind": "port",
                             "sample": "first %d merges of the same run through oracle/swt_oracle.c orc_train_run (the reference's "
                                       "full recount per merge; its cost per merge falls slowly with N_t)" % cpu_sample}}


EMIT = None  # main() installs the function that formats a result as THE JSON line and writes it (rank 0)


def train_bench_guarded(out_so_far, args, torch, dist, rank, world, N, sents, max_vocab, name, limit_s=240.0):
    """train_bench for the headline line.  At world > 1 the sharded runner makes RCCL calls of its own on every rank: a rank
    that fails or hangs there would take the encode half of the line with it (the other ranks wait in a collective for ever).
    So the block runs under a watchdog: if it raises on this rank or does not finish in limit_s, rank 0 writes the line with
    train = {"error": ...} and every rank leaves at once (os._exit: no collective is entered again) -- with exit code 3:
    a training leg that failed, hung or broke parity is a FAILED run, whatever else the line holds."""
    if world == 1:
        return train_bench(args, torch, dist, rank, world, N, sents, max_vocab, name)
    import threading

    def leave(why):
        sys.stderr.write("bench: sharded training on rank %d: %s -- the line goes out without the train block\n" % (rank, why))
        sys.stderr.flush()
        if rank == 0:
            EMIT(dict(out_so_far, train={"error": why, "parallelism": "corpus-sharded x%d" % world}))
        os._exit(3)

    timer = threading.Timer(limit_s, leave, args=("not finished after %.0f s" % limit_s,))
    timer.daemon = True
    timer.start()
    try:
        train = train_bench(args, torch, dist, rank, world, N, sents, max_vocab, name)
    except BaseException as e:  # incl. SystemExit of a parity failure: say so in the line instead of hanging the other ranks
        timer.cancel()
        leave("%s: %s" % (type(e).__name__, e))
    timer.cancel()
    return train


def bench_headline(args, torch, dist, rank, world, local):
    """Both halves of BASELINE.json's metric: configs[1] FastBPE encode (value) + FastBPE.train s/1k-merges ("train")."""
    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd import synth, tokenizers

    N.init(local)
    merges = synth.pretrained_merges()[:8000]
    bpe = tokenizers.FastBPE()
    bpe.merges_list = list(merges)
    bpe._build_table()
    corpora = [("S85k-lex", synth.sentences(85000, 85000 + rank) if rank else synth.s85k()),
               ("S85k-open", synth.sentences_open(85000, 85000 + rank) if rank else synth.s85k_open())]
    runs = [encode_corpus_bench(args, torch, dist, rank, N, bpe, sents, merges, name, cpu_leg=True) for name, sents in corpora]
    # encode: every rank has its own shard-shaped corpus; training: ONE corpus, which the sharded runner cuts into the ranks'
    # sentence ranges (train_sharded) -- so every rank must hold the same sentences, rank 0's
    train_sents = corpora[1][1] if rank == 0 else synth.s85k_open()
    head, other = sorted(runs, key=lambda r: r["mb_s"])
    out = {
        "metric": "FastBPE encode MB/s (value; tokens bit-exact) + BPE train s/1k-merges (train.s_per_1k_merges)",
        "value": round(head["mb_s"], 1), "unit": "MB/s", "ms_per_step": round(head["ms_per_step"], 4), "dtype": "u32",
        "config": {"workload": "configs[1]: FastBPE encode, %s stand-in for train-85k (85,000 sentences, %.2f MB/GPU), first 8,000 "
                               "pretrained merges -- the slower of S85k-lex / S85k-open" % (head["corpus"], head["n_bytes"] / 1e6),
                   "sentences_per_gpu": head["n_sent"], "bytes_per_gpu": head["n_bytes"], "tokens_per_gpu": head["n_tok"],
                   "parallelism": "corpus-sharded x%d, no collective" % world},
        "roofline": head["roofline"], "cpu_baseline": head["cpu"],
        "encode_detail": {r["corpus"]: r["detail"] for r in runs},
        "other_corpus": {"corpus": other["corpus"], "value": round(other["mb_s"], 1), "unit": "MB/s", "ms_per_step": round(other["ms_per_step"], 4),
                         "bytes_per_gpu": other["n_bytes"], "tokens_per_gpu": other["n_tok"], "roofline": other["roofline"],
                         "cpu_baseline": other["cpu"]},
    }
    # the other half of the north star's encode path, and the mixed shard of configs[4]: blocks of their own beside `train`
    # (their own timed regions, same bracket: barrier + synchronize on both sides, MAX over ranks)
    wp = tokenizers.FastWP()
    wp.vocab = set(synth.v30k())
    wp._build_trie()
    out["wp_encode"] = wp_encode_block(args, torch, dist, rank, N, wp, args.sentences or 1000000, max(args.steps // 5, 5), 3)
    out["mixed_encode"] = mixed_encode_block(args, torch, dist, rank, N, bpe, wp, 625000, max(args.steps // 10, 3), 2)
    if out.get("cpu_baseline") is not None:
        out["cpu_baseline"]["reference_python"] = reference_python_baseline("bpe_encode")
    out["train"] = train_bench_guarded(out, args, torch, dist, rank, world, N, train_sents, args.max_vocab or 8000, "S85k-open")
    if isinstance(out["train"], dict) and out["train"].get("cpu_baseline") is not None:
        out["train"]["cpu_baseline"]["reference_python"] = reference_python_baseline("train_extrapolated")
    return out


def bench_bpe_encode(args, torch, dist, rank, world, local):
    """configs[1] alone, on ONE of the two stand-ins (--corpus lex | open): the encode half of the default line"""
    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd import synth, tokenizers

    N.init(local)
    merges = synth.pretrained_merges()[:8000]
    bpe = tokenizers.FastBPE()
    bpe.merges_list = list(merges)
    bpe._build_table()
    if args.corpus == "lex":
        name, sents = "S85k-lex", (synth.sentences(85000, 85000 + rank) if rank else synth.s85k())
    else:
        name, sents = "S85k-open", (synth.sentences_open(85000, 85000 + rank) if rank else synth.s85k_open())
    r = encode_corpus_bench(args, torch, dist, rank, N, bpe, sents, merges, name, cpu_leg=True)
    return {
        "metric": "FastBPE encode throughput (input MB/s, tokens bit-exact)", "value": round(r["mb_s"], 1), "unit": "MB/s",
        "ms_per_step": round(r["ms_per_step"], 4), "dtype": "u32",
        "config": {"workload": "configs[1]: FastBPE encode, %s stand-in for train-85k (85,000 sentences, %.2f MB/GPU), first 8,000 pretrained merges"
                               % (name, r["n_bytes"] / 1e6),
                   "sentences_per_gpu": r["n_sent"], "bytes_per_gpu": r["n_bytes"], "tokens_per_gpu": r["n_tok"],
                   "parallelism": "corpus-sharded x%d, no collective" % world},
        "roofline": r["roofline"], "cpu_baseline": r["cpu"], "encode_detail": {name: r["detail"]},
    }


def reference_python_baseline(key):
    """The reference's OWN Python on the benchmark corpus, measured in the build container by tools/ref_python_baseline.py (the
    reference cannot travel to the GPU box): a static figure read from profiles/, labelled as such.  None when not recorded."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_reference_python_baseline.json")) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    blk = rec.get(key)
    if not blk:
        return None
    blk = dict(blk)
    blk["where"] = rec.get("host")
    blk["static"] = "not measured by this run: read from profiles/r03_reference_python_baseline.json"
    return blk


def wp_encode_block(args, torch, dist, rank, N, wp, n_sent, steps, warmup, cpu_sents=200000, with_e2e=True):
    """configs[2]: K timed steps of the FastWP device path over `n_sent` synthetic sentences resident in HBM; then, outside the
    timed region, the whole-call and dominant-kernel event timings, the oracle on a bounded subsample (parity + cpu_baseline at
    1 thread and all host cores) and the end-to-end rate from list[str]."""
    from subword_tokenizers_amd import synth

    text, off = synth.wp_corpus(n_sent, seed=1000000 + rank, vocab=synth.v30k())
    n_bytes = int(text.size)
    d_text, d_off = to_dev(torch, text), to_dev(torch, off.view(np.int64))
    d_out = torch.empty(n_bytes + 64, dtype=torch.int32, device="cuda")
    d_out_off = torch.empty(n_sent + 1, dtype=torch.int64, device="cuda")
    d_status = torch.empty(n_sent + 8, dtype=torch.uint8, device="cuda")
    d_ntok = torch.zeros(1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        wp._trie.encode_dev(d_text.data_ptr(), n_bytes, d_off.data_ptr(), n_sent, d_out.data_ptr(), d_out_off.data_ptr(),
                            d_status.data_ptr(), d_ntok.data_ptr(), stream)

    for _ in range(warmup):
        step()
    barrier_sync(torch, dist)
    n_tok = int(d_ntok.item())
    N.profile_enable(True)
    N.profile_read()
    barrier_sync(torch, dist)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier_sync(torch, dist)
    elapsed = max_over_ranks(torch, dist, time.perf_counter() - t0)
    kernel_ms, launches = N.profile_read()
    total_bytes = sum_over_ranks(torch, dist, float(n_bytes))
    res = {"value": round(total_bytes * steps / 1e6 / elapsed, 1), "unit": "MB/s", "ms_per_step": round(elapsed / steps * 1e3, 4), "steps": steps,
           "workload": "configs[2]: FastWP failure-link trie encode, V30k vocab, %d synthetic sentences (%.1f MB/GPU)" % (n_sent, n_bytes / 1e6),
           "sentences_per_gpu": n_sent, "bytes_per_gpu": n_bytes, "tokens_per_gpu": n_tok, "roofline": None, "cpu_baseline": None}
    if args.lean:  # counter passes: only the launches of the timed path
        N.profile_enable(False)
        return res
    # outside the timed region: one event pair around ALL kernels of a call (level 2), then around the split + lookup
    # kernel of the dedup path alone (level 3)
    extra = min(max(steps, 1), 10)
    N.profile_enable(2)
    for _ in range(extra):
        step()
    torch.cuda.synchronize()
    call_ms, calls = N.profile_read()
    N.profile_enable(3)
    for _ in range(extra):
        step()
    torch.cuda.synchronize()
    ref_ms, refs = N.profile_read()
    N.profile_enable(False)
    if rank != 0:
        return res
    from oracle import oracle as O

    sub = min(n_sent, cpu_sents)
    orc = O.OracleWP(wp._tokens)
    sents = synth.unpack(text, off, 0, sub)
    blob, boff = O.pack(sents)  # the corpus is lowercase already; packing is excluded on both sides
    cap = 4 * blob.size + 64 * sub + 64
    out = np.zeros(cap, dtype=np.uint32)
    oo = np.zeros(sub + 1, dtype=np.uint64)
    ost = np.zeros(sub + 1, dtype=np.uint8)
    t1 = time.perf_counter()
    O.lib().orc_wp_tokenize_batch(orc._h, O._p32(blob), O._p64(boff), sub, O._p32(out), cap, O._p64(oo), ost.ctypes.data_as(O._u8p))
    cpu1_s = time.perf_counter() - t1
    offs = d_out_off[:sub + 1].cpu().numpy().view(np.uint64)
    ids = d_out[:int(offs[-1])].cpu().numpy().view(np.uint32)
    st = d_status[:sub].cpu().numpy()
    if not (np.array_equal(ids, out[:int(oo[-1])]) and np.array_equal(offs, oo) and np.array_equal(st, ost[:sub])):
        raise SystemExit("PARITY FAILURE: FastWP device ids differ from the oracle on the benchmark subsample")
    cores = host_cores()
    t1 = time.perf_counter()
    mids, moff, mst = orc.tokenize_packed_mt(blob, boff, cores)
    cpun_s = time.perf_counter() - t1
    if not (np.array_equal(ids, mids) and np.array_equal(offs, moff) and np.array_equal(st, mst)):
        raise SystemExit("PARITY FAILURE: the threaded FastWP oracle differs")
    # With the word-level dedup the call is a pipeline (plan, wordref, scan, ureg, wp_encode over the unique chunks, urec,
    # refs-count, scan, refs-write): the roofline line is that of the whole call, the longest kernel (wordref: split + table
    # lookup of every chunk) is reported beside it.  Without dedup (refs == 0: a vocabulary with whitespace inside tokens)
    # the dominant kernel is wp_encode_kernel itself.
    algo = n_bytes + 4.0 * n_tok + 8.0 * (n_sent + 1)
    per_call_s = call_ms / 1e3 / max(calls, 1)
    achieved = algo / per_call_s / 1e9
    dom = ({"name": "wordref_kernel<wp>", "us": round(ref_ms * 1e3 / refs, 2), "launches_timed": int(refs)} if refs else
           {"name": "wp_encode_kernel", "us": round(kernel_ms * 1e3 / max(launches, 1), 2), "launches_timed": int(launches)})
    res["roofline"] = check_frac({
        "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": traffic_from_profile("wp_encode"),
        "kernel": "whole call (dedup pipeline)" if refs else "whole call (plan, wp_encode, scan, gather)", "kernel_us": round(per_call_s * 1e6, 2),
        "algorithmic_bytes_per_launch": int(algo), "launches_timed": int(calls), "dominant_kernel": dom,
        "unique_pass": {"name": "wp_encode_kernel", "us": round(kernel_ms * 1e3 / max(launches, 1), 2)}})
    sub_bytes = int(off[sub])
    res["cpu_baseline"] = {
        "value": round(sub_bytes / 1e6 / cpu1_s, 3), "unit": "MB/s", "cores": 1, "kind": "port",
        "sample": "first %d sentences (%.1f MB) through oracle/swt_oracle.c orc_wp_tokenize_batch" % (sub, sub_bytes / 1e6),
        "all_cores": {"value": round(sub_bytes / 1e6 / cpun_s, 3), "unit": "MB/s", "cores": cores,
                      "sample": "the same sentences, orc_wp_tokenize_batch_mt over %d host threads" % cores},
        "reference_python": reference_python_baseline("wp_encode")}
    if with_e2e:
        all_sents = synth.unpack(text, off)
        wp.encode_ids_batch(all_sents)  # the first call of a size grows the workspaces
        t1 = time.perf_counter()
        e_ids, e_off, e_st = wp.encode_ids_batch(all_sents)
        e2e_s = time.perf_counter() - t1
        full_off = d_out_off.cpu().numpy().view(np.uint64)
        if not (np.array_equal(e_off, full_off) and np.array_equal(e_ids, d_out[:n_tok].cpu().numpy().view(np.uint32)) and not e_st.any()):
            raise SystemExit("PARITY FAILURE: FastWP.encode_ids_batch differs from the device-resident call")
        res["end_to_end_mb_s"] = round(n_bytes / 1e6 / e2e_s, 1)
        res["end_to_end_note"] = "FastWP.encode_ids_batch(list[str]) -> ids, second call of this size: strings -> joined UTF-8, H2D, device lower + encode (swt_wp_encode_joined), ids D2H"
    return res


def bench_wp_encode(args, torch, dist, rank, world, local):
    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd import synth, tokenizers

    N.init(local)
    wp = tokenizers.FastWP()
    wp.vocab = set(synth.v30k())
    wp._build_trie()
    r = wp_encode_block(args, torch, dist, rank, N, wp, args.sentences or 1000000, args.steps, args.warmup)
    return {
        "metric": "FastWP encode throughput (input MB/s, tokens bit-exact)", "value": r["value"],
        "unit": "MB/s", "ms_per_step": r["ms_per_step"], "dtype": "u32",
        "config": {"workload": r["workload"], "sentences_per_gpu": r["sentences_per_gpu"], "bytes_per_gpu": r["bytes_per_gpu"],
                   "tokens_per_gpu": r["tokens_per_gpu"], "parallelism": "corpus-sharded x%d, no collective" % world},
        "roofline": r["roofline"], "cpu_baseline": r["cpu_baseline"], "end_to_end_mb_s": r.get("end_to_end_mb_s"),
    }


def bench_bpe_train(args, torch, dist, rank, world, local):
    """sec / 1k merges alone: FastBPE.train on S85k-open -> vocab 8,000 (the north-star training target)."""
    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd import synth

    N.init(local)
    sents = synth.s85k() if args.corpus == "lex" else synth.s85k_open()
    name = "S85k-lex" if args.corpus == "lex" else "S85k-open"
    tr = train_bench(args, torch, dist, rank, world, N, sents, args.max_vocab or 8000, name, repeats=max(args.steps, 1))
    if tr is None:
        return {"metric": "", "value": 0, "unit": "", "ms_per_step": 0, "dtype": "u32", "config": {}}
    return {"metric": tr["metric"], "value": tr["s_per_1k_merges"], "unit": "s/1k-merges", "higher_is_better": False,
            "ms_per_step": round(tr["train_wall_s"] * 1e3, 3), "dtype": "u32", "scaling": "strong",
            "config": {"workload": tr["workload"], "parallelism": tr["parallelism"]},
            "roofline": tr["roofline"], "cpu_baseline": tr["cpu_baseline"], "us_per_merge_device": tr["us_per_merge_device"],
            "merge_loop_s": tr["merge_loop_s"]}


def mixed_encode_block(args, torch, dist, rank, N, bpe, wp, n_each, steps, warmup, cpu_sents=100000):
    """BASELINE configs[4] per GPU: 10 M sentences over 8 GPUs = 1.25 M per GPU, half through FastBPE (8,000 merges) and half
    through FastWP (V30k); corpus-sharded, no collective.  One step = one FastBPE call + one FastWP call."""
    from subword_tokenizers_amd import synth

    b_sents = synth.sentences(n_each, 10000000 + rank)
    b_text, b_off = N.pack_utf8([s.lower() for s in b_sents])
    w_text, w_off = synth.wp_corpus(n_each, seed=20000000 + rank, vocab=synth.v30k())
    bufs = []
    for text, off in ((b_text, b_off), (w_text, w_off)):
        nb = int(text.size)
        bufs.append((to_dev(torch, text), to_dev(torch, off.view(np.int64)), torch.empty(nb + 64, dtype=torch.int32, device="cuda"),
                     torch.empty(n_each + 1, dtype=torch.int64, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda"), nb))
    d_status = torch.empty(n_each + 8, dtype=torch.uint8, device="cuda")
    # the two halves are independent calls on handles of their own: each goes out on its OWN stream, so that the short kernels
    # of one pipeline (scans, plans, the passes over the unique words) run beside the long ones of the other
    stream = torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    torch.cuda.synchronize()

    def step():
        t, o, out, oo, nt, nb = bufs[0]
        bpe._table.encode_dev(t.data_ptr(), nb, o.data_ptr(), n_each, out.data_ptr(), oo.data_ptr(), nt.data_ptr(), 0, stream)
        t, o, out, oo, nt, nb = bufs[1]
        wp._trie.encode_dev(t.data_ptr(), nb, o.data_ptr(), n_each, out.data_ptr(), oo.data_ptr(), d_status.data_ptr(), nt.data_ptr(), side.cuda_stream)

    for _ in range(warmup):
        step()
    barrier_sync(torch, dist)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier_sync(torch, dist)
    elapsed = max_over_ranks(torch, dist, time.perf_counter() - t0)
    n_bytes = bufs[0][5] + bufs[1][5]
    total_bytes = sum_over_ranks(torch, dist, float(n_bytes))
    res = {"value": round(total_bytes * steps / 1e6 / elapsed, 1), "unit": "MB/s", "ms_per_step": round(elapsed / steps * 1e3, 4), "steps": steps,
           "workload": "configs[4] per GPU: %d sentences through FastBPE (8,000 merges, %.1f MB) + %d through FastWP (V30k, %.1f MB)"
                       % (n_each, bufs[0][5] / 1e6, n_each, bufs[1][5] / 1e6),
           "sentences_per_gpu": 2 * n_each, "bytes_per_gpu": n_bytes, "roofline": None, "cpu_baseline": None}
    if args.lean:
        return res
    N.profile_enable(2)
    N.profile_read()
    for _ in range(min(max(steps, 1), 5)):
        step()
    torch.cuda.synchronize()
    call_ms, calls = N.profile_read()
    N.profile_enable(False)
    if rank != 0:
        return res
    from oracle import oracle as O

    sub = min(n_each, cpu_sents)
    cores = host_cores()
    borc, worc = O.OracleBPE(bpe.merges_list), O.OracleWP(wp._tokens)
    bblob, bboff = O.pack([s.lower() for s in b_sents[:sub]])
    wblob, wboff = O.pack(synth.unpack(w_text, w_off, 0, sub))
    t1 = time.perf_counter()
    bo_ids, bo_off = borc.tokenize_packed_mt(bblob, bboff, 1)
    wo_ids, wo_off, wo_st = worc.tokenize_packed_mt(wblob, wboff, 1)
    cpu1_s = time.perf_counter() - t1
    t1 = time.perf_counter()
    bm_ids, bm_off = borc.tokenize_packed_mt(bblob, bboff, cores)
    wm_ids, wm_off, wm_st = worc.tokenize_packed_mt(wblob, wboff, cores)
    cpun_s = time.perf_counter() - t1
    if not (np.array_equal(bm_ids, bo_ids) and np.array_equal(wm_ids, wo_ids) and np.array_equal(wm_st, wo_st)):
        raise SystemExit("PARITY FAILURE: the threaded oracle differs from the one-thread oracle")
    for (t, o, out, oo, nt, nb), (ids, off) in zip(bufs, ((bo_ids, bo_off), (wo_ids, wo_off))):
        offs = oo[:sub + 1].cpu().numpy().view(np.uint64)
        got = out[:int(offs[-1])].cpu().numpy().view(np.uint32)
        if not (np.array_equal(got, ids) and np.array_equal(offs, off)):
            raise SystemExit("PARITY FAILURE: device ids differ from the oracle on the benchmark subsample (mixed)")
    if not np.array_equal(d_status[:sub].cpu().numpy(), wo_st):
        raise SystemExit("PARITY FAILURE: FastWP statuses differ from the oracle (mixed)")
    n_tok = int(bufs[0][4].item()) + int(bufs[1][4].item())
    algo = n_bytes + 4.0 * n_tok + 8.0 * 2 * (n_each + 1)
    # the two calls overlap (two streams): what a step costs is the timed region's wall per step; the HIP-event spans of the two
    # calls (first kernel .. last kernel of each, summed) are reported beside it
    per_step_s = elapsed / steps
    achieved = algo / per_step_s / 1e9
    res["tokens_per_gpu"] = n_tok
    res["roofline"] = check_frac({
        "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": traffic_from_profile("mixed_encode"),
        "kernel": "one FastBPE call + one FastWP call (both dedup pipelines) on two streams: wall of the timed steps",
        "kernel_us": round(per_step_s * 1e6, 2), "algorithmic_bytes_per_launch": int(algo), "launches_timed": int(steps),
        "sum_of_call_spans_us": round(call_ms * 1e3 / max(calls // 2, 1), 2)})
    sub_bytes = int(b_off[sub]) + int(w_off[sub])
    res["cpu_baseline"] = {
        "value": round(sub_bytes / 1e6 / cpu1_s, 3), "unit": "MB/s", "cores": 1, "kind": "port",
        "sample": "first %d sentences of each half (%.1f MB) through oracle/swt_oracle.c (both batch encoders, one thread)" % (sub, sub_bytes / 1e6),
        "all_cores": {"value": round(sub_bytes / 1e6 / cpun_s, 3), "unit": "MB/s", "cores": cores,
                      "sample": "the same sentences over %d host threads" % cores}}
    return res


def bench_mixed_encode(args, torch, dist, rank, world, local):
    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd import synth, tokenizers

    N.init(local)
    bpe = tokenizers.FastBPE()
    bpe.merges_list = list(synth.pretrained_merges()[:8000])
    bpe._build_table()
    wp = tokenizers.FastWP()
    wp.vocab = set(synth.v30k())
    wp._build_trie()
    r = mixed_encode_block(args, torch, dist, rank, N, bpe, wp, args.sentences or 625000, args.steps, args.warmup)
    return {
        "metric": "mixed FastBPE + FastWP encode throughput (input MB/s, tokens bit-exact)",
        "value": r["value"], "unit": "MB/s", "ms_per_step": r["ms_per_step"], "dtype": "u32",
        "config": {"workload": r["workload"], "sentences_per_gpu": r["sentences_per_gpu"], "bytes_per_gpu": r["bytes_per_gpu"],
                   "parallelism": "corpus-sharded x%d, no collective" % world},
        "roofline": r["roofline"], "cpu_baseline": r["cpu_baseline"],
    }


def bench_bpe_train_words(args, torch, dist, rank, world, local):
    """BASELINE configs[3]: BPE training of a 1 GiB corpus to 32k merges, in the reference's own formulation (bpe.py:73-81:
    deduplicated word types with frequencies): 2,000,000 synthetic types, Zipf(1.05), ~110 M tokens ~ 2^30 bytes."""
    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd import synth

    if world != 1:
        raise SystemExit("bpe_train_1g: single GPU here (the sharded exchange is exercised by --workload bpe_train --gpus N)")
    N.init(local)
    n_types, tokens = args.types or 2_000_000, None
    sym, off, freq = synth.train_words(n_types, 1073741824, total_tokens=(args.types or 2_000_000) * 55)
    n_merges = args.merges or 32000
    corpus_bytes = int(((np.diff(off.astype(np.int64)) + 1) * freq.astype(np.int64)).sum())  # word + one separator, weighted
    times = []
    lefts = rights = counts = None
    for it in range(args.warmup + args.steps):
        if it == args.warmup:
            N.profile_enable(True)
            N.profile_read()
        barrier_sync(torch, dist)
        t0 = time.perf_counter()
        tr = N.BpeTrainer.from_words(sym, off, freq)
        lefts, rights, counts = tr.run(n_merges, N.SYM_BASE)
        info = tr.info()
        trace, stats = tr.step_trace(), tr.stats()
        tr.close()
        barrier_sync(torch, dist)
        if it >= args.warmup:
            times.append(time.perf_counter() - t0)
    kernel_ms, launches = N.profile_read()
    N.profile_enable(False)
    from oracle import oracle as O

    sample = args.parity_merges or 200
    orc = O.OracleBPETrainer.from_words(sym, off, freq)
    n0, w0 = orc.n_symbols, orc.n_words
    t1 = time.perf_counter()
    orc.run(10 ** 9, sample)
    cpu_s = time.perf_counter() - t1
    ids, cnt = orc.merge_ids()
    if not (np.array_equal(np.asarray(lefts[:sample], dtype=np.uint32), ids[:, 0])
            and np.array_equal(np.asarray(rights[:sample], dtype=np.uint32), ids[:, 1])
            and np.array_equal(np.asarray(counts[:sample], dtype=np.uint64), cnt)):
        raise SystemExit("PARITY FAILURE: device merges differ from the oracle on the first %d merges" % sample)
    n_final = int(info["n_symbols"])
    per_merge_s = kernel_ms / 1e3 / max(len(times), 1) / max(len(lefts), 1)  # the event brackets span whole batches of merge steps
    elapsed = sum(times)
    return {
        "metric": "BPE train seconds per 1k merges", "value": round(elapsed / len(times) / max(len(lefts), 1) * 1000, 4),
        "unit": "s/1k-merges", "higher_is_better": False, "ms_per_step": round(elapsed / len(times) * 1e3, 3), "dtype": "u32",
        "config": {"workload": "configs[3] shape: BPE train of %d word types / %d tokens (~%.2f GB of text) to %d merges, "
                               "deduplicated-words-with-frequencies form, %d -> %d symbols" % (w0, int(freq.sum()), corpus_bytes / 1e9, len(lefts), n0, n_final),
                   "parallelism": "single GPU"},
        "scaling": "strong",
        "roofline": train_roofline(trace, stats, n0, w0, n_final, len(lefts), per_merge_s, len(lefts) * len(times),
                                   12.0 * (n0 + n_final) / 2.0 + 8.0 * w0),  # rescan: N_t averaged over the run (trapezoid)
        "cpu_baseline": {"value": round(cpu_s / sample * 1000, 3), "unit": "s/1k-merges", "cores": 1, "kind": "port",
                         "sample": "first %d merges of the same run through oracle/swt_oracle.c (full recount per merge)" % sample},
        "final_symbols": info["n_symbols"], "parity_merges_checked": sample,
    }


def bench_wp_train(args, torch, dist, rank, world, local):
    """NaiveWP.train (wordpiece.py:29-103) on S85k, `--max-vocab` default = initial symbols + 2000"""
    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd import synth, tokenizers

    if world != 1:
        raise SystemExit("wp_train: single GPU only (the sharded exchange exists for BPE training)")
    N.init(local)
    sents = synth.s85k()
    probe = tokenizers.NaiveWP()
    probe.train(sents, 0)
    base = len(probe.vocab)
    probe.reset()
    max_vocab = args.max_vocab or base + 2000
    times = []
    order = []
    for it in range(args.warmup + args.steps):
        if it == args.warmup:
            N.profile_enable(True)
            N.profile_read()
        barrier_sync(torch, dist)
        t0 = time.perf_counter()
        tok = tokenizers.NaiveWP()
        tok.train(sents, max_vocab)
        order = list(tok._merge_order)
        info = tok._trainer.info()
        trace, stats = tok._trainer.step_trace(), tok._trainer.stats()
        tok.reset()
        barrier_sync(torch, dist)
        if it >= args.warmup:
            times.append(time.perf_counter() - t0)
    kernel_ms, launches = N.profile_read()
    N.profile_enable(False)
    from oracle import oracle as O

    sample = 60
    tr = O.OracleWPTrainer(sents)
    n0, w0 = tr.n_symbols, tr.n_words
    t1 = time.perf_counter()
    tr.run(max_vocab, sample)
    cpu_s = time.perf_counter() - t1
    if [tuple(m) for m in tr.merges_list] != [tuple(m) for m in order[:sample]]:
        raise SystemExit("PARITY FAILURE: device merges differ from the oracle on the first %d merges" % sample)
    n_final = int(info["n_symbols"])
    per_merge_s = kernel_ms / 1e3 / max(len(times), 1) / max(len(order), 1)
    elapsed = sum(times)
    return {
        "metric": "WordPiece train seconds per 1k merges", "value": round(elapsed / len(times) / max(len(order), 1) * 1000, 4),
        "unit": "s/1k-merges", "higher_is_better": False, "ms_per_step": round(elapsed / len(times) * 1e3, 3), "dtype": "u32+f64 score",
        "config": {"workload": "NaiveWP.train on S85k to max_vocab=%d: %d initial symbols, %d merges, %d unique words, %d -> %d symbols"
                               % (max_vocab, base, len(order), w0, n0, n_final), "parallelism": "single GPU"},
        "scaling": "strong",
        # rescan: the reference's pair pass + symbol pass + rewrite read/write, per merge
        "roofline": train_roofline(trace, stats, n0, w0, n_final, len(order), per_merge_s, len(order) * len(times),
                                   16.0 * (n0 + n_final) / 2.0 + 8.0 * w0, whole_table=stats["theta"] == 0),  # theta 1: the list of live pairs
        "cpu_baseline": {"value": round(cpu_s / sample * 1000, 3), "unit": "s/1k-merges", "cores": 1, "kind": "port",
                         "sample": "first %d merges of the same run through oracle/swt_oracle.c (orc_wptrain_new + orc_train_run)" % sample},
        "final_symbols": info["n_symbols"],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="headline", choices=["headline", "bpe_encode", "wp_encode", "bpe_train", "wp_train", "bpe_train_1g", "mixed_encode"])
    ap.add_argument("--sentences", type=int, default=None, help="wp_encode: sentences per GPU (default 1,000,000); mixed_encode: per half (default 625,000)")
    ap.add_argument("--max-vocab", type=int, default=None, help="bpe_train: target vocabulary (default 8000)")
    ap.add_argument("--types", type=int, default=None, help="bpe_train_1g: word types (default 2,000,000)")
    ap.add_argument("--merges", type=int, default=None, help="bpe_train_1g: merges (default 32,000)")
    ap.add_argument("--parity-merges", type=int, default=None, help="bpe_train_1g: merges compared with the oracle (default 200)")
    ap.add_argument("--lean", action="store_true", help="bpe_encode / wp_encode / mixed_encode: timed steps only (no extra legs, no parity check): for counter passes")
    ap.add_argument("--corpus", default="open", choices=["open", "lex"], help="bpe_train / bpe_encode: S85k-open (default) or S85k-lex")
    args = ap.parse_args()
    defaults = {"headline": (100, 10), "bpe_encode": (200, 20), "wp_encode": (20, 3), "bpe_train": (2, 1), "wp_train": (2, 1), "bpe_train_1g": (1, 0), "mixed_encode": (10, 2)}[args.workload]
    if args.steps is None:
        args.steps = defaults[0]
    if args.warmup is None:
        args.warmup = defaults[1]

    # ONE line on stdout, whatever the libraries print (RCCL writes a version banner to fd 1 when a communicator is made):
    # everything else goes to stderr, the JSON line to the stdout this process was given
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    torch, dist, rank, world, local = dist_setup(args.gpus)
    fn = {"headline": bench_headline, "bpe_encode": bench_bpe_encode, "wp_encode": bench_wp_encode, "bpe_train": bench_bpe_train,
          "wp_train": bench_wp_train, "bpe_train_1g": bench_bpe_train_words, "mixed_encode": bench_mixed_encode}[args.workload]
    def emit(res):
        res = dict(res)
        line = {"metric": res.pop("metric"), "value": res.pop("value"), "unit": res.pop("unit"), "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": res.pop("ms_per_step"),
                "higher_is_better": res.pop("higher_is_better", True), "scaling": res.pop("scaling", "weak"), "vs_baseline": None,
                "dtype": res.pop("dtype"), "data": "synthetic", "config": res.pop("config")}
        line.update(res)
        print(json.dumps(line), file=real_stdout, flush=True)

    global EMIT
    EMIT = emit
    res = fn(args, torch, dist, rank, world, local)
    if rank == 0:
        emit(res)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
