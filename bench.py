#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X subword-tokenizer hot path, one JSON line on stdout.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bpe_encode|wp_encode|bpe_train|wp_train|bpe_train_1g|mixed_encode]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one process per GPU; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment).  Encode shards the corpus
by sentence: every rank encodes its own S85k-shaped shard with a replicated table and there is NO data-path
collective (weak scaling); torch.distributed/RCCL is only the barrier and the MAX over ranks of the time.

Headline (BASELINE.json configs[1]): FastBPE encode of S85k -- the seeded stand-in for the absent
data/train-85k.json -- with the first 8,000 pretrained merges; metric = MB of input
text (UTF-8, 1 MB = 1e6 B) encoded per second, inputs resident in HBM when the timed region starts.
A "step" is one pass of the whole path (plan + encode + scan + gather kernels) over the batch.

  roofline      timed with HIP events on the launch stream inside the library (swt_profile_*); achieved = algorithmic
                bytes / time; algorithmic bytes per call = input bytes + 4 B per output token + 8 B per sentence
                offset.  FastBPE encode is a pipeline of eight short kernels: its line is that
                of the whole call (first kernel start .. last kernel end) with the longest kernel beside it; the
                other workloads time their dominant kernel.
  cpu_baseline  the C oracle (oracle/, a port of the reference's algorithm) on one host core of this box, on the
                same S85k batch.  The oracle is only the checker/baseline here, never the thing measured.
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


def bench_bpe_encode(args, torch, dist, rank, world, local):
    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd import synth, tokenizers

    N.init(local)
    merges = synth.pretrained_merges()[:8000]
    bpe = tokenizers.FastBPE()
    bpe.merges_list = list(merges)
    bpe._build_table()
    sents = synth.sentences(85000, 85000 + rank) if rank else synth.s85k()
    text, off = N.pack_utf8([s.lower() for s in sents])
    n_bytes, n_sent = int(text.size), len(sents)
    d_text, d_off = to_dev(torch, text), to_dev(torch, off.view(np.int64))
    d_out = torch.empty(n_bytes + 64, dtype=torch.int32, device="cuda")
    d_out_off = torch.empty(n_sent + 1, dtype=torch.int64, device="cuda")
    d_ntok = torch.zeros(1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        bpe._table.encode_dev(d_text.data_ptr(), n_bytes, d_off.data_ptr(), n_sent, d_out.data_ptr(), d_out_off.data_ptr(),
                              d_ntok.data_ptr(), 0, stream)

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
    # outside the timed region: the same step with one event pair around ALL kernels of a call (profile level 2)
    N.profile_enable(2)
    for _ in range(min(max(args.steps, 1), 20)):
        step()
    torch.cuda.synchronize()
    call_ms, calls = N.profile_read()
    N.profile_enable(False)
    total_bytes = sum_over_ranks(torch, dist, float(n_bytes))

    # parity check of what was just measured (rank 0, the whole batch) -- the oracle is the checker only
    roof = cpu = None
    if rank == 0:
        from oracle import oracle as O

        orc = O.OracleBPE(merges)
        ids = d_out[:n_tok].cpu().numpy().view(np.uint32)
        offs = d_out_off.cpu().numpy().view(np.uint64)
        oids, ooff = orc.tokenize_batch_ids(sents)
        # time the C call alone (lower() + packing excluded, as on the GPU side)
        blob, boff = O.pack([s.lower() for s in sents])
        out = np.zeros(max(blob.size, 1), dtype=np.uint32)
        oo = np.zeros(n_sent + 1, dtype=np.uint64)
        t1 = time.perf_counter()
        O.lib().orc_bpe_tokenize_batch(orc._h, O._p32(blob), O._p64(boff), n_sent, O._p32(out), O._p64(oo))
        cpu_s = time.perf_counter() - t1
        if not (np.array_equal(ids, oids) and np.array_equal(offs, ooff)):
            raise SystemExit("PARITY FAILURE: device ids differ from the oracle on the benchmark batch")
        # Since the word-level dedup the batch passes through eight short kernels and none of them touches all of the
        # algorithmic bytes, so the roofline line is that of the whole call: algorithmic bytes of the batch over the
        # time from the first kernel's start to the last kernel's end (HIP events on the launch stream).  The longest
        # single kernel (bpe_encode_kernel over the unique words) is reported beside it.
        algo = n_bytes + 4.0 * n_tok + 8.0 * (n_sent + 1)
        per_call_s = call_ms / 1e3 / max(calls, 1)
        achieved = algo / per_call_s / 1e9
        roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic_from_profile("bpe_encode"),
                "kernel": "dedup pipeline: plan, wordref, scan, ureg, bpe_encode (unique words), refcount, scan, refwrite",
                "kernel_us": round(per_call_s * 1e6, 2),
                "algorithmic_bytes_per_launch": int(algo), "launches_timed": int(calls),
                "dominant_kernel": {"name": "bpe_encode_kernel", "us": round(kernel_ms * 1e3 / max(launches, 1), 2),
                                    "launches_timed": int(launches)}}
        cpu = {"value": round(n_bytes / 1e6 / cpu_s, 3), "unit": "MB/s", "cores": 1, "kind": "port",
               "sample": "the whole S85k batch (%.1f MB), one pass of oracle/swt_oracle.c orc_bpe_tokenize_batch" % (n_bytes / 1e6)}
    return {
        "metric": "FastBPE encode throughput (input MB/s, tokens bit-exact)", "value": round(total_bytes * args.steps / 1e6 / elapsed, 1),
        "unit": "MB/s", "ms_per_step": round(elapsed / args.steps * 1e3, 4), "dtype": "u32",
        "config": {"workload": "configs[1]: FastBPE encode, S85k stand-in for train-85k (85,000 sentences, %.2f MB/GPU), "
                               "first 8,000 pretrained merges" % (n_bytes / 1e6),
                   "sentences_per_gpu": n_sent, "bytes_per_gpu": n_bytes, "tokens_per_gpu": n_tok,
                   "parallelism": "corpus-sharded x%d, no collective" % world},
        "roofline": roof, "cpu_baseline": cpu,
    }


def bench_wp_encode(args, torch, dist, rank, world, local):
    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd import synth, tokenizers

    N.init(local)
    wp = tokenizers.FastWP()
    wp.vocab = set(synth.v30k())
    wp._build_trie()
    n_sent = args.sentences or 1000000
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
    # outside the timed region: one event pair around ALL kernels of a call (level 2), then around the split + lookup
    # kernel of the dedup path alone (level 3)
    extra = min(max(args.steps, 1), 10)
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
    total_bytes = sum_over_ranks(torch, dist, float(n_bytes))
    roof = cpu = None
    if rank == 0:
        from oracle import oracle as O

        sub = min(n_sent, 20000)
        orc = O.OracleWP(wp._tokens)
        sents = synth.unpack(text, off, 0, sub)
        t1 = time.perf_counter()
        oids, ooff, ost = orc.tokenize_batch_ids(sents)
        cpu_s = time.perf_counter() - t1
        offs = d_out_off[:sub + 1].cpu().numpy().view(np.uint64)
        ids = d_out[:int(offs[-1])].cpu().numpy().view(np.uint32)
        st = d_status[:sub].cpu().numpy()
        if not (np.array_equal(ids, oids) and np.array_equal(offs, ooff) and np.array_equal(st, ost)):
            raise SystemExit("PARITY FAILURE: device ids differ from the oracle on the benchmark subsample")
        # With the word-level dedup the call is a pipeline (plan, wordref, scan, ureg, wp_encode over the unique
        # chunks, urec, refs-count, scan, refs-write): the roofline line is that of the whole call, the longest kernel
        # (wordref: split + table lookup of every chunk) is reported beside it.  Without dedup (refs == 0: a vocabulary
        # with whitespace inside tokens) the dominant kernel is wp_encode_kernel itself.
        algo = n_bytes + 4.0 * n_tok + 8.0 * (n_sent + 1)
        per_call_s = call_ms / 1e3 / max(calls, 1)
        achieved = algo / per_call_s / 1e9
        dom = ({"name": "wordref_kernel<wp>", "us": round(ref_ms * 1e3 / refs, 2), "launches_timed": int(refs)} if refs else
               {"name": "wp_encode_kernel", "us": round(kernel_ms * 1e3 / max(launches, 1), 2), "launches_timed": int(launches)})
        roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic_from_profile("wp_encode"),
                "kernel": "whole call (dedup pipeline)" if refs else "whole call (plan, wp_encode, scan, gather)",
                "kernel_us": round(per_call_s * 1e6, 2),
                "algorithmic_bytes_per_launch": int(algo), "launches_timed": int(calls), "dominant_kernel": dom,
                "unique_pass": {"name": "wp_encode_kernel", "us": round(kernel_ms * 1e3 / max(launches, 1), 2)}}
        sub_bytes = int(off[sub])
        cpu = {"value": round(sub_bytes / 1e6 / cpu_s, 3), "unit": "MB/s", "cores": 1, "kind": "port",
               "sample": "first %d sentences (%.1f MB) through oracle/swt_oracle.c orc_wp_tokenize_batch" % (sub, sub_bytes / 1e6)}
    return {
        "metric": "FastWP encode throughput (input MB/s, tokens bit-exact)", "value": round(total_bytes * args.steps / 1e6 / elapsed, 1),
        "unit": "MB/s", "ms_per_step": round(elapsed / args.steps * 1e3, 4), "dtype": "u32",
        "config": {"workload": "configs[2]: FastWP failure-link trie encode, V30k vocab, %d synthetic sentences (%.1f MB/GPU)"
                               % (n_sent, n_bytes / 1e6),
                   "sentences_per_gpu": n_sent, "bytes_per_gpu": n_bytes, "tokens_per_gpu": n_tok,
                   "parallelism": "corpus-sharded x%d, no collective" % world},
        "roofline": roof, "cpu_baseline": cpu,
    }


def bench_bpe_train(args, torch, dist, rank, world, local):
    """sec / 1k merges: S85k -> vocab 8,000 (the north-star training target), single GPU."""
    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd import synth, tokenizers

    N.init(local)
    sents = synth.s85k()
    max_vocab = args.max_vocab or 8000
    times = []
    n_merges = 0
    kernel_ms = launches = 0
    info = {}
    merges = []
    for it in range(args.warmup + args.steps):
        if it == args.warmup:
            N.profile_enable(True)
            N.profile_read()
        barrier_sync(torch, dist)
        t0 = time.perf_counter()
        if world == 1:
            tok = tokenizers.FastBPE()
            tok.train(sents, max_vocab)
            merges = list(tok.merges_list)
            info = tok._trainer.info()
            tok.reset()
        else:
            # corpus-sharded: contiguous sentence ranges, pair-histogram deltas all-gathered over RCCL every merge
            from subword_tokenizers_amd.distributed import ShardedBpeTrainer, TorchGroup

            tr = ShardedBpeTrainer.from_corpus(sents, rank, world, TorchGroup(dist, "cuda"))
            merges = [tuple(m) for m in tr.train(max_vocab)]
            info = tr.engine.t.info()
            tr.engine.t.close()
        barrier_sync(torch, dist)
        dt = max_over_ranks(torch, dist, time.perf_counter() - t0)
        if it >= args.warmup:
            times.append(dt)
        n_merges = len(merges)
    kernel_ms, launches = N.profile_read()
    N.profile_enable(False)
    elapsed = sum(times)
    if rank != 0:
        return {"metric": "", "value": 0, "unit": "", "ms_per_step": 0, "dtype": "u32", "config": {}}
    from oracle import oracle as O

    sample = 200
    tr = O.OracleBPETrainer(sents)
    n0, w0 = tr.n_symbols, tr.n_words
    t1 = time.perf_counter()
    tr.run(max_vocab, sample)
    cpu_s = time.perf_counter() - t1
    if tr.merges_list != merges[:sample]:
        raise SystemExit("PARITY FAILURE: device merges differ from the oracle on the first %d merges" % sample)
    # full-rescan formulation of the reference: ~12*N_t + 8*W bytes per merge; N_t <= N_0
    algo = 12.0 * n0 + 8.0 * w0
    per_launch_s = kernel_ms / 1e3 / max(launches, 1)
    achieved = algo / per_launch_s / 1e9 if per_launch_s else 0.0
    sec_per_1k = elapsed / len(times) / max(n_merges, 1) * 1000
    return {
        "metric": "BPE train seconds per 1k merges", "value": round(sec_per_1k, 4), "unit": "s/1k-merges", "higher_is_better": False,
        "ms_per_step": round(elapsed / len(times) * 1e3, 3), "dtype": "u32",
        "config": {"workload": "FastBPE.train on S85k (stand-in for train-85k) to max_vocab=%d: %d merges, %d unique words, "
                               "%d symbols" % (max_vocab, n_merges, w0, n0),
                   "parallelism": "single GPU" if world == 1 else "corpus-sharded x%d, per-merge delta all-gather (RCCL)" % world},
        "scaling": "strong",
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic_from_profile("bpe_train"),
                     "kernel": "apply_kernel", "kernel_us": round(per_launch_s * 1e6, 2),
                     "algorithmic_bytes_per_launch": int(algo), "launches_timed": int(launches),
                     "note": "algorithmic bytes are those of the reference's full-rescan formulation at N_0; the incremental "
                             "design moves fewer"},
        "cpu_baseline": {"value": round(cpu_s / sample * 1000, 3), "unit": "s/1k-merges", "cores": 1, "kind": "port",
                         "sample": "first %d merges of the same run through oracle/swt_oracle.c orc_train_run" % sample},
        "final_symbols": info["n_symbols"],
    }


def bench_mixed_encode(args, torch, dist, rank, world, local):
    """BASELINE configs[4] per GPU: 10 M sentences over 8 GPUs = 1.25 M per GPU, half through FastBPE (8,000 merges) and half
    through FastWP (V30k); corpus-sharded, no collective.  One step = one FastBPE call + one FastWP call."""
    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd import synth, tokenizers

    N.init(local)
    n_each = args.sentences or 625000
    bpe = tokenizers.FastBPE()
    bpe.merges_list = list(synth.pretrained_merges()[:8000])
    bpe._build_table()
    wp = tokenizers.FastWP()
    wp.vocab = set(synth.v30k())
    wp._build_trie()
    b_sents = synth.sentences(n_each, 10000000 + rank)
    b_text, b_off = N.pack_utf8([s.lower() for s in b_sents])
    w_text, w_off = synth.wp_corpus(n_each, seed=20000000 + rank, vocab=synth.v30k())
    bufs = []
    for text, off in ((b_text, b_off), (w_text, w_off)):
        nb = int(text.size)
        bufs.append((to_dev(torch, text), to_dev(torch, off.view(np.int64)), torch.empty(nb + 64, dtype=torch.int32, device="cuda"),
                     torch.empty(n_each + 1, dtype=torch.int64, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda"), nb))
    d_status = torch.empty(n_each + 8, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        t, o, out, oo, nt, nb = bufs[0]
        bpe._table.encode_dev(t.data_ptr(), nb, o.data_ptr(), n_each, out.data_ptr(), oo.data_ptr(), nt.data_ptr(), 0, stream)
        t, o, out, oo, nt, nb = bufs[1]
        wp._trie.encode_dev(t.data_ptr(), nb, o.data_ptr(), n_each, out.data_ptr(), oo.data_ptr(), d_status.data_ptr(), nt.data_ptr(), stream)

    for _ in range(args.warmup):
        step()
    barrier_sync(torch, dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier_sync(torch, dist)
    elapsed = max_over_ranks(torch, dist, time.perf_counter() - t0)
    N.profile_enable(2)
    N.profile_read()
    for _ in range(min(max(args.steps, 1), 5)):
        step()
    torch.cuda.synchronize()
    call_ms, calls = N.profile_read()
    N.profile_enable(False)
    n_bytes = bufs[0][5] + bufs[1][5]
    total_bytes = sum_over_ranks(torch, dist, float(n_bytes))
    roof = cpu = None
    if rank == 0:
        from oracle import oracle as O

        sub = min(n_each, 10000)
        borc, worc = O.OracleBPE(bpe.merges_list), O.OracleWP(wp._tokens)
        t1 = time.perf_counter()
        bo_ids, bo_off = borc.tokenize_batch_ids(b_sents[:sub])
        wo_ids, wo_off, wo_st = worc.tokenize_batch_ids(synth.unpack(w_text, w_off, 0, sub))
        cpu_s = time.perf_counter() - t1
        for (t, o, out, oo, nt, nb), (ids, off) in zip(bufs, ((bo_ids, bo_off), (wo_ids, wo_off))):
            offs = oo[:sub + 1].cpu().numpy().view(np.uint64)
            got = out[:int(offs[-1])].cpu().numpy().view(np.uint32)
            if not (np.array_equal(got, ids) and np.array_equal(offs, off)):
                raise SystemExit("PARITY FAILURE: device ids differ from the oracle on the benchmark subsample")
        if not np.array_equal(d_status[:sub].cpu().numpy(), wo_st):
            raise SystemExit("PARITY FAILURE: FastWP statuses differ from the oracle")
        n_tok = int(bufs[0][4].item()) + int(bufs[1][4].item())
        algo = n_bytes + 4.0 * n_tok + 8.0 * 2 * (n_each + 1)
        per_step_s = call_ms / 1e3 / max(calls // 2, 1)
        achieved = algo / per_step_s / 1e9
        roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": None, "kernel": "one FastBPE call + one FastWP call (both dedup pipelines), first kernel .. last kernel of each",
                "kernel_us": round(per_step_s * 1e6, 2), "algorithmic_bytes_per_launch": int(algo), "launches_timed": int(calls // 2)}
        sub_bytes = int(b_off[sub]) + int(w_off[sub])
        cpu = {"value": round(sub_bytes / 1e6 / cpu_s, 3), "unit": "MB/s", "cores": 1, "kind": "port",
               "sample": "first %d sentences of each half (%.1f MB) through oracle/swt_oracle.c, lower() + packing included" % (sub, sub_bytes / 1e6)}
    return {
        "metric": "mixed FastBPE + FastWP encode throughput (input MB/s, tokens bit-exact)",
        "value": round(total_bytes * args.steps / 1e6 / elapsed, 1), "unit": "MB/s", "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "dtype": "u32",
        "config": {"workload": "configs[4] per GPU: %d sentences through FastBPE (8,000 merges, %.1f MB) + %d through FastWP (V30k, %.1f MB)"
                               % (n_each, bufs[0][5] / 1e6, n_each, bufs[1][5] / 1e6),
                   "sentences_per_gpu": 2 * n_each, "bytes_per_gpu": n_bytes, "parallelism": "corpus-sharded x%d, no collective" % world},
        "roofline": roof, "cpu_baseline": cpu,
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
        tr.close()
        barrier_sync(torch, dist)
        if it >= args.warmup:
            times.append(time.perf_counter() - t0)
    kernel_ms, launches = N.profile_read()
    N.profile_enable(False)
    from oracle import oracle as O

    sample = 20
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
    algo = 12.0 * n0 + 8.0 * w0
    per_launch_s = kernel_ms / 1e3 / max(launches, 1)
    achieved = algo / per_launch_s / 1e9 if per_launch_s else 0.0
    elapsed = sum(times)
    return {
        "metric": "BPE train seconds per 1k merges", "value": round(elapsed / len(times) / max(len(lefts), 1) * 1000, 4),
        "unit": "s/1k-merges", "higher_is_better": False, "ms_per_step": round(elapsed / len(times) * 1e3, 3), "dtype": "u32",
        "config": {"workload": "configs[3] shape: BPE train of %d word types / %d tokens (~%.2f GB of text) to %d merges, "
                               "deduplicated-words-with-frequencies form, %d symbols" % (w0, int(freq.sum()), corpus_bytes / 1e9, len(lefts), n0),
                   "parallelism": "single GPU"},
        "scaling": "strong",
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "kernel": "apply_kernel",
                     "kernel_us": round(per_launch_s * 1e6, 2), "algorithmic_bytes_per_launch": int(algo),
                     "launches_timed": int(launches),
                     "note": "algorithmic bytes are those of the reference's full-rescan formulation at N_0"},
        "cpu_baseline": {"value": round(cpu_s / sample * 1000, 3), "unit": "s/1k-merges", "cores": 1, "kind": "port",
                         "sample": "first %d merges of the same run through oracle/swt_oracle.c (full recount per merge)" % sample},
        "final_symbols": info["n_symbols"],
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
    algo = 16.0 * n0 + 8.0 * w0  # the reference's formulation: pair pass + symbol pass + rewrite read/write, per merge
    per_launch_s = kernel_ms / 1e3 / max(launches, 1)
    achieved = algo / per_launch_s / 1e9 if per_launch_s else 0.0
    elapsed = sum(times)
    return {
        "metric": "WordPiece train seconds per 1k merges", "value": round(elapsed / len(times) / max(len(order), 1) * 1000, 4),
        "unit": "s/1k-merges", "higher_is_better": False, "ms_per_step": round(elapsed / len(times) * 1e3, 3), "dtype": "u32+f64 score",
        "config": {"workload": "NaiveWP.train on S85k to max_vocab=%d: %d initial symbols, %d merges, %d unique words, %d symbols"
                               % (max_vocab, base, len(order), w0, n0), "parallelism": "single GPU"},
        "scaling": "strong",
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "kernel": "apply_kernel",
                     "kernel_us": round(per_launch_s * 1e6, 2), "algorithmic_bytes_per_launch": int(algo),
                     "launches_timed": int(launches),
                     "note": "algorithmic bytes are those of the reference's full-rescan formulation at N_0"},
        "cpu_baseline": {"value": round(cpu_s / sample * 1000, 3), "unit": "s/1k-merges", "cores": 1, "kind": "port",
                         "sample": "first %d merges of the same run through oracle/swt_oracle.c (orc_wptrain_new + orc_train_run)" % sample},
        "final_symbols": info["n_symbols"],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="bpe_encode", choices=["bpe_encode", "wp_encode", "bpe_train", "wp_train", "bpe_train_1g", "mixed_encode"])
    ap.add_argument("--sentences", type=int, default=None, help="wp_encode: sentences per GPU (default 1,000,000); mixed_encode: per half (default 625,000)")
    ap.add_argument("--max-vocab", type=int, default=None, help="bpe_train: target vocabulary (default 8000)")
    ap.add_argument("--types", type=int, default=None, help="bpe_train_1g: word types (default 2,000,000)")
    ap.add_argument("--merges", type=int, default=None, help="bpe_train_1g: merges (default 32,000)")
    args = ap.parse_args()
    defaults = {"bpe_encode": (200, 20), "wp_encode": (20, 3), "bpe_train": (2, 1), "wp_train": (2, 1), "bpe_train_1g": (1, 0), "mixed_encode": (10, 2)}[args.workload]
    if args.steps is None:
        args.steps = defaults[0]
    if args.warmup is None:
        args.warmup = defaults[1]

    torch, dist, rank, world, local = dist_setup(args.gpus)
    fn = {"bpe_encode": bench_bpe_encode, "wp_encode": bench_wp_encode, "bpe_train": bench_bpe_train,
          "wp_train": bench_wp_train, "bpe_train_1g": bench_bpe_train_words, "mixed_encode": bench_mixed_encode}[args.workload]
    res = fn(args, torch, dist, rank, world, local)
    line = {"metric": res.pop("metric"), "value": res.pop("value"), "unit": res.pop("unit"), "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": res.pop("ms_per_step"),
            "higher_is_better": res.pop("higher_is_better", True), "scaling": res.pop("scaling", "weak"), "vs_baseline": None,
            "dtype": res.pop("dtype"), "data": "synthetic", "config": res.pop("config")}
    line.update(res)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
