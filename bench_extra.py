#!/usr/bin/env python3
"""Secondary measurements (NOT the driver's contract line; that is bench.py): the other
BASELINE.json configurations and a few diagnostics, one JSON object per line.

  python bench_extra.py flat10m | k120 | pcie | latency | refshape | encoder | pipeline [--texts N] | kmeans [--rows N] | ivfpq [--rows N]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402

import bench          # noqa: E402
import shodh_memory_amd as S   # noqa: E402
from shodh_memory_amd import _lib as L   # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps


def flat(rows_n, nq, k, steps=20, scan=L.SCAN_AUTO, tag="flat"):
    rows = bench.synth_rows(torch, rows_n, 384, 1, dev)
    q = bench.synth_rows(torch, nq, 384, 2, dev)
    idx = S.VamanaIndex(S.VamanaConfig(dimension=384, scan_mode=scan, reserve_rows=rows_n))
    idx.build(rows)
    del rows
    out = (torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq,), dtype=torch.int32, device=dev))
    idx.kernel_timing(True)
    dt = timed(lambda: idx.search_batch_device(q, k, out=out), steps)
    km, kmin, kn = idx.kernel_timing(True)
    alg = rows_n * 384 * 4
    print(json.dumps({"bench": tag, "rows": rows_n, "nq": nq, "k": k, "ms_per_step": round(dt * 1e3, 4), "qps": round(nq / dt, 1),
                      "scan_kernel_us_mean": round(km, 1), "scan_kernel_algorithmic_GBs": round(alg * ((nq + 255) // 256 if nq > 4 else 1) / (km * 1e-6) / 1e9 / max(1, (nq + 255) // 256 if nq > 4 else 1), 1),
                      "hbm_frac_of_8TBs": round(alg / (km * 1e-6) / 8e12, 4)}), flush=True)
    return idx, q


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("flat10m", "all"):
        flat(10_000_000, 256, 10, steps=10, tag="flat 10M x 384, batch 256, top-10")
        flat(10_000_000, 1024, 10, steps=5, tag="flat 10M x 384, batch 1024, top-10")
        flat(10_000_000, 1, 10, steps=20, tag="flat 10M x 384, single query")
    if what in ("k120", "all"):
        flat(1_000_000, 256, 120, steps=20, tag="flat 1M, batch 256, k=120 (the index-level k of a top-10 recall, retrieval.rs:913-918)")
        flat(1_000_000, 64, 10, steps=20, tag="flat 1M, batch 64, top-10")
        flat(1_000_000, 1024, 10, steps=10, tag="flat 1M, batch 1024, top-10")
    if what in ("latency", "all"):
        for mode, name in ((L.SCAN_EXACT, "exact-order f32 scan"), (L.SCAN_MFMA, "fp16 MFMA pre-scan + re-score")):
            idx, q = flat(1_000_000, 1, 10, steps=50, scan=mode, tag="flat 1M single query: " + name)
        flat(1_000_000, 256, 10, steps=5, scan=L.SCAN_EXACT, tag="flat 1M, batch 256, top-10, exact-order f32 scan only (SHODH_SCAN_EXACT; also what an unquantisable corpus falls back to)")
        flat(10_000, 1, 10, steps=200, tag="flat 10k x 384 single query (configs[0] size; below 16384 rows the exact-order scan is the path)")
    if what in ("refshape",):
        # the reference's own criterion shapes (benches/memory_benchmarks.rs:156-181, :228-253): populate a small store, then time
        # single recalls at k in {1, 5, 10, 25, 50}; here the vector leg only (one query vector, one synchronised call at a time)
        for n_rows in (100, 1000, 10_000, 100_000):
            rows = bench.synth_rows(torch, n_rows, 384, 1, dev)
            idx = S.VamanaIndex(S.VamanaConfig(dimension=384, reserve_rows=n_rows))
            idx.build(rows)
            q1 = bench.synth_rows(torch, 1, 384, 2, dev)
            res = {}
            for k in (1, 5, 10, 25, 50):
                o = (torch.empty((1, k), dtype=torch.int32, device=dev), torch.empty((1, k), dtype=torch.float32, device=dev), torch.empty((1,), dtype=torch.int32, device=dev))
                ts = []
                for i in range(120):
                    torch.cuda.synchronize(); a = time.perf_counter()
                    idx.search_batch_device(q1, k, out=o)
                    torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
                ts = sorted(ts[20:])
                res["k=%d" % k] = round(ts[len(ts) // 2] * 1e3, 4)
            print(json.dumps({"bench": "reference bench shape: %d memories, single recall (vector leg), p50 ms per call incl. launch + synchronise" % n_rows, "p50_ms": res}), flush=True)
    if what in ("pcie", "all"):
        rows = bench.synth_rows(torch, 1_000_000, 384, 1, dev)
        idx = S.VamanaIndex(S.VamanaConfig(dimension=384, reserve_rows=1_000_000))
        idx.build(rows)
        qh = bench.synth_rows(torch, 256, 384, 2, dev).cpu().numpy()
        for _ in range(3):
            idx.search_batch(qh, 10)
        t = time.perf_counter()
        for _ in range(20):
            idx.search_batch(qh, 10)
        dt = (time.perf_counter() - t) / 20
        print(json.dumps({"bench": "flat 1M, batch 256, top-10, HOST pointers (PCIe H2D queries + D2H results + sync per call)",
                          "ms_per_step": round(dt * 1e3, 4), "qps": round(256 / dt, 1), "stage_us": idx.stage_timings_us()}), flush=True)
    if what in ("encoder", "all"):
        from tests import bert_ref
        for dtype, name in ((L.DTYPE_BF16, "bf16"), (L.DTYPE_FP32, "fp32")):
            e = S.MiniLMEmbedder(synthetic_seed=1234, dtype=dtype)
            b = 4096 if dtype == L.DTYPE_BF16 else 512
            ids, mask = bert_ref.synth_batch(b, 256, seed=5)
            d_ids = ids.to(torch.int32).cuda().contiguous(); d_mask = mask.to(torch.uint8).cuda().contiguous()
            out = torch.empty((b, 384), dtype=torch.float32, device=dev)
            dt = timed(lambda: e.encode_ids_device(d_ids, d_mask, out=out), 5, warmup=2)
            tokens = int(mask.sum())
            lens = mask.sum(1).double()
            flop = float(tokens * 2 * (3 * 384 * 384 + 384 * 384 + 2 * 384 * 1536) * 6 + (lens ** 2).sum() * 4 * 384 * 6)
            print(json.dumps({"bench": "MiniLM-L6 encode, %s, batch %d texts, lengths U[8,128] (real tokens only)" % (name, b),
                              "ms_per_batch": round(dt * 1e3, 3), "texts_per_s": round(b / dt, 1), "tokens_per_s": round(tokens / dt, 1),
                              "tflops": round(flop / dt / 1e12, 2), "device_us": e.stage_timings_us()}), flush=True)
    if what in ("pipeline",):
        # BASELINE.json configs[2]: on-GPU embed + insert + recall end to end (benches/pipeline_benchmarks.rs shape):
        # token ids (synthetic, lengths U[8,128], [CLS] .. [SEP], padded to 256) -> MiniLM bf16 -> add_vectors -> recall
        n_texts = int(sys.argv[sys.argv.index("--texts") + 1]) if "--texts" in sys.argv else 1_000_000
        bsz, ML = 4096, 256
        e = S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_BF16)
        idx = S.VamanaIndex(S.VamanaConfig(dimension=384, reserve_rows=n_texts))
        g = torch.Generator(device=dev).manual_seed(11)
        pos = torch.arange(ML, device=dev)[None, :]

        def synth_ids(b):
            lens = torch.randint(8, 129, (b,), generator=g, device=dev)
            ids = torch.randint(1000, 30522, (b, ML), generator=g, device=dev, dtype=torch.int32)
            mask = (pos < lens[:, None])
            ids = torch.where(mask, ids, torch.zeros_like(ids))
            ids[:, 0] = 101
            ids[torch.arange(b, device=dev), lens - 1] = 102
            return ids.contiguous(), mask.to(torch.uint8).contiguous(), int(lens.sum())

        emb = torch.empty((bsz, 384), dtype=torch.float32, device=dev)
        ids, mask, _ = synth_ids(bsz)
        e.encode_ids_device(ids, mask, out=emb); torch.cuda.synchronize()          # warm-up (weights, workspaces)
        done, tokens, t_enc, t_add = 0, 0, 0.0, 0.0
        t0 = time.perf_counter()
        while done < n_texts:
            b = min(bsz, n_texts - done)
            ids, mask, nt = synth_ids(b)
            torch.cuda.synchronize(); a = time.perf_counter()
            e.encode_ids_device(ids, mask, out=emb[:b])
            torch.cuda.synchronize(); c = time.perf_counter()
            idx.add_vectors(emb[:b])
            torch.cuda.synchronize(); d = time.perf_counter()
            t_enc += c - a; t_add += d - c; tokens += nt; done += b
        t_ingest = time.perf_counter() - t0
        # recall: 256 query texts -> encode -> top-10
        qids, qmask, _ = synth_ids(256)
        qemb = torch.empty((256, 384), dtype=torch.float32, device=dev)
        out = (torch.empty((256, 10), dtype=torch.int32, device=dev), torch.empty((256, 10), dtype=torch.float32, device=dev), torch.empty((256,), dtype=torch.int32, device=dev))

        def recall():
            e.encode_ids_device(qids, qmask, out=qemb)
            idx.search_batch_device(qemb, 10, out=out)
        dt = timed(recall, 50, warmup=20)
        dt_s = timed(lambda: idx.search_batch_device(qemb, 10, out=out), 50, warmup=5)
        idx.search_batch(qemb.cpu().numpy(), 10)      # one host-pointer call: it reads the pre-scan counters back (scan_stats)
        print(json.dumps({"bench": "pipeline (configs[2]): %d synthetic texts, MiniLM-L6 bf16 encode -> add_vectors -> recall top-10, batch 256 query texts" % n_texts,
                          "ingest_s": round(t_ingest, 3), "ingest_texts_per_s": round(n_texts / t_ingest, 1), "encode_s": round(t_enc, 3), "insert_s": round(t_add, 3),
                          "tokens": tokens, "index_rows": idx.len(), "recall_ms_per_batch_incl_query_encode": round(dt * 1e3, 4),
                          "recall_qps_incl_query_encode": round(256 / dt, 1), "search_only_ms_per_batch": round(dt_s * 1e3, 4), "scan_stats": idx.scan_stats()}), flush=True)
    if what in ("kmeans",):
        n = int(sys.argv[sys.argv.index("--rows") + 1]) if "--rows" in sys.argv else 1_000_000
        rows = bench.synth_rows(torch, n, 384, 1, dev).cpu().numpy()
        idx = S.SpannIndex(384, num_probes=32)
        P = idx.compute_partitions(n)
        t = time.perf_counter()
        rng = np.random.default_rng(5)                       # the initial shuffles (thread_rng in the reference): host work, timed apart
        ip = rng.permutation(n).astype(np.uint32)
        pp = [rng.permutation(n).astype(np.uint32) for _ in range(48)]
        t_shuffle = time.perf_counter() - t
        for ivf_it, pq_it in ((2, 1), (6, 3), (25, 20)):     # (25, 20) = the reference's default iteration counts (spann.rs:110, pq.rs:55)
            t = time.perf_counter()
            idx.train(rows, P, ivf_it, pq_it, ivf_perm=ip, pq_perms=pp)
            dt = time.perf_counter() - t
            print(json.dumps({"bench": "device k-means (SpannIndex::build training), %d rows x 384, P = %d, %d IVF + %d PQ iterations (incl. H2D of the rows and shuffles)" % (n, P, ivf_it, pq_it),
                              "seconds": round(dt, 3), "host_shuffles_seconds": round(t_shuffle, 3)}), flush=True)
    if what in ("ivfpq", "all"):
        n = int(sys.argv[sys.argv.index("--rows") + 1]) if "--rows" in sys.argv else 10_000_000
        P, nprobe, nq, k = 4096, 32, 1024, 10
        rows = bench.synth_rows(torch, n, 384, 1, dev)
        g = torch.Generator(device=dev).manual_seed(3)
        cent = rows[torch.randperm(n, generator=g, device=dev)[:P]].contiguous()
        # a few Lloyd steps in torch (any trained state is valid input; parity is defined given the state)
        sample = rows[torch.randperm(n, generator=g, device=dev)[:min(n, 400_000)]]
        for _ in range(4):
            a = (sample @ cent.T).argmax(1)
            cent = torch.zeros_like(cent).index_add_(0, a, sample)
            cnt = torch.bincount(a, minlength=P).clamp(min=1)[:, None]
            cent = torch.nn.functional.normalize(cent / cnt, dim=1)
        sub = sample[:65536].view(-1, 48, 8)
        codebook = torch.stack([sub[torch.randperm(sub.shape[0], generator=g, device=dev)[:256], m] for m in range(48)]).contiguous()
        idx = S.SpannIndex(384, num_probes=nprobe)
        idx.set_trained_state(cent.cpu().numpy(), codebook.cpu().numpy(), np.zeros(P + 1, np.uint64), np.zeros(0, np.uint32), np.zeros((0, 48), np.uint8))
        t = time.perf_counter()
        h_rows = rows.cpu().numpy()
        assign, codes = idx.encode(h_rows)
        t_enc = time.perf_counter() - t
        order = np.argsort(assign, kind="stable")
        off = np.zeros(P + 1, np.uint64); off[1:] = np.cumsum(np.bincount(assign, minlength=P))
        idx.set_trained_state(cent.cpu().numpy(), codebook.cpu().numpy(), off, order.astype(np.uint32), codes[order])
        qh = bench.synth_rows(torch, nq, 384, 2, dev).cpu().numpy()
        for _ in range(2):
            idx.search_batch(qh, k)
        t = time.perf_counter()
        for _ in range(5):
            ids, dist, counts = idx.search_batch(qh, k)
        dt = (time.perf_counter() - t) / 5
        lens = np.diff(off.astype(np.int64))
        print(json.dumps({"bench": "IVF-PQ %d rows, nlist %d, nprobe %d, batch %d, top-%d (host API)" % (n, P, nprobe, nq, k),
                          "ms_per_batch": round(dt * 1e3, 3), "qps": round(nq / dt, 1), "encode_s_for_all_rows": round(t_enc, 2),
                          "list_len_mean": float(lens.mean()), "list_len_max": int(lens.max()),
                          "algorithmic_bytes_per_query": float(nprobe * lens.mean() * 52)}), flush=True)


if __name__ == "__main__":
    main()
