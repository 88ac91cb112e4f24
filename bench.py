#!/usr/bin/env python3
"""bench.py -- recall throughput of the flat index on MI355X (BASELINE.json metric).

A step = one recall pass: one batch of `--nq` (256) query vectors against the whole corpus
(`--rows`, 1M x 384 f32 = BASELINE.json configs[1]), top-`--k` (10), inputs resident in HBM.
`value` = queries/s of the whole job.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: the corpus is row-sharded over the ranks (shodh_memory_amd/distributed.py): every rank scans
its shard for the same query batch, the per-shard top-k are all-gathered over RCCL and merged.
--scaling weak (default) holds --rows PER GPU, so the corpus grows with N (BASELINE.json configs[4] shape, 10M x 8:
the ideal is a constant queries/s while the corpus grows N-fold); --scaling strong keeps the total corpus at --rows.

Besides the driver's fields the JSON line carries
  roofline     -- dominant kernel (MFMA emit scan): algorithmic bytes (live rows x dim x 4) / its mean
                  duration from HIP events recorded by the library on the launch stream
  cpu_baseline -- the CPU oracle's restatement of VamanaIndex::brute_force_search timed on this
                  host's cores on a bounded sample (rank 0, N = 1 only)
  latency_*    -- single-query (nq = 1) recall latency, p50 over 50 calls
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20260926
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16


def synth_rows(torch, n, dim, seed, device):
    """SURVEY 8d corpus, generated in HBM: half correlated rows (shared direction + one strong
    component + noise), half i.i.d. unit rows, shuffled."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn((n, dim), generator=g, device=device, dtype=torch.float32)
    half = n // 2
    x[:half] = x[:half] * 0.3 + 1.0
    idx = torch.arange(half, device=device)
    x[idx, idx % dim] += 0.5 * dim ** 0.5
    x = torch.nn.functional.normalize(x, dim=1)
    perm = torch.randperm(n, generator=g, device=device)
    return x[perm].contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--nq", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--scan", choices=["auto", "exact", "mfma"], default="auto")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="weak",
                    help="N > 1: weak = --rows per GPU (the corpus grows with N, BASELINE configs[4] shape; value = queries x shards per second, the "
                         "answered-query rate is config.end_to_end_queries_per_s, DESIGN.md section 5), strong = --rows in total (value = answered queries/s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-query latency section (keeps a rocprof kernel summary to the batch launches)")
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed recall steps for this long BEFORE the contract's warm-up: the GPU idles at ~100 MHz while the corpus is "
                         "generated and needs a few hundred ms of load to reach its sustained clocks")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample")
    args = ap.parse_args()

    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    backend = os.environ.get("SHODH_BENCH_BACKEND", "nccl")    # "gloo" + one GPU: a functional check of the N > 1 path (not a measurement)
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import shodh_memory_amd as S
    from shodh_memory_amd import _lib as L
    from shodh_memory_amd.distributed import ShardedFlatIndex, shard_range

    n_total = args.rows * (world if args.scaling == "weak" else 1)
    scan_mode = {"auto": L.SCAN_AUTO, "exact": L.SCAN_EXACT, "mfma": L.SCAN_MFMA}[args.scan]
    sh = ShardedFlatIndex(dim=args.dim, n_total=n_total, scan_mode=scan_mode, device=local_rank) if world > 1 else None
    lo, hi = shard_range(n_total, world, rank)
    rows = synth_rows(torch, hi - lo, args.dim, SEED + 1000 * rank, dev)
    queries = synth_rows(torch, args.nq, args.dim, SEED + 1, dev)      # identical on every rank
    if world > 1:
        sh.build_local(rows)
        index = sh.index
        step = lambda: sh.search_batch_device(queries, args.k)        # noqa: E731
    else:
        index = S.VamanaIndex(S.VamanaConfig(dimension=args.dim, scan_mode=scan_mode, device=local_rank, reserve_rows=hi - lo))
        index.build(rows)
        out = (torch.empty((args.nq, args.k), dtype=torch.int32, device=dev), torch.empty((args.nq, args.k), dtype=torch.float32, device=dev),
               torch.empty((args.nq,), dtype=torch.int32, device=dev))
        step = lambda: index.search_batch_device(queries, args.k, out=out)   # noqa: E731

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(int(args.prewarm_ms * 3)):      # ~0.3 ms per step; a fixed count, so that every rank issues the same collectives
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        res = step()
    torch.cuda.synchronize()
    index.kernel_timing(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_mean_us, kern_min_us, kern_n = index.kernel_timing(reset=True)

    # sanity: results are well-formed (full parity is the job of tests/)
    ids = res[0].cpu().numpy().view(np.uint32)
    dd = res[1].cpu().numpy()
    assert (np.diff(dd, axis=1) >= 0).all() and (ids != 0xFFFFFFFF).all()

    qps = args.nq * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    rows_local = hi - lo
    alg_bytes = rows_local * args.dim * 4                 # corpus read once per batch at f32
    # HBM traffic of the dominant kernel from the PMC passes kept under profiles/ (FETCH_SIZE doubled per the
    # gfx950 correction + WRITE_SIZE; collected with rocprofv3 --pmc in separate runs, not live)
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if (pm["config"]["rows"], pm["config"]["dim"], pm["config"]["nq"]) == (rows_local, args.dim, args.nq) and args.scan != "exact":
            traffic = [v["traffic_bytes_per_launch"] for kk, v in pm["kernels"].items() if "mfma_scan_kernel<1" in kk][0]
    except Exception:
        traffic = None
    roof = None
    if kern_n:
        ach = alg_bytes / (kern_mean_us * 1e-6) / 1e9
        flops = 2.0 * rows_local * args.dim * 256         # the pre-scan always multiplies a 256-query pass
        roof = {"bound": "hbm", "kernel": "mfma_scan_kernel<EMIT>" if args.scan != "exact" else "flat_exact_kernel",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": traffic, "launch_us_mean": round(kern_mean_us, 2), "launch_us_min": round(kern_min_us, 2),
                "launches_timed": kern_n, "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_moved_fp16_shadow_per_launch": rows_local * args.dim * 2,
                "mfma_tflops": round(flops / (kern_mean_us * 1e-6) / 1e12, 1), "mfma_frac": round(flops / (kern_mean_us * 1e-6) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)}

    # single-query latency (recall(k) on one query: the reference's own bench shape, benches/memory_benchmarks.rs:228-253)
    lat = None
    if rank == 0 and world == 1 and not args.no_latency:
        q1 = queries[:1].contiguous()
        o1 = (torch.empty((1, args.k), dtype=torch.int32, device=dev), torch.empty((1, args.k), dtype=torch.float32, device=dev),
              torch.empty((1,), dtype=torch.int32, device=dev))
        ts = []
        for i in range(60):
            torch.cuda.synchronize()
            a = time.perf_counter()
            index.search_batch_device(q1, args.k, out=o1)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - a)
        ts = sorted(ts[10:])
        lat = {"nq": 1, "p50_ms": round(ts[len(ts) // 2] * 1e3, 4), "p95_ms": round(ts[int(len(ts) * 0.95)] * 1e3, 4)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O      # CPU baseline leg only: the oracle is the thing being timed here
        cores = os.cpu_count() or 1
        h_rows = rows.cpu().numpy()
        h_q = queries.cpu().numpy()
        # calibrate with one query on one thread, then size the sample to the budget
        s1, _, _ = O.bench_brute_force(h_rows, h_q[:1], args.k, 0, True, 1)
        nq_mt = int(max(cores, min(args.nq, args.cpu_seconds * 0.6 / max(s1, 1e-6) * cores)))
        nq_mt = max(cores, min(nq_mt, args.nq))
        s_mt, c_ids, c_dist = O.bench_brute_force(h_rows, h_q[:nq_mt], args.k, 0, True, cores)
        nq_st = max(1, min(8, int(args.cpu_seconds * 0.2 / max(s1, 1e-6))))
        s_st, _, _ = O.bench_brute_force(h_rows, h_q[:nq_st], args.k, 0, True, 1)
        s_sel, _, _ = O.bench_brute_force(h_rows, h_q[:nq_mt], args.k, 0, False, cores)
        s_avx, _, _ = O.bench_brute_force(h_rows, h_q[:nq_mt], args.k, 1, False, cores)
        parity = bool(np.array_equal(c_ids, ids[:nq_mt]) and c_dist.tobytes() == dd[:nq_mt].tobytes())
        model = ""
        try:
            model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        cpu = {"value": round(nq_mt / s_mt, 3), "unit": "queries/s", "cores": cores, "kind": "port",
               "sample": "%d of the %d queries x all %d rows, top-%d; restated VamanaIndex::brute_force_search (row clone + full sort), "
                         "scalar-4 order, one query per thread" % (nq_mt, args.nq, rows_local, args.k),
               "single_thread_qps": round(nq_st / s_st, 3), "bounded_select_qps": round(nq_mt / s_sel, 3),
               "avx2_order_bounded_select_qps": round(nq_mt / s_avx, 3), "cpu_model": model,
               "gpu_matches_cpu_bit_exact": parity}

    if rank == 0:
        # N > 1, weak scaling (the corpus grows with the GPUs, north_star's "corpus shards across the GPUs"): every rank scans
        # the SAME batch against its own shard, so the whole-job aggregate is queries x shards per second -- "recall queries/s at
        # rows_per_gpu memories", summed over the shards. With that definition value_N / (N * value_1) is the usual weak-scaling
        # efficiency t_1 / t_N. The end-to-end rate (answered queries/s over the N-times larger corpus) is config.end_to_end_queries_per_s.
        weak_multi = world > 1 and args.scaling == "weak"
        value = qps * world if weak_multi else qps
        line = {"metric": ("recall queries/sec @ top-%d, %d-d, %d memories per GPU shard, summed over %d shards" % (args.k, args.dim, rows_local, world))
                          if weak_multi else "recall queries/sec @ top-%d, %d-d, %d memories" % (args.k, args.dim, n_total),
                "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": ("configs[1]: %d memories x %d-d f32, brute-force cosine (-dot) top-%d, batch=%d queries, %d x MI355X"
                                        % (n_total, args.dim, args.k, args.nq, world)) if world == 1 else
                                       ("configs[4] shape (row-sharded corpus, RCCL top-k all-gather): %d memories = %d per GPU x %d MI355X, %d-d f32, "
                                        "brute-force cosine (-dot) top-%d, batch=%d queries (%s scaling)"
                                        % (n_total, hi - lo, world, args.dim, args.k, args.nq, args.scaling)),
                           "rows_total": n_total, "rows_per_gpu": rows_local, "batch": args.nq, "k": args.k, "scan": args.scan,
                           "layout": "row-sharded + RCCL all-gather of per-shard top-k" if world > 1 else "single device",
                           "end_to_end_queries_per_s": round(qps, 1),
                           "value_counts": ("queries x shards: each of the %d ranks scans the same %d-query batch against its own %d-memory shard, "
                                            "then one all-gather + merge; end_to_end_queries_per_s is the answered-query rate over all %d memories"
                                            % (world, args.nq, rows_local, n_total)) if weak_multi else "answered queries",
                           "prescan_dtype": "fp16 MFMA (f32 accumulate) + f32 reference-order re-score"},
                "roofline": roof, "cpu_baseline": cpu, "latency_single_query": lat}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
