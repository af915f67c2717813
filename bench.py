#!/usr/bin/env python3
"""bench.py — test-set predictions/s + MAE of the kNN (k=300) path on the ml-25m-shaped workload.

A "step" is one pass of the reference's timed expression (predict/kNN.scala:42-45) over the
device-resident rating triples: fit (ids, means, deviations, norms) + user x user similarity (MFMA
GEMM) + top-k + exact re-rank + prediction of every test rating + MAE.  Inputs are synthetic (no
MovieLens here or on the GPU box), seeded, ml-25m-shaped; they sit in HBM before the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline      — the DOMINANT kernel of this run by summed launch time (k_tail_select at the ml-25m shape), priced on
                  its COMPULSORY HBM bytes (the similarity panel it reads + one pass over the 4-byte tail entries) against
                  the 8 TB/s HBM peak; the same object carries the kernel's other roofs (`other_roofs`: LDS-atomic
                  updates/s against the measured 9.6 updates/clk/CU of scripts/microbench/lds_atomic_rate.hip) and says
                  which one binds (`binding`).  Every frac can be recomputed from the printed definitions + profiles/.
  roofline_all  — the same for the GEMM (MFMA; frac on SURVEY 8d's each-unordered-pair-once flops, `executed_tflops`
                  beside it), the re-rank and the prediction kernel.
  cpu_baseline  — the fp64 CPU restatement (oracle, 1 thread like the reference's local[1]) timed on a bounded sample
                  of the same workload on this host; cpu_baseline_all_cores — the oracle's threaded bulk form on all
                  host cores (cores stated), same bounded-sample protocol.
  step_ms       — mean / population sigma / min / max over the timed steps (shared/predictions.scala:18-25).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "movie-recommender-system_amd"
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16
HBM_PEAK_TBPS = 8.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# scripts/microbench/lds_atomic_rate.hip on MI355X: 9.6 random-address integer LDS lane-updates per clock per CU
LDS_ATOMIC_UPDATES_PER_CLK_PER_CU = 9.6
CUS, CLOCK_GHZ = 256, 2.4
PMC_PROFILE = "r03_pmc_traffic_syn25m_1gpu.json"  # profiles/: the PMC passes of this exact workload (scripts/profile_round.sh)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_workload(name, synth):
    if name == "syn-25m":
        return synth.syn_25m()
    if name == "syn-100k":
        return synth.syn_100k()
    if name.startswith("syn-scaled:"):  # syn-scaled:users:items:ratings (quick rehearsals)
        u, i, n = (int(x) for x in name.split(":")[1:4])
        return synth.syn_scaled(u, i, n, seed=25)
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(split, k, budget_s, n_test_total):
    """(1-thread literal oracle, all-cores bulk oracle) on the first test users, each for about budget_s seconds."""
    import numpy as np

    from oracle import knncf_oracle as O

    O.build()
    tr, te = split.train, split.test
    O.set_threads(1)  # the fit below is part of the single-threaded baseline
    t0 = time.perf_counter()
    model = O.Model(tr.users, tr.items, tr.ratings)
    t_fit = time.perf_counter() - t0
    O.set_threads(0)
    # first-appearance order of test users
    _, first = np.unique(te.users, return_index=True)
    order = te.users[np.sort(first)]

    # (1) the literal per-pair closures on one thread (the reference forces local[1]); grow the sample until the budget is spent
    pipe = model.pipeline(O.SIM_COSINE, k)
    done_preds, t_pred, n_users = 0, 0.0, 0
    chunk = 4
    while t_pred < budget_s and n_users < len(order):
        users = order[n_users:n_users + chunk]
        mask = np.isin(te.users, users)
        t1 = time.perf_counter()
        pipe.mae(te.users[mask], te.items[mask], te.ratings[mask])
        t_pred += time.perf_counter() - t1
        done_preds += int(mask.sum())
        n_users += len(users)
        chunk = min(chunk * 2, 64)
    # whole-job rate = sample predictions / (their neighbour+predict time + their share of the fit)
    share = t_fit * done_preds / max(1, n_test_total)
    one = {
        "value": done_preds / (t_pred + share), "unit": "predictions/s", "cores": 1, "kind": "port",
        "sample": (f"oracle (C fp64 restatement of shared/predictions.scala, reference is single-threaded local[1]): "
                   f"first {n_users} test users = {done_preds} predictions in {t_pred:.1f} s after a {t_fit:.1f} s fit "
                   f"(fit amortised over all {n_test_total} predictions); host: {O.cpu_model()}, {os.cpu_count()} logical CPUs"),
    }
    # (2) the same closures in the oracle's bulk form (row-wise accumulation, one user per thread).  Thread count: every CPU
    # this process is allowed on — and, because a GPU box hands out a CPU SHARE that the affinity mask does not show (256
    # threads on such a box ran at half the rate of 64), 64 threads as well; the faster of the two is reported with its count
    def bulk_rate(cores):
        done_preds, t_used, n_users = 0, 0.0, 0
        chunk = 256 * min(cores, 64)
        while t_used < budget_s and n_users < len(order):
            users = order[n_users:n_users + chunk]
            mask = np.isin(te.users, users)
            t1 = time.perf_counter()
            table = model.knn_table(k, users=users, threads=cores)
            table.mae(te.users[mask], te.items[mask], te.ratings[mask])
            t_used += time.perf_counter() - t1
            del table
            done_preds += int(mask.sum())
            n_users += len(users)
        share = t_fit * done_preds / max(1, n_test_total)
        return done_preds / (t_used + share), done_preds, t_used, n_users

    allowed = O.host_threads()
    all_cores = None
    try:
        tried = {}
        for cores in sorted({min(64, allowed), allowed}):
            tried[cores] = bulk_rate(cores)
        cores = max(tried, key=lambda c: tried[c][0])
        rate, done_preds, t_used, n_users = tried[cores]
        others = "; ".join(f"{c} threads: {tried[c][0]:.0f} predictions/s" for c in tried if c != cores)
        all_cores = {
            "value": rate, "unit": "predictions/s", "cores": cores, "kind": "port",
            "sample": (f"oracle bulk form (OpenMP, one user per thread; tests/test_oracle_bulk.py pins it bit for bit to the "
                       f"literal closures): first {n_users} test users = {done_preds} predictions in {t_used:.1f} s on {cores} "
                       f"threads ({O.cpu_model()}, {os.cpu_count()} logical CPUs on the host, {allowed} in this process's affinity mask"
                       f"{'; also measured: ' + others if others else ''}), same amortised single-threaded {t_fit:.1f} s fit"),
        }
    except O.OracleError:
        pass  # a user with <= 4 ratings: the bulk form refuses (memo-history dependent)
    return one, all_cores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="syn-25m")
    ap.add_argument("--k", type=int, default=300)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1: nccl (= RCCL, the product path) or gloo "
                    "(rehearsal of the N > 1 host path with several ranks sharing one GPU; collectives staged through the host)")
    ap.add_argument("--force-process-group", action="store_true", help="run the whole collective protocol (process group init with device_id, "
                    "the padded all-gather of the (mean, norm) segments, the collective status, the MAE all-reduce, destroy) even at "
                    "--gpus 1: the one-rank rehearsal of the RCCL path on a single MI355X (tests/test_bench_ranks.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16-leg", action="store_true", help="skip the extra (untimed) bf16-operand steps reported beside the fp16 default")
    ap.add_argument("--head-items", type=int, default=0, help="dense head width of the hybrid similarity (0 = cost model)")
    ap.add_argument("--workspace-bytes", type=int, default=0, help="cap of the similarity-panel workspace (0 = auto); small values = small row blocks")
    ap.add_argument("--engine-flags", type=int, default=0, help="KNNCF_FLAG_* bits (1 verify bound, 2 overlap, 4 bf16 filter operands instead of fp16)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or args.force_process_group:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend=args.backend)

    # the in-tree library is prebuilt; if anything is stale, ONE rank rebuilds it while the others wait
    if rank == 0:
        importlib.import_module(PKG + ".build").build()
    if dist is not None:
        dist.barrier()
    if rank != 0:
        importlib.import_module(PKG + ".build").build()
    kn = importlib.import_module(PKG + ".knncf")
    synth = importlib.import_module(PKG + ".synth")
    sharded = importlib.import_module(PKG + ".sharded")

    t0 = time.perf_counter()
    split = make_workload(args.workload, synth)
    tr, te = split.train, split.test
    log(f"[rank {rank}] {split.name}: {len(tr)} train / {len(te)} test ratings generated in {time.perf_counter() - t0:.1f} s")
    d_tr = (torch.from_numpy(tr.users).to(device), torch.from_numpy(tr.items).to(device), torch.from_numpy(tr.ratings).to(device))
    d_te = (torch.from_numpy(te.users).to(device), torch.from_numpy(te.items).to(device), torch.from_numpy(te.ratings).to(device))
    torch.cuda.synchronize()

    eng = kn.Engine(k=args.k, similarity=kn.SIM_COSINE, device=dev_index, shard_rank=rank, shard_count=world,
                    head_items=args.head_items, flags=args.engine_flags, workspace_bytes=args.workspace_bytes)
    model = sharded.ShardedKnn(sharded.DeviceEngineAdapter(eng, device), dist, rank, world, collective=dist is not None)

    def step():
        model.fit(*d_tr)  # closures are rebuilt every measurement, like the reference's timed region
        return model.mae(kn.PRED_KNN, *d_te)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    mae = float("nan")
    for _ in range(args.warmup):
        mae, _ = step()
    eng.reset_timings()
    barrier()
    t_start = time.perf_counter()
    step_end = []
    for _ in range(args.steps):
        mae, n_pred = step()          # (returns after the device work of the step: the sums come back to the host)
        step_end.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        sharded._all_reduce(dist, tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    step_ms = np.diff(np.array([t_start] + step_end)) * 1e3
    tm = eng.timings()

    if rank == 0:
        steps = max(1, args.steps)
        n_test = len(te)
        ms_per_step = elapsed / steps * 1e3
        gemm_launches = max(1, tm["gemm_launches"])          # ONE per neighbour build on the symmetric path, else one per row block
        launches = max(1, tm["select_launches"])              # row-block launches of select / re-rank
        k = args.k

        # Cache/HBM-side traffic per launch from the committed PMC passes of this workload (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs, corrected as MI355X_MICROARCH.md prescribes: profiles/README.md); null for any other
        # setup — counters cannot be collected from inside this process.  (FETCH_SIZE also counts Infinity-Cache hits: it
        # is an upper bound of the HBM bytes.)
        pmc = {}
        pmc_path = os.path.join(ROOT, "profiles", PMC_PROFILE)
        if world == 1 and split.name == "syn-25m" and args.engine_flags == 0 and args.head_items == 0 and args.workspace_bytes == 0 and args.k == 300 and os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)["kernels"]

        def roof(name, bound, ms, work, peak, unit, note, launches=launches, key=None):
            """achieved = ALGORITHMIC work per launch / average launch duration (HIP events on the kernel's stream)"""
            s_per_launch = ms / launches / 1e3
            ach = (work / launches) / s_per_launch / 1e12 if s_per_launch > 0 else 0.0
            r = {"bound": bound, "kernel": name, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                 "launches_per_step": launches / steps, "avg_launch_ms": s_per_launch * 1e3, "traffic": None,
                 "algorithmic_work": note, "algorithmic_per_launch": work / launches}
            if key in pmc:
                r["traffic"] = pmc[key]["traffic_bytes_per_launch"]
                r["traffic_source"] = f"profiles/{PMC_PROFILE} (2 x FETCH_SIZE + WRITE_SIZE, bytes per launch; includes Infinity-Cache hits)"
            return r

        n_train = len(tr)
        kernels = {
            # each UNORDERED pair once x 2 flops x the columns the dense part contracts (head_items); the kernel
            # executes the full square (both orders of every pair, padded tiles): executed_tflops beside it
            "k_gemm_nt_bf16": roof("k_gemm_nt_ov / k_gemm_nt_bf16 (user x user similarity, dense head, MFMA)", "mfma", tm["gemm_ms"],
                                   0.5 * tm["gemm_flops_algorithmic"], MFMA_BF16_DENSE_PEAK_TFLOPS, "TFLOP/s",
                                   "rows * (U-1) * head_items flops per launch = each unordered (row, user) pair once x 2 flops x head_items "
                                   "(SURVEY 8d with I_c -> the dense head; the sparse tail is k_tail_select's work)", launches=gemm_launches, key="k_gemm_nt_bf16"),
            # compulsory HBM bytes: the similarity panel is read once, the 4-byte tail entries once per launch (their ~450
            # re-reads per launch are served by L2 / Infinity Cache and are priced on the LDS-atomic roof instead)
            "k_tail_select": roof("k_tail_select (sparse tail + histogram select)", "hbm", tm["select_ms"],
                                  tm["select_row_bytes"] + 4.0 * n_train * launches, HBM_PEAK_TBPS, "TB/s",
                                  "compulsory bytes: panel entry size (2 B fp16 / 4 B fp32) * rows * U read once + 4 B * train ratings "
                                  "(one pass over the tail entries) per launch", key="k_tail_select"),
            "k_rerank": roof("k_rerank (exact fp64 re-rank + top-k)", "hbm", tm["rerank_ms"], tm["rerank_row_bytes"],
                             HBM_PEAK_TBPS, "TB/s", "12 B * ratings of every shortlisted candidate (no reuse assumed; the rows are "
                             "re-read from L2 / Infinity Cache, so this is cache-level, not compulsory HBM, traffic)", key="k_rerank"),
            "k_predict_knn": roof("k_predict_knn_items (weighted-sum prediction + MAE; the stage also holds the id lookup and the row sort)",
                                  "hbm", tm["predict_ms"], 12.0 * k * n_test * steps, HBM_PEAK_TBPS, "TB/s",
                                  "12 * k B per prediction (SURVEY 8d)", launches=steps, key="k_predict_knn"),
        }
        g = kernels["k_gemm_nt_bf16"]
        g["executed_tflops"] = tm["gemm_flops_executed"] / (tm["gemm_ms"] / 1e3) / 1e12 if tm["gemm_ms"] > 0 else 0.0
        g["symmetric"] = tm["gemm_launches"] < tm["select_launches"]  # one launch computes the tiles on/above the diagonal and mirrors them
        g["executed_frac"] = g["executed_tflops"] / MFMA_BF16_DENSE_PEAK_TFLOPS
        # the launch's other roof: it writes the whole similarity panel (both triangles on the symmetric path)
        panel_bytes = tm["select_row_bytes"] / gemm_launches  # (what select reads is what the GEMM wrote: entry size x rows x users)
        g["other_roofs"] = {"hbm_write": {"achieved": panel_bytes / (tm["gemm_ms"] / gemm_launches / 1e3) / 1e12 if tm["gemm_ms"] > 0 else 0.0, "peak": HBM_PEAK_TBPS,
                                          "unit": "TB/s", "definition": "similarity panel bytes written per launch / launch time; a store-only kernel of the same "
                                                                        "pattern writes 5.4 - 5.9 TB/s (scripts/microbench/store_pattern.hip): 9.0 ms for the 53 GB panel"}}
        g["binding"] = ("the CU's vector-memory path: at K = 384 the launch is 9.0 ms of panel stores (what the K = 128 launch takes) plus 1.3 ms per k-step "
                        "beyond the second — loads and stores do not overlap although nothing in the instruction stream orders them (one in-order pipe per CU, "
                        "backed up by the HBM write rate); MFMA work hides under the loop's own overhead (ablations in DESIGN.md section 4); the north_star-literal "
                        "dense formulation (all 59 047 columns) runs at 0.47 of the MFMA peak: profiles/r03_dense_all_syn25m_1gpu.json")
        # the tail's other roof: integer LDS atomics (one per tail pair product)
        ts = kernels["k_tail_select"]
        lds_peak = LDS_ATOMIC_UPDATES_PER_CLK_PER_CU * CUS * CLOCK_GHZ * 1e9
        upd_per_s = tm["tail_pair_updates"] / (tm["select_ms"] / 1e3) if tm["select_ms"] > 0 else 0.0
        ts["other_roofs"] = {
            "lds_atomic": {"achieved": upd_per_s / 1e12, "peak": lds_peak / 1e12, "unit": "T lane-updates/s", "frac": upd_per_s / lds_peak,
                           "definition": f"tail pair products per step ({tm['tail_pair_updates'] / steps:.3e}: sum over tail items of raters-in-rows x raters) "
                                         f"/ kernel time; peak = {LDS_ATOMIC_UPDATES_PER_CLK_PER_CU} updates/clk/CU (scripts/microbench/lds_atomic_rate.hip) x {CUS} CUs x {CLOCK_GHZ} GHz"},
            "cache_level_bytes": {"achieved": (tm["select_row_bytes"] + 4.0 * tm["tail_pair_updates"]) / (tm["select_ms"] / 1e3) / 1e12 if tm["select_ms"] > 0 else 0.0,
                                  "unit": "TB/s", "definition": "panel bytes + 4 B per tail pair product (L2 / Infinity-Cache re-reads of the rater lists; NOT HBM bytes)"},
        }
        ts["binding"] = ("neither throughput roof of the line: the kernel alternates a panel scan (HBM: 53 GB per step at ~4.6 TB/s when run "
                         "alone, 11.5 ms) with the tail drain — ~3e8 pieces x 256 B = 77 GB per step out of the 80 MB rater-list array, which "
                         "lives in the Infinity Cache (it cannot live in the 4 MB L2s) — and the drain is bound by the memory LATENCIES a wave's "
                         "window of ~23 pieces pays one after the other, not by bytes or instruction issue.  Evidence (DESIGN.md section 4, "
                         "profiles/README.md): round 3 removed the dummy look-ahead group the drain used to wait for at the end of every window: "
                         "25.7 -> 24.5 ms; pieces aligned to 256 B (two cache lines instead of three, 12 % more pieces): 26.4 ms; 8 -> 6 VALU per "
                         "piece: the marginal rate per pair product did not move (4.2e-13 s, head sweeps 256 .. 640); every piece read out of the "
                         "array's first 16 KiB (timing-only): -2.2 ms of ~11.  Timing-only ablations of the round-2 build at H = 384 (26.7 ms "
                         "kernel): no tail machinery 11.5 ms, + tail set-up and read-out 15.6, + the drain 26.7.  SQ pass of profiles/"
                         + PMC_PROFILE + ": VALU issue and waiting fractions.  HBM frac on compulsory bytes and the LDS-atomic frac are both reported")
        stage_of = {"k_gemm_nt_bf16": "gemm_ms", "k_tail_select": "select_ms", "k_rerank": "rerank_ms", "k_predict_knn": "predict_ms"}
        dominant = max(stage_of, key=lambda n: tm[stage_of[n]])
        out = {
            "metric": "test-set predictions/sec (kNN k=%d, fit+predict+MAE)" % args.k,
            "value": n_test * steps / elapsed,
            "unit": "predictions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "step_ms": {"mean": float(step_ms.mean()), "sigma": float(step_ms.std()), "min": float(step_ms.min()), "max": float(step_ms.max()),
                        "note": "rank 0's wall time per step; population sigma as shared/predictions.scala:19-25"},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": ("bf16" if args.engine_flags & 4 else "fp16") + " MFMA similarity filter (fp32 acc) + f64 exact re-rank/prediction",
            "data": "synthetic",
            "mae": mae,
            "config": {"workload": f"{split.name}: predict.kNN k={args.k}, {eng.num_users} users x {eng.num_items} items, "
                                   f"{len(tr)} train / {n_test} test ratings", "parallelism": f"users block-partitioned x{world}",
                       # one step = knncf_fit + knncf_mae(PRED_KNN): what predict/kNN.scala:43-57 evaluates.  The per-item maps of the
                       # baseline predictors (K4: itemsAvg, itemsAvgDev) are not on that path in the reference either (predictions.scala
                       # :489-585 never calls them); the handle builds them on first use
                       "step": "fit (K0-K3) + neighbourhoods (K5, K6) + predictions + MAE (K7, K8); K4 not on the kNN path",
                       "collectives": (f"torch.distributed/{dist.get_backend()} ({'RCCL' if dist.get_backend() == 'nccl' else 'host-staged rehearsal'}), "
                                       f"world {world}: all-gather of the per-user (mean, norm) segments, all-reduce of the fit status and of "
                                       f"(sum |err|, rows, status)") if dist is not None else "none (one handle)"},
            "roofline": kernels[dominant],  # the dominant kernel of THIS run (by summed launch time)
            "roofline_all": kernels,
            "stage_ms_per_step": {k_: tm[k_] / steps for k_ in ("prep_ms", "densify_ms", "gemm_ms", "tail_ms", "select_ms", "rerank_ms", "predict_ms")},
            # the prediction stage alone (dense-id lookup, row sort, weighted-sum prediction, MAE reduction)
            "predict_stage_predictions_per_s": n_test * steps / (tm["predict_ms"] / 1e3) if tm["predict_ms"] > 0 else None,
            "shortlist_mean": tm["shortlist_total"] / max(1, steps * eng.num_users),
            "fallback_rows_per_step": tm["fallback_rows"] / steps,
            "hybrid": {"head_items": tm["head_items"], "tail_pair_updates_per_step": tm["tail_pair_updates"] / steps},
            # the only throughput the reference publishes (knn-100k.json:53-58: 26.2 s for 20 000 predictions at
            # ml-100k, k = 300, Spark local[1]); another data set and unknown hardware, so vs_baseline stays null
            "reference_published": {"value": 763.0, "unit": "predictions/s", "workload": "ml-100k u2 kNN k=300 (knn-100k.json:53-58)",
                                    "hardware": "unstated"},
        }
        if world == 1 and not args.no_bf16_leg and not (args.engine_flags & 4):
            # north_star says bf16 operands; the default is fp16 (same MFMA rate, 8x narrower error band, identical
            # results).  The bf16 number of the same build, outside the timed region (the fp16 handle is closed first: two
            # handles of this shape do not both fit the one-block configuration in 288 GB, and the leg should measure what a
            # bf16 run of this command measures):
            eng.close()
            eb = kn.Engine(k=args.k, similarity=kn.SIM_COSINE, device=dev_index, head_items=args.head_items,
                           flags=args.engine_flags | kn.FLAG_BF16_FILTER)
            mb = sharded.ShardedKnn(sharded.DeviceEngineAdapter(eb, device), None, 0, 1)
            mb.fit(*d_tr)
            mb.mae(kn.PRED_KNN, *d_te)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            nb = max(1, min(3, args.steps))
            for _ in range(nb):
                mb.fit(*d_tr)
                mae_b, _ = mb.mae(kn.PRED_KNN, *d_te)
            torch.cuda.synchronize()
            dtb = (time.perf_counter() - tb) / nb
            out["bf16_filter"] = {"ms_per_step": dtb * 1e3, "value": n_test / dtb, "unit": "predictions/s", "steps": nb,
                                  "mae": mae_b, "shortlist_mean": eb.timings()["shortlist_total"] / max(1, (nb + 1) * eb.num_users),
                                  "note": "KNNCF_FLAG_BF16_FILTER (bf16 GEMM operands as north_star words it): same neighbours and MAE, "
                                          "wider error band -> longer shortlists -> more re-rank work"}
            eb.close()
        if world == 1 and not args.no_cpu_baseline:
            one, all_cores = cpu_baseline(split, args.k, args.cpu_baseline_seconds, n_test)
            out["cpu_baseline"] = one
            if all_cores:
                out["cpu_baseline_all_cores"] = all_cores
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
