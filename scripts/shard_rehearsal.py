#!/usr/bin/env python3
"""BASELINE config 4 (kNN k = 300 at the ml-25m shape, users sharded x8) rehearsed on ONE GPU: eight shard handles go through
the C-ABI shard protocol one after the other (fit -> view -> exchange of the per-user (mean, norm) -> commit -> partial MAE);
the exchange RCCL's all-gather performs between GPUs is done by device copies.  Two passes: the first allocates, the second
is the steady state the numbers are taken from.  Per shard: the stage timers (device time) and the host wall time of the
three calls.  Writes gpurun_out/shard8_timings.json (-> profiles/rNN_shard8_rehearsal_*.json).

  python scripts/shard_rehearsal.py [--world 8] [--ranks 3,7] [--passes 2] [--check]

--ranks: only these ranks run (the others' (mean, norm) segments come from an unsharded donor handle) — the form to put
         under rocprofv3 for the per-kernel picture of one shard's step.
--check: every prediction against a single unsharded engine, bit for bit (the parity test does the same).
This is a measurement script, not a test: tests/test_gpu_parity.py holds the parity version.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "movie-recommender-system_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--ranks", default="")
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--k", type=int, default=300)
    ap.add_argument("--workload", default="syn-25m")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--head-items", type=int, default=0, help="dense head width (0 = cost model)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "shard8_timings.json"))
    args = ap.parse_args()
    import torch

    kn = importlib.import_module(PKG + ".knncf")
    synth = importlib.import_module(PKG + ".synth")
    sharded = importlib.import_module(PKG + ".sharded")
    if args.workload == "syn-25m":
        d = synth.syn_25m()
    else:
        u, i, n = (int(x) for x in args.workload.split(":")[1:4])
        d = synth.syn_scaled(u, i, n, seed=25)
    dev = torch.device("cuda", 0)
    tr = tuple(torch.from_numpy(a).to(dev) for a in (d.train.users, d.train.items, d.train.ratings))
    te = tuple(torch.from_numpy(a).to(dev) for a in (d.test.users, d.test.items, d.test.ratings))
    world = args.world
    ranks = [int(x) for x in args.ranks.split(",")] if args.ranks else list(range(world))
    partial = len(ranks) < world

    single = None
    ref_pred = None
    if args.check or partial:
        single = kn.Engine(k=args.k)
        single.fit_device(*tr)
    if args.check:
        ref_pred = torch.empty(len(d.test.users), dtype=torch.float64, device=dev)
        rs, rc = single.mae_device(kn.PRED_KNN, *te, pred_out=ref_pred)
    donor = sharded.DeviceEngineAdapter(single, dev).shard_tensors() if partial else None
    if single is not None and not partial:
        single.close()  # (135 GB at this shape: the eight shard handles need the room)
        single = None

    engines = {r: kn.Engine(k=args.k, shard_rank=r, shard_count=world, head_items=args.head_items) for r in ranks}
    report = []
    for p in range(args.passes):
        views, wall = {}, {r: 0.0 for r in ranks}
        for r, e in engines.items():
            e.reset_timings()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.fit_device(*tr)
            wall[r] += time.perf_counter() - t0
            views[r] = sharded.DeviceEngineAdapter(e, dev).shard_tensors()
        for me in ranks:  # the exchange: every other rank's (mean, norm) segment
            mlo, mhi = views[me]["user_range"]
            for key in ("user_avg", "user_norm"):
                if partial:
                    views[me][key][:mlo] = donor[key][:mlo]
                    views[me][key][mhi:] = donor[key][mhi:]
                else:
                    for other in ranks:
                        if other != me:
                            lo, hi = views[other]["user_range"]
                            views[me][key][lo:hi] = views[other][key][lo:hi]
        torch.cuda.synchronize()
        preds = torch.full((len(d.test.users),), float("nan"), dtype=torch.float64, device=dev)
        total, count = 0.0, 0
        report = []
        for r, e in engines.items():
            t0 = time.perf_counter()
            e.shard_commit()
            s, c = e.mae_device(kn.PRED_KNN, *te, pred_out=preds)
            wall[r] += time.perf_counter() - t0
            total += s
            count += c
            t = e.timings()
            stage = {k_: t[k_] for k_ in ("prep_ms", "densify_ms", "gemm_ms", "select_ms", "rerank_ms", "predict_ms")}
            lo, hi = views[r]["user_range"]
            nlo, nhi = views[r]["nnz_range"]
            report.append({"rank": r, "users": hi - lo, "train_ratings": nhi - nlo, "test_rows": c, "stage_ms": stage,
                           "stage_sum_ms": sum(stage.values()), "host_wall_ms": wall[r] * 1e3, "head_items": t["head_items"],
                           "fallback_rows": t["fallback_rows"]})
        if args.check and not partial:
            ok = bool(torch.equal(preds.view(torch.int64), ref_pred.view(torch.int64)))
            print(f"pass {p}: predictions bit-equal to the single engine: {ok}; MAE {total / count!r} vs {rs / rc!r}")
            assert ok and count == rc
    out = {"workload": f"{d.name} k={args.k}, {world} shards rehearsed on one MI355X (one after the other; last pass = steady state)",
           "exchange": "per-user (mean, norm) only: 16 B x users; deviations and preprocessed ratings recomputed by every shard",
           "shards": report,
           "max_stage_sum_ms": max(x["stage_sum_ms"] for x in report), "max_host_wall_ms": max(x["host_wall_ms"] for x in report)}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    for x in report:
        print("rank %d  users %d  stage sum %.2f ms  host wall %.2f ms  " % (x["rank"], x["users"], x["stage_sum_ms"], x["host_wall_ms"]),
              {k_: round(v, 2) for k_, v in x["stage_ms"].items()}, "H", x["head_items"])
    print("max stage sum %.2f ms, max host wall %.2f ms" % (out["max_stage_sum_ms"], out["max_host_wall_ms"]))
    for e in engines.values():
        e.close()


if __name__ == "__main__":
    main()
