"""Derive profiles/<tag>_pmc_traffic_*.json from the three rocprofv3 counter passes (profiles/README.md):

    python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <sq_counter_collection.csv> <out.json>

HBM-side traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B: MI355X_MICROARCH.md's HBM section — FETCH_SIZE / WRITE_SIZE
are in KB and gfx950 reports half of the bytes of wide coalesced reads.  SQ ratios: SQ_WAVE_CYCLES counts quad-cycles per wave,
so SQ_INSTS_VALU / SQ_WAVE_CYCLES x resident waves per SIMD = VALU issue utilisation."""
import csv
import json
import sys
from collections import defaultdict

KERNELS = {"k_tail_select": "k_tail_select", "k_gemm_nt_ov": "k_gemm_nt_bf16", "k_gemm_nt_bf16": "k_gemm_nt_bf16", "k_rerank<": "k_rerank",
           "k_rerankI": "k_rerank", "k_predict_knn_items": "k_predict_knn"}


def per_kernel(path):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(path)):
        for pat, name in KERNELS.items():
            if pat in row["Kernel_Name"]:
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                break
    return acc


def main():
    fetch, write, sq, out = sys.argv[1:5]
    bench_line = sys.argv[5] if len(sys.argv) > 5 else None  # the bench line of one of the passes: names the head the cost model took
    f, w, s = per_kernel(fetch), per_kernel(write), per_kernel(sq)
    kernels = {}
    for name in ("k_tail_select", "k_gemm_nt_bf16", "k_rerank", "k_predict_knn"):
        fs, ws = f[name]["FETCH_SIZE"], w[name]["WRITE_SIZE"]
        fk, wk = sum(fs) / len(fs), sum(ws) / len(ws)
        c = {k: sum(v) for k, v in s[name].items()}
        wc = c["SQ_WAVE_CYCLES"]
        kernels[name] = {
            "launches_profiled": len(fs),
            "FETCH_SIZE_KB_per_launch": fk,
            "WRITE_SIZE_KB_per_launch": wk,
            "traffic_bytes_per_launch": (2 * fk + wk) * 1024,
            "sq": {
                "wait_any_frac": c["SQ_WAIT_ANY"] / wc,
                "wait_inst_any_frac": c["SQ_WAIT_INST_ANY"] / wc,
                "active_inst_any_frac": c["SQ_ACTIVE_INST_ANY"] / wc,
                "valu_insts_per_wave_quadcycle": c["SQ_INSTS_VALU"] / wc,
                "salu_insts_per_wave_quadcycle": c["SQ_INSTS_SALU"] / wc,
                "lds_insts_per_wave_quadcycle": c["SQ_INSTS_LDS"] / wc,
            },
        }
    workload = "syn-25m k=300 1 GPU, default flags (head_items by the cost model)"
    if bench_line:
        try:
            d = json.loads(open(bench_line).read().strip().splitlines()[-1])
            workload = f"{d['config']['workload']}, 1 GPU, default flags (head_items by the cost model: {int(d['hybrid']['head_items'])})"
        except Exception:
            pass
    doc = {
        "workload": workload,
        "source": "rocprofv3 --pmc <one counter set per run> --kernel-trace, python3 bench.py --no-cpu-baseline --steps 1 --warmup 1",
        "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE (KB) reports half of the bytes of wide coalesced reads on gfx950 -> doubled; "
                      "WRITE_SIZE (KB) exact for 16-byte stores; narrow gathers uncalibrated. SQ_* counters: SQ_WAVE_CYCLES counts quad-cycles; "
                      "valu_insts_per_wave_quadcycle x resident waves per SIMD = VALU issue utilisation",
        "kernels": kernels,
    }
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(k, v["launches_profiled"], round(v["traffic_bytes_per_launch"] / 1e9, 2), "GB/launch", {a: round(b, 3) for a, b in v["sq"].items()})


if __name__ == "__main__":
    main()
