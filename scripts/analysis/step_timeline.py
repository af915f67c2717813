"""usage: python scripts/analysis/step_timeline.py <kernel_trace.csv> [t_end_ms]
Prints the kernels of one steady-state step (from the first kernel after a k_reduce_err / the step before's last kernel up to
t_end_ms, default: until the similarity GEMM starts) in start order: queue, start offset, duration — the fit's launch timeline."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows), key=lambda e: e[0])
    gemm = [i for i, e in enumerate(ev) if "k_gemm_nt" in e[2]]
    sel = [i for i, e in enumerate(ev) if "k_tail_select" in e[2]]
    # the step whose GEMM is the second-to-last one: starts after the previous step's last kernel
    g = gemm[-2]
    prev_end = max(i for i in range(g) if "k_predict_knn" in ev[i][2] or "k_reduce" in ev[i][2])
    # first kernel of this step: the first one starting after the previous prediction kernel ended + its reduction
    start_i = prev_end + 1
    while start_i < g and ("reduce" in ev[start_i][2] or "k_predict" in ev[start_i][2]):
        start_i += 1
    t0 = ev[start_i][0]
    t_end = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else ev[g][0] - t0 + 1
    for s, e, n, q in ev[start_i:]:
        if s - t0 > t_end:
            break
        print("q%-3s %9.1f us  +%8.1f us  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, n[:100]))


if __name__ == "__main__":
    main()
