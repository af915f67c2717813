"""usage: python scripts/analysis/gpu_idle.py <kernel_trace.csv> [steps]
Reads a rocprofv3 --kernel-trace CSV of a bench run and reports, for the last `steps` steps (a step starts at k_id_range /
the first kernel of a fit), the wall time, the time at least one kernel was running, and the largest idle gaps with the
kernels on either side — where the host (a sync, a launch chain) leaves the GPU waiting."""
import csv
import sys


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
    # a step = from one k_tail_select to the next (one launch per step at the headline shape)
    marks = [i for i, e in enumerate(ev) if "k_tail_select" in e[2]]
    if len(marks) < 3:
        print("not enough steps")
        return
    a, b = marks[-3], marks[-2]  # one whole step between two select launches
    seg = ev[a:b]
    t0, t1 = seg[0][0], ev[b][0]
    busy = 0
    cur_s, cur_e = seg[0][0], seg[0][1]
    gaps = []
    last_name = seg[0][2]
    for s, e, n in seg[1:] + [ev[b]]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name[:60], n[:60]))
            cur_s, cur_e = s, e
            last_name = n
        else:
            if e > cur_e:
                cur_e = e
                last_name = n
    busy += min(cur_e, t1) - cur_s
    print("step wall %.3f ms, GPU busy %.3f ms, idle %.3f ms, kernels %d" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(seg)))
    for g, p, n in sorted(gaps, reverse=True)[:14]:
        print("  gap %7.1f us  after %-60s before %s" % (g / 1e3, p, n))


if __name__ == "__main__":
    main()
