"""usage: python scripts/install_profiles.py <gpurun_out/TAG dir> <rNN>
Copies what scripts/round_final.sh (profile_round.sh) left under gpurun_out/<tag>/ into profiles/ under the round's names; the
PMC collections are reduced to the dispatches of the four big kernels (the full files list every small launch of the run)."""
import csv
import os
import shutil
import sys

BIG = ("k_tail_select", "k_gemm_nt", "k_rerank", "k_predict_knn")


def reduce_csv(src, dst):
    rows = list(csv.reader(open(src)))
    keep = [rows[0]] + [r for r in rows[1:] if any(b in r[8] for b in BIG)]
    with open(dst, "w", newline="") as f:
        csv.writer(f).writerows(keep)


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    tag = os.path.basename(os.path.normpath(src))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    cp = lambda a, b: shutil.copyfile(os.path.join(src, a), os.path.join(out, b))
    cp("bench.json", f"{rnd}_bench_syn25m_1gpu.json")
    cp("bench_prof.json", f"{rnd}_bench_under_rocprof_syn25m_1gpu.json")
    cp(f"prof/{tag}_kernel_stats.csv", f"{rnd}_kernel_stats_syn25m_1gpu.csv")
    reduce_csv(os.path.join(src, "pmc_fetch/f_counter_collection.csv"), os.path.join(out, f"{rnd}_pmc_fetch_size_syn25m_1gpu.csv"))
    reduce_csv(os.path.join(src, "pmc_write/w_counter_collection.csv"), os.path.join(out, f"{rnd}_pmc_write_size_syn25m_1gpu.csv"))
    reduce_csv(os.path.join(src, "pmc_sq/s_counter_collection.csv"), os.path.join(out, f"{rnd}_pmc_sq_syn25m_1gpu.csv"))
    cp("pmc_traffic.json", f"{rnd}_pmc_traffic_syn25m_1gpu.json")
    if os.path.exists(os.path.join(src, "shard8_timings.json")):
        cp("shard8_timings.json", f"{rnd}_shard8_rehearsal_syn25m_1gpu.json")
    with open(os.path.join(out, f"{rnd}_phase_counters.txt"), "w") as f:
        for k in ("SELECT", "RERANK", "PREDICT"):
            p = os.path.join(src, f"phases_{k}.txt")
            if os.path.exists(p):
                lines = [l for l in open(p).read().splitlines() if "profile" in l or l.strip().startswith("phase")]
                note = " — phase 1 of k_tail_select includes the wait for the profiling atomic issued at the barrier before it" if k == "SELECT" else ""
                f.write(f"== {k}: scripts/phase_profile.sh, -DKNNCF_{k}_PROFILE (thread 0 of every workgroup; the counters slow the kernel{note}) ==\n")
                f.write("\n".join(lines[-13:] if k == "SELECT" else lines[-1:]) + "\n")
    print("installed", rnd, "from", src)


if __name__ == "__main__":
    main()
