"""Builds libknncf.so (HIP, gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libknncf.so")
CLI = os.path.join(HERE, "knncf")
SOURCES = ["api.cpp", "group.cpp", "loader.cpp", "prep.hip", "sort_util.hip", "gemm.hip", "select.hip", "rerank.hip", "predict.hip", "neighbours.hip", "reco.hip"]
HEADERS = ["common.h", "engine.h", os.path.join("..", "..", "include", "knncf.h")]
# -ffp-contract=off: the fp64 kernels must round exactly like the reference's JVM arithmetic (no FMA)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-result", "-x", "hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [_hipcc()] + FLAGS + os.environ.get("KNNCF_EXTRA_HIPCC_FLAGS", "").split() + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for warn in ex.map(compile_one, jobs):
            if warn.strip() and verbose:
                print(warn, file=sys.stderr)
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    # native entry points (predict.Baseline / Personalized / kNN, distributed.DistributedBaseline): plain C++
    # over the C ABI only
    cli_src = os.path.join(CSRC, "cli.cpp")
    if force or _stale(CLI, [cli_src, LIB, os.path.join(HERE, "..", "include", "knncf.h")]):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-o", CLI, cli_src, "-L" + HERE, "-lknncf", "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"cli build failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
