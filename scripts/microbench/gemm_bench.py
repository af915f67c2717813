#!/usr/bin/env python3
"""The similarity GEMM alone (knncf_debug_gemm_bench: not part of the C ABI header): ms per launch at N users x K head columns.
  python scripts/microbench/gemm_bench.py [--n 162560] [--k 384,512,768] [--rows 0 (symmetric) | M] [--iters 5]
KNNCF_GEMM_NO_OVERLAP=1 selects the epilogue-at-the-end kernel for the A/B."""
import argparse
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kn = importlib.import_module("movie-recommender-system_amd.knncf")
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=162560)
ap.add_argument("--k", default="384")
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
L = kn.load_library()
L.knncf_debug_gemm_bench.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
for k in (int(x) for x in a.k.split(",")):
    ms = C.c_double()
    st = L.knncf_debug_gemm_bench(0, a.n, k, a.rows if a.rows else 256, 0 if a.rows else 1, a.iters, C.byref(ms))
    rows = a.rows if a.rows else a.n
    flops = (a.n * (a.n + 256) if not a.rows else 2 * rows * a.n) * k  # executed: tiles on/above the diagonal x 2 flops
    print(f"N {a.n} K {k} rows {rows} {'sym' if not a.rows else 'row-block'}: status {st}  {ms.value:8.3f} ms/launch  {flops / ms.value / 1e9:7.1f} TFLOP/s executed  "
          f"panel {rows * a.n * 2 / 1e9:.1f} GB -> {rows * a.n * 2 / ms.value / 1e9:.2f} TB/s written", flush=True)
