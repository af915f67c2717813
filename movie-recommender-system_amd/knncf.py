"""ctypes binding of libknncf.so (include/knncf.h).

This is the only door into the engine: there is no Python/NumPy/torch compute path behind it.
If the shared library is missing or no gfx950 device is usable the calls raise — they never fall
back to a CPU implementation.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libknncf.so")

OK = 0
E_INVALID, E_NONFINITE, E_DUPLICATE, E_NOMEM, E_HIP, E_STATE, E_UNSUPPORTED, E_NODEVICE, E_RCCL = range(-1, -10, -1)
SIM_COSINE, SIM_ONE, SIM_JACCARD = 0, 1, 2
PRED_GLOBAL_AVG, PRED_USER_AVG, PRED_ITEM_AVG, PRED_BASELINE, PRED_BASELINE_RDD, PRED_KNN, PRED_PERSONALIZED = range(7)
FLAG_VERIFY_BOUND = 1
FLAG_OVERLAP = 2
FLAG_BF16_FILTER = 4  # default filter operand type is fp16 (narrower error band, same MFMA rate)
FLAG_F32_PANEL = 8    # default similarity panel storage is fp16
HEAD_ALL = 0xFFFFFFFF

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("k", C.c_int32),
                ("similarity", C.c_int32), ("shard_rank", C.c_int32), ("shard_count", C.c_int32),
                ("workspace_bytes", C.c_int64), ("flags", C.c_uint32), ("head_items", C.c_uint32)]


class Ratings(C.Structure):
    """knncf_ratings of include/knncf.h"""
    _fields_ = [("n", C.c_int64), ("users", C.POINTER(C.c_int32)), ("items", C.POINTER(C.c_int32)),
                ("ratings", C.POINTER(C.c_double))]


def load_file(path, separator="\t", threads=0, cache=None, info=None):
    """`load` shared/predictions.scala:35-49 through the library's multithreaded parser: (users, items, ratings) in
    file order.  No GPU needed.  cache: path of the binary cache (knncf_load_file_cached): read when it matches the
    file as it is now, (re)written otherwise; info (a dict) then receives {"from_cache": bool}."""
    L = load_library()
    r = Ratings()
    err = C.create_string_buffer(512)
    if cache is None:
        st = L.knncf_load_file(os.fsencode(path), separator.encode(), threads, C.byref(r), err, len(err))
    else:
        hit = C.c_int(0)
        st = L.knncf_load_file_cached(os.fsencode(path), separator.encode(), threads, os.fsencode(cache), C.byref(r),
                                      C.byref(hit), err, len(err))
        if info is not None:
            info["from_cache"] = bool(hit.value)
    if st != 0:
        raise KnncfError(st, err.value.decode(errors="replace"))
    try:
        n = r.n
        u = np.ctypeslib.as_array(r.users, shape=(max(n, 1),))[:n].copy()
        i = np.ctypeslib.as_array(r.items, shape=(max(n, 1),))[:n].copy()
        x = np.ctypeslib.as_array(r.ratings, shape=(max(n, 1),))[:n].copy()
    finally:
        L.knncf_free_ratings(C.byref(r))
    return u, i, x


class Personal(C.Structure):
    """knncf_personal of include/knncf.h"""
    _fields_ = [("n_rows", C.c_int64), ("row_ids", C.POINTER(C.c_int32)), ("row_names", C.POINTER(C.c_char_p)),
                ("name_storage", C.c_void_p), ("ratings", Ratings)]


def load_personal(path, user=944):
    """recommend/Recommender.scala:40-54 through knncf_load_personal: (names, (users, items, ratings)) — `names` is the
    list of (id, title) of every row in file order, the header as (0, "header"); the ratings are the non-zero ones."""
    L = load_library()
    p = Personal()
    err = C.create_string_buffer(512)
    st = L.knncf_load_personal(os.fsencode(path), user, C.byref(p), err, len(err))
    if st != 0:
        raise KnncfError(st, err.value.decode(errors="replace"))
    try:
        names = [(int(p.row_ids[j]), p.row_names[j].decode("utf-8", errors="replace")) for j in range(p.n_rows)]
        n = p.ratings.n
        u = np.ctypeslib.as_array(p.ratings.users, shape=(max(n, 1),))[:n].copy()
        i = np.ctypeslib.as_array(p.ratings.items, shape=(max(n, 1),))[:n].copy()
        x = np.ctypeslib.as_array(p.ratings.ratings, shape=(max(n, 1),))[:n].copy()
    finally:
        L.knncf_free_personal(C.byref(p))
    return names, (u, i, x)


class Timings(C.Structure):
    _fields_ = [("prep_ms", C.c_double), ("densify_ms", C.c_double), ("gemm_ms", C.c_double),
                ("tail_ms", C.c_double), ("select_ms", C.c_double), ("rerank_ms", C.c_double), ("predict_ms", C.c_double),
                ("gemm_launches", C.c_int64), ("gemm_flops_executed", C.c_double),
                ("gemm_flops_algorithmic", C.c_double), ("shortlist_total", C.c_int64),
                ("fallback_rows", C.c_int64), ("max_bound_violation", C.c_double),
                ("head_items", C.c_int64), ("tail_pair_updates", C.c_double),
                ("rerank_row_bytes", C.c_double), ("select_row_bytes", C.c_double), ("select_launches", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class ShardView(C.Structure):
    _fields_ = [("user_begin", C.c_int32), ("user_end", C.c_int32), ("nnz_begin", C.c_int64),
                ("nnz_end", C.c_int64), ("num_users", C.c_int32), ("num_ratings", C.c_int64),
                ("d_user_avg", C.c_void_p), ("d_user_norm", C.c_void_p), ("d_dev", C.c_void_p),
                ("d_pre", C.c_void_p)]


EXPORTS = [
    "knncf_version", "knncf_status_string", "knncf_create", "knncf_destroy", "knncf_last_error",
    "knncf_fit", "knncf_fit_device", "knncf_num_users", "knncf_num_items", "knncf_global_avg",
    "knncf_user_avg", "knncf_item_avg", "knncf_item_avg_dev", "knncf_item_avg_dev_rdd", "knncf_similarity",
    "knncf_knn_similarity", "knncf_neighbors", "knncf_neighbors_batch", "knncf_predict", "knncf_recommend", "knncf_predict_batch",
    "knncf_predict_batch_device", "knncf_mae", "knncf_mae_device", "knncf_shard_view_get",
    "knncf_shard_commit", "knncf_get_timings", "knncf_reset_timings", "knncf_reset_neighbors",
    "knncf_set_k", "knncf_load_file", "knncf_load_file_cached", "knncf_free_ratings", "knncf_load_personal", "knncf_free_personal", "knncf_neighbors_save", "knncf_neighbors_load",
    "knncf_group_create", "knncf_group_destroy", "knncf_group_last_error", "knncf_group_size", "knncf_group_handle",
    "knncf_group_fit", "knncf_group_mae", "knncf_group_predict_batch",
]


class KnncfError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"knncf status {status}: {message}")
        self.status = status


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64; if libknncf.so
    pulled in /opt/rocm's copy first, torch could no longer see the GPU later in the same process.
    Pre-load torch's copy (without importing torch) so both bind to the same runtime."""
    import importlib.util

    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    lib_dir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(lib_dir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def _share_rccl_with_torch():
    """knncf_group_* resolves RCCL at run time and reuses a copy that is already in the process.  Under Python that must
    be the one PyTorch bundles (built against the HIP runtime _share_hip_runtime_with_torch bound the library to): load it
    — importing torch does — before the first group is created, so that /opt/rocm's copy is not pulled in beside it."""
    import importlib.util

    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401  (libtorch_hip links librccl)


def load_library():
    """Load libknncf.so (built in-tree by build.py).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not found: build the HIP extension first (python __graft_entry__.py or "
            f"movie-recommender-system_amd/build.py); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    L.knncf_version.restype = C.c_char_p
    L.knncf_status_string.restype = C.c_char_p
    L.knncf_status_string.argtypes = [C.c_int]
    L.knncf_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.knncf_destroy.argtypes = [C.c_void_p]
    L.knncf_destroy.restype = None
    L.knncf_last_error.argtypes = [C.c_void_p]
    L.knncf_last_error.restype = C.c_char_p
    L.knncf_fit.argtypes = [C.c_void_p, _i32p, _i32p, _f64p, C.c_int64]
    L.knncf_fit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.knncf_num_users.argtypes = [C.c_void_p, _i32p]
    L.knncf_num_items.argtypes = [C.c_void_p, _i32p]
    L.knncf_global_avg.argtypes = [C.c_void_p, _f64p]
    for n in ("knncf_user_avg", "knncf_item_avg", "knncf_item_avg_dev", "knncf_item_avg_dev_rdd"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_int32, _f64p]
    for n in ("knncf_similarity", "knncf_knn_similarity"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_int32, C.c_int32, _f64p]
    L.knncf_neighbors.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _i32p, _f64p, _i32p]
    L.knncf_neighbors_batch.argtypes = [C.c_void_p, _i32p, C.c_int64, C.c_int32, _i32p, _f64p, _i32p]
    L.knncf_predict.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, _f64p]
    L.knncf_recommend.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, _i32p, _f64p, C.POINTER(C.c_int32)]
    L.knncf_predict_batch.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, C.c_int64, _f64p]
    L.knncf_predict_batch_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.knncf_mae.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, _f64p, C.c_int64, _f64p]
    L.knncf_mae_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   _f64p, C.POINTER(C.c_int64), C.c_void_p]
    L.knncf_shard_view_get.argtypes = [C.c_void_p, C.POINTER(ShardView)]
    L.knncf_shard_commit.argtypes = [C.c_void_p]
    L.knncf_get_timings.argtypes = [C.c_void_p, C.POINTER(Timings)]
    L.knncf_reset_timings.argtypes = [C.c_void_p]
    L.knncf_reset_neighbors.argtypes = [C.c_void_p]
    L.knncf_set_k.argtypes = [C.c_void_p, C.c_int32]
    L.knncf_load_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(Ratings), C.c_char_p, C.c_int]
    L.knncf_load_file_cached.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(Ratings), C.POINTER(C.c_int),
                                         C.c_char_p, C.c_int]
    L.knncf_free_ratings.argtypes = [C.POINTER(Ratings)]
    L.knncf_free_ratings.restype = None
    L.knncf_load_personal.argtypes = [C.c_char_p, C.c_int32, C.POINTER(Personal), C.c_char_p, C.c_int]
    L.knncf_free_personal.argtypes = [C.POINTER(Personal)]
    L.knncf_free_personal.restype = None
    L.knncf_neighbors_save.argtypes = [C.c_void_p, C.c_char_p]
    L.knncf_neighbors_load.argtypes = [C.c_void_p, C.c_char_p]
    L.knncf_group_create.argtypes = [C.POINTER(Config), _i32p, C.c_int32, C.POINTER(C.c_void_p)]
    L.knncf_group_destroy.argtypes = [C.c_void_p]
    L.knncf_group_destroy.restype = None
    L.knncf_group_last_error.argtypes = [C.c_void_p]
    L.knncf_group_last_error.restype = C.c_char_p
    L.knncf_group_size.argtypes = [C.c_void_p, _i32p]
    L.knncf_group_handle.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    L.knncf_group_fit.argtypes = [C.c_void_p, _i32p, _i32p, _f64p, C.c_int64]
    L.knncf_group_mae.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, _f64p, C.c_int64, _f64p]
    L.knncf_group_predict_batch.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, C.c_int64, _f64p]
    _lib = L
    return L


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dev_ptr(t, dtype_name):
    """Raw device pointer of a contiguous torch tensor (torch is only plumbing for device memory)."""
    import torch

    want = {"int32": torch.int32, "float64": torch.float64}[dtype_name]
    if not t.is_cuda or t.dtype != want or not t.is_contiguous():
        raise ValueError(f"expected a contiguous cuda tensor of dtype {dtype_name}")
    return C.c_void_p(t.data_ptr())


def _producer_done(t):
    """The engine runs on its own non-blocking HIP streams and does not know the stream that produced a caller's
    tensor: the "_device" entry points require their inputs to be COMPLETE at the call (include/knncf.h).  Work queued
    on torch's current stream of that device is drained here, so tensors produced asynchronously are never read stale."""
    import torch

    torch.cuda.current_stream(t.device).synchronize()


class Engine:
    """One knncf handle == one set of the reference's closures over a training set."""

    def __init__(self, k=300, similarity=SIM_COSINE, device=0, shard_rank=0, shard_count=1,
                 workspace_bytes=0, flags=0, head_items=0):
        self._lib = load_library()
        cfg = Config(C.sizeof(Config), device, k, similarity, shard_rank, shard_count, workspace_bytes, flags,
                     head_items)
        h = C.c_void_p()
        st = self._lib.knncf_create(C.byref(cfg), C.byref(h))
        if st != OK:
            raise KnncfError(st, self._lib.knncf_status_string(st).decode())
        self._h = h
        self.device = device
        self.k = k

    def close(self):
        if getattr(self, "_h", None):
            self._lib.knncf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != OK:
            raise KnncfError(st, self._lib.knncf_last_error(self._h).decode())

    # ---- fit ---------------------------------------------------------------------------
    def fit(self, users, items, ratings):
        u, i, r = _i32(users), _i32(items), _f64(ratings)
        if not (len(u) == len(i) == len(r)):
            raise ValueError("users/items/ratings differ in length")
        self._check(self._lib.knncf_fit(self._h, u.ctypes.data_as(_i32p), i.ctypes.data_as(_i32p),
                                        r.ctypes.data_as(_f64p), len(u)))
        return self

    def fit_device(self, users, items, ratings):
        """users/items int32, ratings float64: contiguous torch tensors on this engine's device."""
        n = users.numel()
        _producer_done(users)
        self._check(self._lib.knncf_fit_device(self._h, _dev_ptr(users, "int32"), _dev_ptr(items, "int32"),
                                               _dev_ptr(ratings, "float64"), n))
        return self

    @property
    def num_users(self):
        v = C.c_int32()
        self._check(self._lib.knncf_num_users(self._h, C.byref(v)))
        return v.value

    @property
    def num_items(self):
        v = C.c_int32()
        self._check(self._lib.knncf_num_items(self._h, C.byref(v)))
        return v.value

    # ---- scalar queries ----------------------------------------------------------------
    def _scalar(self, fn, *args):
        v = C.c_double()
        self._check(fn(self._h, *args, C.byref(v)))
        return v.value

    def global_avg(self):
        return self._scalar(self._lib.knncf_global_avg)

    def user_avg(self, u):
        return self._scalar(self._lib.knncf_user_avg, u)

    def item_avg(self, i):
        return self._scalar(self._lib.knncf_item_avg, i)

    def item_avg_dev(self, i):
        return self._scalar(self._lib.knncf_item_avg_dev, i)

    def item_avg_dev_rdd(self, i):
        return self._scalar(self._lib.knncf_item_avg_dev_rdd, i)

    def similarity(self, u, v):
        return self._scalar(self._lib.knncf_similarity, u, v)

    def knn_similarity(self, u, v):
        return self._scalar(self._lib.knncf_knn_similarity, u, v)

    def neighbors(self, u):
        cap = max(1, self.k)
        ids = np.empty(cap, dtype=np.int32)
        sims = np.empty(cap, dtype=np.float64)
        c = C.c_int32()
        self._check(self._lib.knncf_neighbors(self._h, u, cap, ids.ctypes.data_as(_i32p),
                                              sims.ctypes.data_as(_f64p), C.byref(c)))
        return ids[:c.value].copy(), sims[:c.value].copy()

    def neighbors_batch(self, users):
        """getNeighbors for many users at once (knncf_neighbors_batch): (ids [n, k], sims [n, k], counts [n]); cells past
        a row's count are -1 / nan"""
        u = _i32(users)
        cap = max(1, self.k)
        ids = np.full((len(u), cap), -1, dtype=np.int32)
        sims = np.full((len(u), cap), np.nan, dtype=np.float64)
        counts = np.zeros(len(u), dtype=np.int32)
        self._check(self._lib.knncf_neighbors_batch(self._h, u.ctypes.data_as(_i32p), len(u), cap, ids.ctypes.data_as(_i32p),
                                                    sims.ctypes.data_as(_f64p), counts.ctypes.data_as(_i32p)))
        return ids, sims, counts

    def predict(self, predictor, u, i):
        return self._scalar(self._lib.knncf_predict, predictor, u, i)

    def neighbors_save(self, path):
        """checkpoint of the U x k neighbour table (knncf_neighbors_save)"""
        self._check(self._lib.knncf_neighbors_save(self._h, os.fsencode(path)))

    def neighbors_load(self, path):
        """resume from a checkpoint written by a handle fitted on the same data with the same k (knncf_neighbors_load)"""
        self._check(self._lib.knncf_neighbors_load(self._h, os.fsencode(path)))

    def recommend(self, predictor, user, n):
        """recommendations(train, predictor)(user, n) shared/predictions.scala:651-674: (item ids, predictions)"""
        ids = np.empty(max(1, n), dtype=np.int32)
        preds = np.empty(max(1, n), dtype=np.float64)
        c = C.c_int32()
        self._check(self._lib.knncf_recommend(self._h, predictor, user, n, ids.ctypes.data_as(_i32p),
                                              preds.ctypes.data_as(_f64p), C.byref(c)))
        return ids[:c.value].copy(), preds[:c.value].copy()

    # ---- batch -------------------------------------------------------------------------
    def predict_batch(self, predictor, users, items):
        u, i = _i32(users), _i32(items)
        out = np.empty(len(u), dtype=np.float64)
        self._check(self._lib.knncf_predict_batch(self._h, predictor, u.ctypes.data_as(_i32p),
                                                  i.ctypes.data_as(_i32p), len(u), out.ctypes.data_as(_f64p)))
        return out

    def mae(self, predictor, users, items, ratings):
        u, i, r = _i32(users), _i32(items), _f64(ratings)
        v = C.c_double()
        self._check(self._lib.knncf_mae(self._h, predictor, u.ctypes.data_as(_i32p), i.ctypes.data_as(_i32p),
                                        r.ctypes.data_as(_f64p), len(u), C.byref(v)))
        return v.value

    def mae_device(self, predictor, users, items, ratings, pred_out=None):
        """Partial (sum |r - p|, count) over the rows this shard owns; tensors on the device."""
        s, c = C.c_double(), C.c_int64()
        p = _dev_ptr(pred_out, "float64") if pred_out is not None else None
        _producer_done(users)
        self._check(self._lib.knncf_mae_device(self._h, predictor, _dev_ptr(users, "int32"), _dev_ptr(items, "int32"),
                                               _dev_ptr(ratings, "float64"), users.numel(), C.byref(s), C.byref(c), p))
        return s.value, c.value

    # ---- sharding ----------------------------------------------------------------------
    def shard_view(self):
        v = ShardView()
        self._check(self._lib.knncf_shard_view_get(self._h, C.byref(v)))
        return v

    def shard_commit(self):
        self._check(self._lib.knncf_shard_commit(self._h))

    # ---- introspection -----------------------------------------------------------------
    def timings(self):
        t = Timings()
        self._check(self._lib.knncf_get_timings(self._h, C.byref(t)))
        return t.as_dict()

    def reset_timings(self):
        self._check(self._lib.knncf_reset_timings(self._h))

    def reset_neighbors(self):
        self._check(self._lib.knncf_reset_neighbors(self._h))

    def set_k(self, k):
        self._check(self._lib.knncf_set_k(self._h, k))
        self.k = k


class Group:
    """knncf_group_*: one process, several GPUs — n shard handles + RCCL communicators (ncclCommInitAll) inside the library;
    fit = every shard's fit + ncclAllGather of the (mean, norm) segments + commit, mae = every shard's partial sums +
    ncclAllReduce.  What a JVM binds; here the ctypes mirror for the tests."""

    def __init__(self, devices, k=300, similarity=SIM_COSINE, flags=0, head_items=0, workspace_bytes=0):
        self._lib = load_library()
        _share_rccl_with_torch()
        devs = _i32(devices)
        cfg = Config(C.sizeof(Config), 0, k, similarity, 0, 1, workspace_bytes, flags, head_items)
        g = C.c_void_p()
        st = self._lib.knncf_group_create(C.byref(cfg), devs.ctypes.data_as(_i32p), len(devs), C.byref(g))
        if st != OK:
            raise KnncfError(st, self._lib.knncf_status_string(st).decode())
        self._g = g
        self.k = k
        self.size = len(devs)

    def close(self):
        if getattr(self, "_g", None):
            self._lib.knncf_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != OK:
            raise KnncfError(st, self._lib.knncf_group_last_error(self._g).decode())

    def fit(self, users, items, ratings):
        u, i, r = _i32(users), _i32(items), _f64(ratings)
        self._check(self._lib.knncf_group_fit(self._g, u.ctypes.data_as(_i32p), i.ctypes.data_as(_i32p), r.ctypes.data_as(_f64p), len(u)))
        return self

    def mae(self, predictor, users, items, ratings):
        u, i, r = _i32(users), _i32(items), _f64(ratings)
        v = C.c_double()
        self._check(self._lib.knncf_group_mae(self._g, predictor, u.ctypes.data_as(_i32p), i.ctypes.data_as(_i32p),
                                              r.ctypes.data_as(_f64p), len(u), C.byref(v)))
        return v.value

    def predict_batch(self, predictor, users, items):
        u, i = _i32(users), _i32(items)
        out = np.empty(len(u), dtype=np.float64)
        self._check(self._lib.knncf_group_predict_batch(self._g, predictor, u.ctypes.data_as(_i32p), i.ctypes.data_as(_i32p),
                                                        len(u), out.ctypes.data_as(_f64p)))
        return out

    def shard(self, rank):
        """the shard handle of `rank` as a borrowed Engine (queries, timings); owned by the group"""
        h = C.c_void_p()
        self._check(self._lib.knncf_group_handle(self._g, rank, C.byref(h)))
        e = Engine.__new__(Engine)
        e._lib, e._h, e.device, e.k = self._lib, h, None, self.k
        e.close = lambda: None  # borrowed
        return e
