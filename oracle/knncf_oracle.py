"""ctypes loader for the CPU oracle (oracle/knncf_oracle.c).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libknncf_oracle.so")

SIM_COSINE, SIM_ONE, SIM_JACCARD = 0, 1, 2
KIND_GLOBAL, KIND_USER, KIND_ITEM, KIND_BASELINE, KIND_BASELINE_SPARK = 0, 1, 2, 3, 4

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


def build():
    """Compile the oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "knncf_oracle.c")
    hdr = os.path.join(_HERE, "knncf_oracle.h")
    if os.path.exists(_LIB_PATH) and os.path.getmtime(_LIB_PATH) >= max(
        os.path.getmtime(src), os.path.getmtime(hdr)
    ):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "libknncf_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    L.orc_improve.restype = C.c_uint32
    L.orc_improve.argtypes = [C.c_uint32]
    L.orc_trie_key.restype = C.c_uint32
    L.orc_trie_key.argtypes = [C.c_uint32]
    L.orc_tuple2_hash.restype = C.c_uint32
    L.orc_tuple2_hash.argtypes = [C.c_int32, C.c_int32]
    L.orc_int_set_order.restype = None
    L.orc_int_set_order.argtypes = [_i32p, C.c_int32, _i32p]
    L.orc_scale.restype = C.c_double
    L.orc_scale.argtypes = [C.c_double, C.c_double]
    L.orc_fit.restype = C.c_void_p
    L.orc_fit.argtypes = [_i32p, _i32p, _f64p, C.c_int64, C.POINTER(C.c_int)]
    L.orc_free.restype = None
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_num_users.restype = C.c_int32
    L.orc_num_users.argtypes = [C.c_void_p]
    L.orc_num_items.restype = C.c_int32
    L.orc_num_items.argtypes = [C.c_void_p]
    L.orc_user_iteration_order.restype = None
    L.orc_user_iteration_order.argtypes = [C.c_void_p, _i32p]
    L.orc_average.restype = C.c_double
    L.orc_average.argtypes = [C.c_void_p]
    for name in ("orc_users_avg", "orc_items_avg", "orc_items_avg_dev", "orc_items_avg_dev_spark",
                 "orc_user_weight"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int32, _f64p]
    L.orc_normalized_deviations.restype = _f64p
    L.orc_normalized_deviations.argtypes = [C.c_void_p]
    L.orc_preprocessed.restype = _f64p
    L.orc_preprocessed.argtypes = [C.c_void_p]
    for name in ("orc_predict_global", "orc_predict_user_avg", "orc_predict_item_avg",
                 "orc_predict_item_avg_dev", "orc_predict_baseline", "orc_predict_baseline_spark"):
        f = getattr(L, name)
        f.restype = C.c_double
        f.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.orc_mae_simple.restype = C.c_double
    L.orc_mae_simple.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, _f64p, C.c_int64, _f64p]
    L.orc_pipeline_create.restype = C.c_void_p
    L.orc_pipeline_create.argtypes = [C.c_void_p, C.c_int, C.c_int32]
    L.orc_pipeline_free.restype = None
    L.orc_pipeline_free.argtypes = [C.c_void_p]
    for name in ("orc_pipeline_raw_similarity", "orc_pipeline_knn_similarity", "orc_pipeline_wsd",
                 "orc_pipeline_predict"):
        f = getattr(L, name)
        f.restype = C.c_double
        f.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.orc_fresh_similarity.restype = C.c_double
    L.orc_fresh_similarity.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32]
    L.orc_pipeline_neighbors.restype = C.c_int32
    L.orc_pipeline_neighbors.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _i32p, _f64p]
    L.orc_pipeline_mae.restype = C.c_double
    L.orc_pipeline_mae.argtypes = [C.c_void_p, _i32p, _i32p, _f64p, C.c_int64, _f64p]
    L.orc_recommend.restype = C.c_int32
    L.orc_recommend.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int32, C.c_int32, _i32p, _f64p]
    L.orc_knn_table_build.restype = C.c_void_p
    L.orc_knn_table_build.argtypes = [C.c_void_p, C.c_int32, _i32p, C.c_int32, C.c_int, C.POINTER(C.c_int)]
    L.orc_knn_table_free.restype = None
    L.orc_knn_table_free.argtypes = [C.c_void_p]
    L.orc_knn_table_width.restype = C.c_int32
    L.orc_knn_table_width.argtypes = [C.c_void_p]
    L.orc_knn_table_rows.restype = C.c_int32
    L.orc_knn_table_rows.argtypes = [C.c_void_p]
    L.orc_knn_table_sims.restype = _f64p
    L.orc_knn_table_sims.argtypes = [C.c_void_p]
    L.orc_knn_table_export_ids.restype = None
    L.orc_knn_table_export_ids.argtypes = [C.c_void_p, _i32p, _i32p]
    L.orc_knn_table_predict.restype = C.c_double
    L.orc_knn_table_predict.argtypes = [C.c_void_p, _i32p, _i32p, _f64p, C.c_int64, C.c_int, _f64p, C.POINTER(C.c_int)]
    L.orc_max_threads.restype = C.c_int
    L.orc_set_threads.restype = None
    L.orc_set_threads.argtypes = [C.c_int]
    _lib = L
    return L


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(t)


class OracleError(RuntimeError):
    def __init__(self, status):
        super().__init__(f"oracle status {status}")
        self.status = status


def improve(h):
    return lib().orc_improve(h & 0xFFFFFFFF)


def trie_key(h):
    return lib().orc_trie_key(h & 0xFFFFFFFF)


def tuple2_hash(a, b):
    return lib().orc_tuple2_hash(a, b)


def int_set_order(ids):
    a = _i32(ids)
    out = np.empty_like(a)
    lib().orc_int_set_order(_p(a, _i32p), len(a), _p(out, _i32p))
    return out.tolist()


def scale(x, y):
    return lib().orc_scale(x, y)


class Model:
    """The training set as captured by the reference's closures."""

    def __init__(self, users, items, ratings):
        self.users, self.items, self.ratings = _i32(users), _i32(items), _f64(ratings)
        st = C.c_int(0)
        self._h = lib().orc_fit(_p(self.users, _i32p), _p(self.items, _i32p),
                                _p(self.ratings, _f64p), len(self.users), C.byref(st))
        if not self._h:
            raise OracleError(st.value)
        self.n = len(self.users)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_free(self._h)
            self._h = None

    @property
    def num_users(self):
        return lib().orc_num_users(self._h)

    @property
    def num_items(self):
        return lib().orc_num_items(self._h)

    def user_iteration_order(self):
        out = np.empty(self.num_users, dtype=np.int32)
        lib().orc_user_iteration_order(self._h, _p(out, _i32p))
        return out

    def average(self):
        return lib().orc_average(self._h)

    def _opt(self, fn, key):
        v = C.c_double(0)
        return v.value if fn(self._h, key, C.byref(v)) else None

    def users_avg(self, u):
        return self._opt(lib().orc_users_avg, u)

    def items_avg(self, i):
        return self._opt(lib().orc_items_avg, i)

    def items_avg_dev(self, i):
        return self._opt(lib().orc_items_avg_dev, i)

    def items_avg_dev_spark(self, i):
        return self._opt(lib().orc_items_avg_dev_spark, i)

    def user_weight(self, u):
        return self._opt(lib().orc_user_weight, u)

    def normalized_deviations(self):
        return np.ctypeslib.as_array(lib().orc_normalized_deviations(self._h), shape=(self.n,)).copy()

    def preprocessed(self):
        return np.ctypeslib.as_array(lib().orc_preprocessed(self._h), shape=(self.n,)).copy()

    def predict(self, kind, u, i):
        f = [lib().orc_predict_global, lib().orc_predict_user_avg, lib().orc_predict_item_avg,
             lib().orc_predict_baseline, lib().orc_predict_baseline_spark][kind]
        return f(self._h, u, i)

    def predict_item_avg_dev(self, u, i):
        return lib().orc_predict_item_avg_dev(self._h, u, i)

    def mae(self, kind, users, items, ratings, return_predictions=False):
        u, i, r = _i32(users), _i32(items), _f64(ratings)
        per = np.empty(len(u), dtype=np.float64)
        m = lib().orc_mae_simple(self._h, kind, _p(u, _i32p), _p(i, _i32p), _p(r, _f64p), len(u),
                                 _p(per, _f64p))
        return (m, per) if return_predictions else m

    def fresh_similarity(self, sim_kind, u, v):
        return lib().orc_fresh_similarity(self._h, sim_kind, u, v)

    def recommend(self, kind, user, n):
        """recommendations(ratings, <closed-form predictor kind>)(user, n) shared/predictions.scala:651-674"""
        ids = np.empty(max(1, n), dtype=np.int32)
        preds = np.empty(max(1, n), dtype=np.float64)
        c = lib().orc_recommend(self._h, None, kind, user, n, _p(ids, _i32p), _p(preds, _f64p))
        if c < 0:
            raise OracleError(c)
        return ids[:c].copy(), preds[:c].copy()

    def pipeline(self, sim_kind=SIM_COSINE, k=-1):
        return Pipeline(self, sim_kind, k)

    def knn_table(self, k, users=None, threads=0):
        """The kNN closures (cosine, k) for many users at once on all cores (every user when users is None)."""
        return KnnTable(self, k, users, threads)


def set_threads(n):
    """cap the threads of orc_fit's parallel loops and of the bulk form (n <= 0: all cores)"""
    lib().orc_set_threads(int(n))


def host_threads(cap=None):
    """Threads the bulk evaluation may use: every CPU this process is allowed on (the GPU box hands out a share); cap, if
    given, bounds it."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, n if cap is None else min(n, cap))


def cpu_model():
    """the host CPU's model name (for the baseline's `sample` text)"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


class KnnTable:
    """orc_knn_table: getNeighbors(train, k, adjustedCosine) of many users + the predictor over them, threaded.
    Refused (OracleError) when a user has <= 4 ratings: use Pipeline, which models the memo history."""

    def __init__(self, model, k, users=None, threads=0):
        self.model = model
        self.threads = threads if threads > 0 else host_threads()
        st = C.c_int(0)
        if users is None:
            self._h = lib().orc_knn_table_build(model._h, k, None, 0, self.threads, C.byref(st))
        else:
            u = _i32(users)
            self._h = lib().orc_knn_table_build(model._h, k, _p(u, _i32p), len(u), self.threads, C.byref(st))
        if not self._h:
            raise OracleError(st.value)
        self.rows = lib().orc_knn_table_rows(self._h)
        self.width = lib().orc_knn_table_width(self._h)
        cells = self.rows * self.width
        self.ids = np.empty((self.rows, self.width), dtype=np.int32)   # raw neighbour ids, reference order
        self.row_user = np.empty(self.rows, dtype=np.int32)             # raw user of each row
        lib().orc_knn_table_export_ids(self._h, _p(self.ids, _i32p), _p(self.row_user, _i32p))
        self.sims = (np.ctypeslib.as_array(lib().orc_knn_table_sims(self._h), shape=(max(cells, 1),))[:cells]
                     .reshape(self.rows, self.width))                    # view of the table's memory

    def __del__(self):
        if getattr(self, "_h", None):
            self.sims = None
            lib().orc_knn_table_free(self._h)
            self._h = None

    def mae(self, users, items, ratings):
        """(MAE as the reference's left fold, per-row predictions)"""
        u, i, r = _i32(users), _i32(items), _f64(ratings)
        per = np.empty(len(u), dtype=np.float64)
        st = C.c_int(0)
        m = lib().orc_knn_table_predict(self._h, _p(u, _i32p), _p(i, _i32p), _p(r, _f64p), len(u), self.threads,
                                        _p(per, _f64p), C.byref(st))
        if st.value != 0:
            raise OracleError(st.value)
        return m, per


class Pipeline:
    """predictor(train, weightedSumDeviation(train, F)); F is the similarity
    itself (k < 0) or getSimilarity(train, k, similarity) (k >= 0)."""

    def __init__(self, model, sim_kind, k):
        self.model = model
        self.k = k
        self._h = lib().orc_pipeline_create(model._h, sim_kind, k)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_pipeline_free(self._h)
            self._h = None

    def raw_similarity(self, u, v):
        return lib().orc_pipeline_raw_similarity(self._h, u, v)

    def knn_similarity(self, u, v):
        return lib().orc_pipeline_knn_similarity(self._h, u, v)

    def neighbors(self, u):
        cap = max(1, min(self.k if self.k >= 0 else 0, self.model.num_users))
        ids = np.empty(cap, dtype=np.int32)
        sims = np.empty(cap, dtype=np.float64)
        c = lib().orc_pipeline_neighbors(self._h, u, cap, _p(ids, _i32p), _p(sims, _f64p))
        return ids[:c].copy(), sims[:c].copy()

    def wsd(self, u, i):
        return lib().orc_pipeline_wsd(self._h, u, i)

    def predict(self, u, i):
        return lib().orc_pipeline_predict(self._h, u, i)

    def recommend(self, user, n):
        """recommendations(ratings, predictor(..., this pipeline))(user, n) shared/predictions.scala:651-674"""
        ids = np.empty(max(1, n), dtype=np.int32)
        preds = np.empty(max(1, n), dtype=np.float64)
        c = lib().orc_recommend(self.model._h, self._h, 0, user, n, _p(ids, _i32p), _p(preds, _f64p))
        if c < 0:
            raise OracleError(c)
        return ids[:c].copy(), preds[:c].copy()

    def mae(self, users, items, ratings, return_predictions=False):
        u, i, r = _i32(users), _i32(items), _f64(ratings)
        per = np.empty(len(u), dtype=np.float64)
        m = lib().orc_pipeline_mae(self._h, _p(u, _i32p), _p(i, _i32p), _p(r, _f64p), len(u),
                                   _p(per, _f64p))
        return (m, per) if return_predictions else m
