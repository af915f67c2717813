"""Multi-GPU host logic: one process per GPU, users block-partitioned, collectives by torch.distributed.

Each rank owns a contiguous block of dense users.  Per step:
  1. every rank fits on the (replicated) training triples; the per-user means / norms (ordered fp64
     folds: the expensive, order-sensitive part of K2/K3) are computed for the OWNED users only;
  2. the one exchange step: all-gather of the owned segments of {user mean, user norm} (RCCL
     all-gather over xGMI, 16 B per user).  The normalized deviations and preprocessed ratings of
     the other ranks' users are elementwise functions of (rating, mean) and (deviation, norm): every
     rank recomputes them bit for bit in knncf_shard_commit instead of receiving 16 B per RATING
     (320 MB at the ml-25m shape) — after it every rank can densify any user's row for the
     similarity GEMM's B operand and gather any neighbour's deviation;
  3. every rank builds complete top-k rows for its own users (no cross-GPU merge) and predicts the
     test ratings of its own users;
  4. all-reduce of (sum |r - p|, count) -> MAE.
Failures are collective: a rank whose fit or neighbour build fails makes every rank raise (an
all-reduce of the status code) instead of leaving the others blocked in the next collective.
The engine object only needs: fit_device, shard_tensors, shard_commit, mae_device — the GPU engine
(knncf.Engine through DeviceEngineAdapter) on the GPU box, an oracle-backed double in the CPU tests.
"""
import numpy as np


class _CudaArrayView:
    """Zero-copy view of library-owned device memory for torch (plumbing only)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def wrap_device_f64(ptr, n, device):
    import torch

    if n == 0:
        return torch.empty(0, dtype=torch.float64, device=device)
    return torch.as_tensor(_CudaArrayView(ptr, n, "<f8"), device=device)


class DeviceEngineAdapter:
    """knncf.Engine with the shard buffers exposed as torch tensors."""

    def __init__(self, engine, device):
        self.engine = engine
        self.device = device
        self._wrapped = {}  # (pointer, length) -> tensor view: the library keeps its buffers across re-fits of the same shape

    def fit_device(self, users, items, ratings):
        self.engine.fit_device(users, items, ratings)

    def _wrap(self, ptr, n):
        key = (int(ptr or 0), int(n))
        t = self._wrapped.get(key)
        if t is None:
            if len(self._wrapped) > 64:
                self._wrapped.clear()
            t = self._wrapped[key] = wrap_device_f64(ptr, n, self.device)
        return t

    def shard_tensors(self):
        v = self.engine.shard_view()
        return {
            "user_range": (v.user_begin, v.user_end),
            "nnz_range": (v.nnz_begin, v.nnz_end),
            "user_avg": self._wrap(v.d_user_avg, v.num_users),
            "user_norm": self._wrap(v.d_user_norm, v.num_users),
            "dev": self._wrap(v.d_dev, v.num_ratings),
            "pre": self._wrap(v.d_pre, v.num_ratings),
        }

    def shard_commit(self):
        self.engine.shard_commit()

    def mae_device(self, predictor, users, items, ratings):
        return self.engine.mae_device(predictor, users, items, ratings)


def user_block(num_users, rank, world):
    """Owned dense users [lo, hi): ceil(U / world) per rank, ascending dense index."""
    per = -(-num_users // world)
    return min(per * rank, num_users), min(per * (rank + 1), num_users)


def _staged(dist, t):
    """gloo rehearsals of the multi-GPU path (several ranks sharing one GPU, or CPU tests): gloo moves host memory, so
    device tensors are staged through the host; with nccl (RCCL over xGMI) tensors travel as they are."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_gather_into(dist, recv, send):
    if _staged(dist, send):
        r = recv.cpu()
        dist.all_gather_into_tensor(r, send.cpu())
        recv.copy_(r)
    else:
        dist.all_gather_into_tensor(recv, send)


def _all_reduce(dist, t, op=None):
    kw = {} if op is None else {"op": op}
    if _staged(dist, t):
        c = t.cpu()
        dist.all_reduce(c, **kw)
        t.copy_(c)
    else:
        dist.all_reduce(t, **kw)


class ShardFitError(RuntimeError):
    """Another rank's fit (or neighbour build) failed; every rank raises so that none blocks in the next collective."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


class ShardedKnn:
    """collective=True runs the whole collective protocol even at world 1 (a one-rank process group): the rehearsal of
    the RCCL path on a single GPU — same calls, same tensors, no peer."""

    def __init__(self, engine, dist=None, rank=0, world=1, collective=None):
        self.engine, self.dist, self.rank, self.world = engine, dist, rank, world
        self.collective = (world > 1) if collective is None else bool(collective)
        if self.collective and dist is None:
            raise ValueError("collective mode needs an initialised torch.distributed process group")

    def fit(self, users, items, ratings):
        """fit + the one exchange + commit.  The fit STATUS is collective: a rank whose part of the fit fails (a
        non-finite deviation among ITS users, a device error, out of memory) must not leave the others waiting in the
        all-gather, so every rank first learns whether all of them succeeded and all of them raise otherwise."""
        err = None
        try:
            self.engine.fit_device(users, items, ratings)
        except Exception as e:  # reported collectively below
            err = e
        if self.collective:
            self._raise_together(err, users.device, "fit")
            self.exchange()
        elif err is not None:
            raise err
        self.engine.shard_commit()

    def _raise_together(self, err, device, what):
        import torch

        code = 0 if err is None else int(getattr(err, "status", -1)) or -1
        # gloo moves host memory; with nccl the flag lives on the device (if THAT fails the device is gone and the
        # process group with it: nothing collective is left to do, the local error is raised)
        try:
            flag = torch.tensor([code], dtype=torch.int64, device="cpu" if self.dist.get_backend() == "gloo" else device)
        except Exception:
            if err is not None:
                raise err
            raise
        _all_reduce(self.dist, flag, op=self.dist.ReduceOp.MIN)   # status codes are negative: MIN = "the worst"
        worst = int(flag.item())
        if err is not None:
            raise err
        if worst != 0:
            raise ShardFitError(worst, f"{what} failed on another rank (status {worst}); this rank's part was fine")

    def exchange(self):
        """All-gather of the owned (mean, norm) segments, in place into the library's arrays.  Every rank derives every
        rank's user range from the user count alone (user_block), so no metadata travels; the segments differ in length
        by at most one user and go in ONE padded all_gather_into_tensor."""
        import torch

        t = self.engine.shard_tensors()
        ulo, uhi = t["user_range"]
        avg, norm = t["user_avg"], t["user_norm"]
        U = avg.numel()
        ranges = [user_block(U, r, self.world) for r in range(self.world)]
        if ranges[self.rank] != (ulo, uhi):
            raise RuntimeError(f"rank {self.rank}: the engine owns users [{ulo}, {uhi}), the partition says {ranges[self.rank]}")
        seg = max(h - l for l, h in ranges)
        if seg > 0:
            send = torch.zeros(2 * seg, dtype=avg.dtype, device=avg.device)
            send[:uhi - ulo] = avg[ulo:uhi]
            send[seg:seg + uhi - ulo] = norm[ulo:uhi]
            recv = torch.empty(self.world * 2 * seg, dtype=send.dtype, device=send.device)
            _all_gather_into(self.dist, recv, send)
            for r, (l, h) in enumerate(ranges):
                if r == self.rank:
                    continue
                base = r * 2 * seg
                avg[l:h] = recv[base: base + (h - l)]
                norm[l:h] = recv[base + seg: base + seg + (h - l)]
        # the engine works on its own (non-blocking) HIP stream: the gathered segments must have landed before
        # shard_commit launches the kernels that read them
        if avg.device.type == "cuda":
            torch.cuda.current_stream(avg.device).synchronize()

    def mae(self, predictor, users, items, ratings):
        """All-reduced MAE of the whole test set; every rank receives the same value.  The neighbour build runs inside
        this call (lazily, like the reference's closures) and is the memory-hungry stage: its status is collective too."""
        import torch

        err, s, c = None, 0.0, 0
        try:
            s, c = self.engine.mae_device(predictor, users, items, ratings)
        except Exception as e:
            err = e
        if self.collective:
            # ONE all-reduce carries the partial sums and the status: (sum |r - p|, rows, failed ranks, sum of their codes)
            code = 0 if err is None else int(getattr(err, "status", -1)) or -1
            dev = "cpu" if self.dist.get_backend() == "gloo" else users.device
            try:
                buf = torch.tensor([s, float(c), 0.0 if err is None else 1.0, float(code)], dtype=torch.float64, device=dev)
            except Exception:
                if err is not None:
                    raise err
                raise
            _all_reduce(self.dist, buf)
            s, c, failed, codes = (float(x) for x in buf.tolist())
            c = int(round(c))
            if err is not None:
                raise err
            if failed > 0:
                worst = int(round(codes / failed))  # (exact when the failing ranks agree, e.g. all out of memory)
                raise ShardFitError(worst, f"the neighbour build / prediction failed on {int(failed)} other rank(s) (status {worst})")
        elif err is not None:
            raise err
        return (s / c if c else float("nan")), c
