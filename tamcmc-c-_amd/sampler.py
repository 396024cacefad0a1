"""ctypes binding of include/tamcmc_sampler.h: log-priors (N1) and the batched adaptive-Metropolis +
parallel-tempering sampler (N2) that calls the hot path once per iteration (MALA.cpp:608-692).

The evaluator is either a HIP context (Accel) or, for tests, any Python callable
f(params[n, Nparams], Tcoefs[n]) -> (logL[n], status[n])."""
import ctypes as C

import numpy as np

from . import capi

MAX_LEARN = 8
EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                      C.POINTER(C.c_double), C.POINTER(C.c_int32))
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32)


class SamplerCfg(C.Structure):
    _fields_ = [("Nchains", C.c_int32), ("chain_offset", C.c_int32), ("Nchains_local", C.c_int32),
                ("lambda_temp", C.c_double), ("target_acceptance", C.c_double), ("c0", C.c_double),
                ("epsilon1", C.c_double), ("epsilon2", C.c_double), ("A1", C.c_double), ("dN_mixing", C.c_int64),
                ("n_learn", C.c_int32), ("Nt_learn", C.c_int64 * MAX_LEARN), ("periods_learn", C.c_int64 * MAX_LEARN),
                ("seed", C.c_uint32), ("prior_fct_switch", C.c_int32)]


_BOUND = False


def _lib():
    global _BOUND
    lib = capi.load_library()
    if not _BOUND:
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
        common = [C.c_int32, ip, dp, ip, ip, dp, C.c_int32, dp, dp]
        lib.tamcmc_sampler_create.argtypes = [C.POINTER(vp), C.POINTER(SamplerCfg), EVAL_FN, vp] + common
        lib.tamcmc_sampler_create_hip.argtypes = [C.POINTER(vp), C.POINTER(SamplerCfg), vp] + common
        for name in ("init", "mh_step", "end_iteration", "destroy"):
            getattr(lib, "tamcmc_sampler_" + name).argtypes = [vp]
        lib.tamcmc_sampler_pt_due.argtypes = [vp]
        lib.tamcmc_sampler_pt_draw.argtypes = [vp, ip, dp]
        lib.tamcmc_sampler_pt_local.argtypes = [vp, C.c_int32, C.c_double, ip, dp]
        lib.tamcmc_sampler_pt_record_size.argtypes = [vp]
        lib.tamcmc_sampler_pt_export.argtypes = [vp, C.c_int32, dp]
        lib.tamcmc_sampler_pt_import.argtypes = [vp, C.c_int32, C.c_double, dp, ip, dp]
        lib.tamcmc_sampler_run.argtypes = [vp, C.c_int64, C.POINTER(C.c_uint8), ip]
        lib.tamcmc_sampler_run_sharded.argtypes = [vp, C.c_int64, EXCHANGE_FN, vp, vp, C.POINTER(C.c_uint8), ip, C.POINTER(C.c_int64)]
        lib.tamcmc_shard_block_create.argtypes = [C.POINTER(vp), vp, C.c_int64]
        lib.tamcmc_shard_block_data.argtypes = [vp, C.c_int32, C.POINTER(dp), C.POINTER(C.c_int64)]
        lib.tamcmc_shard_block_count.argtypes = [vp]
        lib.tamcmc_shard_block_count.restype = C.c_int64
        lib.tamcmc_shard_block_reset.argtypes = [vp]
        lib.tamcmc_shard_block_destroy.argtypes = [vp]
        lib.tamcmc_sampler_set_timing.argtypes = [vp, C.c_int32]
        lib.tamcmc_sampler_get_timing.argtypes = [vp, dp, C.POINTER(C.c_int64)]
        lib.tamcmc_sampler_get.argtypes = [vp, C.c_int32, dp, C.c_int64]
        lib.tamcmc_sampler_iteration.argtypes = [vp]
        lib.tamcmc_sampler_iteration.restype = C.c_int64
        lib.tamcmc_sampler_nvars.argtypes = [vp]
        lib.tamcmc_logP_primitive.argtypes = [C.c_int32, dp, C.c_double]
        lib.tamcmc_logP_primitive.restype = C.c_double
        lib.tamcmc_log_prior.argtypes = [C.c_int32, C.c_int32, dp, ip, ip, dp, C.c_int32, dp, ip]
        lib.tamcmc_log_prior.restype = C.c_double
        lib.tamcmc_normals.argtypes = [C.c_uint32, C.c_int32, ip, dp, C.c_int32]
        lib.tamcmc_normals.restype = None
        lib.tamcmc_glibc_rand.argtypes = [C.c_uint32, C.c_int32, ip]
        lib.tamcmc_glibc_rand.restype = None
        lib.tamcmc_glibc_rand_jump.argtypes = [C.c_uint32, C.c_uint64, C.c_int32, ip]
        lib.tamcmc_glibc_rand_jump.restype = None
        _BOUND = True
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def logP_primitive(prior_id, p, x):
    """One primitive prior of primepriors_ctrl.list (stats_dictionary.cpp)."""
    pp = np.zeros(4)
    pp[:len(p)] = p
    return _lib().tamcmc_logP_primitive(int(prior_id), _dp(pp), float(x))


def log_prior(prior_fct_switch, params, plength, priors_names_switch, priors_params, extra_priors):
    params = np.ascontiguousarray(params, dtype=np.float64)
    pl = np.ascontiguousarray(plength, dtype=np.int32)
    sw = np.ascontiguousarray(priors_names_switch, dtype=np.int32)
    pp = np.ascontiguousarray(priors_params, dtype=np.float64)
    ex = np.ascontiguousarray(extra_priors, dtype=np.float64)
    err = C.c_int32(0)
    v = _lib().tamcmc_log_prior(int(prior_fct_switch), params.size, _dp(params), _ip(pl), _ip(sw), _dp(pp), pp.shape[0],
                                _dp(ex), C.byref(err))
    return v, int(err.value)


def glibc_rand(seed, n):
    out = np.empty(n, dtype=np.int32)
    _lib().tamcmc_glibc_rand(int(seed), n, _ip(out))
    return out


def glibc_rand_jump(seed, skip, n):
    out = np.empty(n, dtype=np.int32)
    _lib().tamcmc_glibc_rand_jump(int(seed), int(skip), n, _ip(out))
    return out


def normals(seed, sizes, split=False):
    sz = np.ascontiguousarray(sizes, dtype=np.int32)
    out = np.empty(int(sz.sum()))
    _lib().tamcmc_normals(int(seed), sz.size, _ip(sz), _dp(out), 1 if split else 0)
    return out


def default_cfg(Nchains, chain_offset=0, Nchains_local=None, Tmax=150.0, seed=1, dN_mixing=1, Nt_learn=(500, 1500, 100000),
                periods_learn=(1, 1), prior_fct_switch=0, c0=10.0):
    """config_default.cfg values (target 0.234, c0 10, eps 1e-12, A1 1e14); lambda so that T_max = Tmax."""
    cfg = SamplerCfg()
    cfg.Nchains = Nchains
    cfg.chain_offset = chain_offset
    cfg.Nchains_local = Nchains if Nchains_local is None else Nchains_local
    cfg.lambda_temp = Tmax ** (1.0 / (Nchains - 1)) if Nchains > 1 else 1.0
    cfg.target_acceptance, cfg.c0, cfg.epsilon1, cfg.epsilon2, cfg.A1 = 0.234, c0, 1e-12, 1e-12, 1e14
    cfg.dN_mixing = dN_mixing
    cfg.n_learn = len(Nt_learn)
    for i, v in enumerate(Nt_learn):
        cfg.Nt_learn[i] = v
    for i, v in enumerate(periods_learn):
        cfg.periods_learn[i] = v
    cfg.seed = seed
    cfg.prior_fct_switch = prior_fct_switch
    return cfg


class Sampler:
    def __init__(self, cfg, evaluator, plength, inputs, relax, err, priors_names_switch=None, priors_params=None,
                 extra_priors=(0.0, 1.0, 1e30, 0.0)):
        self._lib = _lib()
        self.cfg = cfg
        self.plength = np.ascontiguousarray(plength, dtype=np.int32)
        self.inputs = np.ascontiguousarray(inputs, dtype=np.float64)
        self.relax = np.ascontiguousarray(relax, dtype=np.int32)
        self.err = np.ascontiguousarray(err, dtype=np.float64)
        self.Nparams = int(self.inputs.size)
        sw = np.zeros(self.Nparams, dtype=np.int32) if priors_names_switch is None else priors_names_switch
        self.sw = np.ascontiguousarray(sw, dtype=np.int32)
        pp = np.zeros((4, self.Nparams)) if priors_params is None else priors_params
        self.pp = np.ascontiguousarray(pp, dtype=np.float64)
        self.extra = np.ascontiguousarray(extra_priors, dtype=np.float64)
        self._h = C.c_void_p()
        self._cb = None
        common = (self.Nparams, _ip(self.plength), _dp(self.inputs), _ip(self.relax), _ip(self.sw), _dp(self.pp),
                  self.pp.shape[0], _dp(self.extra), _dp(self.err))
        if isinstance(evaluator, capi.Accel):
            self._keep = evaluator
            rc = self._lib.tamcmc_sampler_create_hip(C.byref(self._h), C.byref(cfg), evaluator._ctx, *common)
        else:
            def cb(user, n, npar, p_params, p_T, p_logL, p_status):
                try:
                    P = np.ctypeslib.as_array(p_params, shape=(n, npar))
                    T = np.ctypeslib.as_array(p_T, shape=(n,))
                    logL, st = evaluator(P, T)
                    np.ctypeslib.as_array(p_logL, shape=(n,))[:] = logL
                    np.ctypeslib.as_array(p_status, shape=(n,))[:] = st
                    return 0
                except Exception:
                    return capi.E_INVALID
            self._cb = EVAL_FN(cb)
            rc = self._lib.tamcmc_sampler_create(C.byref(self._h), C.byref(cfg), self._cb, None, *common)
        self._check(rc, "tamcmc_sampler_create")
        self.Nvars = int(self._lib.tamcmc_sampler_nvars(self._h))
        self.nloc = int(cfg.Nchains_local)

    def _check(self, rc, where):
        if rc != 0:
            raise capi.AccelError(rc, where, self._lib.tamcmc_strerror(rc).decode())

    def init(self):
        self._check(self._lib.tamcmc_sampler_init(self._h), "tamcmc_sampler_init")

    def run(self, n_iter, history=True):
        moved = np.zeros((n_iter, self.nloc), dtype=np.uint8) if history else None
        swaps = np.zeros(n_iter, dtype=np.int32) if history else None
        rc = self._lib.tamcmc_sampler_run(self._h, n_iter, moved.ctypes.data_as(C.POINTER(C.c_uint8)) if history else None,
                                          _ip(swaps) if history else None)
        self._check(rc, "tamcmc_sampler_run")
        return moved, swaps

    # pieces for sharded runs
    def mh_step(self):
        self._check(self._lib.tamcmc_sampler_mh_step(self._h), "tamcmc_sampler_mh_step")

    def pt_due(self):
        return bool(self._lib.tamcmc_sampler_pt_due(self._h))

    def pt_draw(self):
        A, u = C.c_int32(0), C.c_double(0)
        self._check(self._lib.tamcmc_sampler_pt_draw(self._h, C.byref(A), C.byref(u)), "pt_draw")
        return int(A.value), float(u.value)

    def pt_local(self, A, u):
        sw, r = C.c_int32(0), C.c_double(0)
        self._check(self._lib.tamcmc_sampler_pt_local(self._h, A, u, C.byref(sw), C.byref(r)), "pt_local")
        return bool(sw.value), float(r.value)

    def pt_export(self, chain):
        rec = np.empty(self._lib.tamcmc_sampler_pt_record_size(self._h))
        self._check(self._lib.tamcmc_sampler_pt_export(self._h, chain, _dp(rec)), "pt_export")
        return rec

    def pt_import(self, A, u, peer_record):
        sw, r = C.c_int32(0), C.c_double(0)
        rec = np.ascontiguousarray(peer_record, dtype=np.float64)
        self._check(self._lib.tamcmc_sampler_pt_import(self._h, A, u, _dp(rec), C.byref(sw), C.byref(r)), "pt_import")
        return bool(sw.value), float(r.value)

    def end_iteration(self):
        self._check(self._lib.tamcmc_sampler_end_iteration(self._h), "end_iteration")

    def iteration(self):
        return int(self._lib.tamcmc_sampler_iteration(self._h))

    def get(self, what):
        which = {"vars": 0, "params": 1, "logL": 2, "logPrior": 3, "logPost": 4, "Pmove": 5, "sigma": 6, "mu": 7,
                 "covarmat": 8, "Tcoefs": 9}[what]
        n, nv = self.nloc, self.Nvars
        shape = {0: (n, nv), 1: (n, self.Nparams), 7: (n, nv), 8: (n, nv, nv)}.get(which, (n,))
        out = np.empty(shape)
        self._check(self._lib.tamcmc_sampler_get(self._h, which, _dp(out), out.size), "tamcmc_sampler_get")
        return out

    def set_timing(self, enable=True):
        self._check(self._lib.tamcmc_sampler_set_timing(self._h, 1 if enable else 0), "tamcmc_sampler_set_timing")

    def timing(self):
        """Seconds per phase since set_timing(True) and the iterations they cover (include/tamcmc_sampler.h)."""
        sec, it = np.zeros(8), C.c_int64(0)
        self._check(self._lib.tamcmc_sampler_get_timing(self._h, _dp(sec), C.byref(it)), "tamcmc_sampler_get_timing")
        names = ("proposals", "launch", "priors", "draw_ahead", "wait", "accept", "foreign_draws", "exchange")
        return dict(zip(names, sec.tolist())), int(it.value)

    def run_sharded(self, n_iter, exchange=None, block=None, history=False):
        """n_iter iterations of the sharded loop inside the library (tamcmc_sampler_run_sharded).  exchange(my_chain,
        peer_chain, send: ndarray) -> ndarray is called only for a boundary pair with one end owned here."""
        moved = np.zeros((n_iter, self.nloc), dtype=np.uint8) if history else None
        swaps = np.zeros(n_iter, dtype=np.int32) if history else None
        err = []

        def cb(user, mine, peer, p_send, p_recv, n):
            try:
                out = exchange(int(mine), int(peer), np.ctypeslib.as_array(p_send, shape=(n,)))
                np.ctypeslib.as_array(p_recv, shape=(n,))[:] = out
                return 0
            except BaseException as e:      # noqa: BLE001 -- must not propagate through the C frame
                err.append(e)
                return 1
        fn = EXCHANGE_FN(cb) if exchange is not None else C.cast(None, EXCHANGE_FN)
        done = C.c_int64(0)
        rc = self._lib.tamcmc_sampler_run_sharded(self._h, n_iter, fn, None, block._h if block is not None else None,
                                                  moved.ctypes.data_as(C.POINTER(C.c_uint8)) if history else None,
                                                  _ip(swaps) if history else None, C.byref(done))
        if err:
            raise err[0]
        self._check(rc, "tamcmc_sampler_run_sharded")
        return int(done.value), moved, swaps

    def close(self):
        if self._h and self._h.value:
            self._lib.tamcmc_sampler_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardBlock:
    """This process's share of an output block, filled by Sampler.run_sharded (tamcmc_shard_block_*)."""
    NAMES = {"vars": 0, "stat": 1, "moved": 2, "pt": 3, "sum_sigma": 4, "sum_mu": 5, "sum_covar": 6, "sum_vars": 7}

    def __init__(self, sampler, capacity):
        self._lib = sampler._lib
        self._h = C.c_void_p()
        sampler._check(self._lib.tamcmc_shard_block_create(C.byref(self._h), sampler._h, int(capacity)), "tamcmc_shard_block_create")
        self.nloc, self.nv = sampler.nloc, sampler.Nvars

    def count(self):
        return int(self._lib.tamcmc_shard_block_count(self._h))

    def data(self, what):
        """A copy of one array of the block, shaped (include/tamcmc_sampler.h)."""
        ptr, cnt = C.POINTER(C.c_double)(), C.c_int64(0)
        rc = self._lib.tamcmc_shard_block_data(self._h, self.NAMES[what], C.byref(ptr), C.byref(cnt))
        if rc != 0:
            raise capi.AccelError(rc, "tamcmc_shard_block_data", "invalid block array")
        n, nl, nv = self.count(), self.nloc, self.nv
        shape = {"vars": (n, nl, nv), "stat": (n, 3, nl), "moved": (n, nl), "pt": (n, 4), "sum_sigma": (nl,), "sum_mu": (nl, nv),
                 "sum_covar": (nl, nv, nv), "sum_vars": (nl, nv)}[what]
        if cnt.value == 0:
            return np.empty(shape)
        return np.ctypeslib.as_array(ptr, shape=(int(cnt.value),)).reshape(shape).copy()

    def reset(self):
        self._lib.tamcmc_shard_block_reset(self._h)

    def close(self):
        if self._h and self._h.value:
            self._lib.tamcmc_shard_block_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def neighbour_exchange(dist, chains_per_rank, device=None):
    """The boundary-pair exchange of a sharded run (MALA.cpp:381-445 for a pair whose chains live on two ranks): one
    record each way between the two owners, torch.distributed send/recv -- RCCL over the neighbours' xGMI link with the
    tensors on `device`, or gloo on the CPU.

    Buffers: one send / recv pair per neighbour, allocated at the first exchange with that neighbour and reused (on a GPU:
    a pinned host pair and a device pair; the record travels numpy -> pinned -> device -> peer and back the same way,
    with no allocation and no pageable copy per call).

    Progress: only the two owners of the drawn pair ever enter this function; every other rank goes on with its next
    iteration.  Send and receive are posted together (batch_isend_irecv = one ncclGroup), so neither owner waits for
    the other's receive to be posted.  Each rank walks the iterations in order and at most one pair is drawn per
    iteration, so the exchanges of two neighbours match in iteration order on both sides, and a rank that waits does so
    on a neighbour that is at an EARLIER OR EQUAL iteration and has nothing later to wait for: the waits cannot form a
    cycle (tests/test_exchange_progress_gloo.py: four ranks, one of them slow)."""
    import torch
    bufs = {}

    def buffers(peer, n):
        b = bufs.get(peer)
        if b is None or b[0].numel() != n:
            hs, hr = torch.empty(n, dtype=torch.float64), torch.empty(n, dtype=torch.float64)
            if device is not None:
                hs, hr = hs.pin_memory(), hr.pin_memory()
                ds, dr = torch.empty(n, dtype=torch.float64, device=device), torch.empty(n, dtype=torch.float64, device=device)
            else:
                ds, dr = hs, hr
            b = bufs[peer] = (hs, hr, ds, dr, hs.numpy(), hr.numpy())
        return b

    def exchange(my_chain, peer_chain, send):
        peer = peer_chain // chains_per_rank
        hs, hr, ds, dr, hs_np, hr_np = buffers(peer, int(send.size))
        hs_np[:] = send
        if device is not None:
            ds.copy_(hs, non_blocking=True)
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, ds, peer), dist.P2POp(dist.irecv, dr, peer)]):
            req.wait()
        if device is not None:
            hr.copy_(dr)                      # device -> pinned host; returns when the record is there
        return hr_np
    exchange.buffers = bufs                   # (tests look at the reuse)
    return exchange


def run_sharded(sampler, n_iter, dist, rank, world, chains_per_rank, device=None):
    """n_iter iterations of a sharded run (the loop itself runs in the library): MH step on the local chains, then the
    parallel-tempering attempt; a boundary pair costs one neighbour send/recv each way.  Returns [(A, swapped)] for
    the attempts this rank took part in."""
    _, _, swaps = sampler.run_sharded(n_iter, neighbour_exchange(dist, chains_per_rank, device), history=True)
    return [(int(v) // 2, bool(v & 1)) for v in swaps if v >= 0]
