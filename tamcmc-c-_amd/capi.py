"""ctypes binding of include/tamcmc_accel.h.

There is no CPU fallback: if libtamcmc_accel.so is missing, or no HIP device is usable, every entry
point raises AccelError.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C tamcmc-c-_amd/csrc``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# error codes of tamcmc_accel.h
OK, E_INVALID, E_NODEVICE, E_HIP, E_MODEL_DISABLED, E_UNKNOWN_MODEL, E_NOMEM, E_NOVARS, E_NOGRAD = range(9)
CHAIN_OK, CHAIN_NAN, CHAIN_EMPTY_WINDOW = 0, 1, 2

EXPORTS = [
    "tamcmc_ctx_create", "tamcmc_ctx_set_vars", "tamcmc_ctx_set_spectra", "tamcmc_ctx_set_chain_spectrum",
    "tamcmc_eval_batch", "tamcmc_eval_batch_device",
    "tamcmc_eval_batch_begin", "tamcmc_eval_batch_end",
    "tamcmc_ctx_reserve", "tamcmc_eval_batch_begin_part", "tamcmc_eval_batch_end_part",
    "tamcmc_eval_batch_arm", "tamcmc_eval_batch_fire", "tamcmc_eval_batch_disarm", "tamcmc_eval_batch_poll",
    "tamcmc_model_explicit", "tamcmc_ctx_set_stream", "tamcmc_ctx_synchronize", "tamcmc_ctx_profile",
    "tamcmc_ctx_kernel_time", "tamcmc_ctx_clock_probe_begin", "tamcmc_ctx_clock_probe_end", "tamcmc_ctx_geometry", "tamcmc_ctx_destroy", "tamcmc_device_count",
    "tamcmc_strerror", "tamcmc_last_hip_error", "tamcmc_version",
]


class AccelError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: error {code}: {detail}")


def library_path():
    """In-tree library; TAMCMC_ACCEL_LIB overrides it (developer knob for A/B builds)."""
    return os.environ.get("TAMCMC_ACCEL_LIB") or os.path.join(_HERE, "libtamcmc_accel.so")


def load_library():
    """Load the HIP library (once).  Raises AccelError if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise AccelError(-1, "load_library", f"{path} not found: build it first (no CPU fallback exists)")
    lib = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    vp = C.c_void_p
    lib.tamcmc_ctx_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_double, ip, C.c_int64, dp, dp, dp]
    lib.tamcmc_ctx_set_vars.argtypes = [vp, C.c_int32, ip]
    lib.tamcmc_eval_batch.argtypes = [vp, C.c_int32, C.c_int32, dp, dp, dp, dp, C.c_int32, ip, dp, ip]
    lib.tamcmc_eval_batch_device.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp]
    lib.tamcmc_model_explicit.argtypes = [vp, C.c_int32, dp, dp, ip]
    lib.tamcmc_ctx_reserve.argtypes = [vp, C.c_int32]
    lib.tamcmc_eval_batch_begin_part.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp, dp]
    lib.tamcmc_eval_batch_end_part.argtypes = [vp, C.c_int32, dp, ip]
    lib.tamcmc_ctx_set_stream.argtypes = [vp, vp]
    lib.tamcmc_ctx_synchronize.argtypes = [vp]
    lib.tamcmc_ctx_profile.argtypes = [vp, C.c_int]
    lib.tamcmc_ctx_kernel_time.argtypes = [vp, dp, C.POINTER(C.c_int64)]
    lib.tamcmc_ctx_clock_probe_begin.argtypes = [vp, C.c_double]
    lib.tamcmc_ctx_clock_probe_end.argtypes = [vp, dp, dp]
    lib.tamcmc_ctx_geometry.argtypes = [vp, ip, ip, ip, ip]
    lib.tamcmc_ctx_destroy.argtypes = [vp]
    lib.tamcmc_device_count.argtypes = []
    lib.tamcmc_strerror.argtypes = [C.c_int]
    lib.tamcmc_strerror.restype = C.c_char_p
    lib.tamcmc_last_hip_error.restype = C.c_char_p
    lib.tamcmc_version.restype = C.c_char_p
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is not C.c_char_p:
            fn.restype = C.c_int
    _LIB = lib
    return lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _c64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


class Accel:
    """One context = one star (x, y[, sigma_y]) on one GPU, for one model / likelihood id."""

    def __init__(self, model_case, plength, x, y, sigma_y=None, likelihood_case=0, likelihood_p=1.0, device_id=0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        self.x = _c64(x)
        self.y = _c64(y, self.x.shape)
        sig = _c64(sigma_y, self.x.shape) if sigma_y is not None else None
        self.plength = np.ascontiguousarray(plength, dtype=np.int32)
        if self.plength.shape != (11,):
            raise ValueError("plength must have 11 entries")
        self.Nparams = int(self.plength.sum())
        self.Nx = int(self.x.size)
        self.Nvars = 0
        self.model_case = int(model_case)
        rc = self._lib.tamcmc_ctx_create(C.byref(self._ctx), int(device_id), int(model_case), int(likelihood_case),
                                         float(likelihood_p), _iptr(self.plength), self.Nx, _dptr(self.x), _dptr(self.y),
                                         _dptr(sig))
        self._check(rc, "tamcmc_ctx_create")

    # -- plumbing -------------------------------------------------------------------------------
    def _check(self, rc, where):
        if rc != OK:
            detail = self._lib.tamcmc_strerror(rc).decode()
            if rc == E_HIP:
                detail += " | " + self._lib.tamcmc_last_hip_error().decode()
            raise AccelError(rc, where, detail)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.tamcmc_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- API ------------------------------------------------------------------------------------
    def set_vars(self, index_to_relax):
        idx = np.ascontiguousarray(index_to_relax, dtype=np.int32)
        self._check(self._lib.tamcmc_ctx_set_vars(self._ctx, idx.size, _iptr(idx)), "tamcmc_ctx_set_vars")
        self.Nvars = int(idx.size)

    def set_spectra(self, y, sigma_y=None):
        """Several spectra on the context's grid (rows of y); chains choose theirs with set_chain_spectrum."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        if y.ndim != 2 or y.shape[1] != self.Nx:
            raise ValueError("y must be (Nspectra, Nx)")
        sg = None if sigma_y is None else np.ascontiguousarray(sigma_y, dtype=np.float64)
        if sg is not None and sg.shape != y.shape:
            raise ValueError("sigma_y must have the shape of y")
        self._check(self._lib.tamcmc_ctx_set_spectra(self._ctx, y.shape[0], _dptr(y), _dptr(sg)), "tamcmc_ctx_set_spectra")

    def set_chain_spectrum(self, spectrum_of_chain):
        m = np.ascontiguousarray(spectrum_of_chain, dtype=np.int32)
        self._check(self._lib.tamcmc_ctx_set_chain_spectrum(self._ctx, m.size, _iptr(m)), "tamcmc_ctx_set_chain_spectrum")

    def eval_batch(self, params, Tcoefs, grad=False, model_rows=None):
        """Returns (logL, status[, grad][, models])."""
        params = _c64(params)
        if params.ndim != 2 or params.shape[1] != self.Nparams:
            raise ValueError(f"params must be (Nchains, {self.Nparams})")
        n = params.shape[0]
        T = _c64(Tcoefs, (n,))
        logL = np.empty(n)
        status = np.empty(n, dtype=np.int32)
        g = np.empty((n, self.Nvars)) if grad else None
        rows = np.ascontiguousarray(model_rows, dtype=np.int32) if model_rows is not None else None
        models = np.empty((rows.size, self.Nx)) if rows is not None else None
        rc = self._lib.tamcmc_eval_batch(self._ctx, n, self.Nparams, _dptr(params), _dptr(T), _dptr(logL), _dptr(g),
                                         rows.size if rows is not None else 0, _iptr(rows), _dptr(models), _iptr(status))
        self._check(rc, "tamcmc_eval_batch")
        out = [logL, status]
        if grad:
            out.append(g)
        if rows is not None:
            out.append(models)
        return tuple(out)

    def reserve(self, nchains):
        self._check(self._lib.tamcmc_ctx_reserve(self._ctx, int(nchains)), "tamcmc_ctx_reserve")

    def begin_part(self, part, first, params, Tcoefs):
        """Likelihood evaluation of chains [first, first + len(params)) as part 0 / 1; both parts may be in flight together."""
        params = _c64(params)
        if params.ndim != 2 or params.shape[1] != self.Nparams:
            raise ValueError(f"expected (n, {self.Nparams}) params, got {params.shape}")
        T = _c64(Tcoefs, (params.shape[0],))
        self._part_n = getattr(self, "_part_n", {})
        self._part_n[int(part)] = params.shape[0]
        self._check(self._lib.tamcmc_eval_batch_begin_part(self._ctx, int(part), int(first), params.shape[0], self.Nparams,
                                                           _dptr(params), _dptr(T)), "tamcmc_eval_batch_begin_part")

    def end_part(self, part):
        n = self._part_n[int(part)]
        logL = np.empty(n)
        st = np.empty(n, dtype=np.int32)
        self._check(self._lib.tamcmc_eval_batch_end_part(self._ctx, int(part), _dptr(logL), st.ctypes.data_as(C.POINTER(C.c_int32))),
                    "tamcmc_eval_batch_end_part")
        return logL, st

    def begin(self, params, Tcoefs):
        """Likelihood evaluation of a batch, launched and not waited for (end() collects it)."""
        params = _c64(params)
        if params.ndim != 2 or params.shape[1] != self.Nparams:
            raise ValueError(f"expected (n, {self.Nparams}) params, got {params.shape}")
        T = _c64(Tcoefs, (params.shape[0],))
        self._check(self._lib.tamcmc_eval_batch_begin(self._ctx, params.shape[0], self.Nparams, _dptr(params), _dptr(T)), "tamcmc_eval_batch_begin")
        self._n_flight = params.shape[0]

    def end(self):
        n = self._n_flight
        logL = np.empty(n)
        st = np.empty(n, dtype=np.int32)
        self._check(self._lib.tamcmc_eval_batch_end(self._ctx, n, _dptr(logL), st.ctypes.data_as(C.POINTER(C.c_int32))), "tamcmc_eval_batch_end")
        return logL, st

    def poll(self, chain):
        """(logL, status) of one chain of the batch in flight, or None while it has not arrived."""
        L, st = C.c_double(), C.c_int32()
        rc = self._lib.tamcmc_eval_batch_poll(self._ctx, int(chain), C.byref(L), C.byref(st))
        if rc == -1:
            return None
        self._check(rc, "tamcmc_eval_batch_poll")
        return L.value, st.value

    def arm(self, nchains):
        """The launches of the next likelihood batch go into the stream now, behind a gate; fire() supplies the parameters."""
        self._check(self._lib.tamcmc_eval_batch_arm(self._ctx, int(nchains)), "tamcmc_eval_batch_arm")

    def fire(self, params, Tcoefs):
        params = _c64(params)
        if params.ndim != 2 or params.shape[1] != self.Nparams:
            raise ValueError(f"expected (n, {self.Nparams}) params, got {params.shape}")
        T = _c64(Tcoefs, (params.shape[0],))
        self._check(self._lib.tamcmc_eval_batch_fire(self._ctx, params.shape[0], self.Nparams, _dptr(params), _dptr(T)), "tamcmc_eval_batch_fire")
        self._n_flight = params.shape[0]

    def disarm(self):
        self._check(self._lib.tamcmc_eval_batch_disarm(self._ctx), "tamcmc_eval_batch_disarm")

    def eval_batch_device(self, nchains, d_params, d_T, d_logL, d_grad=0, d_status=0):
        """Device pointers (ints, e.g. torch.Tensor.data_ptr()); enqueued on the ctx stream, no sync."""
        rc = self._lib.tamcmc_eval_batch_device(self._ctx, int(nchains), self.Nparams, C.c_void_p(d_params),
                                                C.c_void_p(d_T), C.c_void_p(d_logL),
                                                C.c_void_p(d_grad) if d_grad else None,
                                                C.c_void_p(d_status) if d_status else None)
        self._check(rc, "tamcmc_eval_batch_device")

    def model_explicit(self, params):
        params = _c64(params, (self.Nparams,))
        out = np.empty(self.Nx)
        st = C.c_int32(0)
        self._check(self._lib.tamcmc_model_explicit(self._ctx, self.Nparams, _dptr(params), _dptr(out), C.byref(st)),
                    "tamcmc_model_explicit")
        return out, int(st.value)

    def set_stream(self, hip_stream):
        self._check(self._lib.tamcmc_ctx_set_stream(self._ctx, C.c_void_p(hip_stream) if hip_stream else None),
                    "tamcmc_ctx_set_stream")

    def synchronize(self):
        self._check(self._lib.tamcmc_ctx_synchronize(self._ctx), "tamcmc_ctx_synchronize")

    def profile(self, enable=True):
        self._check(self._lib.tamcmc_ctx_profile(self._ctx, int(enable)), "tamcmc_ctx_profile")

    def kernel_time(self):
        ms = C.c_double(0.0)
        n = C.c_int64(0)
        self._check(self._lib.tamcmc_ctx_kernel_time(self._ctx, C.byref(ms), C.byref(n)), "tamcmc_ctx_kernel_time")
        return ms.value, n.value

    def clock_probe_begin(self, milliseconds):
        self._check(self._lib.tamcmc_ctx_clock_probe_begin(self._ctx, float(milliseconds)), "tamcmc_ctx_clock_probe_begin")

    def clock_probe_end(self):
        """(core clock in GHz, seconds observed) of the probe started by clock_probe_begin."""
        ghz, sec = C.c_double(0.0), C.c_double(0.0)
        self._check(self._lib.tamcmc_ctx_clock_probe_end(self._ctx, C.byref(ghz), C.byref(sec)), "tamcmc_ctx_clock_probe_end")
        return ghz.value, sec.value

    def geometry(self):
        v = [C.c_int32(0) for _ in range(4)]
        self._check(self._lib.tamcmc_ctx_geometry(self._ctx, *[C.byref(e) for e in v]), "tamcmc_ctx_geometry")
        return dict(bins_per_tile=v[0].value, tiles=v[1].value, threads_per_block=v[2].value, n_multiplets=v[3].value)


def device_count():
    return int(load_library().tamcmc_device_count())


def version():
    return load_library().tamcmc_version().decode()
