"""ctypes loader for oracle/libtamcmc_oracle.so -- TEST INFRASTRUCTURE ONLY.

Allowed importers: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (as the checker /
the reported CPU baseline, never as the product).  The product path (tamcmc-c-_amd) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, NAN, EMPTY_WINDOW, MODEL_DISABLED, UNKNOWN_MODEL, BAD_LAYOUT = 0, 1, 2, 3, 4, 5


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libtamcmc_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_long)
        L.orc_amplitude_ratio.argtypes = [C.c_int, C.c_double, dp]
        L.orc_amplitude_ratio.restype = None
        L.orc_lin_interpol.argtypes = [dp, dp, C.c_long, C.c_double]
        L.orc_lin_interpol.restype = C.c_double
        L.orc_truncation_window.argtypes = [dp, C.c_long, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, lp, lp]
        L.orc_model.argtypes = [C.c_int, dp, ip, dp, C.c_long, dp]
        L.orc_likelihood_chi22p.argtypes = [dp, dp, C.c_long, C.c_double]
        L.orc_likelihood_chi22p.restype = C.c_double
        L.orc_likelihood_chi_square.argtypes = [dp, dp, dp, C.c_long]
        L.orc_likelihood_chi_square.restype = C.c_double
        L.orc_generate_batch.argtypes = [C.c_int, C.c_int, C.c_double, ip, C.c_long, dp, dp, dp, C.c_int, C.c_int,
                                         dp, dp, dp, ip, dp, C.c_int]
        L.orc_grad_fd.argtypes = [C.c_int, C.c_int, C.c_double, ip, C.c_long, dp, dp, dp, C.c_int, dp, C.c_double,
                                  C.c_int, ip, C.c_double, dp]
        L.orc_grad_fd_wide.argtypes = L.orc_grad_fd.argtypes
        L.orc_grad_analytic.argtypes = [C.c_int, C.c_int, C.c_double, ip, C.c_long, dp, dp, dp, C.c_int, dp, C.c_double,
                                        C.c_int, ip, dp, dp, dp]
        L.orc_grad_analytic_batch.argtypes = [C.c_int, C.c_int, C.c_double, ip, C.c_long, dp, dp, dp, C.c_int, C.c_int,
                                              dp, dp, C.c_int, ip, dp, dp, dp, ip, C.c_int]
        L.orc_amplitude_ratio_closed.argtypes = [C.c_int, C.c_double, dp, dp]
        L.orc_amplitude_ratio_closed.restype = None
        L.orc_noise_harvey1985.argtypes = [dp, C.c_long, dp, dp, C.c_long, C.c_int]
        L.orc_noise_harvey1985.restype = None
        L.orc_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def amplitude_ratio(l, beta_deg):
    out = np.empty(2 * l + 1)
    lib().orc_amplitude_ratio(int(l), float(beta_deg), _d(out))
    return out


def lin_interpol(x, y, x_int):
    x, y = _f64(x), _f64(y)
    return lib().orc_lin_interpol(_d(x), _d(y), x.size, float(x_int))


def truncation_window(x, fc, f_s, gamma, l, c):
    x = _f64(x)
    a, b = C.c_long(0), C.c_long(0)
    st = lib().orc_truncation_window(_d(x), x.size, fc, f_s, gamma, int(l), c, C.byref(a), C.byref(b))
    return st, a.value, b.value


def model(model_case, params, plength, x):
    params, x = _f64(params), _f64(x)
    pl = np.ascontiguousarray(plength, dtype=np.int32)
    out = np.empty(x.size)
    st = lib().orc_model(int(model_case), _d(params), _i(pl), _d(x), x.size, _d(out))
    return out, st


def likelihood_chi22p(y, m, p=1.0):
    y, m = _f64(y), _f64(m)
    return lib().orc_likelihood_chi22p(_d(y), _d(m), y.size, float(p))


def likelihood_chi_square(y, m, sigma):
    y, m, sigma = _f64(y), _f64(m), _f64(sigma)
    return lib().orc_likelihood_chi_square(_d(y), _d(m), _d(sigma), y.size)


def generate_batch(model_case, plength, x, y, params, Tcoefs, sigma_y=None, likelihood_case=0, likelihood_p=1.0,
                   want_models=False, nthreads=0):
    x, y, params, T = _f64(x), _f64(y), _f64(params), _f64(Tcoefs)
    sig = _f64(sigma_y) if sigma_y is not None else None
    pl = np.ascontiguousarray(plength, dtype=np.int32)
    n, npar = params.shape
    logL = np.empty(n)
    status = np.empty(n, dtype=np.int32)
    models = np.empty((n, x.size)) if want_models else None
    lib().orc_generate_batch(int(model_case), int(likelihood_case), float(likelihood_p), _i(pl), x.size, _d(x), _d(y),
                             _d(sig), n, npar, _d(params), _d(T), _d(logL), _i(status), _d(models), int(nthreads))
    return (logL, status, models) if want_models else (logL, status)


def grad_fd(model_case, plength, x, y, params_row, Tcoef, index_to_relax, rel_step=1e-6, sigma_y=None,
            likelihood_case=0, likelihood_p=1.0):
    x, y, p = _f64(x), _f64(y), _f64(params_row)
    sig = _f64(sigma_y) if sigma_y is not None else None
    pl = np.ascontiguousarray(plength, dtype=np.int32)
    idx = np.ascontiguousarray(index_to_relax, dtype=np.int32)
    g = np.empty(idx.size)
    st = lib().orc_grad_fd(int(model_case), int(likelihood_case), float(likelihood_p), _i(pl), x.size, _d(x), _d(y),
                           _d(sig), p.size, _d(p), float(Tcoef), idx.size, _i(idx), float(rel_step), _d(g))
    return g, st


def grad_fd_wide(model_case, plength, x, y, params_row, Tcoef, index_to_relax, rel_step=1e-6, sigma_y=None,
                 likelihood_case=0, likelihood_p=1.0):
    """Richardson central differences of the log-likelihood summed in long double (see orc_grad_fd_wide)."""
    x, y, p = _f64(x), _f64(y), _f64(params_row)
    sig = _f64(sigma_y) if sigma_y is not None else None
    pl = np.ascontiguousarray(plength, dtype=np.int32)
    idx = np.ascontiguousarray(index_to_relax, dtype=np.int32)
    g = np.empty(idx.size)
    st = lib().orc_grad_fd_wide(int(model_case), int(likelihood_case), float(likelihood_p), _i(pl), x.size, _d(x), _d(y),
                                _d(sig), p.size, _d(p), float(Tcoef), idx.size, _i(idx), float(rel_step), _d(g))
    return g, st


def grad_analytic(model_case, plength, x, y, params, Tcoefs, index_to_relax, sigma_y=None, likelihood_case=0,
                  likelihood_p=1.0, nthreads=0):
    """Analytic d(logL/T)/dvars of the oracle for a batch of chains.
    Returns (grad[n, Nvars], grad_abs[n, Nvars] = sum of |terms| of every entry, logL[n], status[n])."""
    x, y, params, T = _f64(x), _f64(y), np.atleast_2d(_f64(params)), np.atleast_1d(_f64(Tcoefs))
    sig = _f64(sigma_y) if sigma_y is not None else None
    pl = np.ascontiguousarray(plength, dtype=np.int32)
    idx = np.ascontiguousarray(index_to_relax, dtype=np.int32)
    n, npar = params.shape
    g = np.empty((n, idx.size))
    ga = np.empty((n, idx.size))
    logL = np.empty(n)
    status = np.empty(n, dtype=np.int32)
    lib().orc_grad_analytic_batch(int(model_case), int(likelihood_case), float(likelihood_p), _i(pl), x.size, _d(x), _d(y),
                                  _d(sig), n, npar, _d(params), _d(T), idx.size, _i(idx), _d(g), _d(ga), _d(logL),
                                  _i(status), int(nthreads))
    return g, ga, logL, status


def amplitude_ratio_closed(l, beta_rad):
    v, d = np.empty(2 * l + 1), np.empty(2 * l + 1)
    lib().orc_amplitude_ratio_closed(int(l), float(beta_rad), _d(v), _d(d))
    return v, d


def max_threads():
    return lib().orc_max_threads()
