"""ctypes binding of include/tamcmc_io.h: the reference's file formats (.model, .data, config_default.cfg,
errors_default.cfg, *_ctrl.list) -> the arrays the hot path and the sampler take (Config::setup, config.cpp:80-189)."""
import ctypes as C

import numpy as np

from . import capi
from .sampler import SamplerCfg

IO_OK, IO_E_INVALID, IO_E_OPEN, IO_E_SYNTAX, IO_E_NAME, IO_E_RANGE, IO_E_CAPACITY = range(7)

_BOUND = False


class SetupError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: error {code}: {detail}")


def _lib():
    global _BOUND
    lib = capi.load_library()
    if not _BOUND:
        dp, ip, vp, cp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p, C.c_char_p
        lib.tamcmc_setup_create.argtypes = [C.POINTER(vp), cp]
        lib.tamcmc_setup_create_files.argtypes = [C.POINTER(vp)] + [cp] * 6
        lib.tamcmc_setup_set.argtypes = [vp, cp, cp, cp]
        lib.tamcmc_setup_get.argtypes = [vp, cp, cp, C.c_char_p, C.c_int32]
        lib.tamcmc_setup_apply_phase.argtypes = [vp, cp, C.c_int64, C.c_double]
        lib.tamcmc_setup_load.argtypes = [vp, cp, cp, C.c_int32]
        lib.tamcmc_model_file_slices.argtypes = [cp, dp, C.c_int32, ip]
        lib.tamcmc_setup_sizes.argtypes = [vp, ip, ip, C.POINTER(C.c_int64), ip, ip, ip, ip, dp]
        lib.tamcmc_setup_inputs.argtypes = [vp, dp, ip, ip, dp, dp, dp]
        lib.tamcmc_setup_data.argtypes = [vp, dp, dp, dp]
        lib.tamcmc_setup_name.argtypes = [vp, C.c_int32, C.c_int32, C.c_char_p, C.c_int32]
        lib.tamcmc_setup_scalar.argtypes = [vp, C.c_int32]
        lib.tamcmc_setup_scalar.restype = C.c_double
        lib.tamcmc_setup_sampler_cfg.argtypes = [vp, C.POINTER(SamplerCfg)]
        lib.tamcmc_setup_error.argtypes = [vp]
        lib.tamcmc_setup_error.restype = cp
        lib.tamcmc_setup_log.argtypes = [vp]
        lib.tamcmc_setup_log.restype = cp
        lib.tamcmc_setup_destroy.argtypes = [vp]
        _BOUND = True
    return lib


def model_file_slices(model_file):
    """The `* fmin fmax` lines of a .model file (get_slices_range, main.cpp:379-444) as an (n, 2) array."""
    lib = _lib()
    n = C.c_int32(0)
    rc = lib.tamcmc_model_file_slices(str(model_file).encode(), None, 0, C.byref(n))
    if rc != IO_OK:
        raise SetupError(rc, "tamcmc_model_file_slices", str(model_file))
    out = np.zeros((n.value, 2))
    rc = lib.tamcmc_model_file_slices(str(model_file).encode(), out.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n))
    if rc != IO_OK:
        raise SetupError(rc, "tamcmc_model_file_slices", str(model_file))
    return out


class Setup:
    """Config (config.h:42-182) reduced to what the hot path and its callers read."""

    def __init__(self, config_dir):
        self._lib = _lib()
        self._h = C.c_void_p()
        rc = self._lib.tamcmc_setup_create(C.byref(self._h), str(config_dir).encode())
        if rc != IO_OK:
            self._h = C.c_void_p()
            raise SetupError(rc, "tamcmc_setup_create", str(config_dir))
        self.loaded = False

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.tamcmc_setup_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, where):
        if rc != IO_OK:
            raise SetupError(rc, where, self._lib.tamcmc_setup_error(self._h).decode(errors="replace"))

    def set(self, group, key, value):
        self._check(self._lib.tamcmc_setup_set(self._h, group.encode(), key.encode(), str(value).encode()), "tamcmc_setup_set")

    def get(self, group, key):
        buf = C.create_string_buffer(4096)
        self._check(self._lib.tamcmc_setup_get(self._h, group.encode(), key.encode(), buf, len(buf)), "tamcmc_setup_get")
        return buf.value.decode()

    def apply_phase(self, phase, Nsamples, c0):
        self._check(self._lib.tamcmc_setup_apply_phase(self._h, phase.encode(), int(Nsamples), float(c0)), "tamcmc_setup_apply_phase")

    def load(self, model_file, data_file, slice_ind=0):
        self._check(self._lib.tamcmc_setup_load(self._h, str(model_file).encode(), str(data_file).encode(), int(slice_ind)),
                    "tamcmc_setup_load")
        ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        npar, nvar, mc, lc, pc = (C.c_int32() for _ in range(5))
        nx = C.c_int64()
        lp = C.c_double()
        pl = np.zeros(11, dtype=np.int32)
        self._check(self._lib.tamcmc_setup_sizes(self._h, C.byref(npar), C.byref(nvar), C.byref(nx), pl.ctypes.data_as(ip),
                                                 C.byref(mc), C.byref(lc), C.byref(pc), C.byref(lp)), "tamcmc_setup_sizes")
        self.Nparams, self.Nvars, self.Nx = npar.value, nvar.value, nx.value
        self.plength = pl
        self.model_case, self.likelihood_case, self.prior_case = mc.value, lc.value, pc.value
        self.likelihood_p = lp.value
        self.inputs = np.zeros(self.Nparams)
        self.relax = np.zeros(self.Nparams, dtype=np.int32)
        self.priors_names_switch = np.zeros(self.Nparams, dtype=np.int32)
        self.priors = np.zeros((4, self.Nparams))
        self.extra_priors = np.zeros(4)
        self.err = np.zeros(self.Nvars)
        self._check(self._lib.tamcmc_setup_inputs(self._h, self.inputs.ctypes.data_as(dp), self.relax.ctypes.data_as(ip),
                                                  self.priors_names_switch.ctypes.data_as(ip), self.priors.ctypes.data_as(dp),
                                                  self.extra_priors.ctypes.data_as(dp), self.err.ctypes.data_as(dp)),
                    "tamcmc_setup_inputs")
        self.x, self.y, self.sigma_y = np.zeros(self.Nx), np.zeros(self.Nx), np.zeros(self.Nx)
        self._check(self._lib.tamcmc_setup_data(self._h, self.x.ctypes.data_as(dp), self.y.ctypes.data_as(dp),
                                                self.sigma_y.ctypes.data_as(dp)), "tamcmc_setup_data")
        self.inputs_names = [self._name(0, i) for i in range(self.Nparams)]
        self.priors_names = [self._name(1, i) for i in range(self.Nparams)]
        self.model_fullname, self.ID = self._name(2), self._name(3)
        self.xlabel, self.ylabel, self.xunit, self.yunit = (self._name(k) for k in (4, 5, 6, 7))
        self.Dnu, self.numax, self.C_l, self.fmin, self.fmax, self.resol = (self._lib.tamcmc_setup_scalar(self._h, k) for k in range(6))
        self.index_to_relax = np.flatnonzero(self.relax == 1).astype(np.int32)
        self.log = self._lib.tamcmc_setup_log(self._h).decode(errors="replace")
        self.loaded = True
        return self

    def _name(self, which, i=0):
        buf = C.create_string_buffer(512)
        self._check(self._lib.tamcmc_setup_name(self._h, which, i, buf, len(buf)), "tamcmc_setup_name")
        return buf.value.decode()

    def sampler_cfg(self, seed=0):
        cfg = SamplerCfg()
        self._check(self._lib.tamcmc_setup_sampler_cfg(self._h, C.byref(cfg)), "tamcmc_setup_sampler_cfg")
        cfg.seed = seed
        return cfg
