"""ctypes binding of include/tamcmc_outputs.h: result / restore files in the reference's formats and the
single-process phase driver (MALA::execute, MALA.cpp:608-720), plus readers of those files for post-processing."""
import ctypes as C

import numpy as np

from . import capi
from .setup_io import IO_OK, SetupError

PROGRESS_FN = C.CFUNCTYPE(None, C.c_int64, C.c_int64, C.c_void_p)
_BOUND = False


def _lib():
    global _BOUND
    lib = capi.load_library()
    if not _BOUND:
        vp, dp = C.c_void_p, C.POINTER(C.c_double)
        lib.tamcmc_outputs_create.argtypes = [C.POINTER(vp), vp, C.c_int32, dp, C.c_int64, C.c_int32]
        lib.tamcmc_outputs_record.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_double, C.c_int32]
        lib.tamcmc_outputs_finish.argtypes = [vp, vp]
        lib.tamcmc_outputs_destroy.argtypes = [vp]
        lib.tamcmc_outputs_error.argtypes = [vp]
        lib.tamcmc_outputs_error.restype = C.c_char_p
        lib.tamcmc_restore_apply.argtypes = [vp, vp, C.POINTER(C.c_int64), C.c_char_p, C.c_int32]
        lib.tamcmc_run_phase.argtypes = [vp, vp, PROGRESS_FN, vp, C.c_int32, C.c_char_p, C.c_int32]
        lib.tamcmc_sampler_set.argtypes = [vp, C.c_int32, dp, C.c_int64]
        lib.tamcmc_sampler_set_iteration.argtypes = [vp, C.c_int64]
        lib.tamcmc_sampler_nlocal.argtypes = [vp]
        _BOUND = True
    return lib


def run_phase(setup, sampler, progress=None, restore_precision=6):
    """Restore (if the setup asks for it) -> init -> iterate to Outputs.Nsamples -> result + restore files."""
    lib = _lib()
    err = C.create_string_buffer(1024)
    cb = PROGRESS_FN(progress if progress is not None else (lambda i, n, u: None))
    rc = lib.tamcmc_run_phase(setup._h, sampler._h, cb, None, int(restore_precision), err, len(err))
    if rc != IO_OK:
        raise SetupError(rc, "tamcmc_run_phase", err.value.decode(errors="replace"))


def restore_apply(setup, sampler):
    lib = _lib()
    err = C.create_string_buffer(1024)
    it = C.c_int64(0)
    rc = lib.tamcmc_restore_apply(setup._h, sampler._h, C.byref(it), err, len(err))
    if rc != IO_OK:
        raise SetupError(rc, "tamcmc_restore_apply", err.value.decode(errors="replace"))
    return int(it.value)


# ---------------------------------------------------------------- readers (Diagnostics::read_params_header,
# diagnostics.cpp:809-920; tools/bin2txt_params.cpp; tools/read_stats.cpp)
def read_header(path):
    """ASCII .hdr file -> dict of the `! key= value` lines (values as strings)."""
    out = {}
    for line in open(path):
        if line.startswith("!") and "=" in line:
            k, v = line[1:].split("=", 1)
            out[k.strip()] = v.strip()
    return out


def read_params_bin(root, chain):
    """<root>_chain-<m>.bin with <root>.hdr -> (samples[Nrows, Nvars], header dict)."""
    h = read_header(root + ".hdr")
    nv = int(h["Nvars"])
    a = np.fromfile(f"{root}_chain-{chain}.bin", dtype="<f8")
    return a.reshape(-1, nv), h


def read_stat_criteria_bin(root):
    h = read_header(root + ".hdr")
    nc = int(h["Nchains"])
    a = np.fromfile(root + ".bin", dtype="<f8").reshape(-1, 3 * nc)
    return a[:, :nc], a[:, nc:2 * nc], a[:, 2 * nc:], h


PT_RECORD = np.dtype([("attempt", "u1"), ("chain0", "<i4"), ("Pswitch", "<f8"), ("switched", "u1")])


def read_parallel_tempering_bin(root):
    return np.fromfile(root + ".bin", dtype=PT_RECORD), read_header(root + ".hdr")
