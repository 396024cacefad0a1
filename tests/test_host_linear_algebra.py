"""The sampler's host-side Cholesky factorisation (tamcmc_sampler.cpp: `cholesky`, what the reference gets from
tmpmat.llt().matrixL(), MALA.cpp:344) against the textbook left-looking loops in plain Python floats: the library's
blocked right-looking form promises the SAME operations in the SAME order per element, so the factors must agree bit for
bit -- including which columns are filled in when the matrix is not positive definite."""
import ctypes as C
import math

import numpy as np
import pytest

import tamcmc_amd


def _textbook(A):
    n = A.shape[0]
    L = np.zeros((n, n))
    for j in range(n):
        d = float(A[j, j])
        for k in range(j):
            d = d - float(L[j, k]) * float(L[j, k])
        if not d > 0.0:
            return False, L
        ljj = math.sqrt(d)
        L[j, j] = ljj
        for i in range(j + 1, n):
            t = float(A[i, j])
            for k in range(j):
                t = t - float(L[i, k]) * float(L[j, k])
            L[i, j] = t / ljj
    return True, L


def _lib_cholesky():
    lib = C.CDLL(tamcmc_amd.library_path())
    fn = lib.tamcmc_host_cholesky                     # include/tamcmc_sampler.h
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    return fn


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 9, 21, 43, 44, 45, 50])
def test_cholesky_is_the_textbook_factor_bit_for_bit(n):
    fn = _lib_cholesky()
    rng = np.random.default_rng(100 + n)
    for trial in range(6):
        B = rng.normal(size=(n, n if trial % 3 else max(1, n // 2)))
        A = B @ B.T + (0.5 if trial != 5 else -0.3) * np.eye(n)       # trial 5 (and rank-deficient ones) may fail midway
        A = np.ascontiguousarray(0.5 * (A + A.T))
        L = np.full((n, n), np.nan)
        ok = fn(A.ctypes.data, n, L.ctypes.data) == 0
        ok_ref, L_ref = _textbook(A)
        assert ok == ok_ref
        assert np.array_equal(L, L_ref), f"n={n} trial={trial}: max |diff| {np.max(np.abs(L - L_ref))}"
        if ok:
            assert np.allclose(L @ L.T, A, rtol=1e-12, atol=1e-12 * np.abs(A).max())
