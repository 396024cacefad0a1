"""Host-side mirror of the reference's Model_def plugin surface (tamcmc/headers/model_def.h:23-83,
tamcmc/sources/model_def.cpp) on top of the HIP library.

Same member names and meaning as the reference class:
  params (Nmodels x Nparams), vars (Nmodels x Nvars), model (rows filled on demand), logLikelihood
  (ALREADY divided by the chain temperature, model_def.cpp:302), logPrior, logPosterior, Pmove, moved,
  swaped, Pswap, comparator_MH, comparator_PT, and the methods call_model / call_model_explicit /
  update_params_with_vars / call_likelihood / call_prior / generate_model.
The reference calls generate_model(m) once per chain from an OpenMP loop (MALA.cpp:632-639); here
generate_model(m) is kept for drop-in use and generate_models() evaluates every chain in ONE device
call, which is how the accelerator is meant to be driven.

Error behaviour: where the reference prints and exit()s (unknown switch ids, empty truncation
window) this raises AccelError or reports NaN + a per-chain status; NaN logL stays legal and means
"reject" (MALA.cpp:475).
"""
from dataclasses import dataclass, field

import numpy as np

from . import capi


@dataclass
class Data:
    """data.h:24-36"""
    x: np.ndarray
    y: np.ndarray
    sigma_y: np.ndarray = None
    xlabel: str = ""
    ylabel: str = ""
    xunit: str = ""
    yunit: str = ""
    header: list = field(default_factory=list)

    @property
    def Nx(self):
        return int(self.x.size)


class ModelDef:
    def __init__(self, data, model_fct_name_switch, plength, inputs, relax, Tcoefs, Nchains=None,
                 likelihood_fct_name_switch=0, likelihood_params=1.0, prior_fct=None, device_id=0):
        """Mirrors Model_def::Model_def(Config*, VectorXd Tcoefs, bool) (model_def.cpp:27-181):
        every chain starts from `inputs`; vars = the relax==1 columns (model_def.cpp:81-94)."""
        self.data = data
        self.Tcoefs = np.ascontiguousarray(Tcoefs, dtype=np.float64)
        self.Nmodels = int(Nchains if Nchains is not None else self.Tcoefs.size)
        self.model_fct_name_switch = int(model_fct_name_switch)
        self.likelihood_fct_name_switch = int(likelihood_fct_name_switch)
        self.likelihood_params = float(likelihood_params)
        self.plength = np.ascontiguousarray(plength, dtype=np.int32)
        self.relax = np.ascontiguousarray(relax, dtype=np.int32)
        inputs = np.ascontiguousarray(inputs, dtype=np.float64)
        self.Nparams = int(self.plength.sum())
        if inputs.size != self.Nparams or self.relax.size != self.Nparams:
            raise ValueError("inputs / relax / plength disagree on Nparams")
        self.params = np.tile(inputs, (self.Nmodels, 1))
        self.index_to_relax = np.flatnonzero(self.relax == 1).astype(np.int32)
        self.Nvars = int(self.index_to_relax.size)
        self.Ncons = self.Nparams - self.Nvars
        self.vars = self.params[:, self.index_to_relax].copy()
        self.cons = inputs[self.relax != 1].copy()
        self.model = np.zeros((self.Nmodels, data.Nx))
        self.logLikelihood = np.zeros(self.Nmodels)
        self.logPrior = np.zeros(self.Nmodels)
        self.logPosterior = np.zeros(self.Nmodels)
        self.gradLogLikelihood = np.zeros((self.Nmodels, self.Nvars))   # new: d(logL/T)/dvars
        self.status = np.zeros(self.Nmodels, dtype=np.int32)
        self.Pmove = np.zeros(self.Nmodels)
        self.moved = [False] * self.Nmodels
        self.swaped = False
        self.Pswap = 0.0
        self.comparator_MH = np.zeros(self.Nmodels)
        self.comparator_PT = 0.0
        self.prior_fct = prior_fct   # callable(params_row) -> logPrior; None -> 0
        self._accel = capi.Accel(self.model_fct_name_switch, self.plength, data.x, data.y, data.sigma_y,
                                 self.likelihood_fct_name_switch, self.likelihood_params, device_id)
        if self.Nvars:
            self._accel.set_vars(self.index_to_relax)

    # ---- reference-shaped API -----------------------------------------------------------------
    def update_params_with_vars(self, m):
        """model_def.cpp:370-378"""
        self.params[m, self.index_to_relax] = self.vars[m]

    def call_model(self, data, m):
        """model_def.cpp:210-289: model spectrum of chain m."""
        out, st = self._accel.model_explicit(self.params[m])
        self.status[m] = st
        return out

    def call_model_explicit(self, data, plength0, params0, model_case):
        """model_def.cpp:199-208 (used by tools/getmodel.cpp:111)."""
        with capi.Accel(model_case, plength0, data.x, data.y, data.sigma_y, self.likelihood_fct_name_switch,
                        self.likelihood_params) as acc:
            out, _ = acc.model_explicit(params0)
        return out

    def call_likelihood(self, data, m, Tcoefs):
        """model_def.cpp:291-320: tempered log-likelihood of chain m (recomputes the model)."""
        logL, st = self._accel.eval_batch(self.params[m:m + 1], np.asarray(Tcoefs, dtype=np.float64)[m:m + 1])
        self.status[m] = st[0]
        return float(logL[0])

    def call_prior(self, data, m):
        """model_def.cpp:322-356"""
        return float(self.prior_fct(self.params[m])) if self.prior_fct is not None else 0.0

    def generate_model(self, data, m, Tcoefs, keep_model=True):
        """model_def.cpp:358-367 for one chain."""
        T = np.asarray(Tcoefs, dtype=np.float64)[m:m + 1]
        if keep_model:
            logL, st, mod = self._accel.eval_batch(self.params[m:m + 1], T, model_rows=[0])
            self.model[m] = mod[0]
        else:
            logL, st = self._accel.eval_batch(self.params[m:m + 1], T)
        self.status[m] = st[0]
        self.logLikelihood[m] = logL[0]
        self.logPrior[m] = self.call_prior(data, m)
        self.logPosterior[m] = self.logLikelihood[m] + self.logPrior[m]
        return self.logPosterior[m]

    # ---- batched entry point (one device call per MCMC iteration) -------------------------------
    def generate_models(self, Tcoefs=None, grad=False, model_rows=None):
        T = self.Tcoefs if Tcoefs is None else np.asarray(Tcoefs, dtype=np.float64)
        res = self._accel.eval_batch(self.params, T, grad=grad, model_rows=model_rows)
        self.logLikelihood[:] = res[0]
        self.status[:] = res[1]
        k = 2
        if grad:
            self.gradLogLikelihood[:] = res[k]
            k += 1
        if model_rows is not None:
            for r, m in enumerate(model_rows):
                self.model[m] = res[k][r]
        for m in range(self.Nmodels):
            self.logPrior[m] = self.call_prior(self.data, m)
        self.logPosterior[:] = self.logLikelihood + self.logPrior
        return self.logPosterior

    def close(self):
        self._accel.close()
