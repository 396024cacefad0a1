"""tamcmc-c-_amd -- MI355X (gfx950) accelerator for TAMCMC's hot path.

The directory name is not a Python identifier; import it through the repo-root loader::

    import tamcmc_amd            # tamcmc_amd.py at the repo root

Contents (only what the hot path needs):
  csrc/            hand-written HIP kernels + the extern "C" library (include/tamcmc_accel.h)
  capi.py          ctypes binding of that C ABI (fails loudly if the library is missing)
  model_def.py     host-side mirror of the reference's Model_def plugin surface (model_def.h:23-83)
  sampler.py       binding of include/tamcmc_sampler.h (priors, adaptive Metropolis + parallel tempering)
  setup_io.py      binding of include/tamcmc_io.h (.model / .data / .cfg / .list readers)
  outputs.py       binding of include/tamcmc_outputs.h (result / restore files, phase driver) + readers
  sharded.py       one phase with the chains sharded over processes (one per GPU), files written by rank 0
  synth.py         synthetic spectra / chain parameters of SURVEY.md section 8d
"""
from . import capi, synth, model_def, shard, sampler, setup_io, outputs, sharded  # noqa: F401
from .capi import Accel, AccelError, load_library, library_path  # noqa: F401
from .model_def import ModelDef, Data  # noqa: F401

__all__ = ["capi", "synth", "model_def", "Accel", "AccelError", "load_library", "library_path", "ModelDef", "Data"]
