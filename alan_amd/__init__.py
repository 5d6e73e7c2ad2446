"""
alan_amd -- MI355X-native implementation of alan's tensorised marginal-likelihood hot path
(sample -> per-factor log-probs -> reduce_Ks / logsumexp_dims -> plate sum -> ELBO), behind alan's
own Plate / BoundPlate / Problem / Sample API (src/alan/__init__.py:1-16 of the reference).

Host code (model definition, sampling, per-factor log-probs) is Python on PyTorch-ROCm; the
contraction itself is hand-written HIP for gfx950 in libalan_mi355.so (include/alan_mi355.h).
There is no CPU fallback: evaluating an ELBO needs the library and a GPU.

Where a drop-in user sees something other than the reference -- every one a module switch, listed as
`module.SWITCH = default | value that restores the reference's way` (tests/test_native_abi.py checks this table against
the modules):

  ``engine.FP64_SMALL_FACTORS`` = "fused" | "exact"
      fp64 observations enter the fused fp32 plate step converted (result returned as fp64, within 1e-6 of the reference's
      fp64 log-sum-exp on the BASELINE configurations); "exact": such a problem takes the materialised route, added and
      reduced in fp64 as the reference's promotion does, at roughly twice the time.
  ``split.MERGE_CHUNKS`` = True | False
      ``Split(plate, size)`` evaluates a rank's chunks as one slice while every tensor that needs stays under
      ``split.MERGE_MAX_BYTES``; ``Split(..., merge=False)`` (or the switch) is the reference's per-chunk loop.
  ``dist.DEVICE_NOISE`` = True | False
      ``Problem.sample`` draws the noise of all its Normal variables together, generated inside the launch that uses it
      (Philox4x32-10 keyed by torch's generator: same distribution, reproducible under ``torch.manual_seed``, other
      particles than torch's variable-by-variable ``rsample`` under the same seed).  False: the noise is torch's ``normal_``
      (one call per batch); ``dist.BATCH_NOISE = False`` keeps torch's particles; ``dist.BATCH_DRAWS = False`` issues every
      draw where the model meets it.
  ``posterior.TIMESERIES_POSTERIOR`` = "reference" | "reference"
      (no deviation since round 4: ``importance_sample`` on a Timeseries draws every timestep independently from a
      filtering marginal, as reduce_Ks.py:85-232 does; "smoothing" is the opt-in improvement: exact JOINT trajectories,
      whose moments agree with ``marginals()``.)
"""
from .model import Plate, Group, Data
from .timeseries import Timeseries
from .samplers import CategoricalSampler, PermutationSampler, IndependentSampler
from .dist import *          # noqa: F401,F403  (Normal, Bernoulli, ..., OptParam, QEMParam, TorchDimDist)
from .dist import OptParam, QEMParam, TorchDimDist, new_dist
from .bound import BoundPlate, Problem
from .split import Split, no_checkpoint, checkpoint
from .sample import Sample
from .posterior import ImportanceSample
from .moments import Marginals, RawMoment, CompoundMoment, mean, mean2, var, var_from_raw_moment
from .contract import (reduce_Ks, collect_lps, logsumexp_sum, logsumexp_dims, logmeanexp_dims,
                       chain_logmmexp)

from .training import GraphedStep, GraphedEval
from .optim import Adam
from .sample import EvalPipeline, SamplingPipeline

samplers = [CategoricalSampler, PermutationSampler]

