"""MI355X-native ILMM/OILMM inference hot path: host-side mirror of the LinearMixingModels.jl interface over
liblmm_hip.so (hand-written gfx950 HIP kernels behind the C ABI of include/lmm_hip.h).

The directory name contains a dot, so import it through the root-level loader:  `import lmm_amd`."""
from ._lib import LMMError, PosDefException, init, load, LIB_PATH, SYMBOLS, set_compute_dtype, get_compute_dtype, set_strict_progress, get_strict_progress, set_projection_dtype, get_projection_dtype, wait_stream
from .model import (GP, ILMM, OILMM, FiniteGP, IndependentMOGP, Matern32Kernel, Matern52Kernel,
                    MOInputIsotopicByFeatures, MOInputIsotopicByOutputs, Normal, DeviceNormals, indices_which_reorder_features_to_outputs,
                    indices_which_reorder_outputs_to_features, Orthogonal, SEKernel, get_latent_gp, independent_mogp, logpdf, logpdf_and_gradient,
                    marginals, mean, mean_and_cov, cov, mean_and_var, noise_var, posterior, rand, reshape_y, unpack, var)
from .parallel import select_collective, latent_shard, sharded_logpdf, sharded_mean_and_var, sharded_posterior, sharded_rand

__all__ = [
    "ILMM", "IndependentMOGP", "independent_mogp", "Orthogonal", "OILMM", "get_latent_gp",   # the reference's 6 exports
    "GP", "SEKernel", "Matern32Kernel", "Matern52Kernel", "MOInputIsotopicByOutputs", "MOInputIsotopicByFeatures", "FiniteGP", "Normal", "DeviceNormals",
    "indices_which_reorder_features_to_outputs", "indices_which_reorder_outputs_to_features",
    "logpdf", "logpdf_and_gradient", "posterior", "rand", "marginals", "mean_and_var", "mean_and_cov", "mean", "var", "cov", "noise_var", "reshape_y", "unpack",
    "select_collective", "latent_shard", "sharded_logpdf", "sharded_mean_and_var", "sharded_posterior", "sharded_rand", "init", "load", "PosDefException", "LMMError", "set_compute_dtype", "get_compute_dtype", "set_strict_progress", "get_strict_progress", "set_projection_dtype", "get_projection_dtype", "wait_stream",
]
