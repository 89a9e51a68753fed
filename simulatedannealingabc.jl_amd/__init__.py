"""MI355X-native engine for the particle-population update loop of SimulatedAnnealingABC.jl.

Public surface = the reference's exports (`sabc`, `update_population!` -> `update_population_`,
`RandomWalk`, `DifferentialEvolution`, `StretchMove`) plus the descriptors a device path needs
(`DeviceDistance` simulators and priors as data).  Everything runs in libsabc_hip.so
(hand-written gfx950 HIP kernels behind the C-ABI of include/sabc_hip.h); there is no CPU path.
"""
from ._lib import SABCError, build, lib  # noqa: F401
from .api import (SABCresult, SABCstate, initialization, is_logging, load_result, sabc, save_result,  # noqa: F401
                  update_population_)
from .distributions import (Beta, Exponential, Gamma, HostPrior, LogNormal, MvNormal, Normal, Product, SourcePrior,  # noqa: F401
                            TruncatedNormal, Uniform, from_scipy, product_distribution, truncated)
from .handle import (SabcHandle, op_build_cdf, op_cdf_eval, op_eps_multi, op_eps_single,  # noqa: F401
                     op_normal_pairs, op_philox, op_rng_peak, op_sort)
from .models import DeviceDistance, DeviceSource, GandK, Gaussian2D, GaussianIID, HostDistance, LotkaVolterra  # noqa: F401
from .proposals import DifferentialEvolution, Proposal, RandomWalk, StretchMove  # noqa: F401

__all__ = [
    "sabc", "update_population_", "initialization", "save_result", "load_result", "SABCresult", "SABCstate", "SABCError",
    "RandomWalk", "DifferentialEvolution", "StretchMove", "Proposal",
    "Normal", "Uniform", "Exponential", "LogNormal", "Gamma", "Beta", "TruncatedNormal", "truncated", "MvNormal", "HostPrior", "SourcePrior", "from_scipy", "Product", "product_distribution",
    "DeviceDistance", "DeviceSource", "HostDistance", "GaussianIID", "Gaussian2D", "GandK", "LotkaVolterra",
    "SabcHandle", "op_build_cdf", "op_cdf_eval", "op_eps_single", "op_eps_multi", "op_philox", "op_normal_pairs", "op_rng_peak", "op_sort",
    "build", "lib", "is_logging",
]
