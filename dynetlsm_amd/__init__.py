"""dynetlsm_amd: MI355X-native engine for DynetLSM's Metropolis-within-Gibbs
hot path (network log-likelihoods, latent-position sweep, label block update).

Importing the package does not touch the GPU; the HIP library is loaded on
first use and there is no CPU fallback (``dynetlsm_amd._lib.load`` raises when
``libdynetlsm_hip.so`` has not been built).
"""
from .engine import Chain, SamplerGrid, EngineError  # noqa
from . import network_likelihoods  # noqa
from .lsm import DynamicNetworkLSM  # noqa
from .hdp_lpcm import DynamicNetworkHDPLPCM  # noqa
from .lpcm import DynamicNetworkLPCM  # noqa
from .case_control import DirectedCaseControlSampler  # noqa
from . import metrics  # noqa

__version__ = '0.1.0'
__all__ = ['Chain', 'SamplerGrid', 'EngineError', 'network_likelihoods',
           'DynamicNetworkLSM', 'DynamicNetworkHDPLPCM', 'DynamicNetworkLPCM',
           'DirectedCaseControlSampler']
