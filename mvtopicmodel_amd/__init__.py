"""mvtopicmodel_amd — MI355X-native multi-view HDP collapsed-Gibbs sweep.

Drop-in for the FastQMVWVWorkerRunnable / FastQMVWVUpdaterRunnable hot path of
hmetaxa/MVTopicModel behind FastQMVWVParallelTopicModel.estimate().  The product
is the C-ABI library ``lib/libmvhdp.so`` (include/mvhdp.h); this package is the
host-side mirror used by tests and bench (the reference's host language, Java,
has no toolchain in this image — see INTEGRATION.md for the JNI binding).

There is no CPU fallback: importing works anywhere, but creating a sampler
without a gfx950 device raises.
"""
from ._lib import load_library, LIB_PATH, MvhdpError  # noqa: F401
from .native import NativeGroup, NativeSampler, SweepStats, Hyper  # noqa: F401

__all__ = ["load_library", "LIB_PATH", "MvhdpError", "NativeGroup", "NativeSampler", "SweepStats", "Hyper"]
