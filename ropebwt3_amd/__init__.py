"""ropebwt3_amd -- MI355X-native engine for the incremental FM-index merge of `ropebwt3 build`.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI + host-side C) and
thin ctypes mirrors of the reference's interface for that path.
"""
from . import _build  # noqa: F401
from .gpu import Rb3Gpu, Rb3GpuError, Sorter, PinnedArray, load_library, walker_step, CommGroup, RcclComm, CallbackComm, Shard, ipc_peer_enable, ipc_peer_disable  # noqa: F401

__all__ = ["Rb3Gpu", "Rb3GpuError", "Sorter", "PinnedArray", "load_library", "walker_step"]
