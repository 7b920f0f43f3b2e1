"""ctypes binding of the C ABI in include/rb3gpu.h (librb3gpu.so).

This module mirrors the reference's call surface for the merge path: `Rb3Gpu` stands where
an `mrope_t*` stands in build.c, with `from_plain` = rb3_enc_plain2fmr (fm-index.c:114),
`merge_plain` = rb3_fmi_merge_plain (fm-index.c:279), `get_acc` = rb3_fmi_get_acc and
`export_runs` = the leaf iteration of rb3_enc_fmr2fmd (fm-index.c:31-54).

There is NO CPU fallback: importing works anywhere, but creating a handle raises when the
shared object or a HIP device is missing.
"""
import ctypes
import os

import numpy as np

from . import _build

ASIZE = 6

_ERR = {0: "OK", -1: "ENODEV", -2: "ENOMEM", -3: "EINVAL", -4: "ESYMBOL", -5: "ESTATE", -6: "EINTERNAL", -7: "EUNSUP"}


class Rb3GpuError(RuntimeError):
    def __init__(self, code, what):
        RuntimeError.__init__(self, "%s failed: %s (%d)" % (what, _ERR.get(code, "?"), code))
        self.code = code


class Opt(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("split_log2", ctypes.c_int32), ("verbose", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class Stats(ctypes.Structure):
    _fields_ = [("ms_h2d", ctypes.c_double), ("ms_lf", ctypes.c_double), ("ms_rank", ctypes.c_double),
                ("ms_build", ctypes.c_double), ("ms_export", ctypes.c_double), ("ms_chain", ctypes.c_double),
                ("n_rank_launches", ctypes.c_int64), ("n_lf_steps", ctypes.c_int64), ("n_symbols_merged", ctypes.c_int64),
                ("n_rounds", ctypes.c_int64), ("n_fallbacks", ctypes.c_int64), ("bytes_index", ctypes.c_int64), ("bytes_peak", ctypes.c_int64),
                ("ms_ssa", ctypes.c_double), ("ms_ssa_walk", ctypes.c_double), ("ms_sort", ctypes.c_double), ("n_sort_rounds", ctypes.c_int64),
                ("n_reb_groups", ctypes.c_int64), ("n_reb_groups_window", ctypes.c_int64), ("n_lf_checked", ctypes.c_int64), ("n_long_settles", ctypes.c_int64), ("ms_alloc", ctypes.c_double), ("n_allocs", ctypes.c_int64), ("n_reb_again", ctypes.c_int64), ("bytes_rebuild", ctypes.c_int64), ("n_thinned", ctypes.c_int64), ("tent_mask_bits", ctypes.c_int64), ("n_junctions_checked", ctypes.c_int64), ("n_peer_rounds", ctypes.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


EMIT_F = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64)
EMITW_F = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int64)
EMIT_WORDS_F = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int64)

# name -> (restype, argtypes); every symbol declared in include/rb3gpu.h
SYMBOLS = {
    "rb3gpu_opt_init": (None, [ctypes.POINTER(Opt)]),
    "rb3gpu_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "rb3gpu_create": (ctypes.c_void_p, [ctypes.POINTER(Opt)]),
    "rb3gpu_destroy": (None, [ctypes.c_void_p]),
    "rb3gpu_from_plain": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rb3gpu_merge_plain": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rb3gpu_from_plain_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rb3gpu_merge_plain_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]),
    "rb3gpu_merge_plain_walkers": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rb3gpu_merge_plain_dev_walkers": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]),
    "rb3gpu_mg_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_mg_walk": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rb3gpu_mg_pos_ptr": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]),
    "rb3gpu_mg_finish": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rb3gpu_mg_rank_plain_walkers": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_mg_rank_plain": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_rank1a_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_get_acc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_get_tot": (ctypes.c_int64, [ctypes.c_void_p]),
    "rb3gpu_export_runs": (ctypes.c_int, [ctypes.c_void_p, EMIT_F, ctypes.c_void_p]),
    "rb3gpu_export_run_words": (ctypes.c_int, [ctypes.c_void_p, EMIT_WORDS_F, ctypes.c_void_p]),
    "rb3gpu_export_fmd_words": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]),
    "rb3gpu_host_free": (None, [ctypes.c_void_p]),
    "rb3gpu_export_plain": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_export_plain_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_ssa_dims": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]),
    "rb3gpu_ssa_gen": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_bwt_from_text": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rb3gpu_sort_text": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_merge_text_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]),
    "rb3gpu_merge_text_sa_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]),
    "rb3gpu_sort_text_sa": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_sorter_sort_uploaded_sa": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]),
    "rb3gpu_sorter_sort_sa": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]),
    "rb3gpu_mg_rank_text_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_sorter_sort": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]),
    "rb3gpu_sorter_create": (ctypes.c_void_p, [ctypes.c_int]),
    "rb3gpu_sorter_destroy": (None, [ctypes.c_void_p]),
    "rb3gpu_sorter_bwt": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64, ctypes.c_void_p]),
    "rb3gpu_sorter_release": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_sorter_upload": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rb3gpu_sorter_upload_fwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rb3gpu_sorter_upload_fwd_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rb3gpu_sorter_upload_end": (ctypes.c_int, [ctypes.c_void_p]),
    "rb3gpu_sorter_sort_uploaded": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]),
    "rb3gpu_pinned_alloc": (ctypes.c_void_p, [ctypes.c_int64]),
    "rb3gpu_pinned_free": (None, [ctypes.c_void_p]),
    "rb3gpu_walker_step": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int64, ctypes.c_int64]),
    "rb3gpu_sorter_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "rb3gpu_from_runs": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rb3gpu_from_fmd_words": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_merge_fmd_words": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_merge_index": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_tune": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]),
    "rb3gpu_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Stats)]),
    "rb3gpu_buffer_bytes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int64)]),
    "rb3gpu_stats_reset": (None, [ctypes.c_void_p]),
    "rb3gpu_dev_alloc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]),
    "rb3gpu_dev_upload": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    "rb3gpu_dev_download": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    "rb3gpu_dev_free": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "rb3gpu_dev_copy": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    "rb3gpu_dev_memset": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]),
    "rb3gpu_sh_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_sh_finish": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]),
    "rb3gpu_device_count": (ctypes.c_int, []),
    "rb3gpu_sh_merge": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "rb3gpu_group_create": (ctypes.c_void_p, [ctypes.c_int]),
    "rb3gpu_group_comm": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_group_abort": (None, [ctypes.c_void_p]),
    "rb3gpu_group_destroy": (None, [ctypes.c_void_p]),
    "rb3gpu_rccl_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "rb3gpu_rccl_comm_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_rccl_comm_destroy": (None, [ctypes.c_void_p]),
    "rb3gpu_stream_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "rb3gpu_ipc_peer_enable": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_ipc_peer_disable": (None, [ctypes.c_void_p]),
    "rb3gpu_merge_text_step_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]),
    "rb3gpu_walkers_step_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_shard_split": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_shard_merge": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_shard_gather": (ctypes.c_int, [ctypes.c_void_p]),
    "rb3gpu_shard_destroy": (None, [ctypes.c_void_p]),
    "rb3gpu_shard_rebalance": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rb3gpu_shard_get_acc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_shard_export_runs": (ctypes.c_int, [ctypes.c_void_p, EMIT_F, ctypes.c_void_p]),
    "rb3gpu_shard_export_run_words": (ctypes.c_int, [ctypes.c_void_p, EMITW_F, ctypes.c_void_p]),
    "rb3gpu_sh_merge_text": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "rb3gpu_tprev_from_tw": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_balanced_bounds": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "rb3gpu_export_plain_range_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "rb3gpu_shard_handle": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    "rb3gpu_shard_bounds": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rb3gpu_device_of": (ctypes.c_int, [ctypes.c_void_p]),
    "rb3gpu_stream_of": (ctypes.c_void_p, [ctypes.c_void_p]),
}

# rb3gpu_comm_t (include/rb3gpu.h): the two collectives of the interval-sharded merge
ALL_GATHER_F = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.POINTER(ctypes.c_int64))
ALL_TO_ALL_F = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p)
ABORT_F = ctypes.CFUNCTYPE(None, ctypes.c_void_p)


class CommStruct(ctypes.Structure):
    _fields_ = [("ctx", ctypes.c_void_p), ("rank", ctypes.c_int), ("world", ctypes.c_int),
                ("all_gather", ctypes.c_void_p), ("all_to_all", ctypes.c_void_p), ("abort", ctypes.c_void_p), ("stream_barrier", ctypes.c_void_p),
                ("peer_export", ctypes.c_void_p), ("peer_import", ctypes.c_void_p)]

_libs = {}


def load_library(hooks=False, path=None):
    """Load librb3gpu.so (the in-tree build) and declare every prototype.  Raises if absent.
    hooks=True: the test build (librb3gpu_hooks.so, -DRB3GPU_TEST_HOOKS), whose rb3gpu_tune also knows the keys that
    make a merge pretend a failure; only tests load it."""
    if path is None:
        path = _build.LIB_GPU_HOOKS if hooks else os.environ.get("RB3GPU_LIB", _build.LIB_GPU)  # override for kernel experiments only
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the engine is HIP-only; there is no CPU fallback)" % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        f = getattr(lib, name)  # AttributeError if the header and the library disagree
        f.restype, f.argtypes = res, args
    _libs[path] = lib
    return lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def walker_step(device, length, n_strings, lib=None):
    """rb3gpu_walker_step: the text distance between the LF walkers of a batch (as many walkers as the walker kernel keeps resident)"""
    r = load_library(False, lib).rb3gpu_walker_step(int(device), int(length), int(n_strings))
    if r < 0:
        raise Rb3GpuError(int(r), "rb3gpu_walker_step")
    return int(r)


class PinnedArray:
    """a uint8 numpy array in page-locked host memory (rb3gpu_pinned_alloc): a batch built here goes to HBM with one DMA"""

    def __init__(self, nbytes, lib=None):
        self._lib = load_library(False, lib)
        self._p = self._lib.rb3gpu_pinned_alloc(int(max(nbytes, 1)))
        if not self._p:
            raise MemoryError("rb3gpu_pinned_alloc(%d) failed" % nbytes)
        self.array = np.ctypeslib.as_array(ctypes.cast(self._p, ctypes.POINTER(ctypes.c_uint8)), shape=(int(max(nbytes, 1)),))[:nbytes]

    def free(self):
        if getattr(self, "_p", None):
            self.array = None
            self._lib.rb3gpu_pinned_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Sorter:
    """rb3gpu_sorter_t: the GPU suffix sorter as an object of its own (own HIP stream, scratch, two output buffers): what the
    CLI's sorter thread runs while the batch before is being merged (rb3_build_sais, sais-ss.c:50-56, build.c:55-83)"""

    def __init__(self, device=0, lib=None):
        self._lib = load_library(False, lib)
        self._s = self._lib.rb3gpu_sorter_create(int(device))
        if not self._s:
            raise RuntimeError("rb3gpu_sorter_create failed on device %d (no HIP device: there is no CPU fallback)" % device)

    def _chk(self, r, what):
        if r < 0:
            raise Rb3GpuError(r, what)

    def upload(self, text):
        """the text of a batch host -> HBM (the H2D copy of the merge path); returns when the copy is complete"""
        assert text.dtype == np.uint8 and text.flags["C_CONTIGUOUS"]
        self._chk(self._lib.rb3gpu_sorter_upload(self._s, text.size, text.ctypes.data), "rb3gpu_sorter_upload")

    def upload_fwd(self, text, pair_start):
        """upload of a batch of a few long records on both strands: forward strands only, reverse complements made on the device"""
        assert text.dtype == np.uint8 and text.flags["C_CONTIGUOUS"]
        ps = np.ascontiguousarray(pair_start, dtype=np.int64)
        self._chk(self._lib.rb3gpu_sorter_upload_fwd(self._s, text.size, text.ctypes.data, ps.size, ps.ctypes.data), "rb3gpu_sorter_upload_fwd")

    def upload_fwd_begin(self, text, pair_start):
        """upload_fwd without waiting: the copies are queued on the sorter's stream (page-locked text) and run beside whatever the
        caller does next; the text must not change until upload_end() or sort_uploaded() has returned"""
        assert text.dtype == np.uint8 and text.flags["C_CONTIGUOUS"]
        ps = np.ascontiguousarray(pair_start, dtype=np.int64)
        self._chk(self._lib.rb3gpu_sorter_upload_fwd_begin(self._s, text.size, text.ctypes.data, ps.size, ps.ctypes.data), "rb3gpu_sorter_upload_fwd_begin")

    def upload_end(self):
        self._chk(self._lib.rb3gpu_sorter_upload_end(self._s), "rb3gpu_sorter_upload_end")

    def sort_uploaded_sa(self, length):
        """sort_uploaded with the suffix array: (d_bwt, d_tw, d_sa), all released by release(d_bwt)"""
        p, q, r = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._chk(self._lib.rb3gpu_sorter_sort_uploaded_sa(self._s, int(length), ctypes.byref(p), ctypes.byref(q), ctypes.byref(r)), "rb3gpu_sorter_sort_uploaded_sa")
        return p, q, r

    def sort_uploaded(self, length):
        """suffix-sort the text uploaded last: (d_bwt, d_tw) device pointers, valid until release(d_bwt)"""
        p, q = ctypes.c_void_p(), ctypes.c_void_p()
        self._chk(self._lib.rb3gpu_sorter_sort_uploaded(self._s, int(length), ctypes.byref(p), ctypes.byref(q)), "rb3gpu_sorter_sort_uploaded")
        return p, q

    def sort(self, text):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        p, q = ctypes.c_void_p(), ctypes.c_void_p()
        self._chk(self._lib.rb3gpu_sorter_sort(self._s, text.size, text.ctypes.data, ctypes.byref(p), ctypes.byref(q)), "rb3gpu_sorter_sort")
        return p, q

    def release(self, d_bwt):
        self._chk(self._lib.rb3gpu_sorter_release(self._s, d_bwt), "rb3gpu_sorter_release")

    def stats(self):
        up, so, nb, ns = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
        self._chk(self._lib.rb3gpu_sorter_stats(self._s, ctypes.byref(up), ctypes.byref(so), ctypes.byref(nb), ctypes.byref(ns)), "rb3gpu_sorter_stats")
        return {"ms_upload": up.value, "ms_sort": so.value, "n_batches": nb.value, "n_symbols": ns.value}

    def close(self):
        if getattr(self, "_s", None):
            self._lib.rb3gpu_sorter_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Rb3Gpu:
    """One accumulated BWT resident in the HBM of one MI355X."""

    def __init__(self, device=0, split_log2=0, verbose=1, hooks=False, lib=None):
        self._lib = load_library(hooks, lib)   # lib: another build of the library (kernel experiments, tools/probe_*.py)
        n = self._lib.rb3gpu_device_count()
        if n <= 0:
            raise RuntimeError("no HIP device visible (rb3gpu_device_count=%d); the engine has no CPU fallback" % n)
        opt = Opt()
        self._lib.rb3gpu_opt_init(ctypes.byref(opt))
        opt.device, opt.split_log2, opt.verbose = device, split_log2, verbose
        self._h = self._lib.rb3gpu_create(ctypes.byref(opt))
        if not self._h:
            raise RuntimeError("rb3gpu_create failed on device %d" % device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rb3gpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tune(self, key, value):
        """rb3gpu_tune: a diagnostic switch of this handle (see include/rb3gpu.h)"""
        self._chk(self._lib.rb3gpu_tune(self._h, key.encode(), int(value)), "rb3gpu_tune(%s)" % key)

    def _chk(self, r, what):
        if r < 0:
            raise Rb3GpuError(r, what)
        return r

    # -- the reference's entry points -------------------------------------------------------
    def from_plain(self, bwt):
        bwt = _u8(bwt)
        self._chk(self._lib.rb3gpu_from_plain(self._h, bwt.size, bwt.ctypes.data), "rb3gpu_from_plain")

    def merge_plain(self, bwt):
        bwt = _u8(bwt)
        self._chk(self._lib.rb3gpu_merge_plain(self._h, bwt.size, bwt.ctypes.data), "rb3gpu_merge_plain")

    @staticmethod
    def _walkers(w):
        w = np.ascontiguousarray(w, dtype=np.int64)
        assert w.ndim == 2 and w.shape[1] == 4, "walkers: (n, 4) int64 rows of (row, ka0, nsteps, flags)"
        return w

    def merge_plain_walkers(self, bwt, walkers):
        bwt, w = _u8(bwt), self._walkers(walkers)
        self._chk(self._lib.rb3gpu_merge_plain_walkers(self._h, bwt.size, bwt.ctypes.data, w.shape[0], w.ctypes.data), "rb3gpu_merge_plain_walkers")

    def merge_plain_dev_walkers(self, d_bwt, length, walkers, commit=True):
        if isinstance(walkers, (int, np.integer)):  # the number of strings: one walker per string, made on the device
            self._chk(self._lib.rb3gpu_merge_plain_dev_walkers(self._h, length, d_bwt, int(walkers), None, 1 if commit else 0), "rb3gpu_merge_plain_dev_walkers")
            return
        w = self._walkers(walkers)
        self._chk(self._lib.rb3gpu_merge_plain_dev_walkers(self._h, length, d_bwt, w.shape[0], w.ctypes.data, 1 if commit else 0), "rb3gpu_merge_plain_dev_walkers")

    # staged merge (multi-GPU): begin -> walk (repeatable) -> [collective on pos] -> finish
    def mg_begin(self, d_bwt, length, d_pos_ext=None):
        acc2 = np.zeros(7, dtype=np.int64)
        self._chk(self._lib.rb3gpu_mg_begin(self._h, length, d_bwt, d_pos_ext, acc2.ctypes.data), "rb3gpu_mg_begin")
        return acc2

    def mg_walk(self, walkers=None, stop_row=-1):
        """run walkers; returns the exact value a walker arrived at stop_row with, or -1"""
        if walkers is None:
            self._chk(self._lib.rb3gpu_mg_walk(self._h, 0, None, -1, None), "rb3gpu_mg_walk")
            return -1
        w = self._walkers(walkers)
        arr = np.full(1, -1, dtype=np.int64)
        self._chk(self._lib.rb3gpu_mg_walk(self._h, w.shape[0], w.ctypes.data, stop_row, arr.ctypes.data), "rb3gpu_mg_walk")
        return int(arr[0])

    def mg_pos_ptr(self):
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        self._chk(self._lib.rb3gpu_mg_pos_ptr(self._h, ctypes.byref(p), ctypes.byref(n)), "rb3gpu_mg_pos_ptr")
        return p.value, n.value

    def mg_finish(self, commit=True):
        self._chk(self._lib.rb3gpu_mg_finish(self._h, 1 if commit else 0), "rb3gpu_mg_finish")

    def mg_rank_plain(self, bwt):
        bwt = _u8(bwt)
        pos = np.empty(bwt.size, dtype=np.int64)
        acc2 = np.zeros(7, dtype=np.int64)
        self._chk(self._lib.rb3gpu_mg_rank_plain(self._h, bwt.size, bwt.ctypes.data, pos.ctypes.data, acc2.ctypes.data), "rb3gpu_mg_rank_plain")
        return pos, acc2

    def mg_rank_plain_walkers(self, bwt, walkers):
        bwt, w = _u8(bwt), self._walkers(walkers)
        pos = np.empty(bwt.size, dtype=np.int64)
        acc2 = np.zeros(7, dtype=np.int64)
        self._chk(self._lib.rb3gpu_mg_rank_plain_walkers(self._h, bwt.size, bwt.ctypes.data, w.shape[0], w.ctypes.data, pos.ctypes.data, acc2.ctypes.data), "rb3gpu_mg_rank_plain_walkers")
        return pos, acc2

    def rank1a(self, k):
        k = np.ascontiguousarray(k, dtype=np.int64)
        ok = np.empty((k.size, 6), dtype=np.int64)
        self._chk(self._lib.rb3gpu_rank1a_batch(self._h, k.size, k.ctypes.data, ok.ctypes.data), "rb3gpu_rank1a_batch")
        return ok

    def get_acc(self):
        acc = np.zeros(7, dtype=np.int64)
        self._chk(self._lib.rb3gpu_get_acc(self._h, acc.ctypes.data), "rb3gpu_get_acc")
        return acc

    def get_tot(self):
        return int(self._lib.rb3gpu_get_tot(self._h))

    def export_plain(self):
        out = np.empty(self.get_tot(), dtype=np.uint8)
        self._chk(self._lib.rb3gpu_export_plain(self._h, out.ctypes.data), "rb3gpu_export_plain")
        return out

    def export_fmd_words(self):
        """the data section of the index's .fmd, packed on the GPU (rb3_enc_fmr2fmd + rld_enc, fm-index.c:31-52): uint64 array"""
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        self._chk(self._lib.rb3gpu_export_fmd_words(self._h, ctypes.byref(p), ctypes.byref(n)), "rb3gpu_export_fmd_words")
        out = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint64)), shape=(n.value,)).copy()
        self._lib.rb3gpu_host_free(p)
        return out

    def merge_index(self, other):
        """merge the whole index of another handle (any GPU of the node) into this one (rb3_fmi_merge, fm-index.c:251-277)"""
        self._chk(self._lib.rb3gpu_merge_index(self._h, other._h), "rb3gpu_merge_index")

    def export_plain_dev(self, d_out):
        self._chk(self._lib.rb3gpu_export_plain_dev(self._h, d_out), "rb3gpu_export_plain_dev")

    def bwt_from_text(self, text, step=0):
        """suffix-sort a batch text on the GPU (rb3_build_sais, sais-ss.c:10-56): returns (device pointer of the
        BWT -- free it with dev_free --, ckrow or None)"""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        p = ctypes.c_void_p()
        self._chk(self._lib.rb3gpu_dev_alloc(self._h, text.size + 16, ctypes.byref(p)), "rb3gpu_dev_alloc")
        ck = np.empty((text.size + step - 1) // step, dtype=np.int64) if step > 0 else None
        self._chk(self._lib.rb3gpu_bwt_from_text(self._h, text.size, text.ctypes.data, p, step, ck.ctypes.data if ck is not None else None), "rb3gpu_bwt_from_text")
        return p, ck

    def sort_text(self, text):
        """suffix-sort a batch text on the GPU: returns device pointers (BWT, text-order words) for merge_text_dev;
        free both with dev_free"""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        p, q = ctypes.c_void_p(), ctypes.c_void_p()
        self._chk(self._lib.rb3gpu_dev_alloc(self._h, text.size + 16, ctypes.byref(p)), "rb3gpu_dev_alloc")
        self._chk(self._lib.rb3gpu_dev_alloc(self._h, text.size * 8, ctypes.byref(q)), "rb3gpu_dev_alloc")
        self._chk(self._lib.rb3gpu_sort_text(self._h, text.size, text.ctypes.data, p, q), "rb3gpu_sort_text")
        return p, q

    def sort_text_sa(self, text):
        """sort_text with the suffix array: (BWT, text-order words, suffix array) device pointers; free all three with dev_free"""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        p, q, r = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._chk(self._lib.rb3gpu_dev_alloc(self._h, text.size + 16, ctypes.byref(p)), "rb3gpu_dev_alloc")
        self._chk(self._lib.rb3gpu_dev_alloc(self._h, text.size * 8, ctypes.byref(q)), "rb3gpu_dev_alloc")
        self._chk(self._lib.rb3gpu_dev_alloc(self._h, text.size * 4, ctypes.byref(r)), "rb3gpu_dev_alloc")
        self._chk(self._lib.rb3gpu_sort_text_sa(self._h, text.size, text.ctypes.data, p, q, r), "rb3gpu_sort_text_sa")
        return p, q, r

    def merge_text_dev(self, d_bwt, d_tw, length, walkers, commit=True, d_sa=None):
        """merge a batch given by its BWT and text-order words (walkers by text position: host.walkers_text); d_sa: the batch's
        suffix array if the caller has it (rb3gpu_merge_text_sa_dev)"""
        if isinstance(walkers, (int, np.integer)):  # the number of strings: one walker per string, made on the device
            self._chk(self._lib.rb3gpu_merge_text_sa_dev(self._h, length, d_bwt, d_tw, d_sa, int(walkers), None, 1 if commit else 0), "rb3gpu_merge_text_sa_dev")
            return
        w = self._walkers(walkers)
        self._chk(self._lib.rb3gpu_merge_text_sa_dev(self._h, length, d_bwt, d_tw, d_sa, w.shape[0], w.ctypes.data, 1 if commit else 0), "rb3gpu_merge_text_sa_dev")

    def merge_text_step_dev(self, d_bwt, d_tw, length, n_strings, step, commit=True, d_sa=None):
        """rb3gpu_merge_text_step_dev: the walker list made on the device (a walker per string and one every `step` text positions)"""
        self._chk(self._lib.rb3gpu_merge_text_step_dev(self._h, int(length), d_bwt, d_tw, d_sa, int(n_strings), int(step), 1 if commit else 0), "rb3gpu_merge_text_step_dev")

    def walkers_step_dev(self, d_tw, length, n_strings, step):
        """the walker list rb3gpu_merge_text_step_dev makes on the device, as an (n, 4) int64 array (row = text position, ka0, nsteps, flags)"""
        n, p = ctypes.c_int64(0), ctypes.c_void_p()
        self._chk(self._lib.rb3gpu_walkers_step_dev(self._h, int(length), d_tw, int(n_strings), int(step), ctypes.byref(n), ctypes.byref(p)), "rb3gpu_walkers_step_dev")
        w = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_int64)), shape=(n.value, 4)).copy()
        self._lib.rb3gpu_host_free(p)
        return w

    def mg_rank_text_dev(self, d_bwt, d_tw, length, walkers):
        pos = np.empty(length, dtype=np.int64)
        acc2 = np.zeros(7, dtype=np.int64)
        if isinstance(walkers, (int, np.integer)):
            self._chk(self._lib.rb3gpu_mg_rank_text_dev(self._h, length, d_bwt, d_tw, int(walkers), None, pos.ctypes.data, acc2.ctypes.data), "rb3gpu_mg_rank_text_dev")
            return pos, acc2
        w = self._walkers(walkers)
        self._chk(self._lib.rb3gpu_mg_rank_text_dev(self._h, length, d_bwt, d_tw, w.shape[0], w.ctypes.data, pos.ctypes.data, acc2.ctypes.data), "rb3gpu_mg_rank_text_dev")
        return pos, acc2

    def dev_download(self, p, nbytes):
        out = np.empty(nbytes, dtype=np.uint8)
        self._chk(self._lib.rb3gpu_dev_download(self._h, out.ctypes.data, p, nbytes), "rb3gpu_dev_download")
        return out

    def dev_download_i64(self, p, n):
        return self.dev_download(p, int(n) * 8).view(np.int64)

    def ssa_gen(self, ssa_shift):
        """sampled suffix array of the index (rb3_ssa_gen, ssa.c:54-81): (ms, r2i[m], ssa[n_ssa]) as uint64"""
        m, n_ssa, ms = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        self._chk(self._lib.rb3gpu_ssa_dims(self._h, ssa_shift, ctypes.byref(m), ctypes.byref(n_ssa), ctypes.byref(ms)), "rb3gpu_ssa_dims")
        r2i = np.empty(m.value, dtype=np.uint64)
        ssa = np.empty(max(n_ssa.value, 1), dtype=np.uint64)
        self._chk(self._lib.rb3gpu_ssa_gen(self._h, ssa_shift, r2i.ctypes.data, ssa.ctypes.data), "rb3gpu_ssa_gen")
        return ms.value, r2i, ssa[:n_ssa.value]

    def export_runs(self):
        runs = []

        def emit(_data, c, l):
            runs.append((c, l))
            return 0
        cb = EMIT_F(emit)
        self._chk(self._lib.rb3gpu_export_runs(self._h, cb, None), "rb3gpu_export_runs")
        return runs

    def from_runs(self, runs):
        arr = np.array([(l << 3) | c for c, l in runs], dtype=np.uint64)
        self._chk(self._lib.rb3gpu_from_runs(self._h, arr.size, arr.ctypes.data), "rb3gpu_from_runs")

    # -- device-resident variants ------------------------------------------------------------
    def from_fmd_file(self, path):
        """load an .fmd file, decoding it on the device (rb3gpu_from_fmd_words)"""
        raw = np.fromfile(path, dtype=np.uint8)
        assert raw[:4].tobytes() == b"RLD\x03", "not an FMD file"
        hdr = raw[8:32].view(np.uint64)
        mc = raw[32:80].view(np.uint64).astype(np.int64)
        n_words = int(hdr[1]) // 8
        words = np.ascontiguousarray(raw[80:80 + n_words * 8]).view(np.uint64)
        self._chk(self._lib.rb3gpu_from_fmd_words(self._h, n_words, words.ctypes.data, mc.ctypes.data), "rb3gpu_from_fmd_words")

    def dev_upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = ctypes.c_void_p()
        self._chk(self._lib.rb3gpu_dev_alloc(self._h, arr.nbytes + 16, ctypes.byref(p)), "rb3gpu_dev_alloc")
        self._chk(self._lib.rb3gpu_dev_upload(self._h, p, arr.ctypes.data, arr.nbytes), "rb3gpu_dev_upload")
        return p

    def dev_free(self, p):
        self._chk(self._lib.rb3gpu_dev_free(self._h, p), "rb3gpu_dev_free")

    def from_plain_dev(self, d_bwt, length):
        self._chk(self._lib.rb3gpu_from_plain_dev(self._h, length, d_bwt), "rb3gpu_from_plain_dev")

    def merge_plain_dev(self, d_bwt, length, commit=True):
        self._chk(self._lib.rb3gpu_merge_plain_dev(self._h, length, d_bwt, 1 if commit else 0), "rb3gpu_merge_plain_dev")

    # -- interval-sharded index (multi-GPU; ropebwt3_amd/multi.py) ------------------------------
    def dev_alloc(self, nbytes):
        p = ctypes.c_void_p()
        self._chk(self._lib.rb3gpu_dev_alloc(self._h, int(nbytes), ctypes.byref(p)), "rb3gpu_dev_alloc")
        return p.value

    def dev_copy(self, d_dst, d_src, nbytes):
        self._chk(self._lib.rb3gpu_dev_copy(self._h, d_dst, d_src, int(nbytes)), "rb3gpu_dev_copy")

    def dev_memset(self, d_dst, byte, nbytes):
        self._chk(self._lib.rb3gpu_dev_memset(self._h, d_dst, int(byte), int(nbytes)), "rb3gpu_dev_memset")

    def dev_upload_to(self, d_dst, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self._lib.rb3gpu_dev_upload(self._h, d_dst, arr.ctypes.data, arr.nbytes), "rb3gpu_dev_upload")

    def sh_step(self, n_states, d_in, d_tw, d_ka, adj, bounds, my_iv, d_send):
        """one LF step of the states resident on this rank; returns counts[n_iv + 1] (see include/rb3gpu.h)"""
        adj = np.ascontiguousarray(adj, dtype=np.int64)
        bounds = np.ascontiguousarray(bounds, dtype=np.int64)
        counts = np.zeros(bounds.size, dtype=np.int64)
        self._chk(self._lib.rb3gpu_sh_step(self._h, int(n_states), d_in, d_tw, d_ka, adj.ctypes.data, bounds.size - 1, bounds.ctypes.data, int(my_iv), d_send, counts.ctypes.data), "rb3gpu_sh_step")
        return counts

    def sh_finish(self, jlo, n_rows, d_bwt, d_ka, iv_start, commit=True):
        self._chk(self._lib.rb3gpu_sh_finish(self._h, int(jlo), int(n_rows), d_bwt, d_ka, int(iv_start), 1 if commit else 0), "rb3gpu_sh_finish")

    def sh_merge(self, comm, bounds, d_bwt, d_tw, n2, sent_tp, commit=True):
        """rb3gpu_sh_merge: one batch merged into the interval-sharded index, the lock-step loop inside the library.  comm: a
        GroupComm / RcclComm / CallbackComm of this rank.  Returns (bounds after the merge, lock-step rounds)."""
        b = np.array(bounds, dtype=np.int64)
        tp = np.ascontiguousarray(sent_tp, dtype=np.int64)
        rounds = ctypes.c_int64(0)
        self._chk(self._lib.rb3gpu_sh_merge(self._h, ctypes.addressof(comm.struct), b.ctypes.data, int(n2), d_bwt, d_tw, tp.size, tp.ctypes.data, 1 if commit else 0, ctypes.addressof(rounds)), "rb3gpu_sh_merge")
        return b, int(rounds.value)

    def sh_merge_text(self, comm, bounds, d_tprev, d_tw_slice, n2, sent_tp, commit=True):
        """rb3gpu_sh_merge_text: the same with the BATCH sharded -- d_tprev: the symbol before every text position (1 byte each, whole),
        d_tw_slice: the text-order words of this rank's text range only; the rows come from the owners of the text ranges at the end"""
        b = np.array(bounds, dtype=np.int64)
        tp = np.ascontiguousarray(sent_tp, dtype=np.int64)
        rounds = ctypes.c_int64(0)
        self._chk(self._lib.rb3gpu_sh_merge_text(self._h, ctypes.addressof(comm.struct), b.ctypes.data, int(n2), d_tprev, d_tw_slice, tp.size, tp.ctypes.data, 1 if commit else 0, ctypes.addressof(rounds)), "rb3gpu_sh_merge_text")
        return b, int(rounds.value)

    def tprev_from_tw(self, d_tw, n2):
        """device array of n2 bytes: the symbol before every text position (rb3gpu_tprev_from_tw); free it with dev_free"""
        d = self.dev_alloc(int(n2) + 64)
        self._chk(self._lib.rb3gpu_tprev_from_tw(self._h, int(n2), d_tw, d), "rb3gpu_tprev_from_tw")
        return d

    def balanced_bounds(self, n):
        b = np.zeros(n + 1, dtype=np.int64)
        self._chk(self._lib.rb3gpu_balanced_bounds(self._h, int(n), b.ctypes.data), "rb3gpu_balanced_bounds")
        return b

    def sync(self):
        self._chk(self._lib.rb3gpu_sync(self._h), "rb3gpu_sync")

    def stats(self):
        st = Stats()
        self._chk(self._lib.rb3gpu_stats(self._h, ctypes.byref(st)), "rb3gpu_stats")
        return st.as_dict()

    def stats_reset(self):
        self._lib.rb3gpu_stats_reset(self._h)

    def buffers(self):
        """{name: bytes} of every device buffer the handle holds right now (rb3gpu_buffer_bytes)"""
        out, i = {}, 0
        name, nb = ctypes.c_char_p(), ctypes.c_int64()
        while self._lib.rb3gpu_buffer_bytes(self._h, i, ctypes.byref(name), ctypes.byref(nb)) == 0:
            if nb.value:
                out[name.value.decode()] = int(nb.value)
            i += 1
        return out


class CommGroup:
    """rb3gpu_group_*: the ranks of an interval-sharded index as THREADS of this process, one handle each (barriers + peer copies)"""

    def __init__(self, world, lib=None, hooks=False):
        self._lib = load_library(hooks, lib)
        self.world = int(world)
        self._g = self._lib.rb3gpu_group_create(self.world)
        if not self._g:
            raise Rb3GpuError(-3, "rb3gpu_group_create")

    def comm(self, rank, engine):
        c = GroupComm()
        c.struct = CommStruct()
        r = self._lib.rb3gpu_group_comm(self._g, int(rank), engine._h, ctypes.addressof(c.struct))
        if r < 0:
            raise Rb3GpuError(int(r), "rb3gpu_group_comm")
        c.group = self   # keeps the group alive
        return c

    def abort(self):
        self._lib.rb3gpu_group_abort(self._g)

    def close(self):
        if self._g:
            self._lib.rb3gpu_group_destroy(self._g)
            self._g = None


class GroupComm:
    struct = None


def ipc_peer_enable(engine, comm):
    """rb3gpu_ipc_peer_enable: PEER ROUNDS for ranks that are processes of one node (HIP IPC memory and event handles, a spin barrier in shared memory) on top
    of any communicator of world > 1.  COLLECTIVE: every rank calls it.  True: enabled on every rank; False: not available (the communicator is as it was)."""
    r = engine._lib.rb3gpu_ipc_peer_enable(engine._h, ctypes.addressof(comm.struct))
    if r == 0:
        return True
    if r == -7:
        return False
    raise Rb3GpuError(int(r), "rb3gpu_ipc_peer_enable")


def ipc_peer_disable(engine, comm):
    engine._lib.rb3gpu_ipc_peer_disable(ctypes.addressof(comm.struct))


class RcclComm:
    """rb3gpu_rccl_*: one process per GPU; grouped ncclSend/ncclRecv on the engine's stream.  `uid`: the 128 bytes rank 0 got
    from RcclComm.unique_id() and sent to the other ranks."""

    @staticmethod
    def unique_id(lib=None):
        l = load_library(False, lib)
        buf = ctypes.create_string_buffer(128)
        r = l.rb3gpu_rccl_unique_id(buf)
        if r < 0:
            raise Rb3GpuError(int(r), "rb3gpu_rccl_unique_id")
        return buf.raw

    def __init__(self, engine, rank, world, uid):
        self._lib = engine._lib
        self.struct = CommStruct()
        self.rank, self.world = int(rank), int(world)
        buf = ctypes.create_string_buffer(bytes(uid), 128)
        r = self._lib.rb3gpu_rccl_comm_create(engine._h, self.rank, self.world, buf, ctypes.addressof(self.struct))
        if r < 0:
            raise Rb3GpuError(int(r), "rb3gpu_rccl_comm_create")
        self._open = True

    def close(self):
        if self._open:
            self._lib.rb3gpu_rccl_comm_destroy(ctypes.addressof(self.struct))
            self._open = False


class CallbackComm:
    """a communicator whose two collectives are Python callables (tests: gloo, threads):
    all_gather(vec int64[n]) -> array [world, n];  exchange(d_send, stride, send_counts, d_recv, recv_counts) with device pointers
    as ints (states of 16 bytes; region d of the send buffer starts at d_send + d * stride * 16)."""

    def __init__(self, rank, world, all_gather, exchange, abort=None):
        self.rank, self.world = int(rank), int(world)
        self.error = None

        def _ag(ctx, send, n, recv):
            try:
                out = np.asarray(all_gather(np.ctypeslib.as_array(send, shape=(n,)).copy()), dtype=np.int64).reshape(self.world * n)
                np.ctypeslib.as_array(recv, shape=(self.world * n,))[:] = out
                return 0
            except BaseException as e:   # (an exception must not unwind through the C frames)
                self.error = e
                return -6

        def _a2a(ctx, d_send, stride, send_cnt, d_recv, recv_cnt, stream):
            try:
                sc = np.ctypeslib.as_array(send_cnt, shape=(self.world,)).copy()
                rc = np.ctypeslib.as_array(recv_cnt, shape=(self.world,)).copy()
                # The send regions are complete ON THE ENGINE'S STREAM (include/rb3gpu.h): a callable that reads them with the runtime's synchronous copies -- the null
                # stream, which does not wait for a non-blocking stream -- must not start before that stream is done.  (The merge's last phase, the exchange with the owners
                # of the text ranges, calls this right behind the kernel that fills the regions: eight processes lost rows there, four got away with it.)
                if stream:
                    load_library().rb3gpu_stream_sync(stream)
                exchange(int(d_send or 0), int(stride), sc, int(d_recv or 0), rc)
                return 0
            except BaseException as e:
                self.error = e
                return -6

        def _ab(ctx):
            if abort is not None:
                try:
                    abort()
                except BaseException:
                    pass

        self._keep = (ALL_GATHER_F(_ag), ALL_TO_ALL_F(_a2a), ABORT_F(_ab))   # the C side holds raw pointers to these
        self.struct = CommStruct(None, self.rank, self.world, ctypes.cast(self._keep[0], ctypes.c_void_p), ctypes.cast(self._keep[1], ctypes.c_void_p), ctypes.cast(self._keep[2], ctypes.c_void_p))


class Shard:
    """rb3gpu_shard_*: the index of `engine` cut into n intervals (one handle per device of `devices`), batches merged by n threads
    inside the library, put back together by gather()"""

    def __init__(self, engine, devices):
        self._e, self._lib = engine, engine._lib
        dev = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        opt = Opt()
        self._lib.rb3gpu_opt_init(ctypes.byref(opt))
        opt.verbose = 1
        self._s = self._lib.rb3gpu_shard_split(engine._h, len(devices), dev, ctypes.addressof(opt))
        if not self._s:
            raise Rb3GpuError(-1, "rb3gpu_shard_split")
        self.n = len(devices)

    def bounds(self):
        b = np.zeros(self.n + 1, dtype=np.int64)
        self._lib.rb3gpu_shard_bounds(self._s, b.ctypes.data)
        return b

    def merge(self, d_bwt, d_tw, n2, sent_tp):
        tp = np.ascontiguousarray(sent_tp, dtype=np.int64)
        rounds = ctypes.c_int64(0)
        r = self._lib.rb3gpu_shard_merge(self._s, int(n2), d_bwt, d_tw, tp.size, tp.ctypes.data, ctypes.addressof(rounds))
        if r < 0:
            raise Rb3GpuError(int(r), "rb3gpu_shard_merge")
        return int(rounds.value)

    def get_acc(self):
        a = np.zeros(7, dtype=np.int64)
        r = self._lib.rb3gpu_shard_get_acc(self._s, a.ctypes.data)
        if r < 0:
            raise Rb3GpuError(int(r), "rb3gpu_shard_get_acc")
        return a

    def export_plain(self):
        """the whole index as one byte per symbol, straight from the intervals in rank order (rb3gpu_shard_export_runs: no gather)"""
        out = []
        runs = []

        def emit(_d, c, l):
            runs.append((int(c), int(l)))
            return 0
        cb = EMIT_F(emit)
        r = self._lib.rb3gpu_shard_export_runs(self._s, cb, None)
        if r < 0:
            raise Rb3GpuError(int(r), "rb3gpu_shard_export_runs")
        assert all(runs[i][0] != runs[i + 1][0] for i in range(len(runs) - 1)), "runs that meet at a seam must be joined"
        for c, l in runs:
            out.append(np.full(l, c, dtype=np.uint8))
        return np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)

    def export_run_words(self):
        """(starts, symbols) of the maximal runs of the whole index and its length, from rb3gpu_shard_export_run_words"""
        words, end = [], [-1]

        def emit(_d, n, w, e):
            if n > 0:
                words.append(np.ctypeslib.as_array(w, shape=(n,)).copy())
            if e >= 0:
                end[0] = int(e)
            return 0
        cb = EMITW_F(emit)
        r = self._lib.rb3gpu_shard_export_run_words(self._s, cb, None)
        if r < 0:
            raise Rb3GpuError(int(r), "rb3gpu_shard_export_run_words")
        w = np.concatenate(words) if words else np.zeros(0, dtype=np.uint64)
        return (w >> np.uint64(3)).astype(np.int64), (w & np.uint64(7)).astype(np.uint8), end[0]

    def rebalance(self, pct=25):
        r = self._lib.rb3gpu_shard_rebalance(self._s, int(pct))
        if r < 0:
            raise Rb3GpuError(int(r), "rb3gpu_shard_rebalance")
        return int(r)

    def handle_stats(self, i):
        st = Stats()
        self._lib.rb3gpu_stats(self._lib.rb3gpu_shard_handle(self._s, int(i)), ctypes.byref(st))
        return st.as_dict()

    def destroy(self):
        s, self._s = self._s, None
        if s:
            self._lib.rb3gpu_shard_destroy(s)

    def gather(self):
        s, self._s = self._s, None
        r = self._lib.rb3gpu_shard_gather(s)
        if r < 0:
            raise Rb3GpuError(int(r), "rb3gpu_shard_gather")
