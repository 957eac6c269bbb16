"""ctypes binding of libkgwas.so (the C ABI declared in include/kgwas.h).

The shared library is built in-tree by ``make -C kmersgwas_amd/csrc`` (see __graft_entry__.build()).
There is no Python or CPU fallback for the compute path: if the library is missing, importing this
module raises, and if no HIP device is usable the compute entry points return KGWAS_ERR_DEVICE,
which surfaces here as KgwasError.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KGWAS_LIB: load another build of the same library (kernel experiments, tools/coarse_variants.sh)
LIB_PATH = os.environ.get("KGWAS_LIB") or os.path.join(_HERE, "lib", "libkgwas.so")

KGWAS_OK = 0
KGWAS_ERR_ARG, KGWAS_ERR_IO, KGWAS_ERR_FORMAT, KGWAS_ERR_DEVICE, KGWAS_ERR_STATE, KGWAS_ERR_NOMEM = -1, -2, -3, -4, -5, -6
KERNEL_AUTO, KERNEL_VALU, KERNEL_MFMA, KERNEL_COARSE, KERNEL_NARROW = 0, 1, 2, 3, 4

# Every symbol include/kgwas.h declares (tests check the library exports each one).
SYMBOLS = [
    "kgwas_last_error", "kgwas_version", "kgwas_abi_version", "kgwas_device_count", "kgwas_host_cpu_quota",
    "kgwas_table_open", "kgwas_table_info", "kgwas_table_name", "kgwas_table_column_map", "kgwas_table_read_rows",
    "kgwas_table_close",
    "kgwas_pheno_load", "kgwas_pheno_info", "kgwas_pheno_name", "kgwas_pheno_accession", "kgwas_pheno_values",
    "kgwas_pheno_free", "kgwas_min_count",
    "kgwas_heap_selfcheck", "kgwas_heap_new", "kgwas_heap_add_many", "kgwas_heap_size", "kgwas_heap_pop_all", "kgwas_heap_output_list", "kgwas_heap_rows_sorted", "kgwas_heap_output_to_file", "kgwas_select_check",
    "kgwas_heap_free",
    "kgwas_scan_create", "kgwas_scan_feed_device", "kgwas_scan_feed_host", "kgwas_scan_feed_table", "kgwas_scan_finish", "kgwas_scan_result",
    "kgwas_scan_history", "kgwas_scan_get_stats", "kgwas_scan_reset", "kgwas_scan_lowest", "kgwas_scan_select_mode", "kgwas_scan_debug_residuals", "kgwas_scan_absorb", "kgwas_scan_history_above", "kgwas_scan_heaps_export", "kgwas_scan_heaps_import", "kgwas_scan_expect_finish", "kgwas_scan_history_above_msgs", "kgwas_scan_heaps_export_msgs", "kgwas_scan_destroy", "kgwas_scan_scores_dense",
    "kgwas_merge_shards",
    "kgwas_multiscan_create", "kgwas_multiscan_run_table", "kgwas_multiscan_run_device", "kgwas_multiscan_finish",
    "kgwas_multiscan_result", "kgwas_multiscan_get_stats", "kgwas_multiscan_destroy", "kgwas_kinship_table_multi",
    "kgwas_kinship_create", "kgwas_kinship_feed_device", "kgwas_kinship_feed_host", "kgwas_kinship_feed_table", "kgwas_kinship_partials",
    "kgwas_kinship_from_partials", "kgwas_kinship_get_stats", "kgwas_kinship_destroy", "kgwas_kinship_format",
    "kgwas_write_plink", "kgwas_write_plink_many", "kgwas_table_to_bed",
    "kgwas_snps_open", "kgwas_snps_info", "kgwas_snps_scores", "kgwas_snps_best", "kgwas_snps_write", "kgwas_snps_close",
    "kgwas_synth_rows_device", "kgwas_synth_rows_host",
]


class KgwasError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libkgwas error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


class ScanParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32),
        ("n_acc_file", C.c_uint64), ("n_acc", C.c_uint64), ("col", C.POINTER(C.c_uint64)),
        ("n_pheno", C.c_uint64), ("Y", C.POINTER(C.c_float)), ("topn", C.POINTER(C.c_uint64)),
        ("min_count", C.c_uint64), ("chunk_rows", C.c_uint64),
        ("host_threads", C.c_uint32), ("kernel", C.c_uint32), ("record_history", C.c_uint32),
        ("count_patterns", C.c_uint32),
    ]


class ScanStats(C.Structure):
    _fields_ = [
        ("rows_fed", C.c_uint64), ("rows_tested", C.c_uint64), ("candidates", C.c_uint64),
        ("heap_pushes", C.c_uint64), ("chunks", C.c_uint64), ("score_launches", C.c_uint64),
        ("score_kernel_ms", C.c_double), ("squeeze_kernel_ms", C.c_double), ("replay_ms", C.c_double),
        ("gpu_wait_ms", C.c_double), ("dense_ms", C.c_double),
        ("coarse_kernel_ms", C.c_double), ("coarse_launches", C.c_uint64),
        ("kernel_used", C.c_uint32), ("direct_mode", C.c_uint32), ("patterns", C.c_uint64),
        ("coarse_mode_tiles", C.c_uint32 * 2), ("coarse_mode_lgroups", C.c_uint32 * 2),
        ("coarse_mode_launches", C.c_uint64 * 2), ("coarse_mode_rows", C.c_uint64 * 2),
        ("coarse_mode_ms", C.c_double * 2),
        ("replay_cpu_ms", C.c_double), ("replay_tail_ms", C.c_double),
        ("coarse_mode_tile_slices", C.c_uint32 * 2),
        ("coarse_mx", C.c_uint32), ("coarse_mx_s1_fp6", C.c_uint32), ("coarse_mx_steps", C.c_uint32),
        ("replay_threads", C.c_uint32),
        ("replay_min_ms", C.c_double), ("replay_wall_ms", C.c_double),
        ("replay_splits", C.c_uint64),
        ("columns_popped_ahead", C.c_uint64),
        ("coarse_mx32", C.c_uint32), ("coarse_mx_stream", C.c_uint32),
        ("columns_selected", C.c_uint32), ("columns_replayed_at_finish", C.c_uint32),
    ]

    def as_dict(self):
        d = {}
        for k, _ in self._fields_:
            v = getattr(self, k)
            d[k] = list(v) if hasattr(v, "__len__") else v
        return d


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libkgwas.so not found at %s — build it with `make -C kmersgwas_amd/csrc` "
        "(or __graft_entry__.build()). There is no fallback implementation." % LIB_PATH)



def _load_hip_runtime():
    """A process must hold exactly ONE HIP runtime. libkgwas.so has no DT_NEEDED on libamdhip64 and
    binds to whichever copy is already global: the one bundled with the PyTorch wheel when torch is
    installed (so that device pointers / streams of torch tensors mean the same thing on both sides and
    a later `import torch` finds its runtime already loaded), /opt/rocm's otherwise."""
    import importlib.util
    cands = []
    try:
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cands.append(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so"))
    except Exception:  # pragma: no cover
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cands += [os.path.join(rocm, "lib", "libamdhip64.so.7"), os.path.join(rocm, "lib", "libamdhip64.so"),
              "libamdhip64.so.7", "libamdhip64.so"]
    errs = []
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            return C.CDLL(c, mode=C.RTLD_GLOBAL), c
        except OSError as e:  # pragma: no cover
            errs.append("%s: %s" % (c, e))
    raise ImportError("no HIP runtime (libamdhip64) could be loaded: %s" % "; ".join(errs))


_hip, HIP_RUNTIME_PATH = _load_hip_runtime()
lib = C.CDLL(LIB_PATH)

_vp, _u64, _u32, _i32, _dbl = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_double
_pp = C.POINTER(C.c_void_p)
_pu64 = C.POINTER(C.c_uint64)
_pdbl = C.POINTER(C.c_double)
_pstr = C.POINTER(C.c_char_p)

lib.kgwas_last_error.restype = C.c_char_p
lib.kgwas_version.restype = C.c_int
lib.kgwas_device_count.argtypes = [C.POINTER(C.c_int)]
lib.kgwas_abi_version.argtypes = []
lib.kgwas_abi_version.restype = C.c_uint32
ABI_VERSION = 6  # KGWAS_ABI_VERSION of include/kgwas.h this mirror was written against
if lib.kgwas_abi_version() != ABI_VERSION:
    raise ImportError("libkgwas.so speaks ABI version %d, kmersgwas_amd/capi.py %d: rebuild (make -C kmersgwas_amd/csrc)" % (lib.kgwas_abi_version(), ABI_VERSION))
lib.kgwas_host_cpu_quota.argtypes = []
lib.kgwas_host_cpu_quota.restype = C.c_uint32
lib.kgwas_table_open.argtypes = [C.c_char_p, _u32, _pp]
lib.kgwas_table_info.argtypes = [_vp, _pu64, _pu64, _pu64, C.POINTER(_u32)]
lib.kgwas_table_name.argtypes = [_vp, _u64, _pstr]
lib.kgwas_table_column_map.argtypes = [_vp, _pstr, _u64, _pu64]
lib.kgwas_table_read_rows.argtypes = [_vp, _u64, _u64, _vp]
lib.kgwas_table_close.argtypes = [_vp]
lib.kgwas_table_close.restype = None
lib.kgwas_pheno_load.argtypes = [C.c_char_p, _pp]
lib.kgwas_pheno_info.argtypes = [_vp, _pu64, _pu64]
lib.kgwas_pheno_name.argtypes = [_vp, _u64, _pstr]
lib.kgwas_pheno_accession.argtypes = [_vp, _u64, _pstr]
lib.kgwas_pheno_values.argtypes = [_vp, C.POINTER(C.POINTER(C.c_float))]
lib.kgwas_pheno_free.argtypes = [_vp]
lib.kgwas_pheno_free.restype = None
lib.kgwas_min_count.argtypes = [_u64, _dbl, _u64]
lib.kgwas_min_count.restype = _u64
lib.kgwas_heap_new.argtypes = [_u64, _pp]
lib.kgwas_heap_add_many.argtypes = [_vp, _vp, _vp, _vp, _u64]
lib.kgwas_heap_size.argtypes = [_vp, _pu64, _pu64, _pdbl]
lib.kgwas_heap_pop_all.argtypes = [_vp, _vp, _vp, _vp]
lib.kgwas_heap_output_list.argtypes = [_vp, _vp, _vp, _vp]
lib.kgwas_heap_rows_sorted.argtypes = [_vp, _vp]
lib.kgwas_heap_output_to_file.argtypes = [_vp, C.c_char_p, C.c_int]
lib.kgwas_select_check.argtypes = [C.c_uint64, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int64, C.POINTER(C.c_int), _vp, _vp, _vp, C.POINTER(C.c_uint64)]
lib.kgwas_heap_free.argtypes = [_vp]
lib.kgwas_heap_free.restype = None
lib.kgwas_scan_create.argtypes = [C.POINTER(ScanParams), _pp]
lib.kgwas_scan_feed_device.argtypes = [_vp, _vp, _u64, _u64, _vp]
lib.kgwas_scan_feed_host.argtypes = [_vp, _vp, _u64, _u64]
lib.kgwas_scan_feed_table.argtypes = [_vp, _vp, _u64, _u64]
lib.kgwas_scan_finish.argtypes = [_vp]
lib.kgwas_scan_expect_finish.argtypes = [_vp]
lib.kgwas_scan_result.argtypes = [_vp, _u64, _pu64, C.POINTER(_pu64), C.POINTER(_pdbl), C.POINTER(_pu64)]
lib.kgwas_scan_history.argtypes = [_vp, _u64, _pu64, C.POINTER(_pu64), C.POINTER(_pdbl), C.POINTER(_pu64)]
lib.kgwas_scan_get_stats.argtypes = [_vp, C.POINTER(ScanStats)]
lib.kgwas_scan_reset.argtypes = [_vp]
lib.kgwas_scan_lowest.argtypes = [_vp, _vp, _vp]
lib.kgwas_scan_select_mode.argtypes = [_vp, C.POINTER(C.c_int)]
lib.kgwas_scan_select_mode.restype = C.c_int
lib.kgwas_scan_debug_residuals.argtypes = [_vp, C.c_uint32, C.c_uint64, _vp]
lib.kgwas_scan_debug_residuals.restype = C.c_int
lib.kgwas_heap_selfcheck.argtypes = [C.c_uint32]
lib.kgwas_heap_selfcheck.restype = C.c_int
lib.kgwas_scan_absorb.argtypes = [_vp, _u64, _vp, _pp, _pp, _pp]
lib.kgwas_scan_history_above.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp]
lib.kgwas_scan_heaps_export.argtypes = [_vp, _u64, _vp, _vp, _vp, _vp, _vp]
lib.kgwas_scan_history_above_msgs.argtypes = [_vp, _vp, _u64, _vp, _vp, _vp, _u64, _vp]
lib.kgwas_scan_heaps_export_msgs.argtypes = [_vp, _u64, _vp, _vp, _vp, _u64, _vp]
lib.kgwas_scan_heaps_import.argtypes = [_vp, _u64, _vp, _vp, _vp, _vp, _vp]
lib.kgwas_scan_destroy.argtypes = [_vp]
lib.kgwas_scan_destroy.restype = None
lib.kgwas_scan_scores_dense.argtypes = [_vp, _vp, C.c_int, _u64, _vp, _vp]
lib.kgwas_merge_shards.argtypes = [_u64, _vp, _u64, _vp, _pp, _pp, _pp, _u32, _pp]
lib.kgwas_multiscan_create.argtypes = [C.POINTER(ScanParams), _vp, _u32, _pp]
lib.kgwas_multiscan_run_table.argtypes = [_vp, _vp, _u64, _u64]
lib.kgwas_multiscan_run_device.argtypes = [_vp, _pp, _vp, _vp]
lib.kgwas_multiscan_finish.argtypes = [_vp]
lib.kgwas_multiscan_result.argtypes = [_vp, _u64, _pu64, C.POINTER(_pu64), C.POINTER(_pdbl), C.POINTER(_pu64)]
lib.kgwas_multiscan_get_stats.argtypes = [_vp, C.POINTER(ScanStats), _vp, _pdbl, _pdbl, _pu64]
lib.kgwas_multiscan_destroy.argtypes = [_vp]
lib.kgwas_multiscan_destroy.restype = None
lib.kgwas_kinship_table_multi.argtypes = [_vp, _u32, _vp, _u64, _vp, _pu64]
lib.kgwas_kinship_create.argtypes = [_i32, _u64, _u64, _pp]
lib.kgwas_kinship_feed_device.argtypes = [_vp, _vp, _u64, _vp]
lib.kgwas_kinship_feed_host.argtypes = [_vp, _vp, _u64]
lib.kgwas_kinship_feed_table.argtypes = [_vp, _vp, _u64, _u64]
lib.kgwas_kinship_partials.argtypes = [_vp, _vp, _pu64]
lib.kgwas_kinship_from_partials.argtypes = [_u64, _vp, _u64, _vp]
lib.kgwas_kinship_get_stats.argtypes = [_vp, _pdbl, _pu64, _pu64]
lib.kgwas_kinship_destroy.argtypes = [_vp]
lib.kgwas_kinship_destroy.restype = None
lib.kgwas_kinship_format.argtypes = [_u64, _vp, _u64, C.c_char_p, _u64]
lib.kgwas_kinship_format.restype = _u64
lib.kgwas_write_plink.argtypes = [C.c_char_p, _vp, _vp, _u64, _pstr, _vp, _u64, _vp, _vp]
lib.kgwas_write_plink_many.argtypes = [_u64, _pstr, _vp, _vp, _u64, _pstr, _vp, _vp, _pp, _pp, _u32]
lib.kgwas_snps_open.argtypes = [C.c_char_p, _pstr, _u64, _vp]
lib.kgwas_snps_info.argtypes = [_vp, _vp, _vp, _vp]
lib.kgwas_snps_scores.argtypes = [_vp, _vp, _u64, C.c_double, C.c_int, _vp]
lib.kgwas_snps_best.argtypes = [_vp, _vp, _u64, _u64, C.c_double, C.c_int, _vp, _vp]
lib.kgwas_snps_write.argtypes = [_vp, _u64, _pstr, _vp, _vp, _u64]
lib.kgwas_snps_close.argtypes = [_vp]
lib.kgwas_snps_close.restype = None
lib.kgwas_table_to_bed.argtypes = [_vp, _vp, _u64, _pstr, _vp, _u64, _u64, C.c_int, C.c_char_p, C.c_int, _vp, _vp]
lib.kgwas_synth_rows_device.argtypes = [_vp, _u64, _u64, _u64, _u64, _vp]
lib.kgwas_synth_rows_host.argtypes = [_vp, _u64, _u64, _u64, _u64]


def check(rc: int) -> None:
    if rc != KGWAS_OK:
        raise KgwasError(rc, (lib.kgwas_last_error() or b"").decode(errors="replace"))


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    n = C.c_int(0)
    check(lib.kgwas_device_count(C.byref(n)))
    return n.value
