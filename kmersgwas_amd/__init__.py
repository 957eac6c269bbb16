"""kmersgwas_amd — MI355X-native engine for the kmersGWAS association-scan path.

The product is libkgwas.so (hand-written gfx950 HIP kernels behind the C ABI of include/kgwas.h)
plus the drop-in command-line tools in kmersgwas_amd/bin. This package is the thin ctypes
mirror of the reference's class interface used by the tests, bench.py and multi-GPU plumbing.
Importing it fails loudly if the library has not been built; there is no fallback path.
"""
from . import capi  # noqa: F401  (raises ImportError if libkgwas.so is missing)
from .capi import KgwasError, KERNEL_AUTO, KERNEL_VALU, KERNEL_MFMA, KERNEL_COARSE, KERNEL_NARROW, device_count  # noqa: F401
from .engine import (  # noqa: F401
    AssociationScan, BestAssociationsHeap, Kinship, MultiDeviceScan, kinship_table_multi, KmersTable, Phenotypes, SnpsDataBase, kinship_format, kinship_from_partials,
    merge_shards, min_count, synth_rows_device, synth_rows_host, table_to_bed, write_plink, write_plink_many,
)
