"""pangea_plus_amd — host-side mirror of the reference's command lines over libpangea_hip.so.

The reference (Bioinfo-Tools/PANGEA-plus) is a set of CLI tools; this package exposes one
Python function per reference command line with the same argument meaning, all of them thin
ctypes calls into the C ABI declared in include/pangea_hip.h.  There is no CPU fallback: if the
HIP library is missing or no GPU is visible, every compute call raises PangeaError.
"""
from ._capi import (  # noqa: F401
    PangeaError, lib, lib_path, init, device_count, version,
    SynthCfg, Db, Reads, Hits, Rdp, TaxDb, StageTimes,
    blastn, soap, soap_index, makeblastdb, tax_class, taxcollector, consensus, megaclust2, megaclustable, megaclust_batch, trim2,
)
