"""krust_amd -- MI355X-native canonical k-mer counting (drop-in for krust's hot path).

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C ABI of include/kmerhip.h
  native.py  ctypes binding of that C ABI (plumbing; no CPU fallback)
"""
from . import native  # noqa: F401
from .native import (DeviceCounter, DeviceGroup, KmerHipError, KmerLengthError, canonical, comm_unique_id, lib, owner,  # noqa: F401
                     pack, synth_reads_device, unpack)

__all__ = ["DeviceCounter", "DeviceGroup", "KmerHipError", "KmerLengthError", "canonical", "comm_unique_id", "lib", "owner",
           "pack", "synth_reads_device", "unpack"]
