"""krust_amd -- MI355X-native canonical k-mer counting (drop-in for krust's hot path).

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C ABI of include/kmerhip.h
  native.py  ctypes binding of that C ABI (plumbing; no CPU fallback)
"""
from . import native  # noqa: F401
from .native import (DeviceCounter, DeviceGroup, KmerHipError, KmerLengthError, PinnedArray, canonical, comm_unique_id,  # noqa: F401
                     host_register, host_unregister, lib, owner, pack, synth_reads_device, unpack)

__all__ = ["DeviceCounter", "DeviceGroup", "KmerHipError", "KmerLengthError", "PinnedArray", "canonical", "comm_unique_id",
           "host_register", "host_unregister", "lib", "owner", "pack", "synth_reads_device", "unpack"]
