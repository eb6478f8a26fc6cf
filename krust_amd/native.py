"""ctypes binding of the C ABI in include/kmerhip.h (krust_amd/lib/libkmerhip.so).

This is plumbing only: every call goes to the hand-written HIP path.  There is
no CPU fallback -- if the shared library is missing, or no HIP device is usable,
the calls fail loudly (ImportError / KmerHipError).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", os.environ.get("KMERHIP_LIB", "libkmerhip.so"))  # KMERHIP_LIB: A/B builds

KH_OK = 0
KH_ERR_BAD_K = -1
KH_ERR_BAD_ARG = -2
KH_ERR_NO_DEVICE = -3
KH_ERR_OOM = -4
KH_ERR_TABLE_FULL = -5
KH_ERR_HIP = -6
KH_ERR_STATE = -7
KH_ERR_RANGE = -8
KH_ERR_FORMAT = -9
KH_ERR_RCCL = -10
KH_ERR_PEER = -11
TEXT_FASTA, TEXT_FASTQ = 1, 2


class KhConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("k", C.c_uint32), ("min_quality", C.c_int32),
                ("device", C.c_int32), ("capacity_hint", C.c_uint64), ("stream", C.c_void_p),
                ("flags", C.c_uint32), ("input_mib", C.c_uint32)]


class KhStats(C.Structure):
    _fields_ = [("bases", C.c_uint64), ("kmers", C.c_uint64), ("distinct", C.c_uint64),
                ("table_slots", C.c_uint64), ("grows", C.c_uint64), ("launches", C.c_uint64),
                ("count_kernel_ms", C.c_double), ("h2d_ms", C.c_double), ("part_batches", C.c_uint64),
                ("stage_ms", C.c_double * 8), ("text_scan_ms", C.c_double),
                ("slot_bytes", C.c_uint64)]

class KhUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class KhMergeInfo(C.Structure):
    _fields_ = [("route", C.c_uint32), ("pieces", C.c_uint32), ("unit_bytes", C.c_uint32), ("nranks", C.c_uint32),
                ("local_distinct", C.c_uint64), ("sent_units", C.c_uint64), ("recv_units", C.c_uint64),
                ("owned_distinct", C.c_uint64), ("export_ms", C.c_double), ("wait_ms", C.c_double),
                ("merge_ms", C.c_double), ("total_ms", C.c_double),
                ("nranks_seen", C.c_uint32), ("conserved", C.c_uint32), ("sent_count_sum", C.c_uint64),
                ("merged_count_sum", C.c_uint64)]


ROUTES = ("none", "dense", "regions-heads", "regions-packed", "regions", "pairs")  # KH_ROUTE_*
# names of kh_stats.stage_ms[i] (KH_STAGE_* of the header); bench.py maps them to the kernels that ran
STAGES = ("direct", "unused", "level1", "level2_count", "level2", "region", "misc", "grow")
FLAG_TRACE, FLAG_FORCE_DIRECT, FLAG_FORCE_PARTITION, FLAG_CALLER_STREAM, FLAG_DEFER_TEXT_SCAN = 1, 2, 4, 8, 16


# every symbol include/kmerhip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_U64 = C.c_uint64
SYMBOLS = {
    "kh_abi_version": (C.c_int, []),
    "kh_create": (C.c_int, [C.POINTER(_P), C.POINTER(KhConfig)]),
    "kh_destroy": (None, [_P]),
    "kh_reset": (C.c_int, [_P]),
    "kh_push": (C.c_int, [_P, _P, _P, _U64]),
    "kh_push_device": (C.c_int, [_P, _P, _P, _U64]),
    "kh_push_text": (C.c_int, [_P, _P, _U64, C.c_int]),
    "kh_push_text_device": (C.c_int, [_P, _P, _U64, C.c_int]),
    "kh_finish": (C.c_int, [_P, C.POINTER(KhStats)]),
    "kh_result_size": (C.c_int, [_P, _U64, C.POINTER(_U64)]),
    "kh_result_copy": (C.c_int, [_P, _P, _P, _U64, _U64, C.POINTER(_U64)]),
    "kh_result_copy_device": (C.c_int, [_P, _P, _P, _U64, _U64, C.POINTER(_U64)]),
    "kh_histogram": (C.c_int, [_P, _U64, _P, _P, _U64, C.POINTER(_U64)]),
    "kh_lookup": (C.c_int, [_P, _P, _U64, _P]),
    "kh_owner": (C.c_uint32, [_U64, C.c_uint32, C.c_uint32]),
    "kh_set_shard": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "kh_set_region_window": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "kh_region_unit_counts_device": (C.c_int, [_P, C.c_uint32, _P, _U64, _P]),
    "kh_export_regions_device": (C.c_int, [_P, C.c_uint32, _P, _P, _U64, _P, _U64, _P, C.POINTER(_U64)]),
    "kh_merge_regions_device": (C.c_int, [_P, C.c_uint32, _U64, _P, _P, _P]),
    "kh_export_regions_packed_device": (C.c_int, [_P, C.c_uint32, _P, _U64, _P, _U64, _P, C.POINTER(_U64)]),
    "kh_merge_regions_packed_device": (C.c_int, [_P, C.c_uint32, _U64, _P, _P]),
    "kh_export_regions_heads_device": (C.c_int, [_P, C.c_uint32, _P, _U64, _P, _U64, _P, C.POINTER(_U64)]),
    "kh_merge_regions_heads_device": (C.c_int, [_P, C.c_uint32, _U64, _P, _P]),
    "kh_export_dense_device": (C.c_int, [_P, _P, _U64]),
    "kh_merge_dense_device": (C.c_int, [_P, _P, _U64, C.c_uint32, C.c_uint32]),
    "kh_export_by_owner_device": (C.c_int, [_P, C.c_uint32, _P, _P, _U64, _P]),
    "kh_merge_pairs_device": (C.c_int, [_P, _P, _P, _U64]),
    "kh_merge_pairs": (C.c_int, [_P, _P, _P, _U64]),
    "kh_comm_unique_id": (C.c_int, [C.POINTER(KhUniqueId)]),
    "kh_comm_init": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(KhUniqueId)]),
    "kh_merge_across": (C.c_int, [_P, C.POINTER(KhMergeInfo)]),
    "kh_group_create": (C.c_int, [C.POINTER(_P), C.POINTER(KhConfig), C.POINTER(C.c_int32), C.c_uint32]),
    "kh_group_ctx": (_P, [_P, C.c_uint32]),
    "kh_group_size": (C.c_uint32, [_P]),
    "kh_group_merge": (C.c_int, [_P, C.POINTER(KhMergeInfo)]),
    "kh_group_destroy": (None, [_P]),
    "kh_host_alloc": (C.c_int, [C.POINTER(_P), _U64]),
    "kh_host_free": (C.c_int, [_P]),
    "kh_host_register": (C.c_int, [_P, _U64]),
    "kh_host_unregister": (C.c_int, [_P]),
    "kh_pack": (C.c_int, [C.c_char_p, C.c_uint32, C.POINTER(_U64), C.POINTER(C.c_uint32)]),
    "kh_unpack": (C.c_int, [_U64, C.c_uint32, C.c_char_p]),
    "kh_canonical": (C.c_int, [_U64, C.c_uint32, C.POINTER(_U64), C.POINTER(C.c_int)]),
    "kh_strerror": (C.c_char_p, [C.c_int]),
    "kh_last_error": (C.c_char_p, [_P]),
    "kh_synth_reads_device": (C.c_int, [C.c_int, _P, _U64, _U64, C.c_uint32, _U64, _U64, _P, _P]),
}


class KmerHipError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = lib().kh_strerror(status).decode()
        super().__init__(f"kmerhip error {status}: {msg}" + (f" ({detail})" if detail else ""))


class KmerLengthError(ValueError):
    """Mirror of kmerust::error::KmerLengthError (src/error.rs:86-95)."""

    def __init__(self, k):
        self.k, self.min, self.max = k, 1, 32
        super().__init__(f"k-mer length {k} is out of range: must be between 1 and 32")


_lib = None
ABI_VERSION = 2  # KMERHIP_ABI_VERSION of the include/kmerhip.h this file mirrors (tests/test_abi.py)


def lib():
    """Loads the HIP library; raises ImportError (never falls back) if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C krust_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        # torch bundles its own libamdhip64.so.7; importing it first makes the dynamic linker
        # resolve our DT_NEEDED entry to that already-loaded copy, so the process holds ONE HIP
        # runtime and tensor.data_ptr() addresses are valid in our kernels.  (torch is plumbing:
        # device memory, streams, torch.distributed.)  Without torch, /opt/rocm/lib is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        if L.kh_abi_version() != ABI_VERSION:  # (the structs below are laid out for this version of include/kmerhip.h)
            raise ImportError(f"{LIB_PATH} speaks ABI version {L.kh_abi_version()}, this binding {ABI_VERSION}: rebuild the library")
        _lib = L
    return _lib


def _addr(a):
    """numpy array / bytes / int address / None -> (address or None, keepalive)."""
    if a is None:
        return None, None
    if isinstance(a, int):
        return a, None
    if isinstance(a, np.ndarray):
        a = np.ascontiguousarray(a)
        return a.ctypes.data, a
    arr = np.frombuffer(a, dtype=np.uint8)
    return arr.ctypes.data, arr


def _text_format(fmt):
    return {"fasta": TEXT_FASTA, "fastq": TEXT_FASTQ, TEXT_FASTA: TEXT_FASTA, TEXT_FASTQ: TEXT_FASTQ}[fmt]


class DeviceCounter:
    """One kh_ctx: a GPU-resident canonical k-mer count table.

    Stands where the reference has `KmerMap` (src/run.rs:491-583): build()/
    build_with_quality() -> push()/push_device(); into_hashmap() -> result()."""

    def __init__(self, k, min_quality=None, capacity_hint=0, device=-1, stream=None, trace=False, path=None, defer_text_scan=False,
                 input_bytes=0):
        """stream: None = the context creates its own non-blocking stream (NOT ordered with torch's streams:
        synchronise buffers you hand over yourself); an integer = launch on that hipStream_t, where 0 is the
        legacy default stream (torch.cuda.current_stream().cuda_stream is usually 0)."""
        if not (1 <= int(k) <= 32):
            raise KmerLengthError(int(k))
        cfg = KhConfig(C.sizeof(KhConfig), int(k), -1 if min_quality is None else int(min_quality),
                       int(device), int(capacity_hint), stream or None,
                       (FLAG_TRACE if trace else 0) | (FLAG_CALLER_STREAM if stream is not None else 0)
                       | (FLAG_DEFER_TEXT_SCAN if defer_text_scan else 0)
                       | {None: 0, "auto": 0, "direct": FLAG_FORCE_DIRECT, "partition": FLAG_FORCE_PARTITION}[path],
                       min(0xFFFFFFFF, (int(input_bytes) + (1 << 20) - 1) >> 20))
        h = _P()
        rc = lib().kh_create(C.byref(h), C.byref(cfg))
        if rc != KH_OK:
            raise KmerHipError(rc)
        self._h = h
        self._owned = True
        self.k = int(k)
        self.min_quality = min_quality

    @classmethod
    def _adopt(cls, handle, k, min_quality):
        """A context that belongs to a DeviceGroup (destroyed with the group, not here)."""
        self = cls.__new__(cls)
        self._h, self._owned, self.k, self.min_quality = _P(handle), False, int(k), min_quality
        return self

    # -- lifecycle ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                lib().kh_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != KH_OK:
            raise KmerHipError(rc, lib().kh_last_error(self._h).decode())

    def reset(self):
        self._check(lib().kh_reset(self._h))

    # -- input -------------------------------------------------------------
    def push(self, bases, qual=None):
        """Flat host buffer(s); records separated by >=1 non-ACGT byte."""
        bp, kb = _addr(bases)
        n = kb.size if kb is not None else 0
        qp, kq = _addr(qual)
        if kq is not None and kq.size != n:
            raise ValueError("qual must have the same length as bases")
        self._check(lib().kh_push(self._h, bp, qp, n))

    def push_device(self, d_bases, d_qual, n):
        """Device-resident flat buffers given as integer addresses (e.g. tensor.data_ptr())."""
        self._check(lib().kh_push_device(self._h, d_bases, d_qual, int(n)))

    def push_text(self, text, fmt):
        """Raw FASTA / FASTQ text (whole records) scanned on the device; fmt: "fasta" | "fastq".
        Raises KmerHipError(status=KH_ERR_FORMAT) for layouts the device scanner does not take."""
        tp, kt = _addr(text)
        self._check(lib().kh_push_text(self._h, tp, kt.size if kt is not None else 0, _text_format(fmt)))

    def push_text_device(self, d_text, n, fmt):
        self._check(lib().kh_push_text_device(self._h, d_text, int(n), _text_format(fmt)))

    def finish(self):
        st = KhStats()
        self._check(lib().kh_finish(self._h, C.byref(st)))
        d = {f: getattr(st, f) for f, _ in KhStats._fields_ if f != "stage_ms"}
        d["stage_ms"] = {name: st.stage_ms[i] for i, name in enumerate(STAGES)}
        return d

    # -- output ------------------------------------------------------------
    def result_size(self, min_count=1):
        n = _U64(0)
        self._check(lib().kh_result_size(self._h, int(min_count), C.byref(n)))
        return int(n.value)

    def result(self, min_count=1, sort=True, out=None):
        """(keys, counts) as uint64 arrays; packed canonical keys.  out = (keys, counts) arrays to fill (e.g. views of
        PinnedArray memory: the copy is then one DMA each)."""
        n = self.result_size(min_count)
        if out is not None:
            keys, cnts = out[0][:n], out[1][:n]
            assert keys.size == n and cnts.size == n, "out arrays too small"
        else:
            keys = np.empty(n, dtype=np.uint64)
            cnts = np.empty(n, dtype=np.uint64)
        got = _U64(0)
        self._check(lib().kh_result_copy(self._h, keys.ctypes.data, cnts.ctypes.data, n, int(min_count), C.byref(got)))
        assert got.value == n
        if sort:
            o = np.argsort(keys, kind="stable")
            keys, cnts = keys[o], cnts[o]
        return keys, cnts

    def result_device(self, d_keys, d_counts, cap, min_count=1):
        got = _U64(0)
        self._check(lib().kh_result_copy_device(self._h, d_keys, d_counts, int(cap), int(min_count), C.byref(got)))
        return int(got.value)

    def as_dict(self, min_count=1):
        k, c = self.result(min_count)
        return dict(zip(k.tolist(), c.tolist()))

    def as_str_dict(self, min_count=1):
        """HashMap<String,u64> of into_hashmap (src/run.rs:573-582)."""
        return {unpack(key, self.k): c for key, c in self.as_dict(min_count).items()}

    def histogram(self, min_count=1):
        cap = 1 << 12
        while True:
            cnt = np.empty(cap, dtype=np.uint64)
            frq = np.empty(cap, dtype=np.uint64)
            n = _U64(0)
            rc = lib().kh_histogram(self._h, int(min_count), cnt.ctypes.data, frq.ctypes.data, cap, C.byref(n))
            if rc == KH_ERR_RANGE:
                cap *= 16
                continue
            self._check(rc)
            return list(zip(cnt[: n.value].tolist(), frq[: n.value].tolist()))

    def lookup(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(keys.size, dtype=np.uint64)
        self._check(lib().kh_lookup(self._h, keys.ctypes.data, keys.size, out.ctypes.data))
        return out

    # -- multi-GPU merge ---------------------------------------------------
    def comm_init(self, nranks, rank, unique_id):
        """Joins the RCCL communicator named by `unique_id` (128 bytes from comm_unique_id() of rank 0) as
        `rank` of `nranks`: collective, blocks until every rank has called it."""
        uid = KhUniqueId()
        C.memmove(C.byref(uid), bytes(unique_id), 128)
        self._check(lib().kh_comm_init(self._h, int(nranks), int(rank), C.byref(uid)))

    def merge_across(self):
        """kh_merge_across: the whole exchange inside the library (collective over the communicator's ranks).
        Afterwards this context holds the keys with owner(key, k, nranks) == rank.  Returns the merge info."""
        info = KhMergeInfo()
        self._check(lib().kh_merge_across(self._h, C.byref(info)))
        return merge_info_dict(info)

    def export_by_owner_device(self, nparts, d_keys, d_counts, cap):
        parts = np.zeros(nparts, dtype=np.uint64)
        self._check(lib().kh_export_by_owner_device(self._h, int(nparts), d_keys, d_counts, int(cap), parts.ctypes.data))
        return parts

    def set_shard(self, index, count):
        """Make the (empty) table shard `index` of `count` (power of two) by hash range."""
        self._check(lib().kh_set_shard(self._h, int(index), int(count)))

    def set_region_window(self, piece=0, npieces=1):
        """The next region-ordered exports / merges cover piece `piece` of `npieces` of every owner's
        region range (kh_set_region_window); (0, 1) = everything."""
        self._check(lib().kh_set_region_window(self._h, int(piece), int(npieces)))

    def region_unit_counts_device(self, unit_bytes, d_region_counts, region_cap):
        """Exchange units (4: heads, 8: packed pairs, 16: pairs) per region of the whole table; returns
        the number of table regions, or None where the matching export would not be representable."""
        nreg = _U64(0)
        rc = lib().kh_region_unit_counts_device(self._h, int(unit_bytes), C.c_void_p(int(d_region_counts)), int(region_cap), C.byref(nreg))
        if rc == KH_ERR_RANGE:
            return None
        self._check(rc)
        return int(nreg.value)

    def export_regions_device(self, nparts, d_keys, d_counts, cap, d_region_counts, region_cap):
        """Live pairs in region order (grouped by owner) + per-region live counts.
        Returns (pairs per owner as uint64 array, number of table regions)."""
        parts = np.zeros(nparts, dtype=np.uint64)
        nreg = _U64(0)
        self._check(lib().kh_export_regions_device(self._h, int(nparts), d_keys, d_counts, int(cap), d_region_counts,
                                                   int(region_cap), parts.ctypes.data, C.byref(nreg)))
        return parts, int(nreg.value)

    def merge_regions_device(self, sender_regions, d_keys, d_counts, d_region_counts):
        """d_keys / d_counts / d_region_counts: one device address per sender."""
        n = len(d_keys)
        assert len(d_counts) == n and len(d_region_counts) == n
        arr = lambda xs: (_P * n)(*[_P(int(x)) for x in xs])
        ak, ac, ar = arr(d_keys), arr(d_counts), arr(d_region_counts)
        self._check(lib().kh_merge_regions_device(self._h, n, int(sender_regions), ak, ac, ar))

    def export_regions_packed_device(self, nparts, d_pairs, cap, d_region_counts, region_cap):
        """Packed export (one u64 per pair).  Returns (parts, regions), or None when the table is not
        representable in the packed form (large k for the table size, or a count >= 2^32)."""
        parts = np.zeros(nparts, dtype=np.uint64)
        nreg = _U64(0)
        rc = lib().kh_export_regions_packed_device(self._h, int(nparts), d_pairs, int(cap), d_region_counts,
                                                   int(region_cap), parts.ctypes.data, C.byref(nreg))
        if rc == KH_ERR_RANGE:
            return None
        self._check(rc)
        return parts, int(nreg.value)

    def export_regions_heads_device(self, nparts, d_heads, cap, d_region_counts, region_cap):
        """32-bit heads export.  Returns (heads per owner, regions), or None when not representable
        (k too large for the table size, a very large count, or more than `cap` heads)."""
        parts = np.zeros(nparts, dtype=np.uint64)
        nreg = _U64(0)
        rc = lib().kh_export_regions_heads_device(self._h, int(nparts), d_heads, int(cap), d_region_counts,
                                                  int(region_cap), parts.ctypes.data, C.byref(nreg))
        if rc == KH_ERR_RANGE:
            return None
        self._check(rc)
        return parts, int(nreg.value)

    def merge_regions_heads_device(self, sender_regions, d_heads, d_region_counts):
        n = len(d_heads)
        assert len(d_region_counts) == n
        arr = lambda xs: (_P * n)(*[_P(int(x)) for x in xs])
        ah, ar = arr(d_heads), arr(d_region_counts)
        self._check(lib().kh_merge_regions_heads_device(self._h, n, int(sender_regions), ah, ar))

    def merge_regions_packed_device(self, sender_regions, d_pairs, d_region_counts):
        n = len(d_pairs)
        assert len(d_region_counts) == n
        arr = lambda xs: (_P * n)(*[_P(int(x)) for x in xs])
        ap, ar = arr(d_pairs), arr(d_region_counts)
        self._check(lib().kh_merge_regions_packed_device(self._h, n, int(sender_regions), ap, ar))

    def export_dense_device(self, d_dense, n_entries):
        """Small k (2k <= 26): d_dense[key] = count over the whole 4^k key space."""
        self._check(lib().kh_export_dense_device(self._h, d_dense, int(n_entries)))

    def merge_dense_device(self, d_dense, n_entries, owner, nparts):
        self._check(lib().kh_merge_dense_device(self._h, d_dense, int(n_entries), int(owner), int(nparts)))

    def merge_pairs_device(self, d_keys, d_counts, n):
        self._check(lib().kh_merge_pairs_device(self._h, d_keys, d_counts, int(n)))

    def merge_pairs(self, keys, counts):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        assert keys.size == counts.size
        self._check(lib().kh_merge_pairs(self._h, keys.ctypes.data, counts.ctypes.data, keys.size))


def merge_info_dict(info):
    d = {f: getattr(info, f) for f, _ in KhMergeInfo._fields_}
    d["path"] = ROUTES[info.route] + (f"-x{info.pieces}" if info.pieces > 1 else "")
    return d


def comm_unique_id():
    """128 opaque bytes (an ncclUniqueId) that name a new communicator; rank 0 makes it, every rank gets a copy."""
    uid = KhUniqueId()
    rc = lib().kh_comm_unique_id(C.byref(uid))
    if rc != KH_OK:
        raise KmerHipError(rc)
    return bytes(C.string_at(C.byref(uid), 128))


class DeviceGroup:
    """kh_group: one context per listed device inside ONE process, one host thread per context for the merge.
    Distinct devices exchange over RCCL; a device listed twice (1-GPU test boxes) makes the group exchange by
    device-to-device copies inside the process."""

    def __init__(self, k, devices, min_quality=None, capacity_hint=0, trace=False, path=None):
        if not (1 <= int(k) <= 32):
            raise KmerLengthError(int(k))
        cfg = KhConfig(C.sizeof(KhConfig), int(k), -1 if min_quality is None else int(min_quality), -1,
                       int(capacity_hint), None,
                       (FLAG_TRACE if trace else 0) | {None: 0, "auto": 0, "direct": FLAG_FORCE_DIRECT,
                                                       "partition": FLAG_FORCE_PARTITION}[path], 0)
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        g = _P()
        rc = lib().kh_group_create(C.byref(g), C.byref(cfg), devs, len(devices))
        if rc != KH_OK:
            raise KmerHipError(rc)
        self._g = g
        self.counters = [DeviceCounter._adopt(lib().kh_group_ctx(g, i), k, min_quality) for i in range(len(devices))]

    def __len__(self):
        return len(self.counters)

    def __getitem__(self, i):
        return self.counters[i]

    def merge(self):
        infos = (KhMergeInfo * len(self.counters))()
        rc = lib().kh_group_merge(self._g, infos)
        if rc != KH_OK:
            detail = "; ".join(lib().kh_last_error(c._h).decode() for c in self.counters)
            raise KmerHipError(rc, detail)
        return [merge_info_dict(i) for i in infos]

    def close(self):
        if getattr(self, "_g", None):
            for c in self.counters:
                c._h = None
            lib().kh_group_destroy(self._g)
            self._g = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---- host memory the device reaches directly --------------------------------
class PinnedArray:
    """A numpy view of kh_host_alloc'ed (pinned) memory: push() / push_text() from it and result(out=...) into it move
    by DMA, without the staging copies pageable memory needs.  Free with close() (or the context manager)."""

    def __init__(self, n, dtype=np.uint8):
        self.dtype = np.dtype(dtype)
        self.nbytes = int(n) * self.dtype.itemsize
        p = _P()
        rc = lib().kh_host_alloc(C.byref(p), max(self.nbytes, 1))
        if rc != KH_OK:
            raise KmerHipError(rc)
        self._p = p
        buf = (C.c_uint8 * max(self.nbytes, 1)).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(n))

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            lib().kh_host_free(self._p)
            self._p = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def host_register(arr):
    """hipHostRegister the memory of a contiguous numpy array (kh_host_register); pair with host_unregister."""
    rc = lib().kh_host_register(arr.ctypes.data, arr.nbytes)
    if rc != KH_OK:
        raise KmerHipError(rc)


def host_unregister(arr):
    rc = lib().kh_host_unregister(arr.ctypes.data)
    if rc != KH_OK:
        raise KmerHipError(rc)


# ---- pure helpers (host) ----------------------------------------------------

def pack(seq):
    if isinstance(seq, str):
        seq = seq.encode()
    out = _U64(0)
    pos = C.c_uint32(0)
    rc = lib().kh_pack(seq, len(seq), C.byref(out), C.byref(pos))
    if rc == KH_ERR_BAD_K:
        raise KmerLengthError(len(seq))
    if rc != KH_OK:
        b = seq[pos.value]  # Display of InvalidBaseError, src/error.rs:106-122
        if 0x20 <= b < 0x7F:
            raise ValueError(f"invalid base '{chr(b)}' (0x{b:02x}) at position {pos.value}")
        raise ValueError(f"invalid base 0x{b:02x} at position {pos.value}")
    return int(out.value)


def unpack(bits, k):
    if not (1 <= k <= 32):
        raise KmerLengthError(k)
    buf = C.create_string_buffer(k)
    lib().kh_unpack(int(bits), k, buf)
    return buf.raw.decode()


def canonical(bits, k):
    out = _U64(0)
    rc_flag = C.c_int(0)
    rc = lib().kh_canonical(int(bits), k, C.byref(out), C.byref(rc_flag))
    if rc == KH_ERR_BAD_K:
        raise KmerLengthError(k)
    return int(out.value), bool(rc_flag.value)


def owner(key, k, nparts):
    """Owner shard of a packed canonical k-mer (fast-range of the top bits of the table hash)."""
    return int(lib().kh_owner(int(key), int(k), int(nparts)))


def synth_reads_device(d_bases, d_qual, seed, genome_len, read_len, first_read, n_reads, device=-1, stream=None):
    rc = lib().kh_synth_reads_device(int(device), stream, int(seed), int(genome_len), int(read_len),
                                     int(first_read), int(n_reads), d_bases, d_qual)
    if rc != KH_OK:
        raise KmerHipError(rc)
