"""ctypes binding of oracle/libkmer_oracle.so -- the CHECKER used by tests,
smoke() and bench.py's cpu_baseline leg.  Never imported by krust_amd/."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libkmer_oracle.so")

_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)


def _load():
    lib = C.CDLL(_SO)
    lib.ko_kmer_length_ok.argtypes = [C.c_uint64]
    lib.ko_kmer_length_ok.restype = C.c_int
    lib.ko_from_sub.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, _u8p, C.POINTER(C.c_size_t)]
    lib.ko_from_sub.restype = C.c_int
    lib.ko_pack_bytes.argtypes = [C.c_char_p, C.c_size_t]
    lib.ko_pack_bytes.restype = C.c_uint64
    lib.ko_canonical.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    lib.ko_canonical.restype = C.c_uint64
    lib.ko_unpack.argtypes = [C.c_uint64, C.c_size_t, C.c_char_p]
    lib.ko_unpack.restype = None
    lib.ko_map_new.restype = C.c_void_p
    lib.ko_map_free.argtypes = [C.c_void_p]
    lib.ko_map_len.argtypes = [C.c_void_p]
    lib.ko_map_len.restype = C.c_uint64
    lib.ko_map_add.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    lib.ko_map_get.argtypes = [C.c_void_p, C.c_uint64]
    lib.ko_map_get.restype = C.c_uint64
    lib.ko_map_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    lib.ko_map_dump.restype = C.c_uint64
    lib.ko_map_total.argtypes = [C.c_void_p]
    lib.ko_map_total.restype = C.c_uint64
    for name in ("ko_process_sequence", "ko_process_sequence_rolling"):
        f = getattr(lib, name)
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        f.restype = None
    lib.ko_count_valid_windows.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    lib.ko_count_valid_windows.restype = C.c_uint64
    lib.ko_histogram.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    lib.ko_histogram.restype = C.c_uint64
    lib.ko_crc32.argtypes = [C.c_char_p, C.c_size_t]
    lib.ko_crc32.restype = C.c_uint32
    lib.ko_count_records_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_uint64, C.c_size_t, C.c_int, C.c_int]
    lib.ko_count_records_mt.restype = C.c_uint64
    lib.ko_count_records_mt2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_uint64, C.c_size_t, C.c_int, C.c_int, C.c_uint64, C.c_void_p]
    lib.ko_count_records_mt2.restype = C.c_uint64
    lib.ko_scan_flat_sampled_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int,
                                            C.c_uint64, C.c_int]
    lib.ko_scan_flat_sampled_mt.restype = C.c_uint64
    lib.ko_count_flat_radix_mt.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.ko_count_flat_radix_mt.restype = C.c_uint64
    lib.ko_hist_flat_radix_mt.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                          C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.ko_hist_flat_radix_mt.restype = C.c_uint64
    lib.ko_synth_hg.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
    lib.ko_synth_hg.restype = None
    lib.ko_write_fasta.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
    lib.ko_write_fasta.restype = C.c_int
    lib.ko_map_digest.argtypes = [C.c_void_p]
    lib.ko_map_digest.restype = C.c_uint64
    lib.ko_mix64.argtypes = [C.c_uint64]
    lib.ko_mix64.restype = C.c_uint64
    lib.ko_synth_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64,
                                   C.c_void_p, C.c_void_p]
    lib.ko_synth_reads.restype = None
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _buf(b):
    """bytes / bytearray / np.uint8 array -> (address, length, keepalive)."""
    if b is None:
        return None, 0, None
    a = np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, dtype=np.uint8)
    return a.ctypes.data, a.size, a


class OracleMap:
    """u64 -> u64 count map (stands in for krust's DashMap, src/run.rs:489)."""

    def __init__(self):
        self._m = lib().ko_map_new()

    def __del__(self):
        if getattr(self, "_m", None):
            lib().ko_map_free(self._m)
            self._m = None

    def __len__(self):
        return int(lib().ko_map_len(self._m))

    def add(self, key, addend=1):
        lib().ko_map_add(self._m, key, addend)

    def get(self, key):
        return int(lib().ko_map_get(self._m, key))

    def total(self):
        return int(lib().ko_map_total(self._m))

    def process(self, seq, k, qual=None, min_quality=None, rolling=False):
        sp, sn, _ks = _buf(seq)
        qp, qn, _kq = _buf(qual)
        if qual is not None:
            assert qn >= sn
        f = lib().ko_process_sequence_rolling if rolling else lib().ko_process_sequence
        f(self._m, sp, sn, qp, k, -1 if min_quality is None else int(min_quality))

    def arrays(self):
        n = len(self)
        keys = np.empty(n, dtype=np.uint64)
        cnts = np.empty(n, dtype=np.uint64)
        got = lib().ko_map_dump(self._m, keys.ctypes.data, cnts.ctypes.data, n)
        assert got == n
        order = np.argsort(keys, kind="stable")
        return keys[order], cnts[order]

    def as_dict(self):
        k, c = self.arrays()
        return dict(zip(k.tolist(), c.tolist()))

    def as_str_dict(self, k):
        return {unpack(key, k): c for key, c in self.as_dict().items()}

    def histogram(self, min_count=1):
        n = max(len(self), 1)
        cnt = np.empty(n, dtype=np.uint64)
        frq = np.empty(n, dtype=np.uint64)
        nd = lib().ko_histogram(self._m, min_count, cnt.ctypes.data, frq.ctypes.data, n)
        return list(zip(cnt[:nd].tolist(), frq[:nd].tolist()))

    def scan_flat(self, seq, k, qual=None, min_quality=None, sample_mask=0, nthreads=1):
        """Threaded rolling scan of a flat buffer; returns total valid windows."""
        sp, sn, _ks = _buf(seq)
        qp, _, _kq = _buf(qual)
        return int(lib().ko_scan_flat_sampled_mt(self._m, sp, sn, qp, k, -1 if min_quality is None else int(min_quality),
                                                 sample_mask, nthreads))

    def digest(self):
        return int(lib().ko_map_digest(self._m))

    def count_records_mt(self, seq, offs, lens, k, qual=None, min_quality=None, nthreads=1, expect_distinct=0, stats=None):
        """krust-equivalent threaded count.  stats: a dict that receives the lock statistics."""
        sp, _, _ks = _buf(seq)
        qp, _, _kq = _buf(qual)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        st = np.zeros(4, dtype=np.uint64)
        n = int(lib().ko_count_records_mt2(self._m, sp, qp, offs.ctypes.data, lens.ctypes.data,
                                           len(offs), k, -1 if min_quality is None else int(min_quality),
                                           nthreads, int(expect_distinct), st.ctypes.data))
        if stats is not None:
            stats.update(contended=int(st[0]), wait_ns=int(st[1]), upserts=int(st[2]), rehashes_under_lock=int(st[3]))
        return n


def count_records(records, k, quals=None, min_quality=None, rolling=False):
    """records: list of bytes -> OracleMap (one process_sequence per record)."""
    m = OracleMap()
    for i, r in enumerate(records):
        q = None if quals is None else quals[i]
        m.process(r, k, qual=q, min_quality=min_quality, rolling=rolling)
    return m


def count_flat_radix(seq, k, qual=None, min_quality=None, nthreads=1):
    """Optimised CPU formulation (rolling scan + two-phase radix count): (kmers, distinct, digest)."""
    sp, sn, _ks = _buf(seq)
    qp, _, _kq = _buf(qual)
    d = C.c_uint64(0)
    g = C.c_uint64(0)
    tot = lib().ko_count_flat_radix_mt(sp, sn, qp, k, -1 if min_quality is None else int(min_quality), nthreads,
                                       C.byref(d), C.byref(g))
    return int(tot), int(d.value), int(g.value)


def hist_flat_radix(seq, k, qual=None, min_quality=None, nthreads=1, npasses=1, min_count=1):
    """Full count of a flat buffer in `npasses` bounded-memory passes: (kmers, distinct, digest, histogram) with
    the histogram as ascending (count, frequency) pairs after the min_count filter (src/run.rs:447-450,471-481)."""
    sp, sn, _ks = _buf(seq)
    qp, _, _kq = _buf(qual)
    d, g, n = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    cap = 1 << 17
    while True:
        cnt = np.empty(cap, dtype=np.uint64)
        frq = np.empty(cap, dtype=np.uint64)
        tot = lib().ko_hist_flat_radix_mt(sp, sn, qp, k, -1 if min_quality is None else int(min_quality), nthreads,
                                          npasses, int(min_count), cnt.ctypes.data, frq.ctypes.data, cap,
                                          C.byref(n), C.byref(d), C.byref(g))
        if n.value <= cap:
            break
        cap = int(n.value)
    return int(tot), int(d.value), int(g.value), list(zip(cnt[: n.value].tolist(), frq[: n.value].tolist()))


# hg38 (GRCh38.p14 primary assembly) chromosome lengths: chr1..22, X, Y, M -- 3.09 Gbp, chr1 = 248,956,422 bp
HG38_LENGTHS = (248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
                133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
                58617616, 64444167, 46709983, 50818468, 156040895, 57227415, 16569)


def synth_hg(seed, lens, nthreads=1):
    """'hg-like' synthetic assembly: flat uint8 buffer, record r = lens[r] bases followed by '\\n'."""
    lens = np.ascontiguousarray(lens, dtype=np.uint64)
    out = np.empty(int(lens.sum()) + lens.size, dtype=np.uint8)
    lib().ko_synth_hg(int(seed), lens.ctypes.data, lens.size, out.ctypes.data, int(nthreads))
    return out


def write_fasta(path, flat, lens, width=60):
    lens = np.ascontiguousarray(lens, dtype=np.uint64)
    flat = np.ascontiguousarray(flat, dtype=np.uint8)
    assert flat.size == int(lens.sum()) + lens.size
    if lib().ko_write_fasta(os.fsencode(path), flat.ctypes.data, lens.ctypes.data, lens.size, int(width)) != 0:
        raise OSError(f"writing {path} failed")


def pack(seq):
    return int(lib().ko_pack_bytes(seq, len(seq)))


def unpack(bits, k):
    out = C.create_string_buffer(k + 1)
    lib().ko_unpack(bits, k, out)
    return out.value.decode()


def canonical(seq):
    rc = C.c_int(0)
    bits = lib().ko_canonical(seq, len(seq), C.byref(rc))
    return int(bits), bool(rc.value)


def from_sub(seq):
    norm = C.create_string_buffer(len(seq) + 1)
    eb = C.c_uint8(0)
    ep = C.c_size_t(0)
    r = lib().ko_from_sub(seq, len(seq), norm, C.byref(eb), C.byref(ep))
    if r == 0:
        return norm.raw[: len(seq)], None
    return None, (chr(eb.value), int(ep.value))


def valid_windows(seq, k, qual=None, min_quality=None):
    sp, sn, _ks = _buf(seq)
    qp, _, _kq = _buf(qual)
    return int(lib().ko_count_valid_windows(sp, sn, qp, k, -1 if min_quality is None else int(min_quality)))


def crc32(data):
    return int(lib().ko_crc32(data, len(data)))


def mix64(z):
    return int(lib().ko_mix64(z & 0xFFFFFFFFFFFFFFFF))


def synth_reads(seed, genome_len, read_len, first_read, n_reads, with_qual=True):
    stride = read_len + 1
    bases = np.empty(n_reads * stride, dtype=np.uint8)
    qual = np.empty(n_reads * stride, dtype=np.uint8) if with_qual else None
    lib().ko_synth_reads(seed, genome_len, read_len, first_read, n_reads, bases.ctypes.data,
                         qual.ctypes.data if with_qual else None)
    return bases, qual


# ---- which shard owns a key, for whole arrays (test infrastructure) -------------------------------------------------
# kh_owner (include/kmerhip.h) answers one key per call; the exchange tests ask for a million: a Python loop around the C ABI
# was most of their run time (round 5).  This is the same arithmetic in numpy -- the table hash of krust_amd/csrc/kmer_bits.h
# (kh_hash_n) and the fast-range of its top bits -- and every call holds a sample of its answers to the C ABI's.
_FC = (0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D, 0x27D4EB2F)


def table_hash_np(keys, k):
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    mask = np.uint64((1 << k) - 1)
    m32 = np.uint64(0xFFFFFFFF)
    L = (keys >> np.uint64(k)) & mask
    R = keys & mask
    for i, c in enumerate(_FC):
        if i % 2:  # rounds 2 and 4 (kh_feistel_g): bits k .. 2k-1 of the full product (R, c < 2^32: it fits 64 bits)
            t = ((R * np.uint64(c)) >> np.uint64(k)) & mask
        else:
            cc = np.uint64((c & 0xFFFFFF) | 1) if 16 <= k <= 24 else np.uint64(c)
            t = (R * cc) & m32
            if k < 32:
                t = t >> np.uint64(32 - k)
        L, R = R, (L ^ t) & mask
    return (L << np.uint64(k)) | R


def owners(K, keys, k, nparts):
    """kh_owner(key, k, nparts) for every key of an array; K = the krust_amd module (its C ABI checks a sample)."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    H = table_hash_np(keys, k) << np.uint64(64 - 2 * k)     # left-aligned, as kh_table_hash
    own = (((H >> np.uint64(32)) * np.uint64(nparts)) >> np.uint64(32)).astype(np.int64)
    if keys.size:
        probe = np.unique(np.linspace(0, keys.size - 1, num=min(keys.size, 97)).astype(np.int64))
        assert [int(own[i]) for i in probe] == [K.owner(int(keys[i]), k, nparts) for i in probe], "numpy owner() differs from kh_owner"
    return own
