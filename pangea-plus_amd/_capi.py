"""ctypes binding of include/pangea_hip.h and the Python mirror of the reference's CLI verbs."""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(_HERE, "lib", "libpangea_hip.so")


class PangeaError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("pangea_hip status %d: %s" % (status, msg))
        self.status = status


_lib = None


def lib():
    """The HIP library. Fails loudly when it has not been built: there is no other backend."""
    global _lib
    if _lib is None:
        if not os.path.exists(lib_path):
            raise PangeaError(-3, "libpangea_hip.so is not built (run `python pangea-plus_amd/build.py`); "
                                  "this package has no CPU fallback")
        # PyTorch-ROCm wheels bundle their own HIP runtime; whichever runtime is loaded first owns the GPU for
        # the process, so when torch is installed let it load first and bind this library to the same one
        # (bench.py and the broadcast path use torch tensors that alias this library's HBM).
        if "torch" not in sys.modules and not os.environ.get("PGX_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        _lib = C.CDLL(lib_path)
        _declare(_lib)
    return _lib


class SynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_seq", C.c_int64), ("seq_len", C.c_int32), ("n_genus", C.c_int64),
                ("read_seed", C.c_uint64), ("read_len", C.c_int32)]

    @classmethod
    def default(cls, **kw):
        c = cls()
        lib().pgx_synth_default(C.byref(c))
        for k, v in kw.items():
            setattr(c, k, v)
        return c


class StageTimes(C.Structure):
    _fields_ = [("seed_extend_ms", C.c_float), ("group_ms", C.c_float), ("sort_ms", C.c_float),
                ("consensus_ms", C.c_float), ("total_ms", C.c_float), ("probes", C.c_int64), ("postings", C.c_int64),
                ("candidates", C.c_int64), ("hits", C.c_int64), ("survivors", C.c_int64), ("gapped_ms", C.c_float),
                ("gapped_wide", C.c_int64), ("dust_ms", C.c_float), ("attempts", C.c_int32)]


class _DevArray(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ptr", C.c_void_p), ("bytes", C.c_size_t)]


class _DbShape(C.Structure):
    _fields_ = [("n_seq", C.c_int64), ("n_bases", C.c_int64), ("has_amb", C.c_int32), ("index_bits", C.c_int32),
                ("n_postings", C.c_int64), ("synthetic_ids", C.c_int32)]


class _BlastnOpts(C.Structure):
    _fields_ = [("query_path", C.c_char_p), ("db_prefix", C.c_char_p), ("out_path", C.c_char_p), ("outfmt", C.c_int),
                ("rank", C.c_int), ("world_size", C.c_int), ("ungapped", C.c_int), ("no_dust", C.c_int)]


class _SoapOpts(C.Structure):
    _fields_ = [("reads_path", C.c_char_p), ("db_prefix", C.c_char_p), ("out_path", C.c_char_p),
                ("unmapped_path", C.c_char_p), ("match_mode", C.c_int), ("repeat_mode", C.c_int), ("max_n", C.c_int),
                ("report_id", C.c_int), ("reads_b_path", C.c_char_p), ("unpaired_path", C.c_char_p), ("min_insert", C.c_int),
                ("max_insert", C.c_int)]


HIT_DTYPE = np.dtype([("read", "<i4"), ("subject", "<i4"), ("qstart", "<i4"), ("qend", "<i4"), ("sstart", "<i4"),
                      ("send", "<i4"), ("score", "<i4"), ("mismatch", "<u2"), ("gapopen", "<u2")])
REC_DTYPE = np.dtype([("hit", "<i4"), ("matches", "<i4")])
VOTE_DTYPE = np.dtype([("depth", "<i4"), ("name", "<u4", (7,)), ("votes", "u1", (7,)), ("pad", "u1")])

# every symbol include/pangea_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "pgx_last_error", "pgx_version", "pgx_init", "pgx_device_count", "pgx_current_device", "pgx_db_build", "pgx_db_open",
    "pgx_db_from_fasta", "pgx_db_close", "pgx_db_num_seqs", "pgx_db_num_bases", "pgx_db_seq_id",
    "pgx_db_device_arrays", "pgx_db_get_shape", "pgx_db_alloc_like", "pgx_db_finish_import", "pgx_db_checksum", "pgx_blastn_run", "pgx_db_set_ungapped", "pgx_db_set_dust", "pgx_db_set_dust_each_search",
    "pgx_soap_index", "pgx_soap_run", "pgx_tax_create", "pgx_tax_open", "pgx_tax_close", "pgx_tax_gi2taxid",
    "pgx_tax_node", "pgx_tax_names", "pgx_tax_format_node", "pgx_tax_format_name", "pgx_tax_cli", "pgx_free",
    "pgx_tax_lineage_batch", "pgx_taxcollect_file", "pgx_consensus_file", "pgx_synth_default", "pgx_db_from_synth",
    "pgx_synth_write_taxdump", "pgx_reads_from_fasta", "pgx_reads_from_fasta_text", "pgx_reads_from_synth", "pgx_reads_write_fasta", "pgx_reads_redo_dust", "pgx_rdp_write_file", "pgx_reads_close", "pgx_reads_count",
    "pgx_reads_get", "pgx_blast_search", "pgx_hits_close", "pgx_hits_count", "pgx_hits_copy",
    "pgx_hits_read_offsets", "pgx_hits_read_counts", "pgx_hits_slice", "pgx_hits_format", "pgx_db_bind_taxonomy", "pgx_db_subject_lineage",
    "pgx_rdp_from_file", "pgx_rdp_from_synth", "pgx_rdp_close", "pgx_consensus_batch", "pgx_classify_consensus", "pgx_classify_consensus_tri", "pgx_vote3_batch", "pgx_vote3_format",
    "pgx_consensus_format", "pgx_consensus_format_file", "pgx_last_stage_times", "pgx_megaclust_file", "pgx_megaclust_batch", "pgx_megaclustable", "pgx_trim_file", "pgx_blast_score_columns", "pgx_blast_score_columns_v", "pgx_probe_gather", "pgx_probe_issue", "pgx_probe_issue_name",
]


def _declare(L):
    def sig(name, restype=C.c_int, argtypes=None):
        f = getattr(L, name, None)
        if f is None:  # an older build of the library: the call site will fail loudly when used
            return
        f.restype = restype
        if argtypes is not None:
            f.argtypes = argtypes
    V, I64, I32, S = C.c_void_p, C.c_int64, C.c_int32, C.c_char_p
    sig("pgx_last_error", S)
    sig("pgx_version", S)
    sig("pgx_db_num_seqs", I64, [V])
    sig("pgx_db_num_bases", I64, [V])
    sig("pgx_db_seq_id", S, [V, I64])
    sig("pgx_db_subject_lineage", S, [V, I64])
    sig("pgx_db_set_ungapped", C.c_int, [V, C.c_int])
    sig("pgx_db_set_dust_each_search", C.c_int, [V, C.c_int])
    sig("pgx_reads_count", I64, [V])
    sig("pgx_hits_count", I64, [V])
    sig("pgx_reads_from_synth", C.c_int, [V, I64, I64, V])
    sig("pgx_reads_from_fasta", C.c_int, [S, I64, I64, V])
    sig("pgx_reads_from_fasta_text", C.c_int, [C.c_char_p, C.c_size_t, I64, I64, V])
    sig("pgx_reads_get", C.c_int, [V, I64, V, I32, V])
    sig("pgx_reads_write_fasta", C.c_int, [V, S])
    sig("pgx_reads_redo_dust", C.c_int, [V])
    sig("pgx_rdp_write_file", C.c_int, [V, V, V, S])
    sig("pgx_hits_copy", C.c_int, [V, V, I64])
    sig("pgx_hits_read_offsets", C.c_int, [V, V, I64])
    sig("pgx_hits_read_counts", C.c_int, [V, V, I64])
    sig("pgx_rdp_from_synth", C.c_int, [V, I64, I64, V, V])
    sig("pgx_consensus_batch", C.c_int, [V, V, V, V, I64])
    sig("pgx_classify_consensus", C.c_int, [V, V, V, V, V, I64])
    sig("pgx_classify_consensus_tri", C.c_int, [V, V, V, C.c_char_p, V, V, I64])
    sig("pgx_vote3_batch", C.c_int, [V, V, V, V, C.c_char_p, V, I64])
    sig("pgx_vote3_format", C.c_int, [V, V, V, I64, V, V])
    sig("pgx_consensus_format", C.c_int, [V, V, V, V, I64, V, V])
    sig("pgx_consensus_format_file", C.c_int, [V, V, V, V, I64, S, V])
    sig("pgx_tax_lineage_batch", C.c_int, [V, V, I64, V, V, V])
    sig("pgx_megaclust_file", C.c_int, [V, V])
    sig("pgx_megaclust_batch", C.c_int, [V, V, V, V, I64, V, V, V, V])
    sig("pgx_megaclustable", C.c_int, [C.c_int, V, V])
    sig("pgx_trim_file", C.c_int, [V, V, V, V, V])
    sig("pgx_probe_gather", C.c_int, [C.c_uint64, C.c_int, V, V])
    sig("pgx_probe_issue", C.c_int, [C.c_int, C.c_int, V])
    sig("pgx_blast_score_columns", C.c_int, [C.c_int32, I64, I64, I64, C.c_char_p, C.c_char_p])
    sig("pgx_blast_score_columns_v", C.c_int, [C.c_int32, I64, I64, I64, C.c_int, C.c_char_p, C.c_char_p])
    sig("pgx_free", None, [V])
    for name in ("pgx_db_close", "pgx_reads_close", "pgx_hits_close", "pgx_rdp_close", "pgx_tax_close"):
        sig(name, None, [V])


def _check(rc):
    if rc < 0:
        raise PangeaError(rc, lib().pgx_last_error().decode("utf-8", "replace"))
    return rc


def _b(s):
    return None if s is None else os.fsencode(s)


def _take_text(ptr, length=None):
    """malloc'd C string -> bytes, then pgx_free."""
    if not ptr:
        return b""
    if length is not None and length >= (1 << 31):  # ctypes.string_at takes a C int
        data = bytes((C.c_char * length).from_address(ptr))
    else:
        data = C.string_at(ptr, length) if length is not None else C.string_at(ptr)
    lib().pgx_free(ptr)
    return data


def init(device=0):
    _check(lib().pgx_init(int(device)))


def device_count():
    return lib().pgx_device_count()


def version():
    return lib().pgx_version().decode()


class _Handle:
    _close = None

    def __init__(self, ptr):
        self.ptr = ptr

    def close(self):
        if self.ptr:
            getattr(lib(), self._close)(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class _CudaArray:
    """Zero-copy view of library-owned HBM for torch.as_tensor (used for the RCCL broadcast)."""

    def __init__(self, ptr, nbytes, owner):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        self.owner = owner


class Db(_Handle):
    """Sequence database in HBM (makeblastdb / 2bwt-builder + the device seed index)."""
    _close = "pgx_db_close"

    @classmethod
    def open(cls, prefix):
        p = C.c_void_p()
        _check(lib().pgx_db_open(_b(prefix), C.byref(p)))
        return cls(p)

    @classmethod
    def from_fasta(cls, path):
        p = C.c_void_p()
        _check(lib().pgx_db_from_fasta(_b(path), C.byref(p)))
        return cls(p)

    @classmethod
    def from_synth(cls, cfg):
        p = C.c_void_p()
        _check(lib().pgx_db_from_synth(C.byref(cfg), C.byref(p)))
        return cls(p)

    @classmethod
    def alloc_like(cls, shape):
        s = _DbShape(*shape)
        p = C.c_void_p()
        _check(lib().pgx_db_alloc_like(C.byref(s), C.byref(p)))
        return cls(p)

    def finish_import(self):
        _check(lib().pgx_db_finish_import(self.ptr))

    @property
    def num_seqs(self):
        return lib().pgx_db_num_seqs(self.ptr)

    @property
    def num_bases(self):
        return lib().pgx_db_num_bases(self.ptr)

    def seq_id(self, i):
        return lib().pgx_db_seq_id(self.ptr, i).decode()

    def shape(self):
        s = _DbShape()
        _check(lib().pgx_db_get_shape(self.ptr, C.byref(s)))
        return (s.n_seq, s.n_bases, s.has_amb, s.index_bits, s.n_postings, s.synthetic_ids)

    def checksum(self):
        """wrapping 64-bit sums of (packed bases, sequence offsets, bucket offsets, postings) as they stand on this GPU"""
        out = (C.c_uint64 * 4)()
        _check(lib().pgx_db_checksum(self.ptr, out))
        return tuple(int(x) for x in out)

    def device_arrays(self):
        arr = (_DevArray * 16)()
        n = _check(lib().pgx_db_device_arrays(self.ptr, arr, 16))
        return [(arr[i].name.decode(), _CudaArray(arr[i].ptr, arr[i].bytes, self)) for i in range(n)]

    def set_ungapped(self, flag=True):
        """`blastn -ungapped` for searches through this handle: stop after the ungapped stage (spec v1)."""
        _check(lib().pgx_db_set_ungapped(self.ptr, 1 if flag else 0))

    def set_dust(self, flag=True):
        """`blastn -dust no` (flag False) for searches through this handle."""
        _check(lib().pgx_db_set_dust(self.ptr, 1 if flag else 0))

    def set_dust_each_search(self, flag):
        """S3d (query masking) recomputed inside every search through this handle, as BLAST runs it"""
        _check(lib().pgx_db_set_dust_each_search(self.ptr, 1 if flag else 0))

    def bind_taxonomy(self, tax):
        _check(lib().pgx_db_bind_taxonomy(self.ptr, tax.ptr))

    def subject_lineage(self, i):
        s = lib().pgx_db_subject_lineage(self.ptr, i)
        return None if s is None else s.decode("latin-1")


class Reads(_Handle):
    _close = "pgx_reads_close"

    @classmethod
    def from_synth(cls, cfg, first, count):
        p = C.c_void_p()
        _check(lib().pgx_reads_from_synth(C.byref(cfg), first, count, C.byref(p)))
        return cls(p)

    @classmethod
    def from_fasta(cls, path, first=0, count=-1):
        p = C.c_void_p()
        _check(lib().pgx_reads_from_fasta(_b(path), first, count, C.byref(p)))
        return cls(p)

    @classmethod
    def from_fasta_text(cls, text, first=0, count=-1):
        """A batch from FASTA text in memory (bytes), e.g. the FASTA trim2() returns."""
        p = C.c_void_p()
        _check(lib().pgx_reads_from_fasta_text(text, len(text), first, count, C.byref(p)))
        return cls(p)

    def write_fasta(self, path):
        _check(lib().pgx_reads_write_fasta(self.ptr, _b(path)))

    def redo_dust(self):
        """Recompute the batch's DUST window bits (what BLAST does inside every search)."""
        _check(lib().pgx_reads_redo_dust(self.ptr))

    def __len__(self):
        return lib().pgx_reads_count(self.ptr)

    def get(self, i, cap=1 << 16):
        buf = np.zeros(cap, dtype=np.uint8)
        n = C.c_int32()
        _check(lib().pgx_reads_get(self.ptr, i, buf.ctypes.data, cap, C.byref(n)))
        return buf[:n.value].copy()


class Hits(_Handle):
    _close = "pgx_hits_close"

    def __len__(self):
        return lib().pgx_hits_count(self.ptr)

    def to_numpy(self):
        n = len(self)
        out = np.zeros(n, dtype=HIT_DTYPE)
        if n:
            _check(lib().pgx_hits_copy(self.ptr, out.ctypes.data, n))
        return out

    def read_offsets(self, n_reads):
        out = np.zeros(n_reads + 1, dtype=np.int64)
        _check(lib().pgx_hits_read_offsets(self.ptr, out.ctypes.data, n_reads + 1))
        return out

    def read_counts(self, n_reads):
        out = np.zeros(n_reads, dtype=np.int64)
        _check(lib().pgx_hits_read_counts(self.ptr, out.ctypes.data, n_reads))
        return out

    def rows(self, n_reads):
        """(slots, slot offsets, row mask): the rows a text would show are slots[mask] (see pgx_hits in the header)."""
        h, off, cnt = self.to_numpy(), self.read_offsets(n_reads), self.read_counts(n_reads)
        idx = np.arange(len(h)) - np.repeat(off[:-1], np.diff(off))
        return h, off, idx < np.repeat(cnt, np.diff(off))

    def format(self, db, reads):
        txt, ln = C.c_void_p(), C.c_size_t()
        _check(lib().pgx_hits_format(self.ptr, db.ptr, reads.ptr, C.byref(txt), C.byref(ln)))
        return _take_text(txt.value, ln.value)

    def slice(self, first_read, n_reads):
        """The rows of reads [first_read, +n_reads) as a table of its own (read numbers and offsets from 0)."""
        p = C.c_void_p()
        _check(lib().pgx_hits_slice(self.ptr, C.c_int64(first_read), C.c_int64(n_reads), C.byref(p)))
        return Hits(p)


class Rdp(_Handle):
    _close = "pgx_rdp_close"

    @classmethod
    def from_synth(cls, cfg, first, count, db):
        p = C.c_void_p()
        _check(lib().pgx_rdp_from_synth(C.byref(cfg), first, count, db.ptr, C.byref(p)))
        return cls(p)

    @classmethod
    def from_file(cls, path, reads, db):
        p = C.c_void_p()
        _check(lib().pgx_rdp_from_file(_b(path), reads.ptr, db.ptr, C.byref(p)))
        return cls(p)

    def write_file(self, path, reads, db):
        _check(lib().pgx_rdp_write_file(self.ptr, reads.ptr, db.ptr, _b(path)))


class TaxDb(_Handle):
    """Tax_class/ncbitc.c in process: the .bin files of a directory + device parent/rank arrays."""
    _close = "pgx_tax_close"

    @staticmethod
    def create(directory):
        _check(lib().pgx_tax_create(_b(directory)))

    @classmethod
    def open(cls, directory):
        p = C.c_void_p()
        _check(lib().pgx_tax_open(_b(directory), C.byref(p)))
        return cls(p)

    def gi2taxid(self, gi):
        t = C.c_int()
        _check(lib().pgx_tax_gi2taxid(self.ptr, int(gi), C.byref(t)))
        return t.value

    def lineage_batch(self, gis):
        gis = np.ascontiguousarray(gis, dtype=np.int32)
        n = len(gis)
        lin = np.zeros((n, 16), dtype=np.int32)
        cnt = np.zeros(n, dtype=np.int32)
        st = np.zeros(n, dtype=np.int32)
        _check(lib().pgx_tax_lineage_batch(self.ptr, gis.ctypes.data, n, lin.ctypes.data, cnt.ctypes.data,
                                            st.ctypes.data))
        return lin, cnt, st

    def collect_file(self, in_path, out_path):
        rep = C.c_void_p()
        rc = lib().pgx_taxcollect_file(self.ptr, _b(in_path), _b(out_path), C.byref(rep))
        text = _take_text(rep.value)
        _check(rc)
        return text


def blast_search(db, reads):
    p = C.c_void_p()
    _check(lib().pgx_blast_search(db.ptr, reads.ptr, C.byref(p)))
    return Hits(p)


def classify_consensus(db, reads, rdp, want_records=True, want_hits=True, soap=None):
    """The fused hot path; `soap` = path of the SOAP classification (the -s stream of the reference's Consensus)."""
    n = len(reads)
    recs = np.zeros(n, dtype=REC_DTYPE) if want_records else None
    hp = C.c_void_p()
    if soap is None:
        _check(lib().pgx_classify_consensus(db.ptr, reads.ptr, rdp.ptr, C.byref(hp) if want_hits else None,
                                            recs.ctypes.data if want_records else None, n))
    else:
        _check(lib().pgx_classify_consensus_tri(db.ptr, reads.ptr, rdp.ptr, _b(soap), C.byref(hp) if want_hits else None,
                                                recs.ctypes.data if want_records else None, n))
    return (Hits(hp) if want_hits else None), recs


def vote3(db, reads, hits, rdp, soap):
    """Opt-in three-way vote (pgx-vote3 v1): (records, text).  `soap` = path of the SOAP table of the same reads."""
    n = len(reads)
    recs = np.zeros(n, dtype=VOTE_DTYPE)
    _check(lib().pgx_vote3_batch(db.ptr, reads.ptr, hits.ptr, rdp.ptr, _b(soap), recs.ctypes.data, n))
    txt, ln = C.c_void_p(), C.c_size_t()
    _check(lib().pgx_vote3_format(db.ptr, reads.ptr, recs.ctypes.data, n, C.byref(txt), C.byref(ln)))
    return recs, _take_text(txt.value, ln.value)


def consensus_format(db, reads, hits, recs):
    txt, ln = C.c_void_p(), C.c_size_t()
    recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
    _check(lib().pgx_consensus_format(db.ptr, reads.ptr, hits.ptr, recs.ctypes.data, len(recs), C.byref(txt),
                                      C.byref(ln)))
    return _take_text(txt.value, ln.value)


def consensus_format_file(db, reads, hits, recs, path):
    """the Consensus text of a batch straight into `path` (rendered and written piece by piece); returns its size"""
    ln = C.c_size_t()
    recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
    _check(lib().pgx_consensus_format_file(db.ptr, reads.ptr, hits.ptr, recs.ctypes.data, len(recs), str(path).encode(), C.byref(ln)))
    return ln.value


class _MegaclustOpts(C.Structure):
    _fields_ = [("in_path", C.c_char_p), ("out_path", C.c_char_p), ("s", C.c_char_p), ("e", C.c_char_p), ("b", C.c_char_p),
                ("c", C.c_char_p), ("d", C.c_char_p), ("help", C.c_int)]


def _mc_opts(i=None, o=None, s=None, e=None, b=None, c=None, d=None, h=False):
    t = lambda v: None if v is None else str(v).encode()  # noqa: E731  option texts, as on the command line
    return _MegaclustOpts(_b(i), _b(o), t(s), t(e), t(b), t(c), t(d), 1 if h else 0)


def megaclust_batch(db, reads, hits, recs, s=None, e=None, b=None, c=None, d=None):
    """megaclust2.pl's table for the consensus records of a batch in HBM: (csv bytes, stdout bytes)."""
    recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
    o = _mc_opts(None, None, s, e, b, c, d)
    txt, ln, log = C.c_void_p(), C.c_size_t(), C.c_void_p()
    rc = lib().pgx_megaclust_batch(db.ptr, reads.ptr, hits.ptr, recs.ctypes.data, len(recs), C.byref(o), C.byref(txt),
                                   C.byref(ln), C.byref(log))
    out = _take_text(log.value)
    _check(rc)
    return _take_text(txt.value, ln.value), out


def stage_times():
    t = StageTimes()
    _check(lib().pgx_last_stage_times(C.byref(t)))
    return t


# ------------------------------------------------------------------ mirrors of the reference command lines
def makeblastdb(infile, out):
    """`makeblastdb -in <infile> -out <out> -dbtype nucl` (reference README.md:62)."""
    _check(lib().pgx_db_build(_b(infile), _b(out)))


def blastn(query, db, out, outfmt=6, rank=0, world_size=1, ungapped=False, dust=True):
    """`blastn -query F -db DB -outfmt 6 -out O [-ungapped] [-dust no]` (reference README.md:96)."""
    o = _BlastnOpts(_b(query), _b(db), _b(out), int(outfmt), rank, world_size, 1 if ungapped else 0, 0 if dust else 1)
    _check(lib().pgx_blastn_run(C.byref(o)))


def soap_index(fasta):
    """`2bwt-builder ref.fasta` (reference README.md:130)."""
    _check(lib().pgx_soap_index(_b(fasta)))


def soap(a, D, o, u=None, M=4, r=1, n=5, t=False, b=None, unpaired=None, m=400, x=600):
    """`soap -a reads -D ref.index -o out [-u unmapped] -M 4 -r 1 -n 5` (reference README.md:134); paired-end with
    `b` (-b), `unpaired` (-2), `m` / `x` (-m / -x), soap.man:29-50."""
    opts = _SoapOpts(_b(a), _b(D), _b(o), _b(u), M, r, n, int(bool(t)), _b(b), _b(unpaired), m, x)
    _check(lib().pgx_soap_run(C.byref(opts)))


def tax_class(args, cwd="."):
    """`tax_class {-c,-s GI,-g GI,-t TAXID,-n TAXID,-v,-h}` (ncbitc.c:860-1004): (status, stdout, stderr)."""
    argv = [b"tax_class"] + [os.fsencode(a) for a in args]
    arr = (C.c_char_p * (len(argv) + 1))(*argv, None)
    out, err = C.c_void_p(), C.c_void_p()
    rc = lib().pgx_tax_cli(len(argv), arr, _b(cwd), C.byref(out), C.byref(err))
    return rc, _take_text(out.value), _take_text(err.value)


def taxcollector(f, o, taxdir="./Tax_class"):
    """`perl NCBI-taxcollector-0.01.pl -f in -o out` (reference README.md:109); returns the stdout report."""
    with TaxDb.open(taxdir) as db:
        return db.collect_file(f, o)


def megaclust2(i, o, s=None, e=None, b=None, c=None, d=None, h=False):
    """`perl Megaclust/megaclust2.pl -i IN -o OUT [-s -e -b -d -c -h]` (reference README.md:176); returns its stdout."""
    opts = _mc_opts(i, o, s, e, b, c, d, h)
    log = C.c_void_p()
    rc = lib().pgx_megaclust_file(C.byref(opts), C.byref(log))
    text = _take_text(log.value)
    _check(rc)
    return text


def megaclustable(argv):
    """`perl Megaclustable/megaclustable.pl -m A.csv B.csv ... -t LEVEL -o OUT` (reference README.md:185); `argv` is the
    word list after the script name; returns its stdout."""
    words = [os.fsencode(str(w)) for w in argv]
    arr = (C.c_char_p * max(1, len(words)))(*words)
    log = C.c_void_p()
    rc = lib().pgx_megaclustable(len(words), arr, C.byref(log))
    text = _take_text(log.value)
    _check(rc)
    return text


def blast_score_columns(score, qlen, db_len, db_nseq, gapped=True):
    """(e-value text, bit-score text) of an -outfmt 6 row with raw score `score`, as the row formatter prints them
    (gapped=False: the statistics of `blastn -ungapped`, spec S4u)."""
    ev, bs = C.create_string_buffer(32), C.create_string_buffer(32)
    _check(lib().pgx_blast_score_columns_v(int(score), int(qlen), int(db_len), int(db_nseq), 1 if gapped else 0, ev, bs))
    return ev.value.decode(), bs.value.decode()


class _TrimOpts(C.Structure):
    _fields_ = [("a", C.c_char_p), ("b", C.c_char_p), ("g", C.c_char_p), ("t", C.c_char_p), ("q", C.c_char_p), ("j", C.c_int)]


TRIM_NONE, TRIM_FASTQ, TRIM_QSEQ, TRIM_UNKNOWN, TRIM_FASTA_QUAL, TRIM_FASTA_JOIN = 0, 1, 2, 3, 4, 5


def trim2(a, b=None, g=None, t=None, q=None, j=False):
    """`perl Trim/trim2.3.pl -a READS_1 [-b READS_2] [-g GAP] [-t TRUNCATE]` (reference README.md:34) on FASTQ or QSEQ
    input.  Returns (messages, runblast FASTA bytes or None, mode): the script writes the FASTA to
    output_files/trim2/<basename of a>_runblast.fasta, which is left to the caller (bin/trim2 does it); its stdout is
    the FASTA followed by the messages when mode == TRIM_FASTQ, the messages alone otherwise."""
    txt = lambda v: None if v is None else str(v).encode()  # noqa: E731  option texts, as on the command line
    opts = _TrimOpts(_b(a), _b(b), txt(g), txt(t), txt(q), 1 if j else 0)
    log, fasta, ln, mode = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_int()
    rc = lib().pgx_trim_file(C.byref(opts), C.byref(log), C.byref(fasta), C.byref(ln), C.byref(mode))
    messages = _take_text(log.value)
    made = bool(fasta.value)
    text = _take_text(fasta.value, ln.value) if made else None
    _check(rc)
    return messages, text, mode.value


def consensus(b, r, o, s=None):
    """`perl Consensus_BLAST_SOAP_RDP-1.1.pl -b B -r R [-s S] -o O` (reference README.md:152)."""
    log = C.c_void_p()
    rc = lib().pgx_consensus_file(_b(b), _b(r), _b(s), _b(o), C.byref(log))
    text = _take_text(log.value)
    _check(rc)
    return text
