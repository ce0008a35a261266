"""Readers/writers for the Kaldi and OpenFst binary formats that sit at the boundary of MFA's alignment path.

The reference never parses these itself — it hands paths to kalpy (``read_gmm_model``,
``read_transition_model``, ``FstArchive``, ``AlignmentArchive``, ``MatrixArchive``:
MFA/alignment/multiprocessing.py:26-42, MFA/models.py:481-491).  Formats: SURVEY.md Appendix A.5, A.11, A.13.
Host-side plumbing only: nothing here touches the GPU.
"""
from __future__ import annotations

import io
import struct
import zipfile
from dataclasses import dataclass, field
from pathlib import Path
from typing import BinaryIO, Dict, Iterator, List, Optional, Tuple

import numpy as np


class KaldiFormatError(RuntimeError):
    pass


class BinaryReader:
    """Cursor over a Kaldi binary stream."""

    def __init__(self, data: bytes, pos: int = 0):
        self.d = data
        self.p = pos

    def peek(self, n: int = 1) -> bytes:
        return self.d[self.p : self.p + n]

    def read(self, n: int) -> bytes:
        b = self.d[self.p : self.p + n]
        if len(b) != n:
            raise KaldiFormatError("unexpected end of data")
        self.p += n
        return b

    def expect_binary_header(self) -> None:
        if self.read(2) != b"\0B":
            raise KaldiFormatError("not a Kaldi binary object (text mode is not supported)")

    def token(self) -> str:
        end = self.d.index(b" ", self.p)
        t = self.d[self.p : end].decode("ascii")
        self.p = end + 1
        return t

    def expect(self, tok: str) -> None:
        t = self.token()
        if t != tok:
            raise KaldiFormatError(f"expected token {tok!r}, got {t!r}")

    def int32(self) -> int:
        sz = self.read(1)[0]
        if sz != 4:
            raise KaldiFormatError(f"expected int32 size marker 4, got {sz}")
        return struct.unpack("<i", self.read(4))[0]

    def uint32_or_int32(self) -> int:
        sz = struct.unpack("<b", self.read(1))[0]
        if sz == 4:
            return struct.unpack("<i", self.read(4))[0]
        if sz == -4:
            return struct.unpack("<I", self.read(4))[0]
        raise KaldiFormatError(f"bad integer size marker {sz}")

    def float32(self) -> float:
        sz = self.read(1)[0]
        if sz == 4:
            return struct.unpack("<f", self.read(4))[0]
        if sz == 8:
            return struct.unpack("<d", self.read(8))[0]
        raise KaldiFormatError(f"bad float size marker {sz}")

    def int_vector(self) -> np.ndarray:
        sz = self.read(1)[0]
        if sz != 4:
            raise KaldiFormatError("expected int32 vector")
        n = struct.unpack("<i", self.read(4))[0]
        return np.frombuffer(self.read(4 * n), dtype="<i4").copy()

    def vector(self) -> np.ndarray:
        t = self.token()
        if t not in ("FV", "DV"):
            raise KaldiFormatError(f"expected FV/DV, got {t!r}")
        n = self.int32()
        dt = "<f4" if t == "FV" else "<f8"
        return np.frombuffer(self.read(n * int(dt[2])), dtype=dt).copy()

    def matrix(self) -> np.ndarray:
        t = self.token()
        if t in ("FM", "DM"):
            r, c = self.int32(), self.int32()
            dt = "<f4" if t == "FM" else "<f8"
            return np.frombuffer(self.read(r * c * int(dt[2])), dtype=dt).reshape(r, c).copy()
        if t in ("CM", "CM2", "CM3"):
            return _read_compressed(self, t)
        raise KaldiFormatError(f"expected matrix token, got {t!r}")


def _read_compressed(r: BinaryReader, fmt: str) -> np.ndarray:
    """Kaldi CompressedMatrix (MFA/corpus/features.py:235,364 write these)."""
    min_value, rng, rows, cols = struct.unpack("<ffii", r.read(16))
    if fmt == "CM":
        hdr = np.frombuffer(r.read(8 * cols), dtype="<u2").reshape(cols, 4).astype(np.float32)
        data = np.frombuffer(r.read(rows * cols), dtype=np.uint8).reshape(cols, rows)
        p = min_value + rng * (1.0 / 65535.0) * hdr  # p0, p25, p75, p100
        out = np.empty((rows, cols), dtype=np.float32)
        for c in range(cols):
            v = data[c].astype(np.float32)
            p0, p25, p75, p100 = p[c]
            out[:, c] = np.where(
                v <= 64, p0 + (p25 - p0) * v * (1 / 64.0),
                np.where(v <= 192, p25 + (p75 - p25) * (v - 64) * (1 / 128.0), p75 + (p100 - p75) * (v - 192) * (1 / 63.0)),
            )
        return out
    if fmt == "CM2":
        data = np.frombuffer(r.read(2 * rows * cols), dtype="<u2").reshape(rows, cols)
        return (min_value + data.astype(np.float32) * (rng / 65535.0)).astype(np.float32)
    data = np.frombuffer(r.read(rows * cols), dtype=np.uint8).reshape(rows, cols)
    return (min_value + data.astype(np.float32) * (rng / 255.0)).astype(np.float32)


# --------------------------------------------------------------------------- writers
def _w_token(f: BinaryIO, tok: str) -> None:
    f.write(tok.encode("ascii") + b" ")


def _w_int32(f: BinaryIO, v: int) -> None:
    f.write(b"\x04" + struct.pack("<i", v))


def write_matrix(f: BinaryIO, m: np.ndarray) -> None:
    m = np.ascontiguousarray(m)
    if m.dtype == np.float64:
        _w_token(f, "DM")
    else:
        m = m.astype("<f4")
        _w_token(f, "FM")
    _w_int32(f, m.shape[0])
    _w_int32(f, m.shape[1])
    f.write(m.tobytes())


def write_compressed_matrix(f: BinaryIO, m: np.ndarray) -> None:
    """Kaldi CompressedMatrix, method kOneByteWithColHeaders ("CM"): what ``compute_mfccs_for_export(seg, compress=True)``
    and FinalFeatureFunction write into feats.*.ark (MFA/corpus/features.py:235, :364; Kaldi matrix/compressed-matrix.cc:
    ComputeGlobalHeader, ComputeColHeader, FloatToChar).  Columns are stored one after another, one
    byte per value, quantised piecewise-linearly between the column's 0/25/75/100-th percentiles (uint16 on a global range).
    Columns shorter than five rows use Kaldi's small-matrix rule (percentiles from the sorted values directly)."""
    m = np.ascontiguousarray(m, dtype=np.float32)
    rows, cols = m.shape
    if rows == 0 or cols == 0:
        raise ValueError("cannot compress an empty matrix")
    mn, mx = float(m.min()), float(m.max())
    if mx == mn:
        mx = mn + (1.0 + abs(mn))
    mn32, rng32 = np.float32(mn), np.float32(np.float32(mx) - np.float32(mn))
    _w_token(f, "CM")
    f.write(struct.pack("<ffii", float(mn32), float(rng32), rows, cols))

    def to_u16(v):
        fr = (np.float32(v) - mn32) / rng32
        fr = min(max(float(fr), 0.0), 1.0)
        return int(fr * 65535 + 0.499)

    hdr = np.zeros((cols, 4), dtype=np.uint16)
    data = np.zeros((cols, rows), dtype=np.uint8)
    inc = np.float32(rng32 * np.float32(1.0 / 65535.0))
    for c in range(cols):
        col = m[:, c]
        sd = np.sort(col)
        if rows >= 5:
            q = rows // 4
            p0 = min(to_u16(sd[0]), 65532)
            p25 = min(max(to_u16(sd[q]), p0 + 1), 65533)
            p75 = min(max(to_u16(sd[3 * q]), p25 + 1), 65534)
            p100 = max(to_u16(sd[rows - 1]), p75 + 1)
        else:
            p0 = min(to_u16(sd[0]), 65532)
            p25 = min(max(to_u16(sd[1]) if rows > 1 else p0 + 1, p0 + 1), 65533)
            p75 = min(max(to_u16(sd[2]) if rows > 2 else p25 + 1, p25 + 1), 65534)
            p100 = max(to_u16(sd[3]) if rows > 3 else p75 + 1, p75 + 1)
        hdr[c] = (p0, p25, p75, p100)
        f0, f25, f75, f100 = (np.float32(mn32 + inc * np.float32(x)) for x in (p0, p25, p75, p100))
        lo = np.clip(((col - f0) / (f25 - f0) * 64 + 0.5).astype(np.int64), 0, 64)
        mid = np.clip(64 + ((col - f25) / (f75 - f25) * 128 + 0.5).astype(np.int64), 64, 192)
        hi = np.clip(192 + ((col - f75) / (f100 - f75) * 63 + 0.5).astype(np.int64), 192, 255)
        data[c] = np.where(col < f25, lo, np.where(col < f75, mid, hi)).astype(np.uint8)
    f.write(hdr.astype("<u2").tobytes())
    f.write(data.tobytes())


def compress_round_trip(m: np.ndarray) -> np.ndarray:
    """What a feature matrix looks like after it has been written to and read back from a Kaldi CompressedMatrix table
    (8-bit, per-column headers) — the corpus path of the reference stores raw MFCCs and the CMVN-applied features this way
    (MFA/corpus/features.py:235, :356-365), so `mfa align` aligns features quantised twice."""
    buf = io.BytesIO()
    write_compressed_matrix(buf, m)
    r = BinaryReader(buf.getvalue())
    fmt = r.token()
    return _read_compressed(r, fmt)


def write_vector(f: BinaryIO, v: np.ndarray) -> None:
    v = np.ascontiguousarray(v)
    if v.dtype == np.float64:
        _w_token(f, "DV")
    else:
        v = v.astype("<f4")
        _w_token(f, "FV")
    _w_int32(f, v.shape[0])
    f.write(v.tobytes())


def write_int_vector(f: BinaryIO, v) -> None:
    v = np.ascontiguousarray(v, dtype="<i4")
    f.write(b"\x04" + struct.pack("<i", v.shape[0]) + v.tobytes())


# --------------------------------------------------------------------------- model objects
@dataclass
class HmmState:
    forward_pdf_class: int
    self_loop_pdf_class: int
    transitions: List[Tuple[int, float]]  # (dst state, prob)


@dataclass
class HmmTopology:
    phones: np.ndarray
    phone2idx: np.ndarray
    entries: List[List[HmmState]]

    def entry_for_phone(self, phone: int) -> List[HmmState]:
        return self.entries[int(self.phone2idx[phone])]


@dataclass
class RawTransitionModel:
    """What ``final.mdl`` stores before Kaldi's ComputeDerived."""

    topo: HmmTopology
    tuples: np.ndarray  # [n,4] phone, hmm_state, forward_pdf, self_loop_pdf
    log_probs: np.ndarray  # [num_tids+1] float32, index 0 unused


@dataclass
class RawAmDiagGmm:
    dim: int
    gconsts: List[np.ndarray]
    weights: List[np.ndarray]
    means_invvars: List[np.ndarray]
    inv_vars: List[np.ndarray]

    @property
    def num_pdfs(self) -> int:
        return len(self.gconsts)


def read_topology(r: BinaryReader) -> HmmTopology:
    r.expect("<Topology>")
    phones = r.int_vector()
    phone2idx = r.int_vector()
    sz = r.int32()
    is_hmm = True
    if sz == -1:
        is_hmm = False
        sz = r.int32()
    entries: List[List[HmmState]] = []
    for _ in range(sz):
        n_states = r.int32()
        states = []
        for _ in range(n_states):
            fwd = r.int32()
            slf = fwd if is_hmm else r.int32()
            n_tr = r.int32()
            trans = []
            for _ in range(n_tr):
                dst = r.int32()
                p = r.float32()
                trans.append((dst, float(np.float32(p))))
            states.append(HmmState(fwd, slf, trans))
        entries.append(states)
    r.expect("</Topology>")
    return HmmTopology(phones, phone2idx, entries)


def read_transition_model(r: BinaryReader) -> RawTransitionModel:
    r.expect("<TransitionModel>")
    topo = read_topology(r)
    tok = r.token()
    if tok not in ("<Triples>", "<Tuples>"):
        raise KaldiFormatError(f"expected <Triples>/<Tuples>, got {tok}")
    n = r.int32()
    tuples = np.zeros((n, 4), dtype=np.int32)
    for i in range(n):
        tuples[i, 0] = r.int32()
        tuples[i, 1] = r.int32()
        tuples[i, 2] = r.int32()
        tuples[i, 3] = r.int32() if tok == "<Tuples>" else tuples[i, 2]
    r.expect("</Triples>" if tok == "<Triples>" else "</Tuples>")
    r.expect("<LogProbs>")
    log_probs = r.vector().astype(np.float32)
    r.expect("</LogProbs>")
    r.expect("</TransitionModel>")
    return RawTransitionModel(topo, tuples, log_probs)


def read_am_diag_gmm(r: BinaryReader) -> RawAmDiagGmm:
    r.expect("<DIMENSION>")
    dim = r.int32()
    r.expect("<NUMPDFS>")
    n = r.int32()
    am = RawAmDiagGmm(dim, [], [], [], [])
    for _ in range(n):
        tok = r.token()
        if tok != "<DiagGMM>":
            raise KaldiFormatError(f"expected <DiagGMM>, got {tok}")
        tok = r.token()
        if tok == "<GCONSTS>":
            am.gconsts.append(r.vector().astype(np.float32))
            tok = r.token()
        else:
            am.gconsts.append(None)
        if tok != "<WEIGHTS>":
            raise KaldiFormatError(f"expected <WEIGHTS>, got {tok}")
        am.weights.append(r.vector().astype(np.float32))
        r.expect("<MEANS_INVVARS>")
        am.means_invvars.append(r.matrix().astype(np.float32))
        r.expect("<INV_VARS>")
        am.inv_vars.append(r.matrix().astype(np.float32))
        r.expect("</DiagGMM>")
    return am


def read_model(data: bytes) -> Tuple[RawTransitionModel, RawAmDiagGmm]:
    """Parse a ``final.mdl`` / ``final.alimdl`` byte string."""
    r = BinaryReader(data)
    r.expect_binary_header()
    tm = read_transition_model(r)
    am = read_am_diag_gmm(r)
    return tm, am


def write_model(f: BinaryIO, tm: RawTransitionModel, am: RawAmDiagGmm) -> None:
    """Write ``final.mdl`` in Kaldi binary form (<Triples> when forward == self-loop pdf everywhere)."""
    f.write(b"\0B")
    _w_token(f, "<TransitionModel>")
    _w_token(f, "<Topology>")
    write_int_vector(f, tm.topo.phones)
    write_int_vector(f, tm.topo.phone2idx)
    is_hmm = all(s.forward_pdf_class == s.self_loop_pdf_class for e in tm.topo.entries for s in e)
    if not is_hmm:
        _w_int32(f, -1)
    _w_int32(f, len(tm.topo.entries))
    for e in tm.topo.entries:
        _w_int32(f, len(e))
        for s in e:
            _w_int32(f, s.forward_pdf_class)
            if not is_hmm:
                _w_int32(f, s.self_loop_pdf_class)
            _w_int32(f, len(s.transitions))
            for dst, p in s.transitions:
                _w_int32(f, dst)
                f.write(b"\x04" + struct.pack("<f", p))
    _w_token(f, "</Topology>")
    triples = bool(np.all(tm.tuples[:, 2] == tm.tuples[:, 3]))
    _w_token(f, "<Triples>" if triples else "<Tuples>")
    _w_int32(f, tm.tuples.shape[0])
    for row in tm.tuples:
        for k in range(3 if triples else 4):
            _w_int32(f, int(row[k]))
    _w_token(f, "</Triples>" if triples else "</Tuples>")
    _w_token(f, "<LogProbs>")
    write_vector(f, tm.log_probs.astype(np.float32))
    _w_token(f, "</LogProbs>")
    _w_token(f, "</TransitionModel>")
    _w_token(f, "<DIMENSION>")
    _w_int32(f, am.dim)
    _w_token(f, "<NUMPDFS>")
    _w_int32(f, am.num_pdfs)
    for i in range(am.num_pdfs):
        _w_token(f, "<DiagGMM>")
        _w_token(f, "<GCONSTS>")
        write_vector(f, am.gconsts[i])
        _w_token(f, "<WEIGHTS>")
        write_vector(f, am.weights[i])
        _w_token(f, "<MEANS_INVVARS>")
        write_matrix(f, am.means_invvars[i])
        _w_token(f, "<INV_VARS>")
        write_matrix(f, am.inv_vars[i])
        _w_token(f, "</DiagGMM>")


def read_matrix_file(data: bytes) -> np.ndarray:
    """``lda.mat`` / single-matrix files."""
    r = BinaryReader(data)
    r.expect_binary_header()
    return r.matrix()


# --------------------------------------------------------------------------- decision tree (A.11)
@dataclass
class EventMap:
    kind: str  # "CE" | "TE" | "SE"
    answer: int = -1
    key: int = 0
    table: List[Optional["EventMap"]] = field(default_factory=list)
    yes_set: frozenset = frozenset()
    yes: Optional["EventMap"] = None
    no: Optional["EventMap"] = None

    def map(self, event: Dict[int, int]) -> Optional[int]:
        node = self
        while node is not None:
            if node.kind == "CE":
                return node.answer
            if node.key not in event:
                return None
            v = event[node.key]
            if node.kind == "TE":
                node = node.table[v] if 0 <= v < len(node.table) else None
            else:
                node = node.yes if v in node.yes_set else node.no
        return None

    def leaves(self) -> List[int]:
        if self.kind == "CE":
            return [self.answer]
        out: List[int] = []
        if self.kind == "TE":
            for t in self.table:
                if t is not None:
                    out += t.leaves()
        else:
            out += self.yes.leaves() + self.no.leaves()
        return out


def _read_event_map(r: BinaryReader) -> Optional[EventMap]:
    tok = r.token()
    if tok == "NULL":
        return None
    if tok == "CE":
        return EventMap("CE", answer=r.int32())
    if tok == "TE":
        key = r.int32()
        size = r.uint32_or_int32()
        r.expect("(")
        table = [_read_event_map(r) for _ in range(size)]
        r.expect(")")
        return EventMap("TE", key=key, table=table)
    if tok == "SE":
        key = r.int32()
        yes = frozenset(int(x) for x in r.int_vector())
        r.expect("{")
        y = _read_event_map(r)
        n = _read_event_map(r)
        r.expect("}")
        return EventMap("SE", key=key, yes_set=yes, yes=y, no=n)
    raise KaldiFormatError(f"bad EventMap token {tok!r}")


@dataclass
class ContextDependency:
    """Kaldi ``tree`` file: context width N, central position P, EventMap to pdf-ids (key -1 = pdf-class)."""

    context_width: int
    central_position: int
    to_pdf: EventMap

    def compute(self, phone_window: List[int], pdf_class: int) -> int:
        ev = {-1: pdf_class}
        for i, p in enumerate(phone_window):
            ev[i] = p
        ans = self.to_pdf.map(ev)
        if ans is None:
            raise KeyError(f"tree has no leaf for context {phone_window} pdf-class {pdf_class}")
        return ans


def read_tree(data: bytes) -> ContextDependency:
    r = BinaryReader(data)
    r.expect_binary_header()
    r.expect("ContextDependency")
    n = r.int32()
    p = r.int32()
    r.expect("ToPdf")
    em = _read_event_map(r)
    r.expect("EndContextDependency")
    return ContextDependency(n, p, em)


# --------------------------------------------------------------------------- OpenFst VectorFst<StdArc> (A.13)
FST_MAGIC = 0x7EB2FDD6
ARC_DTYPE = np.dtype([("ilabel", "<i4"), ("olabel", "<i4"), ("weight", "<f4"), ("nextstate", "<i4")])


@dataclass
class Fst:
    """A tropical-semiring FST in CSR form (the layout the device consumes)."""

    start: int
    arc_offsets: np.ndarray  # int64 [S+1]
    arcs: np.ndarray  # ARC_DTYPE [A]
    final: np.ndarray  # float32 [S], +inf = non-final

    @property
    def num_states(self) -> int:
        return int(self.final.shape[0])

    @property
    def num_arcs(self) -> int:
        return int(self.arcs.shape[0])


def _fst_string(r: BinaryReader) -> str:
    n = struct.unpack("<i", r.read(4))[0]
    return r.read(n).decode("ascii")


def read_fst(r: BinaryReader) -> Fst:
    magic = struct.unpack("<i", r.read(4))[0]
    if magic & 0xFFFFFFFF != FST_MAGIC:
        raise KaldiFormatError("bad OpenFst magic")
    fst_type = _fst_string(r)
    arc_type = _fst_string(r)
    if fst_type != "vector" or arc_type != "standard":
        raise KaldiFormatError(f"only VectorFst<StdArc> is supported, got {fst_type}/{arc_type}")
    _version, flags = struct.unpack("<ii", r.read(8))
    _props, start, num_states, _num_arcs = struct.unpack("<Qqqq", r.read(32))
    if flags & 0x3:
        raise KaldiFormatError("FST with embedded symbol tables is not supported")
    finals = np.empty(num_states, dtype=np.float32)
    offs = np.zeros(num_states + 1, dtype=np.int64)
    chunks = []
    for s in range(num_states):
        finals[s] = struct.unpack("<f", r.read(4))[0]
        narcs = struct.unpack("<q", r.read(8))[0]
        chunks.append(np.frombuffer(r.read(16 * narcs), dtype=ARC_DTYPE))
        offs[s + 1] = offs[s] + narcs
    arcs = np.concatenate(chunks) if chunks else np.zeros(0, dtype=ARC_DTYPE)
    return Fst(int(start), offs, arcs.copy(), finals)


def write_fst(f: BinaryIO, fst: Fst) -> None:
    f.write(struct.pack("<i", FST_MAGIC))
    for s in ("vector", "standard"):
        f.write(struct.pack("<i", len(s)) + s.encode("ascii"))
    f.write(struct.pack("<ii", 2, 0))
    f.write(struct.pack("<Qqqq", 0x0000000000000003, fst.start, fst.num_states, 0))  # kExpanded|kMutable
    for s in range(fst.num_states):
        a0, a1 = int(fst.arc_offsets[s]), int(fst.arc_offsets[s + 1])
        f.write(struct.pack("<fq", float(fst.final[s]), a1 - a0))
        f.write(np.ascontiguousarray(fst.arcs[a0:a1]).tobytes())


# --------------------------------------------------------------------------- ark / scp tables
def read_ark(data: bytes, kind: str) -> Iterator[Tuple[str, object]]:
    """Iterate a binary ark held in memory.  kind: 'matrix' | 'vector' | 'int_vector' | 'fst'."""
    r = BinaryReader(data)
    n = len(data)
    while r.p < n:
        end = data.index(b" ", r.p)
        key = data[r.p : end].decode("utf8")
        r.p = end + 1
        if kind == "fst":
            yield key, read_fst(r)
            continue
        r.expect_binary_header()
        if kind == "matrix":
            yield key, r.matrix()
        elif kind == "vector":
            yield key, r.vector()
        elif kind == "int_vector":
            yield key, r.int_vector()
        else:
            raise ValueError(kind)


def write_ark_entry(f: BinaryIO, key: str, obj, kind: str) -> int:
    """Append one entry; returns the byte offset an scp line should point at."""
    f.write(key.encode("utf8") + b" ")
    off = f.tell()
    if kind == "fst":
        write_fst(f, obj)
        return off
    f.write(b"\0B")
    if kind == "matrix":
        write_matrix(f, obj)
    elif kind == "compressed_matrix":
        write_compressed_matrix(f, obj)
    elif kind == "vector":
        write_vector(f, obj)
    elif kind == "int_vector":
        write_int_vector(f, obj)
    else:
        raise ValueError(kind)
    return off


def write_table(ark_path, entries, kind: str, scp_path=None) -> None:
    """Write a binary ark (and its scp: ``key path:offset``) from (key, object) pairs — Kaldi ``ark,scp:`` wspecifier."""
    ark_path = Path(ark_path)
    lines = []
    with open(ark_path, "wb") as f:
        for key, obj in entries:
            off = write_ark_entry(f, key, obj, kind)
            lines.append(f"{key} {ark_path}:{off}\n")
    if scp_path is not None:
        Path(scp_path).write_text("".join(lines), encoding="utf8")


def read_scp(scp_path) -> List[Tuple[str, str, int]]:
    """(key, ark path, byte offset) per line of a Kaldi scp."""
    out = []
    for line in Path(scp_path).read_text(encoding="utf8").splitlines():
        line = line.strip()
        if not line:
            continue
        key, rest = line.split(None, 1)
        path, _, off = rest.rpartition(":")
        out.append((key, path, int(off)))
    return out


def read_scp_object(ark_cache: Dict[str, bytes], path: str, offset: int, kind: str):
    """Object an scp line points at (``offset`` = first byte after the key's space)."""
    if path not in ark_cache:
        ark_cache[path] = Path(path).read_bytes()
    r = BinaryReader(ark_cache[path])
    r.p = offset
    if kind == "fst":
        return read_fst(r)
    r.expect_binary_header()
    return {"matrix": r.matrix, "vector": r.vector, "int_vector": r.int_vector}[kind]()


def load_acoustic_model_archive(path) -> Dict[str, bytes]:
    """Members of an MFA acoustic-model zip keyed by base name (MFA/models.py:367-379 lists them)."""
    out: Dict[str, bytes] = {}
    with zipfile.ZipFile(path) as zf:
        for info in zf.infolist():
            if info.is_dir():
                continue
            out[Path(info.filename).name] = zf.read(info.filename)
    return out


def read_wav_pcm16(path) -> Tuple[np.ndarray, int]:
    """PCM16 RIFF reader (stdlib only).  Returns (int16 [channels, N], sample_rate).

    The reference goes through librosa/soundfile with resampling; the parity domain of this
    engine is native 16 kHz PCM16, so other encodings are rejected loudly rather than approximated.
    """
    import wave

    with wave.open(str(path), "rb") as w:
        if w.getsampwidth() != 2:
            raise KaldiFormatError(f"{path}: only 16-bit PCM wav is supported (sampwidth={w.getsampwidth()})")
        ch, sr, n = w.getnchannels(), w.getframerate(), w.getnframes()
        pcm = np.frombuffer(w.readframes(n), dtype="<i2").reshape(-1, ch).T.copy()
    return pcm, sr
