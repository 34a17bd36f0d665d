"""SAM text -> decoded 32-byte alignment records (gf_alnrec).  Only the nine columns the reference reads are used
(collect_reads_for_gaps.py:76-91).  RNEXT '=' and an RNEXT equal to RNAME both decode to mate_ref == ref (samtools
always writes '=' in that case)."""
import numpy as np

from . import _lib as B

NO_REF = 0xFFFFFFFF


def clip_flag(cigar):
    """GapReadsCollector.is_clipped (collect_reads_for_gaps.py:13-26): +2 when the CIGAR ends in S/H, +1 when its first
    operation is S/H."""
    f = 2 if cigar[-1] in "SH" else 0
    for ch in cigar:
        if ch.isdigit():
            continue
        if ch in "SH":
            f += 1
        break
    return f


def decode(lines, scaffold_index):
    """lines: iterable of SAM alignment lines.  Returns (records, columns) with columns[i] = the first 9 fields."""
    cols = []
    for line in lines:
        if not line or line[0] == "@":
            continue
        f = line.split(None, 9)
        if len(f) < 9:
            continue
        cols.append(f[:9])
    recs = np.zeros(len(cols), dtype=B.ALNREC)
    for i, f in enumerate(cols):
        ref = scaffold_index.get(f[2], NO_REF)
        mref = ref if f[6] == "=" else scaffold_index.get(f[6], NO_REF)
        recs[i] = (int(f[3]), int(f[7]), int(f[8]), ref, mref, int(f[1]) & 0xFFFF, min(255, int(f[4])), clip_flag(f[5]), i)
    return recs, cols


class LazyCols:
    """The first nine columns of record i, cut from the SAM text on demand (only the records that produce a hit need them)."""

    def __init__(self, text, line_begin):
        self.text, self.lb = text, line_begin

    def __len__(self):
        return len(self.lb)

    def __getitem__(self, i):
        a = int(self.lb[i])
        b = self.text.find(b"\n", a)
        return self.text[a:b if b >= 0 else len(self.text)].decode().split(None, 9)[:9]


def decode_on_device(gf, sam_lines, names):
    """Same records as decode(sam_lines, {name: index}), parsed by gf_sam_pack on the GPU; columns come back lazily.
    names: list indexed like the scaffold index (the .fai order)."""
    text = sam_lines if isinstance(sam_lines, (bytes, bytearray)) else "\n".join(l.rstrip("\n") for l in sam_lines).encode()
    recs, lb = gf.sam_pack(text, names)
    return recs, LazyCols(bytes(text), lb)


def read_fai(path):
    names = []
    with open(path) as f:
        for line in f:
            if line.strip():
                names.append(line.split()[0])
    return names


def read_gap_positions(path, scaffold_index):
    """gap_positions.txt -> (structured gap array, per-gap key '{scaffoldIdx}_{n}').  n restarts at 1 whenever the scaffold
    changes (collect_reads_for_gaps.py:34-63, merge_reads.py:27-41)."""
    rows, keys = [], []
    cnt, pre = 1, None
    with open(path) as f:
        for line in f:
            fl = line.split()
            if len(fl) < 4:
                continue
            if fl[3] != pre:
                cnt = 1
            rows.append((scaffold_index[fl[3]], int(fl[0]), int(fl[1]), cnt))
            keys.append("%d_%d" % (scaffold_index[fl[3]], cnt))
            cnt += 1
            pre = fl[3]
    return np.array(rows, dtype=B.GAP), keys
