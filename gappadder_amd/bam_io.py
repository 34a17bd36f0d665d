"""BAM file -> decoded 32-byte alignment records (gf_alnrec) without samtools: the BGZF blocks are inflated and the records
decoded on the GPU (gf_bgzf_inflate + gf_bam_pack, csrc/bam.hip); the host only reads the header and cuts the text columns of
the few records that produce a hit.  The records and columns equal what sam_io.decode gives for the `samtools view` lines of
the same file (the reference's input, collect_reads_for_gaps.py:76-91), in file order — one pass over the whole BAM instead
of one `samtools view <bam> <scaffold>` pipe per scaffold (run_multi_threads_collect_reads.py:30-32)."""
import io
import struct

import numpy as np

NO_REF = 0xFFFFFFFF
CIGAR_OPS = "MIDNSHP=X"
SEQ_CODES = "=ACMGRSVTWYHKDBN"
BUILTIN = "builtin"
_NIBBLE_PAIRS = np.array([(ord(SEQ_CODES[b >> 4]) | (ord(SEQ_CODES[b & 15]) << 8)) for b in range(256)], dtype="<u2")   # SEQ byte -> its two characters


def is_builtin(samtools_path):
    """software_path.samtools = "builtin" in the configuration JSON selects this module instead of a samtools executable."""
    return samtools_path is not None and str(samtools_path).rstrip("/") == BUILTIN


def parse_header(stream):
    """Inflated bytes from the start of the file -> (reference names, offset of the first alignment record), or None when
    the header is not complete yet (SAMv1 §4.2: magic, l_text, text, n_ref, then l_name/name/l_ref per reference)."""
    b = bytes(stream[:12]) if len(stream) >= 12 else b""
    if len(b) < 12:
        return None
    if b[:4] != b"BAM\x01":
        raise ValueError("not a BAM stream (magic %r)" % b[:4])
    l_text = struct.unpack_from("<i", b, 4)[0]
    o = 8 + l_text
    if len(stream) < o + 4:
        return None
    n_ref = struct.unpack_from("<i", bytes(stream[o:o + 4]))[0]
    o += 4
    names = []
    for _ in range(n_ref):
        if len(stream) < o + 4:
            return None
        l_name = struct.unpack_from("<i", bytes(stream[o:o + 4]))[0]
        if len(stream) < o + 4 + l_name + 4:
            return None
        names.append(bytes(stream[o + 4:o + 4 + l_name - 1]).decode())
        o += 4 + l_name + 4
    return names, o


class BamCols:
    """The SAM columns of record i as `samtools view` prints them, cut on demand from the record's bytes.  Host-backed
    (`stream` = inflated bytes) or device-backed (`stream` = None, `gf` given): the records asked for are gathered from the
    stream the GPU still holds (gf_bam_fetch) — valid until the next gf.bgzf_inflate / gf.bam_pack call; prefetch(indices)
    gets many in one gather."""

    def __init__(self, stream, rec_begin, ref_names, gf=None, stream_end=None):
        self.s, self.rb, self.names, self.gf = stream, np.asarray(rec_begin, dtype=np.uint64), ref_names, gf
        self.end = int(stream_end if stream_end is not None else (len(stream) if stream is not None else 0))
        self.cache = {}
        self.on_device = stream is None   # the decoded records are still on the GPU too (gf.tag_alignments_bam)

    def __len__(self):
        return len(self.rb)

    def prefetch(self, indices):
        if self.s is not None:
            return
        idx = np.array(sorted(set(int(i) for i in indices) - set(self.cache)), dtype=np.int64)
        if not len(idx):
            return
        begin = self.rb[idx]
        nxt = np.append(self.rb[1:], np.uint64(self.end))
        end = nxt[idx]
        blob = self.gf.bam_fetch(begin, end).tobytes()
        o = 0
        for i, ln in zip(idx, (end - begin)):
            self.cache[int(i)] = blob[o:o + int(ln)]
            o += int(ln)

    def _rec(self, i):
        i = int(i)
        if self.s is not None:
            o = int(self.rb[i])
            return bytes(self.s[o:o + 4 + struct.unpack_from("<i", bytes(self.s[o:o + 4]))[0]])
        if i not in self.cache:
            self.prefetch([i])
        return self.cache[i]

    def __getitem__(self, i):
        r = self._rec(i)
        ref, pos, l_name, mapq, _bin, n_cig, flag, _l_seq, mref, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", r, 4)
        qname = r[36:36 + l_name - 1].decode()
        ops = np.frombuffer(r[36 + l_name:36 + l_name + 4 * n_cig], dtype="<u4")
        cigar = "".join("%d%s" % (v >> 4, CIGAR_OPS[v & 15]) for v in ops) or "*"
        rname = self.names[ref] if ref >= 0 else "*"
        rnext = "*" if mref < 0 else ("=" if mref == ref else self.names[mref])
        return [qname, str(flag), rname, str(pos + 1), str(mapq), cigar, rnext, str(mpos + 1), str(tlen)]

    def seq_qual(self, i):
        """SEQ and QUAL columns of record i ('*' for an absent one), as `samtools view` prints them."""
        r = self._rec(i)
        l_name, n_cig, l_seq = r[12], r[16] | (r[17] << 8), struct.unpack_from("<i", r, 20)[0]
        p = 36 + l_name + 4 * n_cig
        if l_seq == 0:
            return "*", "*"
        nib = np.frombuffer(r[p:p + (l_seq + 1) // 2], dtype=np.uint8)
        seq = _NIBBLE_PAIRS[nib].tobytes()[:l_seq].decode()          # two bases per byte through a 256-entry table
        q = np.frombuffer(r[p + (l_seq + 1) // 2:p + (l_seq + 1) // 2 + l_seq], dtype=np.uint8)
        qual = "*" if len(q) and q[0] == 0xFF else (q + 33).tobytes().decode()
        return seq, qual


def decode_chunks(gf, chunks, fai_names):
    """chunks: iterable of consecutive pieces of a BAM file (any sizes).  Yields (records, BamCols) per piece that completed
    at least one record; record indices (rec.read) restart at 0 in every yield, like one sam_io.decode call per piece.
    Only file bytes go to the GPU and only records (32 B each), the header and the records later asked of BamCols come back:
    the inflated stream never crosses PCIe.  A BamCols is valid until the next piece is requested."""
    index = {n: i for i, n in enumerate(fai_names)}
    ref_names, ref_map = None, None
    file_carry, rec_carry = b"", b""
    for piece in chunks:
        data = file_carry + bytes(piece)
        n_stream, used = gf.bgzf_inflate(data, rec_carry, want_host=False)
        file_carry = data[used:]
        first = 0
        if ref_names is None:
            hdr, want = None, 1 << 16
            while hdr is None and n_stream:
                got = min(want, n_stream)
                hdr = parse_header(gf.bam_fetch([0], [got]).tobytes())
                if got == n_stream:
                    break
                want *= 4
            if hdr is None:          # header longer than the pieces so far: keep everything inflated and go on
                rec_carry = gf.bam_fetch([0], [n_stream]).tobytes() if n_stream else b""
                continue
            ref_names, first = hdr
            ref_map = np.array([index.get(n, NO_REF) for n in ref_names], dtype=np.uint32)
        recs, rb, consumed = gf.bam_pack(None, first, ref_map, n_bytes=n_stream)
        rec_carry = gf.bam_fetch([consumed], [n_stream]).tobytes() if consumed < n_stream else b""
        if len(recs):
            yield recs, BamCols(None, rb, ref_names, gf=gf, stream_end=consumed)
    if file_carry:
        raise ValueError("BAM file ends inside a BGZF block (%d stray bytes)" % len(file_carry))
    if rec_carry and ref_names is not None:
        raise ValueError("BAM file ends inside an alignment record (%d stray bytes)" % len(rec_carry))


def read_file_chunks(path, chunk_bytes=256 << 20):
    with open(path, "rb") as f:
        while True:
            b = f.read(chunk_bytes)
            if not b:
                return
            yield b


def decode_file(gf, path, fai_names, chunk_bytes=256 << 20):
    return decode_chunks(gf, read_file_chunks(path, chunk_bytes), fai_names)


def _fai_rows_slow(data):
    rows, cur, off = [], None, 0
    for line in io.BytesIO(data):
        if line.startswith(b">"):
            if cur:
                rows.append(cur)
            cur = [line[1:].split()[0].decode(), 0, off + len(line), 0, 0]
        elif cur is not None and line.strip():
            if cur[3] == 0:
                cur[3], cur[4] = len(line.rstrip(b"\r\n")), len(line)
            cur[1] += len(line.rstrip(b"\r\n"))
        off += len(line)
    if cur:
        rows.append(cur)
    return rows


def write_fai(sf_fasta):
    """`samtools faidx` stand-in for the builtin mode: NAME LENGTH OFFSET LINEBASES LINEWIDTH per sequence.  Plain records (LF line
    ends, no blank or white-space lines) are measured with byte searches — a 250-Mb draft is 4 M lines; anything else takes the
    line loop."""
    with open(sf_fasta, "rb") as f:
        data = f.read()
    # (bytes.strip() takes " \t\n\r\v\f": a line of those alone is skipped by the line loop)
    if any(c in data for c in (b"\r", b"\t", b"\x0b", b"\x0c", b" \n", b"\n\n")) or not data.startswith(b">"):
        rows = _fai_rows_slow(data)
    else:
        rows, at = [], 0
        while at < len(data):
            nxt = data.find(b"\n>", at)
            end = nxt + 1 if nxt >= 0 else len(data)
            h_end = data.find(b"\n", at, end)
            if h_end < 0:                          # a header without a line end closes the file
                rows.append([data[at + 1:end].split()[0].decode(), 0, end, 0, 0])
                break
            body0 = h_end + 1
            l1 = data.find(b"\n", body0, end)
            n_nl = data.count(b"\n", body0, end)
            if body0 == end:
                lb = lw = 0
            elif l1 < 0:
                lb = lw = end - body0
            else:
                lb, lw = l1 - body0, l1 - body0 + 1
            rows.append([data[at + 1:h_end].split()[0].decode(), end - body0 - n_nl, body0, lb, lw])
            at = end
    with open(sf_fasta + ".fai", "w") as f:
        f.write("".join("%s\t%d\t%d\t%d\t%d\n" % tuple(r) for r in rows))
