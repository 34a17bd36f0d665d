"""Test-side BAM writer and reader (SAMv1 §4): SAM lines -> BAM bytes in BGZF blocks, and BGZF -> bytes through Python's zlib.
Only used to make inputs for, and to check, the GPU BAM path — the product never imports this."""
import random
import struct
import zlib

CIGAR_OPS = "MIDNSHP=X"
SEQ_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def bgzf_block(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    assert len(data) <= 65536
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    cdata = co.compress(data) + co.flush()
    bsize = len(cdata) + 25
    assert bsize <= 65536
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + cdata +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


BGZF_EOF = bgzf_block(b"")


def bgzf_compress(data, block=0xFF00, seed=None, levels=(6,), eof=True):
    """block: bytes of input per BGZF block (seed given: random sizes in [1, block]); levels are cycled per block: 0 = stored
    DEFLATE blocks, 'fixed' = fixed Huffman code, 'huffman' / 'rle' = zlib's Z_HUFFMAN_ONLY / Z_RLE strategies, 1..9 = zlib
    levels (dynamic codes)."""
    rng = random.Random(seed)
    out, p, i = [], 0, 0
    while p < len(data):
        n = block if seed is None else rng.randint(1, block)
        lv = levels[i % len(levels)]
        chunk = data[p:p + n]
        if lv == "fixed":
            out.append(bgzf_block(chunk, 6, zlib.Z_FIXED))
        elif lv == "huffman":      # Huffman coding only: no matches, long literal codes
            out.append(bgzf_block(chunk[:40000], 6, zlib.Z_HUFFMAN_ONLY))
            chunk = chunk[:40000]
        elif lv == "rle":          # matches of distance 1 only
            out.append(bgzf_block(chunk, 6, zlib.Z_RLE))
        else:
            if lv == 0 and len(chunk) > 65000:   # stored blocks add 5 bytes per 64 KiB - keep the BGZF block under 64 KiB
                chunk = chunk[:65000]
            out.append(bgzf_block(chunk, lv))
        p += len(chunk)
        i += 1
    if eof:
        out.append(BGZF_EOF)
    return b"".join(out)


def bgzf_decompress(data):
    out, p = [], 0
    while p < len(data):
        xlen = struct.unpack_from("<H", data, p + 10)[0]
        bsize = struct.unpack_from("<H", data, p + 16)[0] + 1
        out.append(zlib.decompress(data[p + 12 + xlen:p + bsize - 8], -15))
        p += bsize
    return b"".join(out)


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def sam_record(line, ref_index):
    f = line.rstrip("\n").split("\t")
    qname, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = f[:11]
    ref = ref_index.get(rname, -1) if rname != "*" else -1
    mref = ref if rnext == "=" else (ref_index.get(rnext, -1) if rnext != "*" else -1)
    ops, num = [], ""
    if cigar != "*":
        for ch in cigar:
            if ch.isdigit():
                num += ch
            else:
                ops.append((int(num) << 4) | CIGAR_OPS.index(ch))
                num = ""
    if seq == "*":
        seq = ""
    l_seq = len(seq)
    codes = [SEQ_CODE.get(c.upper(), 15) for c in seq] + [0]
    packed = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, l_seq, 2))
    q = bytes([0xFF] * l_seq) if qual == "*" else bytes(ord(c) - 33 for c in qual)
    tags = b""
    for t in f[11:]:
        tag, ty, val = t.split(":", 2)
        if ty == "i":
            tags += tag.encode() + b"i" + struct.pack("<i", int(val))
        elif ty == "A":
            tags += tag.encode() + b"A" + val.encode()
        else:
            tags += tag.encode() + b"Z" + val.encode() + b"\0"
    p0 = int(pos) - 1
    reflen = sum(v >> 4 for v in ops if (v & 15) in (0, 2, 3, 7, 8)) or 1
    body = struct.pack("<iiBBHHHiiii", ref, p0, len(qname) + 1, int(mapq), _reg2bin(max(p0, 0), max(p0, 0) + reflen), len(ops), int(flag),
                       l_seq, mref, int(pnext) - 1, int(tlen))
    body += qname.encode() + b"\0" + b"".join(struct.pack("<I", v) for v in ops) + packed + q + tags
    return struct.pack("<i", len(body)) + body


def sam_to_bam_stream(sam_lines, ref_names, ref_lens, header_text=None):
    """uncompressed BAM bytes for the alignment lines (header lines are skipped; the @SQ text is generated)."""
    if header_text is None:
        header_text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in zip(ref_names, ref_lens))
    out = [b"BAM\x01", struct.pack("<i", len(header_text)), header_text.encode(), struct.pack("<i", len(ref_names))]
    for n, l in zip(ref_names, ref_lens):
        out.append(struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l))
    idx = {n: i for i, n in enumerate(ref_names)}
    for line in sam_lines:
        if line and line[0] != "@" and len(line.split("\t")) >= 11:
            out.append(sam_record(line, idx))
    return b"".join(out)
