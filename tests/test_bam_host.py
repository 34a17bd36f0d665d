"""Host side of the builtin BAM mode (gappadder_amd/bam_io.py) without a GPU: header parsing, the text columns cut from an
inflated stream, the .fai writer — against the SAM fixtures the reference's lists were generated from."""
import os

import bam_util as U
from golden_util import CASES, Case
from gappadder_amd import bam_io


def test_header_and_columns_round_trip():
    for name in CASES:
        case = Case(name)
        names = [l.split()[0] for l in case.fai.splitlines() if l.strip()]
        for lib in case.libs:
            lines = lib["sam"].splitlines()[:400] + ["qz\t77\t*\t0\t0\t*\t*\t0\t0\tACGTNACGT\tIIII#IIII\tNM:i:1",
                                                     "qw\t141\t*\t0\t0\t*\t*\t0\t0\tACG\t*"]
            stream = U.sam_to_bam_stream(lines, names, [1000] * len(names))
            assert U.bgzf_decompress(U.bgzf_compress(stream, block=3000, seed=1, levels=(0, 6, "fixed"))) == stream
            got_names, first = bam_io.parse_header(stream)
            assert got_names == names
            for cut in (0, 3, 11, first - 1):
                assert bam_io.parse_header(stream[:cut]) is None          # incomplete header: ask for more bytes
            rb, o = [], first
            while o < len(stream):
                rb.append(o)
                o += 4 + int.from_bytes(stream[o:o + 4], "little")
            assert o == len(stream) and len(rb) == len(lines)
            cols = bam_io.BamCols(stream, rb, got_names)
            for i, l in enumerate(lines):
                f = l.split("\t")
                assert cols[i] == f[:9] and list(cols.seq_qual(i)) == f[9:11], l


def test_write_fai_equals_samtools_faidx_fixture(tmp_path):
    for name in CASES:
        case = Case(name)
        p = os.path.join(str(tmp_path), name + ".fa")
        open(p, "w").write(case.draft_fa)
        bam_io.write_fai(p)
        assert open(p + ".fai").read() == case.fai


def test_builtin_switch():
    assert bam_io.is_builtin("builtin") and bam_io.is_builtin("builtin/") and not bam_io.is_builtin("samtools") and not bam_io.is_builtin(None)
