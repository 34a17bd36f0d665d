"""Host-side pieces of gappadder_amd/device_collect.py that need no GPU: FASTQ files cut into pieces of whole records."""
import os

import pytest


@pytest.mark.parametrize("chunk", [1, 7, 64, 1000, 1 << 20])
def test_fastq_pieces_hold_whole_records_and_cover_the_file(tmp_path, chunk):
    pytest.importorskip("torch")
    from gappadder_amd.device_collect import _fastq_chunks, _guess_read_len
    recs = ["@r%d/1 extra\n%s\n+\n%s\n" % (i, "ACGTN"[i % 5] * (20 + i % 7), "@" * (20 + i % 7)) for i in range(57)]   # '@' in the qualities
    text = "".join(recs)[:-1]                                                                                   # no newline at the end
    p = os.path.join(str(tmp_path), "x.fq")
    open(p, "w").write(text)
    got, pos = [], 0
    for off, data in _fastq_chunks(p, chunk):
        assert off == pos and data
        pos += len(data)
        got.append(data)
    assert b"".join(got).decode() == text
    for piece in got[:-1]:
        assert piece.count(b"\n") % 4 == 0 and piece.startswith(b"@r") and piece.endswith(b"\n")
    assert _guess_read_len([p]) == 26
