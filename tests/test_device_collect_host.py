"""Host-side pieces of gappadder_amd/device_collect.py that need no GPU."""
import os

import pytest


def test_read_length_guess_takes_the_longest_of_the_first_records(tmp_path):
    pytest.importorskip("torch")
    from gappadder_amd.device_collect import _guess_read_len
    recs = ["@r%d/1 extra\n%s\n+\n%s\n" % (i, "ACGTN"[i % 5] * (20 + i % 7), "@" * (20 + i % 7)) for i in range(57)]   # '@' in the qualities
    p = os.path.join(str(tmp_path), "x.fq")
    open(p, "w").write("".join(recs)[:-1])
    assert _guess_read_len([p]) == 26
    assert _guess_read_len([p], n_records=3) == 22
