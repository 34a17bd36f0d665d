"""The C-ABI library loads and exports every symbol include/gapfill_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "gapfill_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as G
    G.build()
    from gappadder_amd import _lib as B
    names = _declared()
    assert len(names) >= 20
    L = ctypes.CDLL(B.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    B.lib()  # the typed binding resolves too


def test_struct_sizes_match_header():
    from gappadder_amd import _lib as B
    assert B.GAP.itemsize == 16 and B.ALNREC.itemsize == 32 and B.TAGHIT.itemsize == 12 and B.HIT.itemsize == 8
    assert B.ALNREC.fields["read"][1] == 24 and B.ALNREC.fields["flag"][1] == 20


def test_pack_reads_layout_is_kmerutils_msb_first():
    from gappadder_amd.hip_api import GapFill
    from oracle import gp_oracle as O
    seq = "ACGTTGCANACGTACGTTTGACCAGGATTACAN"
    packed, nm = GapFill.pack_reads([seq], len(seq), with_mask=True)
    b = packed[0]
    # first 32 bases == KmerUtils 64-bit value, big-endian bytes
    assert int.from_bytes(bytes(b[:8]), "big") == O.pack_kmer64(seq, 0, 32)
    assert [i for i in range(len(seq)) if (nm[0][i // 32] >> (i % 32)) & 1] == [8, 32]


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    try:
        GapFill(0)
    except B.GapFillError as e:
        assert e.code == B.GF_E_NODEV
    else:
        raise AssertionError("gf_init must fail without a device")
