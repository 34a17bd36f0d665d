"""Seeded contig sets around flank anchors for the picker tests (host and device against oracle/gp_oracle.py::pick_gap)."""
import numpy as np


def rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def picker_cases(seed, n_gaps, flank_len=120, per_gap=(1, 4)):
    """[(left flank, right flank, [contig, ...])]: every gap gets contigs built from its own 30- and 15-base anchors in every
    arrangement the picker distinguishes — in order / reverse-complemented / repeated anchors / wrong order / back to back
    (span 0) / one anchor only / overlapping anchors / no anchor / anchors of BOTH orientations in one contig — plus noise."""
    rng = np.random.default_rng(seed)
    out = []
    for g in range(n_gaps):
        l, r = rand_seq(rng, flank_len), rand_seq(rng, flank_len)
        if g % 17 == 7:
            l, r = "ACGT" * 5, "TTGA" * 5                  # shorter than anchor 30 (and 20 >= 15)
        if g % 17 == 8:
            l = l[:-3] + "NNN"                             # non-ACGT inside the anchors: never anchored
        if g % 17 == 9:
            l, r = l[-30:], r[:30]                         # flank == anchor: unclipped hits at 30
        contigs = []
        for _ in range(int(rng.integers(per_gap[0], per_gap[1] + 1))):
            a = int(rng.choice([30, 15]))
            if len(l) < a:
                a = 15
            la, ra = l[-a:], r[:a]
            mid = rand_seq(rng, int(rng.integers(0, 300)))
            kind = int(rng.integers(0, 10))
            if kind == 0:
                s = rand_seq(rng, 50) + la + mid + ra + rand_seq(rng, 40)
            elif kind == 1:
                s = _rc(rand_seq(rng, 20) + la + mid + ra + rand_seq(rng, 5))
            elif kind == 2:
                s = la + rand_seq(rng, 30) + la + mid + ra + rand_seq(rng, 17) + ra
            elif kind == 3:
                s = ra + mid + la
            elif kind == 4:
                s = rand_seq(rng, 9) + la + ra
            elif kind == 5:
                s = rand_seq(rng, 60) + la + mid
            elif kind == 6:
                s = la[:-5] + ra
            elif kind == 7:
                s = mid + ra + rand_seq(rng, 30)
            elif kind == 8:                                # forward pair and reverse pair in ONE contig: the forward pair is taken
                s = la + mid[:40] + ra + rand_seq(rng, 10) + _rc(la + mid + ra)
            else:
                s = rand_seq(rng, int(rng.integers(10, 25)))
            contigs.append(s)
        contigs.append(rand_seq(rng, 200))
        out.append((l, r, contigs))
    return out
