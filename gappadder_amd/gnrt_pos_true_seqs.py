"""Preprocess stage: gap positions and flank sequences of the draft (mirrors DGProcessor, gnrt_pos_true_seqs.py:7-100).
Host-side and tiny in the reference and here; the flanks also seed the GPU's flank k-mer index."""
import os
import re


def read_fasta(path):
    """(name, sequence) per record: name = first word behind '>', sequence = the record's lines joined (only '\n' is stripped, as
    in the reference's line loop); what stands before the first header is dropped.  The file is cut at its header lines and a
    record's line ends are removed in one pass each — a 250-Mb draft is 4 M lines."""
    with open(path) as f:
        text = f.read()
    at = 1 if text.startswith(">") else text.find("\n>") + 2
    if at == 1 and not text.startswith(">"):      # (find gave -1: no header at all)
        return
    while True:
        nxt = text.find("\n>", at)
        rec = text[at:nxt] if nxt >= 0 else text[at:]
        head, _, body = rec.partition("\n")
        yield head.split()[0], body.replace("\n", "")
        if nxt < 0:
            return
        at = nxt + 2


_RUN = re.compile(r"N[^ACGT]*")   # a gap starts at an 'N' and runs to the next UPPER-case A/C/G/T


def scan_gaps(seq, min_gap):
    """[(start, end)]: first 'N' .. first upper-case ACGT after it (gnrt_pos_true_seqs.py:19-48); kept when
    end-start >= min_gap (:53); the search resumes at end+2 (:56); a run reaching the end of the sequence is dropped."""
    out, pos = [], 0
    while True:
        m = _RUN.search(seq, pos)
        if m is None or m.end() == len(seq):
            break
        if m.end() - m.start() >= min_gap:
            out.append((m.start(), m.end()))
        pos = m.end() + 2
    return out


def flanks(seq, start, end, flank_length):
    """gnrt_pos_true_seqs.py:94-99 (Python slice semantics included: start-5 may be negative)."""
    left = seq[0:start - 5] if start < flank_length else seq[start - flank_length:start - 5]
    return left, seq[end + 5:end + flank_length]


class DGProcessor:
    def __init__(self, ref_path, sf_pos):
        self.ref_path = ref_path
        self.sf_pos = sf_pos
        self._records = None      # (path, mtime, size) -> the draft's records: the two steps below read the same file (a 250-Mb draft: 0.3 s a pass)

    def _draft(self, path):
        st = os.stat(path)
        key = (os.path.abspath(path), st.st_mtime_ns, st.st_size)
        if self._records is None or self._records[0] != key:
            self._records = (key, list(read_fasta(path)))
        return self._records[1]

    def gnrt_gap_positions(self, min_gap_lenth):
        with open(self.sf_pos, "w") as out:
            for name, seq in self._draft(self.ref_path):
                for s, e in scan_gaps(seq, min_gap_lenth):
                    out.write("%d %d %d %s\n" % (s, e, e - s, name))

    def get_gap_flank_seqs(self, ref_path, sf_gap_pos, frame_length, sf_fai, working_folder):
        idx = {}
        with open(sf_fai) as f:
            for i, line in enumerate(l for l in f if l.strip()):
                idx[line.split()[0]] = i
        folder = working_folder + "flank_regions"
        os.makedirs(folder, exist_ok=True)
        gaps = {}
        with open(sf_gap_pos) as f:
            for line in f:
                fl = line.split()
                gaps.setdefault(fl[3], []).append((int(fl[0]), int(fl[1])))
        for name, seq in self._draft(ref_path):
            for num, (s, e) in enumerate(gaps.get(name, []), 1):
                gid = "%d_%d" % (idx[name], num)
                l, r = flanks(seq, s, e, frame_length)
                with open("%s/%s.fa" % (folder, gid), "w") as out:
                    out.write(">%s_left\n%s\n>%s_right\n%s\n" % (gid, l, gid, r))
        self._records = None      # (the draft is not needed again: the memory goes back)
