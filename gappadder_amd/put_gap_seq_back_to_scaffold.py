"""Write the closed gap sequences back into the draft scaffolds (mirrors put_gap_seq_back_to_scaffold.py:8-95).

Host-only text work.  Reproduced as the reference does it, quirks included, and pinned by a reference-generated fixture
(tests/golden/twolib/writeback.json.gz):
  * a picked record `>{scaffoldIdx}_{gapIdx}_...` fills gap gapIdx (1-based, gap_positions.txt order) of scaffold scaffoldIdx
    (.fai order); a later record for the same gap replaces an earlier one (:11-20);
  * a filled gap [start, end) is replaced by the sequence and copying resumes at end + 1 — the base at `end` (the first non-N
    base after the run) is DROPPED (:88), mirroring the one right-flank base the picker keeps (pick_contigs.py slice);
  * an unfilled gap of a scaffold that has other filled gaps keeps its N run (:80-82); a scaffold with no filled gap, or no gap at
    all, is written unchanged, header description included (:63-75); rebuilt scaffolds get the bare id as header (:93);
  * FASTA out in Biopython's writer convention: 60 columns.
usage: python -m gappadder_amd.put_gap_seq_back_to_scaffold scaffolds.fa gap_positions.txt picked_seqs.fa new_scaffolds.fa"""
import sys


def _records(path):
    """(id, full header line without '>', sequence) per FASTA record."""
    out, name, desc, chunks = [], None, None, []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if name is not None:
                    out.append((name, desc, "".join(chunks)))
                name, desc, chunks = line[1:].split()[0], line[1:], []
            else:
                chunks.append(line)
    if name is not None:
        out.append((name, desc, "".join(chunks)))
    return out


def _wrap(title, seq):
    return ">" + title + "\n" + "".join(seq[i:i + 60] + "\n" for i in range(0, len(seq), 60))


def put_gap_seq_back_to_scaffold(sf_scf, sf_gap_pos, sf_gap_seq, sf_new_scf):
    gap_seq = {}
    for name, _, seq in _records(sf_gap_seq):
        f = name.split("_")
        gap_seq.setdefault(int(f[0]), {})[int(f[1])] = seq
    scaffold_id = {}
    with open(sf_scf + ".fai") as fin:
        for cnt, line in enumerate(fin):
            scaffold_id[line.split()[0]] = cnt
    gap_pos, cnt, pre_id = {}, 1, 0
    with open(sf_gap_pos) as fin:
        for line in fin:
            f = line.split()
            sid = scaffold_id[f[3]]
            if pre_id != sid:
                cnt = 1
            gap_pos.setdefault(sid, {})[cnt] = (int(f[0]), int(f[1]))
            cnt += 1
            pre_id = sid
    with open(sf_new_scf, "w") as out:
        for name, desc, seq in _records(sf_scf):
            sid = scaffold_id[name]
            if sid not in gap_pos or not gap_pos[sid] or sid not in gap_seq:
                out.write(_wrap(desc, seq))
                continue
            pre_pos, parts = 0, []
            for gid in range(1, len(gap_pos[sid]) + 1):
                if gid not in gap_seq[sid]:
                    print("Gap ", gid, "of Scaffold ", sid, "does not exist!!!")
                    continue
                start, end = gap_pos[sid][gid]
                parts.append(seq[pre_pos:start])
                parts.append(gap_seq[sid][gid])
                pre_pos = end + 1
            parts.append(seq[pre_pos:])
            out.write(_wrap(name, "".join(parts)))


if __name__ == "__main__":
    put_gap_seq_back_to_scaffold(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4])
