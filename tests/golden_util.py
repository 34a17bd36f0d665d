"""Loaders for the committed golden fixtures (tests/golden/<case>/{inputs,expected.tar.gz})."""
import gzip
import io
import json
import os
import tarfile

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gz(path):
    with gzip.open(path, "rt") as f:
        return f.read()


class Case:
    def __init__(self, name):
        d = os.path.join(GOLDEN, name)
        self.name = name
        self.dir = d
        self.meta = json.load(open(os.path.join(d, "inputs", "meta.json")))
        self.fai = open(os.path.join(d, "inputs", "draft.fa.fai")).read()
        self.fai_names = [l.split()[0] for l in self.fai.splitlines()]
        self.libs = []
        if self.meta.get("regenerate"):
            # large inputs are a pure function of the seed (tests/golden/synth_text.py, this repo's own generator); the digest
            # recorded when the reference ran on them pins the regeneration
            import hashlib
            import sys
            sys.path.insert(0, GOLDEN)
            from synth_text import make_case
            gen = make_case(name, self.meta["seed"])
            digest = hashlib.sha256("".join([gen["draft_fa"]] + [l[x] for l in gen["libs"] for x in ("sam", "fq1", "fq2")]).encode()).hexdigest()
            assert digest == self.meta["inputs_sha256"], "regenerated inputs of %s differ from the ones the reference ran on" % name
            self.draft_fa = gen["draft_fa"]
            for lib in gen["libs"]:
                self.libs.append({"is": lib["is"], "sd": lib["sd"], "sam": lib["sam"], "fq1": lib["fq1"], "fq2": lib["fq2"],
                                  "folder": "%d_is%d" % (len(self.libs) + 1, lib["is"])})
        else:
            self.draft_fa = _gz(os.path.join(d, "inputs", "draft.fa.gz"))
            for i, lib in enumerate(self.meta["libs"]):
                self.libs.append({"is": lib["is"], "sd": lib["sd"],
                                  "sam": _gz(os.path.join(d, "inputs", "lib%d.sam.gz" % i)),
                                  "fq1": _gz(os.path.join(d, "inputs", "lib%d_1.fq.gz" % i)),
                                  "fq2": _gz(os.path.join(d, "inputs", "lib%d_2.fq.gz" % i)),
                                  "folder": "%d_is%d" % (i + 1, lib["is"])})
        self.expected = {}
        with gzip.open(os.path.join(d, "expected.tar.gz"), "rb") as g:
            with tarfile.open(fileobj=io.BytesIO(g.read())) as tf:
                for m in tf.getmembers():
                    if m.isfile():
                        self.expected[m.name] = tf.extractfile(m).read().decode()

    def fasta_records(self):
        recs, name, chunks = [], None, []
        for line in self.draft_fa.splitlines():
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(chunks)))
                name, chunks = line[1:].split()[0], []
            else:
                chunks.append(line)
        if name is not None:
            recs.append((name, "".join(chunks)))
        return recs

    def exp_lines(self, rel):
        return self.expected[rel].splitlines()

    def exp_dir(self, prefix):
        return {k[len(prefix):]: v for k, v in self.expected.items() if k.startswith(prefix)}


CASES = ["twolib", "edge", "bounds", "c1"]   # c1 = BASELINE.json configs[0]
