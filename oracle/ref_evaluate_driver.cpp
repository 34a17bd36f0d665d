// TEST INFRASTRUCTURE (oracle/): driver around the REFERENCE's own pairwise overlap evaluation
// (ContigsCompactor::Evaluate, /root/reference/ContigsCompactor-v0.2.0/ContigsMerger/ContigsCompactor.cpp:1572-1976, a private
// member: the header is included with `private` opened), compiled with the reference's sources where they lie by oracle/Makefile
// target `ref` into oracle/_ref/evaluate_kat (-O0).  Pins the oracle's restatement (tests/golden/evaluate_kat.json);
// never linked into the product.
//
// usage: evaluate_kat <contigs.fa> <mismatch> <indel> <max_clip> <frac_min_overlap> <frac_loss> <min_overlap> <min_overlap_scaffold>
//   -> one line per ordered pair (i, j), i != j, of the node list [c0, c0_R, c1, c1_R, ...]:
//      "i j res" and, when res != 0, " posEndSeq1 nclip overlap mergedLen containment"
#define private public
#include "ContigsCompactor.h"
#undef private
#include "fastaMultiSeqs.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    if (argc != 9) return 2;
    MultiFastqSeqs contigs;
    contigs.ReadFromFile(argv[1]);
    ContigsCompactor cc;
    cc.SetVerbose(false);
    cc.SetMismatchScore(atof(argv[2]));
    cc.SetIndelScore(atof(argv[3]));
    cc.SetMaxOverlapLenClip(atof(argv[4]));
    cc.SetMinOverlap(atof(argv[5]));
    cc.SetFractionLossScore(atof(argv[6]));
    cc.SetMinOverlapLen(atof(argv[7]));
    cc.SetMinOverlapLenWithScaffold(atof(argv[8]));
    std::vector<FastaSequence*> nodes;
    for (int i = 0; i < (int)contigs.GetNumOfSeqs(); ++i) {   // CompactVer3, ContigsCompactor.cpp:782-800
        FastaSequence* rc = new FastaSequence(*contigs.GetSeq(i));
        rc->RevsereComplement();
        nodes.push_back(contigs.GetSeq(i));
        nodes.push_back(rc);
    }
    for (size_t i = 0; i < nodes.size(); ++i)
        for (size_t j = 0; j < nodes.size(); ++j) {
            if (i == j) continue;
            ContigsCompactorAction act;
            const int res = cc.Evaluate(nodes[i], nodes[j], act);
            if (res == 0) printf("%zu %zu 0\n", i, j);
            else printf("%zu %zu %d %d %d %d %d %d\n", i, j, res, act.GetPosEndSeq1(), act.GetOneEndClipLenth(), act.GetOverlapSize(), act.GetMergedLen(),
                        act.IsContainment() ? 1 : 0);
        }
    return 0;
}
