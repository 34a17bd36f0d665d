// TEST INFRASTRUCTURE (oracle/): driver around the REFERENCE's own contig-merging prefilter
// (QuickCheckerContigsMatch, /root/reference/ContigsCompactor-v0.2.0/ContigsMerger/ContigsCompactor.cpp:1982-2095), compiled with
// the reference's sources where they lie by oracle/Makefile target `ref` into oracle/_ref/quickcheck_kat (-O0: the tree has
// missing-return UB that breaks at -O1+).  Pins the oracle's restatement of the all-pairs 10-mer check
// (tests/golden/quickcheck_kat.json); never linked into the product.
//
// usage: quickcheck_kat <contigs.fa> <k>   ->  one line "i j" per feasible pair of the node list [c0, c0_R, c1, c1_R, ...],
// i <= j, exactly the pairs MultiThreadQuickChecker::threadQuickCheck visits (ContigsCompactor.cpp:1073-1098), in (i, j) order.
#include "ContigsCompactor.h"
#include "fastaMultiSeqs.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    MultiFastqSeqs contigs;
    contigs.ReadFromFile(argv[1]);
    const int k = atoi(argv[2]);
    std::vector<FastaSequence*> nodes;
    for (int i = 0; i < (int)contigs.GetNumOfSeqs(); ++i) {   // CompactVer3, ContigsCompactor.cpp:782-800
        FastaSequence* rc = new FastaSequence(*contigs.GetSeq(i));
        rc->RevsereComplement();
        nodes.push_back(contigs.GetSeq(i));
        nodes.push_back(rc);
    }
    for (size_t i = 0; i < nodes.size(); ++i) {
        QuickCheckerContigsMatch qc(nodes[i], k);
        for (size_t j = i; j < nodes.size(); ++j)
            if (qc.IsMatchFeasible(nodes[j])) printf("%zu %zu\n", i, j);
    }
    return 0;
}
