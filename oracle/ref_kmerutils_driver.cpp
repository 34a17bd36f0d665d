// TEST INFRASTRUCTURE (oracle/): driver around the REFERENCE's own KmerUtils.cpp
// (/root/reference/ContigsCompactor-v0.2.0/ContigsMerger/KmerUtils.cpp), compiled where it lies by
// oracle/Makefile target `ref` into oracle/_ref/kmerutils_kat.  Used only to pin the oracle's 2-bit
// k-mer layout (tests/golden/kmerutils_kat.json); never linked into the product.
#include "KmerUtils.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::string cmd = argv[1];
    if (cmd == "kmers" && argc == 4) {            // kmers <seq> <k> : GetAllKmersFromSeq (KmerUtils.cpp:90-115)
        std::vector<KmerTypeShort> v;
        GetAllKmersFromSeq(argv[2], (int)strlen(argv[2]), atoi(argv[3]), v);
        for (size_t i = 0; i < v.size(); ++i) printf("%016llx\n", (unsigned long long)v[i]);
        return 0;
    }
    if (cmd == "tostr" && argc == 4) {            // tostr <hex> <k> : ConvKmerToString (KmerUtils.cpp:127-169)
        KmerTypeShort km = strtoull(argv[2], 0, 16);
        char buf[80];
        ConvKmerToString(km, atoi(argv[3]), buf);
        printf("%s\n", buf);
        return 0;
    }
    if (cmd == "pred" && argc == 6) {             // pred <src> <read> <k> <thr> : IsReadContainingFreqKmers (:215-241)
        int k = atoi(argv[4]);
        std::vector<KmerTypeShort> v;
        GetAllKmersFromSeq(argv[2], (int)strlen(argv[2]), k, v);
        MapShortKmerFreq m;
        for (size_t i = 0; i < v.size(); ++i) AddShortKmerToHashMap(v[i], m, 1.0);
        printf("%d\n", IsReadContainingFreqKmers(argv[3], (int)strlen(argv[3]), k, atoi(argv[5]), m) ? 1 : 0);
        return 0;
    }
    return 2;
}
