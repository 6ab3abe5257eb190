// ref_preprocess.cpp -- TEST INFRASTRUCTURE.  A 10-line driver around the
// REFERENCE's own, unmodified DataLoader (graph-server/graph/dataloader.cpp:225-330)
// compiled from the sources where they lie under /root/reference (see Makefile,
// target _ref/ref_preprocess).  No reference source is copied into this repo and
// no stand-in headers are used: graph/{graph,dataloader,vertex,edge}.cpp,
// utils/utils.cpp and common/utils.cpp build with the stock toolchain.
//
//   ref_preprocess <datasetDir/> <nodeId> <numNodes> <undirected 0|1>
// reads  <datasetDir/>graph.bsnap.edges + graph.bsnap.parts
// writes <datasetDir/>graph.<nodeId>.bin      (format: SURVEY.md A.4)
#include <cstdlib>
#include <string>
#include "graph-server/graph/dataloader.hpp"
int main(int argc, char **argv) {
    if (argc != 5) return 2;
    DataLoader dl(std::string(argv[1]), (unsigned)atoi(argv[2]),
                  (unsigned)atoi(argv[3]), atoi(argv[4]) != 0);
    dl.preprocess();
    return 0;
}
