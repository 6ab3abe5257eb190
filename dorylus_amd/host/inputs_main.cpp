// inputs_main.cpp -- the reference's offline data-prep tools (inputs/, driven by
// inputs/prepare:1-87) as one multi-call binary: the first argument -- or the name the
// binary is invoked under -- selects
//   graphtobinary    [file] --snapfile=<text edge list> --undirected=<0/1> --header=<0/1>   -> <file>.bsnap
//   featurestobinary --featuresfile=<text> --featuredimension=<F>                          -> <file>.bsnap
//   labelstobinary   --labelsfile=<text> --labelkinds=<K>                                  -> <file>.bsnap
//   partitioner      <GraphBsnapFile> <NumVertices> <NumPartitions> [--method=block|hash|bfs|ldg]
//                                            -> parts_<P>/<graph>.parts (+ .comm with the edge cut)
// Same command lines, file names and byte formats as inputs/graphToBinary.cpp:47-160,
// featuresToBinary.cpp:31-99, labelsToBinary.cpp:31-92, partitioner.cpp:33-135.  The reference
// partitions with METIS 5.1.0 (third-party, absent here): `.parts` is just
// an input of the hot path, so this tool offers deterministic METIS-free methods instead --
// block (contiguous, balanced by vertex count; default), hash, bfs (breadth-first regions), ldg (restreamed linear
// deterministic greedy: each vertex joins the partition that already holds most of its neighbours, damped by how full
// that partition is; --passes=N restreaming passes, ten by default; balanced within 5 % -- the one to use when the graph has
// community structure: on the Amazon-size 50-community graph with shuffled ids it cuts 14.9 % of the edges where contiguous
// blocks cut 87.5 % and the generator's own community order 17.4 %).
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <queue>
#include <sstream>
#include <string>
#include <vector>

struct BSHeader {  // graph/dataloader.hpp:11-15
    int sizeOfVertexType;
    unsigned numVertices;
    unsigned long long numEdges;
};

static std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    return s.substr(a, s.find_last_not_of(" \t\r\n") - a + 1);
}

static int graphtobinary(int argc, char **argv) {
    std::string snap;
    bool undirected = false, withheader = false;
    for (int i = 0; i < argc; ++i) {
        if (!strncmp("--snapfile=", argv[i], 11)) snap = argv[i] + 11;
        if (!strncmp("--undirected=", argv[i], 13)) undirected = atoi(argv[i] + 13) != 0;
        if (!strncmp("--header=", argv[i], 9)) withheader = atoi(argv[i] + 9) != 0;
    }
    if (snap.empty()) {
        std::cout << "Usage: graphtobinary --snapfile=<SnapFile> --undirected=<0/1> --header=<0/1>" << std::endl;
        return -1;
    }
    BSHeader h{(int)sizeof(unsigned), 0, 0};
    auto each_edge = [&](const std::function<void(unsigned, unsigned)> &fn) -> bool {
        std::ifstream in(snap);
        if (!in.good()) {
            fprintf(stderr, "Cannot open graph snap file: %s [Reason: %s]\n", snap.c_str(), strerror(errno));
            return false;
        }
        std::string line;
        while (std::getline(in, line)) {
            if (!line.empty() && (line[0] == '#' || line[0] == '%')) continue;
            std::istringstream iss(line);
            unsigned src, dst;
            if (!(iss >> src >> dst)) break;       // graphToBinary.cpp:45,88: stops at the first malformed line
            if (src == dst) continue;              // self edges are removed (:47,91)
            fn(src, dst);
        }
        return true;
    };
    if (withheader) {
        bool any = false;
        if (!each_edge([&](unsigned s, unsigned d) {
                h.numVertices = std::max(h.numVertices, std::max(s, d));
                ++h.numEdges;
                any = true;
            }))
            return 1;
        (void)any;
        ++h.numVertices;
        if (undirected) h.numEdges *= 2;
        std::cout << "Graph info - Vertices: " << h.numVertices << ", Edges: " << h.numEdges << std::endl;
    }
    std::ofstream out(snap + ".bsnap", std::ios::binary);
    if (withheader) out.write((const char *)&h, sizeof(h));
    if (!each_edge([&](unsigned s, unsigned d) {
            out.write((const char *)&s, 4);
            out.write((const char *)&d, 4);
            if (undirected) {
                out.write((const char *)&d, 4);
                out.write((const char *)&s, 4);
            }
        }))
        return 1;
    return 0;
}

static int featurestobinary(int argc, char **argv) {
    std::string file;
    unsigned dim = 0;
    for (int i = 0; i < argc; ++i) {
        if (!strncmp("--featuresfile=", argv[i], 15)) file = argv[i] + 15;
        if (!strncmp("--featuredimension=", argv[i], 19)) dim = (unsigned)atoi(argv[i] + 19);
    }
    if (file.empty() || dim == 0) {
        std::cout << "Usage: featurestobinary --featuresfile=<FeatureFile> --featuredimension=<FeatureDimension>" << std::endl;
        return -1;
    }
    std::ifstream in(file);
    if (!in.good()) {
        fprintf(stderr, "Cannot open feature file: %s [Reason: %s]\n", file.c_str(), strerror(errno));
        return 1;
    }
    std::ofstream out(file + ".bsnap", std::ios::binary);
    out.write((const char *)&dim, 4);
    std::string line;
    unsigned long long row = 0;
    while (std::getline(in, line)) {
        line = trim(line);
        // featuresToBinary.cpp:52-53: only lines that start with a digit are rows
        if (line.empty() || line[0] < '0' || line[0] > '9') continue;
        unsigned n = 0;
        size_t p = 0;
        while (p < line.size()) {                 // split on any run of ',' / ' ' (token_compress_on)
            size_t q = line.find_first_of(", ", p);
            if (q == std::string::npos) q = line.size();
            if (q > p) {
                float f = std::stof(line.substr(p, q - p));
                out.write((const char *)&f, 4);
                ++n;
            }
            p = q + 1;
        }
        if (n != dim) {
            fprintf(stderr, "features row %llu has %u values, expected %u\n", row, n, dim);
            return 1;
        }
        ++row;
    }
    return 0;
}

static int labelstobinary(int argc, char **argv) {
    std::string file;
    unsigned kinds = 0;
    for (int i = 0; i < argc; ++i) {
        if (!strncmp("--labelsfile=", argv[i], 13)) file = argv[i] + 13;
        if (!strncmp("--labelkinds=", argv[i], 13)) kinds = (unsigned)atoi(argv[i] + 13);
    }
    if (file.empty() || kinds == 0) {
        std::cout << "Usage: labelstobinary --labelsfile=<LabelFile> --labelkinds=<LabelKinds>" << std::endl;
        return -1;
    }
    std::ifstream in(file);
    if (!in.good()) {
        fprintf(stderr, "Cannot open labels file: %s [Reason: %s]\n", file.c_str(), strerror(errno));
        return 1;
    }
    std::ofstream out(file + ".bsnap", std::ios::binary);
    out.write((const char *)&kinds, 4);
    std::string line;
    while (std::getline(in, line)) {
        line = trim(line);
        if (line.empty() || line[0] < '0' || line[0] > '9') continue;   // labelsToBinary.cpp:49-50
        unsigned l = (unsigned)std::stoul(line);
        out.write((const char *)&l, 4);
    }
    return 0;
}

static int partitioner(int argc, char **argv) {
    std::vector<std::string> pos;
    std::string method = "block";
    int passes = 10;   // (three passes leave half of the edges of a 50-community graph cut: 0.49; ten: 0.149 -- profiles/r06_partition_quality_amazon_community.json)
    for (int i = 1; i < argc; ++i) {
        if (!strncmp("--method=", argv[i], 9)) method = argv[i] + 9;
        else if (!strncmp("--passes=", argv[i], 9)) passes = std::max(1, atoi(argv[i] + 9));   // ldg: restreaming passes
        else pos.push_back(argv[i]);
    }
    if (pos.size() != 3) {
        std::cout << "Usage: partitioner <GraphBsnapFile> <NumVertices> <NumPartitions> [--method=block|hash|bfs|ldg] [--passes=N]" << std::endl;
        return -1;
    }
    const std::string graph = pos[0];
    const unsigned P = (unsigned)atoi(pos[2].c_str());
    if (P == 0 || P > 32767) {   // ids are parsed as short by the graph server (dataloader.cpp:61)
        fprintf(stderr, "bad partition count\n");
        return 1;
    }
    FILE *f = fopen(graph.c_str(), "rb");
    if (!f) {
        fprintf(stderr, "Cannot open graph bsnap file: %s [Reason: %s]\n", graph.c_str(), strerror(errno));
        return 1;
    }
    BSHeader h;
    if (fread(&h, sizeof(h), 1, f) != 1 || h.sizeOfVertexType != 4) {
        fclose(f);
        fprintf(stderr, "bad bsnap header\n");
        return 1;
    }
    const unsigned V = h.numVertices;
    std::vector<unsigned> e;
    {
        unsigned buf[2];
        while (fread(buf, 4, 2, f) == 2) {
            e.push_back(buf[0]);
            e.push_back(buf[1]);
        }
    }
    fclose(f);
    std::cout << "Number of vertices: " << V << std::endl << "Number of edges: " << e.size() / 2 << std::endl;
    std::vector<int> parts(V, 0);
    if (method == "hash") {
        for (unsigned v = 0; v < V; ++v) parts[v] = (int)((v * 2654435761ull % 4294967296ull) % P);
    } else if (method == "bfs") {
        // breadth-first order from vertex 0 (restarting at the lowest unvisited id), cut into P equal chunks
        std::vector<unsigned long long> ptr(V + 1, 0);
        for (size_t i = 0; i < e.size(); i += 2) { ++ptr[e[i] + 1]; ++ptr[e[i + 1] + 1]; }
        for (unsigned v = 0; v < V; ++v) ptr[v + 1] += ptr[v];
        std::vector<unsigned> adj(ptr[V]);
        std::vector<unsigned long long> cur(ptr.begin(), ptr.end() - 1);
        for (size_t i = 0; i < e.size(); i += 2) { adj[cur[e[i]]++] = e[i + 1]; adj[cur[e[i + 1]]++] = e[i]; }
        std::vector<char> seen(V, 0);
        std::vector<unsigned> order;
        order.reserve(V);
        for (unsigned s = 0; s < V; ++s) {
            if (seen[s]) continue;
            std::queue<unsigned> q;
            q.push(s);
            seen[s] = 1;
            while (!q.empty()) {
                unsigned u = q.front();
                q.pop();
                order.push_back(u);
                for (unsigned long long k = ptr[u]; k < ptr[u + 1]; ++k)
                    if (!seen[adj[k]]) { seen[adj[k]] = 1; q.push(adj[k]); }
            }
        }
        for (unsigned i = 0; i < V; ++i) parts[order[i]] = (int)((unsigned long long)i * P / V);
    } else if (method == "ldg") {
        // Linear deterministic greedy (Stanton & Kliot), restreamed: pass 0 sees only the vertices placed so far, later
        // passes see everyone's previous placement.  Undirected view of the graph, vertices in id order, ties -> lowest id.
        std::vector<unsigned long long> ptr(V + 1, 0);
        for (size_t i = 0; i < e.size(); i += 2)
            if (e[i] < V && e[i + 1] < V) { ++ptr[e[i] + 1]; ++ptr[e[i + 1] + 1]; }
        for (unsigned v = 0; v < V; ++v) ptr[v + 1] += ptr[v];
        std::vector<unsigned> adj(ptr[V]);
        std::vector<unsigned long long> cur(ptr.begin(), ptr.end() - 1);
        for (size_t i = 0; i < e.size(); i += 2)
            if (e[i] < V && e[i + 1] < V) { adj[cur[e[i]]++] = e[i + 1]; adj[cur[e[i + 1]]++] = e[i]; }
        const double cap = (double)V / P * 1.05 + 1.0;
        std::vector<int> where(V, -1);
        std::vector<unsigned> size(P, 0);
        std::vector<unsigned> hits(P, 0);
        for (int pass = 0; pass < passes; ++pass) {
            for (unsigned v = 0; v < V; ++v) {
                if (where[v] >= 0) --size[where[v]];
                std::fill(hits.begin(), hits.end(), 0u);
                for (unsigned long long k = ptr[v]; k < ptr[v + 1]; ++k)
                    if (where[adj[k]] >= 0 && adj[k] != v) ++hits[where[adj[k]]];
                int best = -1;
                double best_score = -1.0;
                for (unsigned q = 0; q < P; ++q) {
                    if ((double)size[q] + 1.0 > cap) continue;
                    const double score = ((double)hits[q] + 1e-3) * (1.0 - (double)size[q] / cap);   // 1e-3: empty partitions fill evenly
                    if (score > best_score) { best_score = score; best = (int)q; }
                }
                if (best < 0) best = (int)(std::min_element(size.begin(), size.end()) - size.begin());
                where[v] = best;
                ++size[best];
            }
        }
        parts = where;
    } else if (method == "block") {
        for (unsigned v = 0; v < V; ++v) parts[v] = (int)((unsigned long long)v * P / V);
    } else {
        fprintf(stderr, "unknown method %s\n", method.c_str());
        return 1;
    }
    unsigned long long cut = 0;
    for (size_t i = 0; i < e.size(); i += 2)
        if (e[i] < V && e[i + 1] < V && parts[e[i]] != parts[e[i + 1]]) ++cut;
    const std::string dir = "./parts_" + pos[2] + "/";
    mkdir(dir.c_str(), 0777);
    std::string base = graph.substr(graph.find_last_of('/') == std::string::npos ? 0 : graph.find_last_of('/') + 1);
    {
        std::ofstream c(dir + base + ".comm");
        c << "Communication cost: " << cut << std::endl;
    }
    std::ofstream pf(dir + base + ".parts");
    for (unsigned v = 0; v < V; ++v) pf << parts[v] << "\n";
    std::cout << "Writing partitioning results... edge cut " << cut << std::endl;
    return 0;
}

int main(int argc, char **argv) {
    std::string name = argv[0];
    name = name.substr(name.find_last_of('/') == std::string::npos ? 0 : name.find_last_of('/') + 1);
    int shift = 0;
    const char *tools[] = {"graphtobinary", "featurestobinary", "labelstobinary", "partitioner"};
    bool known = false;
    for (auto t : tools) known |= name == t;
    if (!known && argc > 1) {
        name = argv[1];
        shift = 1;
    }
    argc -= shift;
    argv += shift;
    if (name == "graphtobinary") return graphtobinary(argc, argv);
    if (name == "featurestobinary") return featurestobinary(argc, argv);
    if (name == "labelstobinary") return labelstobinary(argc, argv);
    if (name == "partitioner") return partitioner(argc, argv);
    fprintf(stderr, "usage: dory-inputs {graphtobinary|featurestobinary|labelstobinary|partitioner} ...\n");
    return 2;
}
