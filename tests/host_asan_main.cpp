// host_asan_main.cpp -- the host side of the library (partition builder, graph.<id>.bin IO, halo plan, input file
// readers) under AddressSanitizer + UBSan, without a device: tests/test_host_sanitizers.py compiles
// dorylus_amd/host/{partition,formats}.cpp together with this driver and runs it.  The three device-library entry
// points partition.cpp refers to are never reached here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../include/dorylus_host.h"

extern "C" {
int dory_graph_upload(dory_ctx *, uint32_t, uint32_t, uint32_t, uint64_t, const uint64_t *, const uint32_t *, const float *,
                      uint64_t, const uint64_t *, const uint32_t *, const float *, const float *) { return DORY_ERR_NODEVICE; }
int dory_halo_plan(dory_ctx *, int, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *) { return DORY_ERR_NODEVICE; }
int dory_set_option(dory_ctx *, const char *, int64_t) { return DORY_ERR_NODEVICE; }
}

#define REQUIRE(c)                                                                  \
    do {                                                                            \
        if (!(c)) {                                                                 \
            fprintf(stderr, "REQUIRE failed: %s (%s:%d) [%s]\n", #c, __FILE__, __LINE__, dory_host_last_error()); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

template <typename T>
static bool same(const T *a, const T *b, size_t n) { return n == 0 || memcmp(a, b, n * sizeof(T)) == 0; }

int main(int argc, char **argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    std::mt19937 rng(7);
    for (int undirected = 0; undirected < 2; ++undirected) {
        for (uint32_t P : {1u, 3u, 8u}) {
            const uint32_t V = 400 + 37 * P;
            const uint64_t E = 6000;
            std::vector<uint32_t> s(E), d(E);
            std::vector<int32_t> parts(V);
            for (auto &x : s) x = rng() % V;
            for (auto &x : d) x = rng() % V;
            for (uint64_t i = 0; i < 50; ++i) d[i] = s[i];                       // self loops are dropped
            for (uint32_t v = 0; v < V; ++v) parts[v] = (int32_t)(rng() % P);
            if (P == 3) for (auto &p : parts) if (p == 2) p = 0;                 // an empty partition
            for (uint32_t r = 0; r < P; ++r) {
                dory_partition *p = nullptr, *q = nullptr;
                REQUIRE(dory_partition_build(s.data(), d.data(), E, parts.data(), V, r, P, undirected, &p) == 0);
                const std::string path = tmp + "/asan_graph." + std::to_string(r) + ".bin";
                REQUIRE(dory_partition_save(p, path.c_str()) == 0);
                REQUIRE(dory_partition_load(path.c_str(), &q) == 0);
                dory_partition_view a, b;
                REQUIRE(dory_partition_get(p, &a) == 0 && dory_partition_get(q, &b) == 0);
                REQUIRE(a.local_vtx_cnt == b.local_vtx_cnt && a.src_ghost_cnt == b.src_ghost_cnt && a.dst_ghost_cnt == b.dst_ghost_cnt);
                REQUIRE(a.local_in_edge_cnt == b.local_in_edge_cnt && a.local_out_edge_cnt == b.local_out_edge_cnt);
                const uint32_t N = a.local_vtx_cnt;
                REQUIRE(same(a.local_to_global, b.local_to_global, N) && same(a.norms, b.norms, N));
                REQUIRE(same(a.column_ptrs, b.column_ptrs, (size_t)N + 1) && same(a.row_ptrs, b.row_ptrs, (size_t)N + 1));
                REQUIRE(same(a.row_idxs, b.row_idxs, a.local_in_edge_cnt) && same(a.csc_values, b.csc_values, a.local_in_edge_cnt));
                REQUIRE(same(a.column_idxs, b.column_idxs, a.local_out_edge_cnt) && same(a.csr_values, b.csr_values, a.local_out_edge_cnt));
                REQUIRE(same(a.src_ghosts, b.src_ghosts, a.src_ghost_cnt) && same(a.dst_ghosts, b.dst_ghosts, a.dst_ghost_cnt));
                for (uint64_t e = 0; e < a.local_in_edge_cnt; ++e) REQUIRE(a.row_idxs[e] < N + a.src_ghost_cnt);
                for (uint64_t e = 0; e < a.local_out_edge_cnt; ++e) REQUIRE(a.column_idxs[e] < N + a.dst_ghost_cnt);
                for (int dir = 0; dir < 2; ++dir) {                               // receive plan: a permutation of the ghost slots
                    const uint32_t G = dir == 0 ? a.src_ghost_cnt : a.dst_ghost_cnt;
                    std::vector<uint32_t> cnt(P), slots(G + 1);
                    REQUIRE(dory_partition_recv_plan(p, parts.data(), dir, cnt.data(), slots.data()) == 0);
                    std::vector<char> seen(G, 0);
                    uint32_t total = 0;
                    for (uint32_t x : cnt) total += x;
                    REQUIRE(total == G && cnt[r] == 0);
                    for (uint32_t i = 0; i < G; ++i) {
                        REQUIRE(slots[i] < G && !seen[slots[i]]);
                        seen[slots[i]] = 1;
                    }
                }
                REQUIRE(dory_partition_upload(nullptr, p, parts.data()) != 0);   // no context: refused, nothing dereferenced
                // input files of the graph server
                const uint32_t F = 5, C = 3;
                const std::string fpath = tmp + "/asan_features.bsnap", lpath = tmp + "/asan_labels.bsnap", cpath = tmp + "/asan_layers.config";
                {
                    FILE *f = fopen(fpath.c_str(), "wb");
                    REQUIRE(f);
                    fwrite(&F, 4, 1, f);
                    for (uint32_t v = 0; v < V; ++v)
                        for (uint32_t c = 0; c < F; ++c) { float x = (float)(v * F + c); fwrite(&x, 4, 1, f); }
                    fclose(f);
                    f = fopen(lpath.c_str(), "wb");
                    REQUIRE(f);
                    fwrite(&C, 4, 1, f);
                    for (uint32_t v = 0; v < V; ++v) { uint32_t l = v % C; fwrite(&l, 4, 1, f); }
                    fclose(f);
                    f = fopen(cpath.c_str(), "w");
                    REQUIRE(f);
                    fprintf(f, "%u\n4\n%u\n", F, C);
                    fclose(f);
                }
                uint32_t dims[8], nd = 0;
                REQUIRE(dory_read_layer_config(cpath.c_str(), dims, 8, &nd) == 0 && nd == 3 && dims[0] == F && dims[2] == C);
                REQUIRE(dory_read_layer_config(cpath.c_str(), dims, 2, &nd) != 0);          // too many layers for the caller's array
                std::vector<float> loc((size_t)N * F + 1), gh((size_t)a.src_ghost_cnt * F + 1);
                REQUIRE(dory_read_features(fpath.c_str(), p, F, r, nullptr, loc.data(), gh.data()) == 0);
                for (uint32_t i = 0; i < N; ++i) REQUIRE(loc[(size_t)i * F + 2] == (float)(a.local_to_global[i] * F + 2));
                for (uint32_t i = 0; i < a.src_ghost_cnt; ++i) REQUIRE(gh[(size_t)i * F] == (float)(a.src_ghosts[i] * F));
                REQUIRE(dory_read_features(fpath.c_str(), p, F + 1, r, nullptr, loc.data(), gh.data()) != 0);   // wrong width
                std::vector<uint32_t> lab(N + 1);
                REQUIRE(dory_read_labels(lpath.c_str(), p, C, lab.data()) == 0);
                for (uint32_t i = 0; i < N; ++i) REQUIRE(lab[i] == a.local_to_global[i] % C);
                REQUIRE(dory_partition_free(p) == 0 && dory_partition_free(q) == 0);
                remove(path.c_str());
            }
        }
    }
    dory_partition *bad = nullptr;
    REQUIRE(dory_partition_load("/nonexistent/graph.0.bin", &bad) != 0 && bad == nullptr);
    printf("host sanitizer run ok\n");
    return 0;
}
