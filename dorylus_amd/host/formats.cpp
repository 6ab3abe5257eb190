// formats.cpp -- readers for the input files next to the hot path, compatible with the
// reference graph server (src/graph-server/engine/utils.cpp:460-596, engine.hpp:30-37):
// layer configuration (run/*.config), features.bsnap (+ the per-node cache
// feats<F>.<id>.bin), labels.bsnap.  Errors are status codes, never assert/exit.
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/dorylus_host.h"

static thread_local std::string g_fmt_err;
extern "C" const char *dory_formats_last_error(void) { return g_fmt_err.c_str(); }
static int ferr(int code, const std::string &m) {
    g_fmt_err = m;
    return code;
}

static std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}

extern "C" {

// Engine::readLayerConfigFile (engine/utils.cpp:460-479): one unsigned per non-empty line
int dory_read_layer_config(const char *path, uint32_t *dims, uint32_t max_dims, uint32_t *count) {
    if (!path || !dims || !count) return ferr(DORY_ERR_ARG, "read_layer_config: bad arguments");
    std::ifstream f(path);
    if (!f.good()) return ferr(DORY_ERR_IO, std::string("cannot open layer configuration file ") + path + ": " + std::strerror(errno));
    std::string line;
    uint32_t n = 0;
    while (std::getline(f, line)) {
        line = trim(line);
        if (line.empty()) continue;
        char *end = nullptr;
        unsigned long v = std::strtoul(line.c_str(), &end, 10);
        if (end == line.c_str() || v == 0) return ferr(DORY_ERR_IO, "layer configuration: bad line '" + line + "'");
        if (n >= max_dims) return ferr(DORY_ERR_ARG, "layer configuration: too many layers");
        dims[n++] = (uint32_t)v;
    }
    if (n < 2) return ferr(DORY_ERR_IO, "layer configuration needs at least two widths");
    *count = n;
    return DORY_OK;
}

// Engine::readFeaturesFile (engine/utils.cpp:486-552): u32 numFeatures, then one row of
// numFeatures floats per global vertex; rows of local vertices go to `local`
// (N x F, local-id order), rows of source ghosts to `ghost` (Gsrc x F, ghost-slot order).
// With cache_dir != NULL the reference's cache file <cache_dir>feats<F>.<node>.bin is
// used / created (local block followed by ghost block).
int dory_read_features(const char *path, const dory_partition *p, uint32_t expect_dim, uint32_t node_id,
                       const char *cache_dir, float *local, float *ghost) {
    if (!path || !p || !local) return ferr(DORY_ERR_ARG, "read_features: bad arguments");
    struct dory_partition_view v;
    dory_partition_get(p, &v);
    const size_t F = expect_dim;
    std::string cache;
    if (cache_dir) {
        cache = std::string(cache_dir) + "feats" + std::to_string(expect_dim) + "." + std::to_string(node_id) + ".bin";
        if (FILE *c = fopen(cache.c_str(), "rb")) {
            const size_t nl = (size_t)v.local_vtx_cnt * F, ng = (size_t)v.src_ghost_cnt * F;
            bool ok = (nl == 0 || fread(local, 4, nl, c) == nl) && (ng == 0 || (ghost && fread(ghost, 4, ng, c) == ng));
            char extra;
            ok = ok && fread(&extra, 1, 1, c) == 0;   // a cache of another partitioning with more rows is not this one's
            fclose(c);
            if (ok) return DORY_OK;
        }
    }
    FILE *f = fopen(path, "rb");
    if (!f) return ferr(DORY_ERR_IO, std::string("cannot open features file ") + path + ": " + std::strerror(errno));
    uint32_t nf = 0;
    if (fread(&nf, 4, 1, f) != 1 || nf != expect_dim) {
        fclose(f);
        return ferr(DORY_ERR_IO, "features header does not match layer 0 width");
    }
    // global id -> destination row (local or ghost), both lists ascend with global id
    std::vector<float> row(F);
    uint32_t li = 0, gi = 0, gvid = 0;
    while (fread(row.data(), 4, F, f) == F) {
        if (gi < v.src_ghost_cnt && v.src_ghosts[gi] == gvid) {
            if (ghost) memcpy(ghost + (size_t)gi * F, row.data(), 4 * F);
            ++gi;
        } else if (li < v.local_vtx_cnt && v.local_to_global[li] == gvid) {
            memcpy(local + (size_t)li * F, row.data(), 4 * F);
            ++li;
        }
        ++gvid;
    }
    fclose(f);
    if (gvid != v.global_vtx_cnt) return ferr(DORY_ERR_IO, "features file row count != globalVtxCnt");
    // the single forward scan relies on both id lists ascending (graph.cpp:34-60 writes them so): rows left
    // unfilled mean a partition file whose ghost / local ids are in another order -- refuse, do not cache zeros
    if (gi != v.src_ghost_cnt || li != v.local_vtx_cnt)
        return ferr(DORY_ERR_IO, "features: local or ghost global ids of the partition do not ascend (rows left unfilled)");
    if (!cache.empty()) {
        if (FILE *c = fopen(cache.c_str(), "wb")) {
            fwrite(local, 4, (size_t)v.local_vtx_cnt * F, c);
            if (ghost) fwrite(ghost, 4, (size_t)v.src_ghost_cnt * F, c);
            fclose(c);
        }
    }
    return DORY_OK;
}

// Engine::readLabelsFile (engine/utils.cpp:559-596): u32 labelKinds, then one u32 per
// global vertex; `labels` receives the class id of every local vertex (local-id order).
int dory_read_labels(const char *path, const dory_partition *p, uint32_t expect_kinds, uint32_t *labels) {
    if (!path || !p || !labels) return ferr(DORY_ERR_ARG, "read_labels: bad arguments");
    struct dory_partition_view v;
    dory_partition_get(p, &v);
    FILE *f = fopen(path, "rb");
    if (!f) return ferr(DORY_ERR_IO, std::string("cannot open labels file ") + path + ": " + std::strerror(errno));
    uint32_t kinds = 0;
    if (fread(&kinds, 4, 1, f) != 1 || kinds != expect_kinds) {
        fclose(f);
        return ferr(DORY_ERR_IO, "labels header does not match the last layer width");
    }
    std::vector<uint32_t> all(v.global_vtx_cnt);
    const size_t got = fread(all.data(), 4, v.global_vtx_cnt, f);
    uint32_t extra;
    const bool more = fread(&extra, 4, 1, f) == 1;
    fclose(f);
    if (got != v.global_vtx_cnt || more) return ferr(DORY_ERR_IO, "labels file entry count != globalVtxCnt");
    for (uint32_t l = 0; l < v.local_vtx_cnt; ++l) {
        const uint32_t c = all[v.local_to_global[l]];
        if (c >= kinds) return ferr(DORY_ERR_IO, "label out of range");
        labels[l] = c;
    }
    return DORY_OK;
}

}  // extern "C"
