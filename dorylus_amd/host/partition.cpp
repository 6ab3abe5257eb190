// partition.cpp -- host-side partition builder and graph.<id>.bin reader/writer.
//
// Index-identical re-design of the reference's DataLoader::preprocess
// (src/graph-server/graph/dataloader.cpp:53-330): the reference walks every edge
// through std::map lookups and per-vertex edge vectors (~3.5 us/edge); this is a
// two-pass counting build over flat arrays (count, prefix-sum, fill), O(E + V),
// which keeps record order inside each column/row and therefore reproduces the
// reference's CSC/CSR arrays byte for byte (tests/test_partition_builder.py
// compares against graph.<id>.bin files written by the reference's own code).
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dorylus_host.h"

struct dory_partition {
    uint32_t N = 0, V = 0, Gsrc = 0, Gdst = 0, P = 0;
    uint64_t nin = 0, nout = 0, nglobal = 0;
    std::vector<uint32_t> l2g;
    std::vector<float> norm;
    std::vector<uint32_t> srcGhost, dstGhost;
    std::vector<uint32_t> fwdCnt, fwdList, bwdCnt, bwdList;
    std::vector<uint64_t> colPtr, rowPtr;
    std::vector<uint32_t> rowIdx, colIdx;
    std::vector<float> cscVal, csrVal;
    // 0: built from directed records -> csrVal is exactly cscVal transposed (across partitions too);
    // 1: built with undirected = 1 -> ghost norms count file records only (dataloader.cpp:192-218), so the
    //    values of one edge differ between its two owners;  -1: read from graph.<id>.bin, unknown
    int undirected = -1;
};

static thread_local std::string g_err;
static int herr(int code, const std::string &m) {
    g_err = m;
    return code;
}

// float vtxNorm = std::pow(deg, -.5)  (dataloader.cpp:155-156): double pow, narrowed
static inline float inv_sqrt_deg(uint64_t deg) { return (float)std::pow((double)deg, -.5); }

extern "C" {

const char *dory_host_last_error(void) { return g_err.c_str(); }

int dory_partition_build(const uint32_t *src, const uint32_t *dst, uint64_t nrec, const int32_t *parts,
                         uint32_t V, uint32_t me, uint32_t P, int undirected, dory_partition **out) {
    if (!out || !parts || (nrec && (!src || !dst)) || P == 0 || me >= P)
        return herr(DORY_ERR_ARG, "partition_build: bad arguments");
    for (uint64_t i = 0; i < nrec; ++i)
        if (src[i] >= V || dst[i] >= V) return herr(DORY_ERR_ARG, "partition_build: vertex id out of range");
    std::unique_ptr<dory_partition> pp(new dory_partition());
    dory_partition &g = *pp;
    g.V = V;
    g.P = P;
    g.undirected = undirected ? 1 : 0;

    // readPartsFile (dataloader.cpp:53-87): local ids ascend with global id
    const uint32_t NONE = 0xFFFFFFFFu;
    std::vector<uint32_t> g2l(V, NONE);
    for (uint32_t v = 0; v < V; ++v) {
        if (parts[v] < 0 || (uint32_t)parts[v] >= P) return herr(DORY_ERR_ARG, "partition_build: partition id out of range");
        if ((uint32_t)parts[v] == me) {
            g2l[v] = (uint32_t)g.l2g.size();
            g.l2g.push_back(v);
        }
    }
    const uint32_t N = g.N = (uint32_t)g.l2g.size();

    // The three passes over the edge records are parallel over *vertex ownership*: thread t
    // owns the global ids [t*V/T, (t+1)*V/T), scans all records in file order and applies only
    // the updates that belong to vertices it owns -- no atomics, and the record order inside
    // every column/row (the reference's edge-file order) is preserved by construction.
    unsigned T = std::thread::hardware_concurrency();
    if (T > 32) T = 32;
    if ((uint64_t)nrec < (1u << 20)) T = 1;                       // not worth the threads
    if (const char *e = getenv("DORY_BUILD_THREADS")) T = (unsigned)atoi(e);   // explicit choice wins
    if (T == 0) T = 1;
    if (T > 256) T = 256;
    auto owner_lo = [&](unsigned t) { return (uint32_t)((uint64_t)V * t / T); };
    auto run_parallel = [&](const std::function<void(unsigned, uint32_t, uint32_t)> &fn) {
        if (T == 1) { fn(0, 0, V); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back(fn, t, owner_lo(t), owner_lo(t + 1));
        for (auto &x : th) x.join();
    };

    // pass 1: degrees, ghost membership, per-peer destination tables
    std::vector<uint64_t> inCnt(N, 0), outCnt(N, 0);
    std::vector<uint32_t> indegFile(V, 0);             // findGhostDegrees: file records only
    std::vector<uint8_t> isInGhost(V, 0), isOutGhost(V, 0);
    std::vector<std::vector<uint8_t>> fwdTab(P), bwdTab(P);
    for (uint32_t p = 0; p < P; ++p)
        if (p != me) {
            fwdTab[p].assign(N, 0);
            bwdTab[p].assign(N, 0);
        }
    std::vector<uint64_t> tin(T, 0), tout(T, 0), tglob(T, 0);
    run_parallel([&](unsigned t, uint32_t lo, uint32_t hi) {
        uint64_t nin = 0, nout = 0, nglob = 0;
        auto own = [&](uint32_t v) { return v >= lo && v < hi; };
        auto count_edge = [&](uint32_t from, uint32_t to) {  // processEdge (dataloader.cpp:94-146)
            const bool of = own(from), ot = own(to);
            if (!of && !ot) return;                // ownership first: no table lookups for foreign records
            const uint32_t pf = (uint32_t)parts[from], pt = (uint32_t)parts[to];
            if (of) {                              // updates keyed by the source vertex
                if (pf == me) {
                    const uint32_t lf = g2l[from];
                    ++outCnt[lf];
                    ++nout;
                    if (pt != me) fwdTab[pt][lf] = 1;
                }
                if (pt == me && pf != me) isInGhost[from] = 1;
            }
            if (ot) {                              // updates keyed by the destination vertex
                if (pt == me) {
                    const uint32_t lt = g2l[to];
                    ++inCnt[lt];
                    ++nin;
                    if (pf != me) bwdTab[pf][lt] = 1;
                }
                if (pf == me && pt != me) isOutGhost[to] = 1;
            }
        };
        for (uint64_t i = 0; i < nrec; ++i) {
            const uint32_t s = src[i], d = dst[i];
            if (s == d) continue;                            // dataloader.cpp:268-269
            count_edge(s, d);
            if (undirected) count_edge(d, s);
            if (t == 0) ++nglob;
            if (own(d)) ++indegFile[d];                      // dataloader.cpp:204-214 (dst occurrences)
        }
        tin[t] = nin; tout[t] = nout; tglob[t] = nglob;
    });
    for (unsigned t = 0; t < T; ++t) { g.nin += tin[t]; g.nout += tout[t]; g.nglobal += tglob[t]; }

    // ghost ranks in ascending global id (std::map order, dataloader.cpp:311-322)
    std::vector<uint32_t> ghostRankIn(V, NONE), ghostRankOut(V, NONE);
    for (uint32_t v = 0; v < V; ++v) {
        if (isInGhost[v]) {
            ghostRankIn[v] = (uint32_t)g.srcGhost.size();
            g.srcGhost.push_back(v);
        }
        if (isOutGhost[v]) {
            ghostRankOut[v] = (uint32_t)g.dstGhost.size();
            g.dstGhost.push_back(v);
        }
    }
    g.Gsrc = (uint32_t)g.srcGhost.size();
    g.Gdst = (uint32_t)g.dstGhost.size();

    // send lists (dataloader.cpp:277-297)
    g.fwdCnt.assign(P, 0);
    g.bwdCnt.assign(P, 0);
    for (uint32_t p = 0; p < P; ++p) {
        if (p == me) continue;
        for (uint32_t j = 0; j < N; ++j)
            if (fwdTab[p][j]) { g.fwdList.push_back(j); ++g.fwdCnt[p]; }
    }
    for (uint32_t p = 0; p < P; ++p) {
        if (p == me) continue;
        for (uint32_t j = 0; j < N; ++j)
            if (bwdTab[p][j]) { g.bwdList.push_back(j); ++g.bwdCnt[p]; }
    }
    fwdTab.clear();
    bwdTab.clear();

    // norms (setEdgeNormalizations, dataloader.cpp:153-185)
    std::vector<float> vnorm(N);
    g.norm.resize(N);
    for (uint32_t v = 0; v < N; ++v) {
        vnorm[v] = inv_sqrt_deg(inCnt[v] + 1);
        g.norm[v] = vnorm[v] * vnorm[v];
    }

    // prefix sums
    g.colPtr.assign((size_t)N + 1, 0);
    g.rowPtr.assign((size_t)N + 1, 0);
    for (uint32_t v = 0; v < N; ++v) {
        g.colPtr[v + 1] = g.colPtr[v] + inCnt[v];
        g.rowPtr[v + 1] = g.rowPtr[v] + outCnt[v];
    }
    g.rowIdx.resize(g.nin);
    g.cscVal.resize(g.nin);
    g.colIdx.resize(g.nout);
    g.csrVal.resize(g.nout);

    // pass 2: CSC (in-edges, owner = destination) and CSR (out-edges, owner = source), record
    // order inside each column / row (graph.hpp:167-215)
    {
        std::vector<uint64_t> curIn(g.colPtr.begin(), g.colPtr.end() - 1), curOut(g.rowPtr.begin(), g.rowPtr.end() - 1);
        run_parallel([&](unsigned, uint32_t lo, uint32_t hi) {
            auto own = [&](uint32_t v) { return v >= lo && v < hi; };
            auto fill = [&](uint32_t from, uint32_t to) {
                const bool of = own(from), ot = own(to);
                if (!of && !ot) return;
                const bool fl = (uint32_t)parts[from] == me, tl = (uint32_t)parts[to] == me;
                if (tl && ot) {                      // in-edge of local vertex `to`
                    const uint32_t lt = g2l[to];
                    const uint64_t pos = curIn[lt]++;
                    float srcNorm;
                    if (fl) {
                        g.rowIdx[pos] = g2l[from];
                        srcNorm = vnorm[g2l[from]];
                    } else {
                        g.rowIdx[pos] = N + ghostRankIn[from];
                        srcNorm = inv_sqrt_deg((uint64_t)indegFile[from] + 1);
                    }
                    g.cscVal[pos] = srcNorm * vnorm[lt];         // e.setData(srcNorm * vtxNorm)
                }
                if (fl && of) {                      // out-edge of local vertex `from`
                    const uint32_t lf = g2l[from];
                    const uint64_t pos = curOut[lf]++;
                    float dstNorm;
                    if (tl) {
                        g.colIdx[pos] = g2l[to];
                        dstNorm = vnorm[g2l[to]];
                    } else {
                        g.colIdx[pos] = N + ghostRankOut[to];
                        dstNorm = inv_sqrt_deg((uint64_t)indegFile[to] + 1);
                    }
                    g.csrVal[pos] = vnorm[lf] * dstNorm;         // e.setData(vtxNorm * dstNorm)
                }
            };
            for (uint64_t i = 0; i < nrec; ++i) {
                const uint32_t s = src[i], d = dst[i];
                if (s == d) continue;
                fill(s, d);
                if (undirected) fill(d, s);
            }
        });
    }
    *out = pp.release();
    return DORY_OK;
}

int dory_partition_build_from_files(const char *dataset_dir, uint32_t me, uint32_t P, int undirected,
                                    dory_partition **out) {
    if (!dataset_dir || !out) return herr(DORY_ERR_ARG, "partition_build_from_files: bad arguments");
    const std::string dir(dataset_dir);
    // .parts: one id per line, lines not starting with a digit skipped (dataloader.cpp:66-72)
    std::vector<int32_t> parts;
    {
        std::ifstream f(dir + "graph.bsnap.parts");
        if (!f.good()) return herr(DORY_ERR_IO, "cannot open " + dir + "graph.bsnap.parts: " + std::strerror(errno));
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] < '0' || line[0] > '9') continue;
            std::istringstream iss(line);
            short id;
            if (!(iss >> id)) break;
            parts.push_back(id);
        }
    }
    // .edges: BSHeaderType + (u32,u32) records (dataloader.hpp:11-15)
    FILE *f = fopen((dir + "graph.bsnap.edges").c_str(), "rb");
    if (!f) return herr(DORY_ERR_IO, "cannot open " + dir + "graph.bsnap.edges: " + std::strerror(errno));
    struct { int32_t sizeOfVertexType; uint32_t numVertices; uint64_t numEdges; } h;
    if (fread(&h, sizeof(h), 1, f) != 1 || h.sizeOfVertexType != 4) {
        fclose(f);
        return herr(DORY_ERR_IO, "bad graph.bsnap.edges header");
    }
    std::vector<uint32_t> rec;
    {
        long pos = ftell(f);
        fseek(f, 0, SEEK_END);
        long endp = ftell(f);
        fseek(f, pos, SEEK_SET);
        const size_t n = (size_t)(endp - pos) / 8;  // the reference reads until EOF, not numEdges
        rec.resize(n * 2);
        if (n && fread(rec.data(), 8, n, f) != n) {
            fclose(f);
            return herr(DORY_ERR_IO, "short read on graph.bsnap.edges");
        }
    }
    fclose(f);
    const size_t n = rec.size() / 2;
    std::vector<uint32_t> s(n), d(n);
    for (size_t i = 0; i < n; ++i) {
        s[i] = rec[2 * i];
        d[i] = rec[2 * i + 1];
    }
    return dory_partition_build(s.data(), d.data(), n, parts.data(), (uint32_t)parts.size(), me, P, undirected, out);
}

// ---- graph.<id>.bin -------------------------------------------------------
int dory_partition_save(const dory_partition *p, const char *path) {
    if (!p || !path) return herr(DORY_ERR_ARG, "partition_save: bad arguments");
    FILE *f = fopen(path, "wb");
    if (!f) return herr(DORY_ERR_IO, std::string("cannot open ") + path + ": " + std::strerror(errno));
    auto w = [&](const void *b, size_t n) { return n == 0 || fwrite(b, 1, n, f) == n; };
    bool ok = w(&p->N, 4) && w(&p->V, 4) && w(&p->Gsrc, 4) && w(&p->Gdst, 4) && w(&p->nin, 8) && w(&p->nout, 8) &&
              w(&p->nglobal, 8) && w(p->l2g.data(), 4 * (size_t)p->N) && w(p->norm.data(), 4 * (size_t)p->N);
    for (uint32_t k = 0; ok && k < p->Gsrc; ++k) {
        uint32_t pr[2] = {p->srcGhost[k], p->N + k};
        ok = w(pr, 8);
    }
    for (uint32_t k = 0; ok && k < p->Gdst; ++k) {
        uint32_t pr[2] = {p->dstGhost[k], p->N + k};
        ok = w(pr, 8);
    }
    ok = ok && w(&p->P, 4);
    size_t off = 0;
    for (uint32_t q = 0; ok && q < p->P; ++q) {
        ok = w(&p->fwdCnt[q], 4) && w(p->fwdList.data() + off, 4 * (size_t)p->fwdCnt[q]);
        off += p->fwdCnt[q];
    }
    off = 0;
    for (uint32_t q = 0; ok && q < p->P; ++q) {
        ok = w(&p->bwdCnt[q], 4) && w(p->bwdList.data() + off, 4 * (size_t)p->bwdCnt[q]);
        off += p->bwdCnt[q];
    }
    ok = ok && w(&p->N, 4) && w(&p->nin, 8) && w(p->cscVal.data(), 4 * p->nin) &&
         w(p->colPtr.data(), 8 * ((size_t)p->N + 1)) && w(p->rowIdx.data(), 4 * p->nin);
    ok = ok && w(&p->N, 4) && w(&p->nout, 8) && w(p->csrVal.data(), 4 * p->nout) &&
         w(p->rowPtr.data(), 8 * ((size_t)p->N + 1)) && w(p->colIdx.data(), 4 * p->nout);
    fclose(f);
    return ok ? DORY_OK : herr(DORY_ERR_IO, std::string("write failed on ") + path);
}

int dory_partition_load(const char *path, dory_partition **out) {
    if (!path || !out) return herr(DORY_ERR_ARG, "partition_load: bad arguments");
    FILE *f = fopen(path, "rb");
    if (!f) return herr(DORY_ERR_IO, std::string("cannot open ") + path + ": " + std::strerror(errno));
    std::unique_ptr<dory_partition> pp(new dory_partition());
    dory_partition &g = *pp;
    auto r = [&](void *b, size_t n) { return n == 0 || fread(b, 1, n, f) == n; };
    bool ok = r(&g.N, 4) && r(&g.V, 4) && r(&g.Gsrc, 4) && r(&g.Gdst, 4) && r(&g.nin, 8) && r(&g.nout, 8) && r(&g.nglobal, 8);
    if (ok) {
        g.l2g.resize(g.N);
        g.norm.resize(g.N);
        ok = r(g.l2g.data(), 4 * (size_t)g.N) && r(g.norm.data(), 4 * (size_t)g.N);
    }
    auto ghosts = [&](std::vector<uint32_t> &v, uint32_t n) {
        v.resize(n);
        for (uint32_t k = 0; ok && k < n; ++k) {
            uint32_t pr[2];
            ok = r(pr, 8);
            if (ok && pr[1] != g.N + k) ok = false;   // local id must be N + rank
            v[k] = pr[0];
        }
    };
    if (ok) ghosts(g.srcGhost, g.Gsrc);
    if (ok) ghosts(g.dstGhost, g.Gdst);
    ok = ok && r(&g.P, 4) && g.P > 0 && g.P <= 65536;
    auto lists = [&](std::vector<uint32_t> &cnt, std::vector<uint32_t> &lst) {
        cnt.assign(g.P, 0);
        for (uint32_t q = 0; ok && q < g.P; ++q) {
            ok = r(&cnt[q], 4);
            if (!ok || cnt[q] > g.N) { ok = false; break; }
            const size_t o = lst.size();
            lst.resize(o + cnt[q]);
            ok = r(lst.data() + o, 4 * (size_t)cnt[q]);
        }
    };
    if (ok) lists(g.fwdCnt, g.fwdList);
    if (ok) lists(g.bwdCnt, g.bwdList);
    uint32_t cc = 0;
    uint64_t nnz = 0;
    ok = ok && r(&cc, 4) && r(&nnz, 8) && cc == g.N && nnz == g.nin;
    if (ok) {
        g.cscVal.resize(nnz);
        g.colPtr.resize((size_t)g.N + 1);
        g.rowIdx.resize(nnz);
        ok = r(g.cscVal.data(), 4 * nnz) && r(g.colPtr.data(), 8 * ((size_t)g.N + 1)) && r(g.rowIdx.data(), 4 * nnz);
    }
    ok = ok && r(&cc, 4) && r(&nnz, 8) && cc == g.N && nnz == g.nout;
    if (ok) {
        g.csrVal.resize(nnz);
        g.rowPtr.resize((size_t)g.N + 1);
        g.colIdx.resize(nnz);
        ok = r(g.csrVal.data(), 4 * nnz) && r(g.rowPtr.data(), 8 * ((size_t)g.N + 1)) && r(g.colIdx.data(), 4 * nnz);
    }
    fclose(f);
    if (!ok) return herr(DORY_ERR_IO, std::string("malformed graph bin ") + path);
    *out = pp.release();
    return DORY_OK;
}

int dory_partition_free(dory_partition *p) {
    delete p;
    return DORY_OK;
}

int dory_partition_get(const dory_partition *p, struct dory_partition_view *v) {
    if (!p || !v) return herr(DORY_ERR_ARG, "partition_get: bad arguments");
    v->local_vtx_cnt = p->N; v->global_vtx_cnt = p->V; v->src_ghost_cnt = p->Gsrc; v->dst_ghost_cnt = p->Gdst;
    v->num_nodes = p->P; v->local_in_edge_cnt = p->nin; v->local_out_edge_cnt = p->nout; v->global_edge_cnt = p->nglobal;
    v->local_to_global = p->l2g.data(); v->norms = p->norm.data();
    v->src_ghosts = p->srcGhost.data(); v->dst_ghosts = p->dstGhost.data();
    v->fwd_counts = p->fwdCnt.data(); v->fwd_lists = p->fwdList.data();
    v->bwd_counts = p->bwdCnt.data(); v->bwd_lists = p->bwdList.data();
    v->column_ptrs = p->colPtr.data(); v->row_idxs = p->rowIdx.data(); v->csc_values = p->cscVal.data();
    v->row_ptrs = p->rowPtr.data(); v->column_idxs = p->colIdx.data(); v->csr_values = p->csrVal.data();
    return DORY_OK;
}

int dory_partition_recv_plan(const dory_partition *p, const int32_t *parts, int dir, uint32_t *recv_counts,
                             uint32_t *recv_slots) {
    if (!p || !parts || !recv_counts || (dir != 0 && dir != 1)) return herr(DORY_ERR_ARG, "recv_plan: bad arguments");
    const std::vector<uint32_t> &ghost = dir == 0 ? p->srcGhost : p->dstGhost;
    std::vector<std::vector<uint32_t>> per(p->P);
    for (uint32_t k = 0; k < ghost.size(); ++k) {
        const int32_t o = parts[ghost[k]];
        if (o < 0 || (uint32_t)o >= p->P) return herr(DORY_ERR_ARG, "recv_plan: ghost owner out of range");
        per[(uint32_t)o].push_back(k);
    }
    size_t off = 0;
    for (uint32_t q = 0; q < p->P; ++q) {
        recv_counts[q] = (uint32_t)per[q].size();
        for (uint32_t k : per[q]) recv_slots[off++] = k;
    }
    return DORY_OK;
}

int dory_partition_upload(dory_ctx *ctx, const dory_partition *p, const int32_t *parts) {
    if (!ctx || !p) return herr(DORY_ERR_ARG, "partition_upload: bad arguments");
    int rc = dory_graph_upload(ctx, p->N, p->Gsrc, p->Gdst, p->nin, p->colPtr.data(), p->rowIdx.data(),
                               p->cscVal.data(), p->nout, p->rowPtr.data(), p->colIdx.data(),
                               p->csrVal.data(), p->norm.data());
    if (rc) return rc;
    // the transform-first GCN order needs csrVal == cscVal transposed: only known for directed builds
    if ((rc = dory_set_option(ctx, "adjacency_values_asymmetric", p->undirected == 0 ? 0 : 1))) return rc;
    if (!parts || p->P <= 1) return rc;
    // receive side of the plan: peer q's k-th row lands in the k-th ghost slot owned by q
    for (int dir = 0; dir < 2; ++dir) {
        const std::vector<uint32_t> &ghost = dir == 0 ? p->srcGhost : p->dstGhost;
        std::vector<uint32_t> rcnt(p->P, 0), rslots(ghost.size() + 1, 0);
        if ((rc = dory_partition_recv_plan(p, parts, dir, rcnt.data(), rslots.data()))) return rc;
        const std::vector<uint32_t> &scnt = dir == 0 ? p->fwdCnt : p->bwdCnt;
        const std::vector<uint32_t> &slist = dir == 0 ? p->fwdList : p->bwdList;
        uint32_t dummy = 0;
        rc = dory_halo_plan(ctx, dir, scnt.data(), slist.empty() ? &dummy : slist.data(), rcnt.data(), rslots.data());
        if (rc) return rc;
    }
    return DORY_OK;
}

}  // extern "C"
