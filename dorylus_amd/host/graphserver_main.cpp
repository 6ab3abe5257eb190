// graphserver_main.cpp -- a `graphserver` for the hip backend that accepts the reference
// graph server's command line (src/graph-server/engine/utils.cpp:313-452; the argv that
// run/run-onnode:154-179 builds) and runs the synchronous epoch loop on MI355X GPUs of one
// node: Engine::init -> run -> output -> destroy (src/graph-server/main.cpp:16-41).
//
// Flags that only drive the ZeroMQ / Lambda machinery (ports, wserveripfile, numlambdas,
// staleness, timeout_ratio, cthreads, dthreads, pipeline, MODE) are accepted and ignored,
// like unregistered options are in the reference (allow_unregistered()).
// Node identity: --dshmachinesfile + --pripfile when both are readable (the reference's
// way, nodemanager.cpp:290-311), else RANK / WORLD_SIZE / LOCAL_RANK from the environment
// (torchrun / mpirun launchers), else a single node.
#include <sys/stat.h>
#include <unistd.h>

#include <cstdarg>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dorylus_host.h"

extern "C" const char *dory_formats_last_error(void);
extern "C" int dory_read_layer_config(const char *, uint32_t *, uint32_t, uint32_t *);
extern "C" int dory_read_features(const char *, const dory_partition *, uint32_t, uint32_t, const char *, float *, float *);
extern "C" int dory_read_labels(const char *, const dory_partition *, uint32_t, uint32_t *);

static unsigned g_node = 0;
static void printLog(const char *fmt, ...) {  // "[ Node %3u ]  ..." (utils/utils.cpp:17-30)
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "[ Node %3u ]  ", g_node);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}

static std::map<std::string, std::string> parse_args(int argc, char **argv) {
    // --key value | --key=value ; unknown keys kept (allow_unregistered); unique-prefix
    // matching is resolved by the lookups below (run-onnode passes --numEpoch)
    std::map<std::string, std::string> m;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a.rfind("--", 0) != 0) continue;
        a = a.substr(2);
        size_t eq = a.find('=');
        if (eq != std::string::npos) m[a.substr(0, eq)] = a.substr(eq + 1);
        else if (i + 1 < argc && std::string(argv[i + 1]).rfind("--", 0) != 0) m[a] = argv[++i];
        else m[a] = "1";
    }
    return m;
}
static bool lookup(const std::map<std::string, std::string> &m, const std::string &name, std::string *out) {
    auto it = m.find(name);
    if (it != m.end()) { *out = it->second; return true; }
    for (auto &kv : m)  // boost::program_options accepts unambiguous prefixes
        if (!kv.first.empty() && name.rfind(kv.first, 0) == 0 && kv.first.size() >= 4) { *out = kv.second; return true; }
    return false;
}

static std::vector<std::string> read_lines(const std::string &path) {
    std::vector<std::string> v;
    std::ifstream f(path);
    std::string l;
    while (std::getline(f, l)) {
        size_t a = l.find_first_not_of(" \t\r\n");
        if (a == std::string::npos) continue;
        size_t b = l.find_last_not_of(" \t\r\n");
        v.push_back(l.substr(a, b - a + 1));
    }
    return v;
}

#define DIE(...)                       \
    do {                               \
        printLog(__VA_ARGS__);         \
        return 1;                      \
    } while (0)

int main(int argc, char **argv) {
    struct timespec proc_start;
    clock_gettime(CLOCK_REALTIME, &proc_start);   // id files older than this belong to another job
    auto args = parse_args(argc, argv);
    std::string datasetDir, featuresFile, layerFile, labelsFile, tmpDir = "/tmp", gnn = "GCN", s;
    if (args.count("help") || !lookup(args, "datasetdir", &datasetDir) || !lookup(args, "featuresfile", &featuresFile) ||
        !lookup(args, "layerfile", &layerFile) || !lookup(args, "labelsfile", &labelsFile)) {
        fprintf(stderr, "usage: graphserver --datasetdir D/ --featuresfile F --layerfile L --labelsfile Y "
                        "[--numEpochs N] [--gnn GCN|GAT] [--undirected 0|1] [--tmpdir T] [--lr 0.01] [--dory-<option> <int>] "
                        "[reference flags are accepted]\n");
        return 2;
    }
    lookup(args, "tmpdir", &tmpDir);
    lookup(args, "gnn", &gnn);
    unsigned numEpochs = 10, undirected = 0;
    if (lookup(args, "numEpochs", &s)) numEpochs = (unsigned)atoi(s.c_str());
    if (lookup(args, "undirected", &s)) undirected = (unsigned)atoi(s.c_str());
    float lr = 0.01f;   // run/run-onnode:226 LEARNING_RATE default (a weight-server argument there)
    if (lookup(args, "lr", &s)) lr = (float)atof(s.c_str());

    // ---- node identity -----------------------------------------------------------------
    unsigned nodeId = 0, numNodes = 1;
    int localRank = 0;
    std::string dsh, prip;
    bool have = false;
    if (lookup(args, "dshmachinesfile", &dsh) && lookup(args, "pripfile", &prip)) {
        auto machines = read_lines(dsh);
        auto me = read_lines(prip);
        if (!machines.empty() && !me.empty()) {
            for (size_t i = 0; i < machines.size(); ++i) {
                std::string ip = machines[i].substr(machines[i].find('@') == std::string::npos ? 0 : machines[i].find('@') + 1);
                if (ip == me[0]) { nodeId = (unsigned)i; have = true; }
            }
            if (have) numNodes = (unsigned)machines.size();
        }
    }
    if (!have) {
        if (const char *e = getenv("RANK")) nodeId = (unsigned)atoi(e);
        if (const char *e = getenv("WORLD_SIZE")) numNodes = (unsigned)atoi(e);
    }
    if (const char *e = getenv("LOCAL_RANK")) localRank = atoi(e); else localRank = (int)nodeId;
    g_node = nodeId;
    if (numNodes > 256) DIE("at most 256 nodes (engine.cpp:52)");

    const auto t_init = std::chrono::steady_clock::now();
    time_t start_time = time(nullptr);

    // ---- Engine::init (engine.cpp:40-167) ----------------------------------------------
    uint32_t dims[64], nd = 0;
    if (dory_read_layer_config(layerFile.c_str(), dims, 64, &nd)) DIE("%s", dory_formats_last_error());
    const uint32_t L = nd - 1;

    // graph.<id>.bin: preprocess once, then load (engine.cpp:62-72)
    char name[64];
    snprintf(name, sizeof(name), "graph.%u.bin", nodeId);
    const std::string binPath = datasetDir + name;
    dory_partition *part = nullptr;
    if (access(binPath.c_str(), R_OK) != 0) {
        printLog("Preprocessing... Output to %s", binPath.c_str());
        if (dory_partition_build_from_files(datasetDir.c_str(), nodeId, numNodes, (int)undirected, &part)) DIE("%s", dory_host_last_error());
        if (dory_partition_save(part, binPath.c_str())) printLog("warning: %s", dory_host_last_error());
        printLog("Finish preprocessing!");
    } else if (dory_partition_load(binPath.c_str(), &part)) {
        DIE("%s", dory_host_last_error());
    }
    struct dory_partition_view v;
    dory_partition_get(part, &v);
    if (v.num_nodes != numNodes) DIE("graph bin was built for %u nodes, running with %u", v.num_nodes, numNodes);
    printLog("<GM>: %u global vertices, %llu global edges,\n\t\t%u local vertices, %llu local in-edges, %llu local out-edges\n\t\t%u out ghost vertices, %u in ghost vertices",
             v.global_vtx_cnt, (unsigned long long)v.global_edge_cnt, v.local_vtx_cnt,
             (unsigned long long)v.local_in_edge_cnt, (unsigned long long)v.local_out_edge_cnt, v.dst_ghost_cnt, v.src_ghost_cnt);

    std::vector<int32_t> parts;
    if (numNodes > 1) {
        for (auto &l : read_lines(datasetDir + "graph.bsnap.parts"))
            if (l[0] >= '0' && l[0] <= '9') parts.push_back(atoi(l.c_str()));
        if (parts.size() != v.global_vtx_cnt) DIE("graph.bsnap.parts does not match the graph");
    }

    const bool gat = gnn == "GAT";
    if (!gat && gnn != "GCN") DIE("Unsupported GNN type: %s", gnn.c_str());
    std::vector<float> x((size_t)v.local_vtx_cnt * dims[0]), fg((size_t)v.src_ghost_cnt * dims[0] + 1);
    std::vector<uint32_t> labels(v.local_vtx_cnt);
    if (dory_read_features(featuresFile.c_str(), part, dims[0], nodeId, datasetDir.c_str(), x.data(), fg.data())) DIE("%s", dory_formats_last_error());
    if (dory_read_labels(labelsFile.c_str(), part, dims[L], labels.data())) DIE("%s", dory_formats_last_error());

    dory_ctx *ctx = nullptr;
    if (dory_create(localRank, &ctx)) DIE("%s", dory_last_error(nullptr));
#define CK(call) if (call) DIE("%s: %s", #call, dory_last_error(ctx))
    CK(dory_configure(ctx, gat ? DORY_GAT : DORY_GCN, L, dims, v.global_vtx_cnt, nodeId, numNodes));
    // library options that have no reference flag: --dory-<option> <integer>, e.g. --dory-gcn_transform_first 2
    for (auto &kv : args)
        if (kv.first.rfind("dory-", 0) == 0) CK(dory_set_option(ctx, kv.first.substr(5).c_str(), atoll(kv.second.c_str())));
    CK(dory_partition_upload(ctx, part, numNodes > 1 ? parts.data() : nullptr));
    CK(dory_preallocate(ctx));
    CK(dory_tensor_upload(ctx, 0, gat ? "h" : "x", x.data()));
    if (!gat && v.src_ghost_cnt) CK(dory_tensor_upload(ctx, 0, "fg", fg.data()));
    CK(dory_labels_upload(ctx, labels.data()));
    CK(dory_weights_init_xavier(ctx));
    CK(dory_adam_config(ctx, lr));
    if (numNodes > 1) {  // RCCL bootstrap over a file in tmpdir (single node, shared filesystem)
        // One id file per job.  The file carries the job's nonce behind the 128-byte id and a rank accepts it by CONTENT:
        //  * DORY_JOB_ID set (run/run-dorylus-hip: launcher pid + start time -- unique per job): the nonce must match; when
        //    a rank starts relative to rank 0 does not matter (staggered per-machine launches), nor does any clock;
        //  * no job id: the rendezvous port is the nonce, which a crashed earlier job may have used too -- then the file must
        //    also be no older than the job's start (DORY_JOB_START, epoch seconds, from the launcher) or, lacking that,
        //    than ten minutes before this rank started (ranks of one job do not start further apart than that).
        const char *job = getenv("DORY_JOB_ID");
        const bool unique_job = job && *job;
        const char *nonce = unique_job ? job : getenv("MASTER_PORT");
        if (!nonce || !*nonce) nonce = "default";
        char nonce_buf[64];
        memset(nonce_buf, 0, sizeof(nonce_buf));
        strncpy(nonce_buf, nonce, sizeof(nonce_buf) - 1);
        const std::string idFile = tmpDir + "/dorylus_rccl_id." + nonce_buf + ".bin";
        unsigned char id[128];
        if (nodeId == 0) {
            if (dory_comm_unique_id(id)) DIE("ncclGetUniqueId failed");
            remove(idFile.c_str());   // stale file of an earlier job with the same nonce
            const std::string tmp = idFile + ".tmp";
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, 128, f) != 128 || fwrite(nonce_buf, 1, sizeof(nonce_buf), f) != sizeof(nonce_buf)) DIE("cannot write %s", tmp.c_str());
            fclose(f);
            rename(tmp.c_str(), idFile.c_str());
        } else {
            time_t not_before = proc_start.tv_sec - 600;
            if (const char *js = getenv("DORY_JOB_START")) { if (*js) not_before = (time_t)strtoll(js, nullptr, 10); }
            bool ok = false;
            for (int tries = 0; tries < 600 && !ok; ++tries) {
                struct stat sb;
                // the job's nonce is what identifies the file; the age bound stays even then (a constant DORY_JOB_ID reused after a
                // crashed job must not hand out that job's id: a day is far more than any launcher's skew, DORY_JOB_START tightens it)
                const time_t bound = unique_job && !getenv("DORY_JOB_START") ? proc_start.tv_sec - 86400 : not_before;
                if (stat(idFile.c_str(), &sb) == 0 && sb.st_mtim.tv_sec >= bound) {
                    if (FILE *f = fopen(idFile.c_str(), "rb")) {
                        char got[64];
                        ok = fread(id, 1, 128, f) == 128 && fread(got, 1, sizeof(got), f) == sizeof(got) &&
                             memcmp(got, nonce_buf, sizeof(got)) == 0;
                        fclose(f);
                    }
                }
                if (!ok) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
            if (!ok) DIE("timed out waiting for %s (job nonce '%s')", idFile.c_str(), nonce_buf);
        }
        CK(dory_comm_init(ctx, id, (int)nodeId, (int)numNodes));
        if (nodeId == 0) { std::this_thread::sleep_for(std::chrono::seconds(2)); unlink(idFile.c_str()); }
    }
    dory_engine *eng = nullptr;
    CK(dory_engine_create(ctx, &eng));
    const double timeInit = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_init).count();
    printLog("Engine initialization complete.");

    // ---- Engine::run (engine.cpp:223-235): numEpochs synchronous epochs -----------------------
    std::vector<double> epochMs(numEpochs);
    float acc = 0, loss = 0;
    uint32_t valRows = 0;
    for (unsigned ep = 0; ep < numEpochs; ++ep) {
        printLog("Sync Epoch %u starts...", ep + 1);
        CK(dory_engine_run(eng, 1, &epochMs[ep]));
        if (ep > 0) printLog("Time for epoch %u: %.2lfms", ep, epochMs[ep]);   // pipeline.cpp:118-127 skips epoch 0
        if (!gat) {
            CK(dory_train_stat(ctx, &acc, &loss, &valRows));
            if (valRows) printLog("batch Acc: %f, Loss: %f", acc / valRows, loss / valRows);  // CPU_comm.cpp:116
            // the weight servers' sum over the nodes (weightserver.cpp:190-262), logged by node 0 in their words
            float gacc = 0, gloss = 0;
            uint32_t gRows = 0;
            CK(dory_train_stat_global(ctx, &gacc, &gloss, &gRows));
            if (nodeId == 0 && gRows) printLog("Epoch %u, acc: %.4f, loss: %.4f", ep + 1, gacc / gRows, gloss / gRows);
        }
    }
    time_t end_time = time(nullptr);

    // ---- Engine::output (engine/utils.cpp:109-212, 219-291) -------------------------------------
    double sum = 0;
    for (unsigned i = 1; i < numEpochs; ++i) sum += epochMs[i];
    const double avg = numEpochs > 1 ? sum / (numEpochs - 1) : (numEpochs ? epochMs[0] : 0.0);
    const uint64_t edgesPerEpoch = 2 * v.local_in_edge_cnt + v.local_out_edge_cnt;
    char rep[2048];
    dory_engine_report(eng, rep, sizeof(rep));
    fputs(rep, stderr);
    printLog("<EM>: Initialization takes %.3lf ms", timeInit);
    printLog("<EM>: Aggregated edges/sec (this node) %.3lf M", avg > 0 ? edgesPerEpoch / avg / 1e3 : 0.0);
    printLog("<EM>: Final accuracy %.3lf", valRows ? acc / valRows : 0.0);
    {
        std::ofstream out(tmpDir + "/output_" + std::to_string(nodeId));
        out << "<EM>: Run start time: " << std::ctime(&start_time);
        out << "<EM>: Run end time: " << std::ctime(&end_time);
        out << "<EM>: Using 1 lambdas\n";
        char b[256];
        snprintf(b, sizeof(b), "<EM>: Initialization takes %.3lf ms\n", timeInit);
        out << b;
        snprintf(b, sizeof(b), "<EM>: Average epoch time %.3lf ms\n", avg);
        out << b;
        snprintf(b, sizeof(b), "<EM>: Final accuracy %.3lf\n", valRows ? acc / valRows : 0.0);
        out << b << "Relaunched Lambda Cnt: 0\n";
    }
    dory_engine_destroy(eng);
    dory_destroy(ctx);
    dory_partition_free(part);
    return 0;
}
