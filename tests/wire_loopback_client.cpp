// wire_loopback_client.cpp -- the graph-server side of tests/test_wire_loopback.py: a ZeroMQ DEALER (libzmq's C API; the
// socket is the caller's, include/dorylus_wire.h) that sends exactly the frame lists dory_wire_build_pull / _push / _accloss
// emit, one zmq frame per list entry, to oracle/_ref/ref_wire_peer, and reads the replies with dory_wire_parse_pull_reply.
// What MessageService::prefetchWeightsMatrix / sendWeightUpdate / sendAccloss do in the reference
// (commmanager/message_service.cpp:142-245).  Prints one JSON object; exit code 0 = every check on this side passed.
#include <zmq.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/dorylus_wire.h"

static void *sock;
static int fails = 0;
#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); ++fails; } \
    } while (0)

static void send_frames(const uint8_t *buf, const size_t *off, int n) {
    for (int i = 0; i < n; ++i) {
        const int rc = zmq_send(sock, buf + off[i], off[i + 1] - off[i], i + 1 < n ? ZMQ_SNDMORE : 0);
        CHECK(rc == (int)(off[i + 1] - off[i]));
    }
}

struct Frame { std::vector<uint8_t> b; bool more; };
static Frame recv_frame() {
    zmq_msg_t m;
    zmq_msg_init(&m);
    Frame f;
    if (zmq_msg_recv(&m, sock, 0) < 0) { fprintf(stderr, "recv: %s\n", zmq_strerror(zmq_errno())); exit(2); }
    f.b.assign((uint8_t *)zmq_msg_data(&m), (uint8_t *)zmq_msg_data(&m) + zmq_msg_size(&m));
    f.more = zmq_msg_more(&m) != 0;
    zmq_msg_close(&m);
    return f;
}

int main(int argc, char **argv) {
    const int port = argc > 1 ? atoi(argv[1]) : 55431;
    const uint32_t nodeId = 3, epoch = 5;
    void *ctx = zmq_ctx_new();
    sock = zmq_socket(ctx, ZMQ_DEALER);
    int timeout_ms = 20000;
    zmq_setsockopt(sock, ZMQ_RCVTIMEO, &timeout_ms, sizeof(timeout_ms));
    zmq_setsockopt(sock, ZMQ_SNDTIMEO, &timeout_ms, sizeof(timeout_ms));
    char identity[8];
    memcpy(identity, &nodeId, 4);
    memcpy(identity + 4, "hipX", 4);                       // MessageService::setUpWeightSocket: node id + address bytes
    zmq_setsockopt(sock, ZMQ_IDENTITY, identity, sizeof(identity));
    char addr[64];
    snprintf(addr, sizeof(addr), "tcp://127.0.0.1:%d", port);
    CHECK(zmq_connect(sock, addr) == 0);

    std::vector<uint8_t> buf((size_t)8 << 20);
    size_t off[16];
    const uint32_t dims[3] = {602, 128, 41};
    // ---- prefetchWeightsMatrix: one PULL of "w" per layer ----
    std::string pulled = "";
    for (uint32_t l = 0; l < 2; ++l) {
        dory_wire_chunk c{0, nodeId, 0, 0, l, 0 /* FORWARD */, epoch, 1};
        const char *names[1] = {"w"};
        const int nf = dory_wire_build_pull(&c, names, 1, buf.data(), buf.size(), off, 15);
        CHECK(nf == 2);
        send_frames(buf.data(), off, nf);
        Frame h = recv_frame();
        CHECK(h.b.size() == DORY_WIRE_TENSOR_HDR_SIZE && h.more);
        Frame d = recv_frame();
        CHECK(!d.more);
        char name[9];
        uint32_t rows = 0, cols = 0;
        const int rc = dory_wire_parse_pull_reply(h.b.data(), d.b.size(), name, &rows, &cols);
        CHECK(rc == 0 && rows == dims[l] && cols == dims[l + 1] && !strcmp(name, "w"));
        const float *w = (const float *)d.b.data();
        bool same = d.b.size() == (size_t)rows * cols * 4;
        for (size_t i = 0; same && i < (size_t)rows * cols; ++i) same = w[i] == (float)l + (float)(i % 1000) * 1e-3f;
        CHECK(same);
        char line[160];
        snprintf(line, sizeof(line), "%s{\"layer\": %u, \"rc\": %d, \"name\": \"%s\", \"rows\": %u, \"cols\": %u, \"payload_ok\": %s}", l ? ", " : "", l, rc, name, rows, cols, same ? "true" : "false");
        pulled += line;
    }
    // ---- a tensor the server does not hold: ERR_HEADER_FIELD ----
    int err_rc;
    {
        dory_wire_chunk c{0, nodeId, 0, 0, 1, 0, epoch, 1};
        const char *names[1] = {"nosuch"};
        const int nf = dory_wire_build_pull(&c, names, 1, buf.data(), buf.size(), off, 15);
        send_frames(buf.data(), off, nf);
        Frame h = recv_frame();
        char name[9];
        uint32_t rows = 0, cols = 0;
        err_rc = dory_wire_parse_pull_reply(h.b.data(), 0, name, &rows, &cols);
        CHECK(err_rc == 1 && !h.more);
    }
    // ---- sendWeightUpdate: PUSH of the layer-1 gradient; then read it back bit for bit ----
    std::vector<float> grad((size_t)dims[1] * dims[2]);
    double wsum = 0;
    for (size_t i = 0; i < grad.size(); ++i) { grad[i] = std::sin((float)i) * 0.25f; wsum += (double)grad[i] * (double)(1 + i % 7); }
    {
        dory_wire_chunk c{0, nodeId, 0, 0, 1, 1 /* BACKWARD */, epoch, 1};
        const char *names[1] = {"w"};
        const uint32_t rows[1] = {dims[1]}, cols[1] = {dims[2]};
        const float *data[1] = {grad.data()};
        const int nf = dory_wire_build_push(&c, names, rows, cols, data, 1, buf.data(), buf.size(), off, 15);
        CHECK(nf == 3);
        send_frames(buf.data(), off, nf);
    }
    bool readback = false;
    {
        dory_wire_chunk c{0, nodeId, 0, 0, 1, 0, epoch, 1};
        const char *names[1] = {"w_upd"};
        const int nf = dory_wire_build_pull(&c, names, 1, buf.data(), buf.size(), off, 15);
        send_frames(buf.data(), off, nf);
        Frame h = recv_frame();
        Frame d = recv_frame();
        char name[9];
        uint32_t rows = 0, cols = 0;
        const int rc = dory_wire_parse_pull_reply(h.b.data(), d.b.size(), name, &rows, &cols);
        readback = rc == 0 && rows == dims[1] && cols == dims[2] && d.b.size() == grad.size() * 4 && !memcmp(d.b.data(), grad.data(), d.b.size());
        CHECK(readback);
    }
    // ---- sendAccloss ----
    {
        const int nf = dory_wire_build_accloss(nodeId, epoch, 153431, 0.9375f, 1.25f, buf.data(), buf.size(), off, 15);
        CHECK(nf == 2);
        send_frames(buf.data(), off, nf);
    }
    // ---- OP::TERM ----
    {
        uint8_t hdr[DORY_WIRE_HEADER_SIZE];
        dory_wire_chunk c{0, nodeId, 0, 0, 0, 0, epoch, 1};
        dory_wire_pack_chunk_header(hdr, DORY_OP_TERM, &c);
        CHECK(zmq_send(sock, hdr, sizeof(hdr), 0) == (int)sizeof(hdr));
    }
    printf("{\"fails\": %d, \"pulled\": [%s], \"err_reply_rc\": %d, \"readback_bit_identical\": %s, \"pushed_weighted_sum\": %.17g}\n", fails, pulled.c_str(),
           err_rc, readback ? "true" : "false", wsum);
    int linger = 2000;
    zmq_setsockopt(sock, ZMQ_LINGER, &linger, sizeof(linger));
    zmq_close(sock);
    zmq_ctx_term(ctx);
    return fails ? 1 : 0;
}
