// ref_wire_peer.cpp -- TEST INFRASTRUCTURE.  A weight-server stand-in for the loopback test of the wire formats
// (tests/test_wire_loopback.py): a ZeroMQ ROUTER on tcp://127.0.0.1:<port> that answers the graph-server <-> weight-server
// conversation of the reference -- PULL, PUSH, EVAL, TERM -- and reads every incoming frame with the REFERENCE's own
// header-only code, compiled from /root/reference in place: common/utils.hpp (Chunk, OP, HEADER_SIZE, TENSOR_HDR_SIZE,
// parse<>, parseName, populateHeader) and the C++ binding the reference vendors, common/zmq.hpp, over the image's real
// libzmq (/opt/conda: zmq.h + libzmq.so.5).  No reference source is copied, no stand-in header is used.
//
// The real weight server (weight-server/weightserver.cpp, serverworker.cpp) cannot be built here: it needs
// boost/algorithm/string/trim.hpp and cblas.h, which this image lacks.  What this peer keeps from it is the loop shape
// of ServerWorker::work / sendTensors / recvTensors / recvEvalData (serverworker.cpp:30-160), restated around the
// reference's parse calls with a plain map of float vectors instead of WeightTensor.  It proves that the frame lists
// include/dorylus_wire.h emits are understood by the reference's parsing code over a real socket, and that the replies
// the reference's populateHeader builds are understood by dory_wire_parse_pull_reply; it proves nothing about the
// server's averaging / Adam.
//
// usage: ref_wire_peer <port>    -- prints one JSON object with everything it parsed, after OP::TERM
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <zmq.hpp>          // the reference's vendored binding (src/common/zmq.hpp), on the image's zmq.h
#include "common/utils.hpp"

struct Tensor { unsigned rows, cols; std::vector<float> data; };

int main(int argc, char **argv) {
    const int port = argc > 1 ? atoi(argv[1]) : 55431;
    zmq::context_t ctx(1);
    zmq::socket_t sock(ctx, ZMQ_ROUTER);        // the frontend of weightserver.cpp (ROUTER <-> DEALER workers), in one socket
    char addr[64];
    snprintf(addr, sizeof(addr), "tcp://127.0.0.1:%d", port);
    sock.bind(addr);
    // weightsStore[layer]["w"]: a known function of (layer, index) the client checks on its side
    std::map<unsigned, std::map<std::string, Tensor>> store;
    const unsigned dims[3] = {602, 128, 41};    // run/reddit.config
    for (unsigned l = 0; l < 2; ++l) {
        Tensor t{dims[l], dims[l + 1], std::vector<float>((size_t)dims[l] * dims[l + 1])};
        for (size_t i = 0; i < t.data.size(); ++i) t.data[i] = (float)l + (float)(i % 1000) * 1e-3f;
        store[l]["w"] = t;
    }
    std::string log = "";
    unsigned n_pull = 0, n_push = 0, n_eval = 0;
    char line[512];
    while (true) {
        zmq::message_t identity, header;
        sock.recv(&identity);
        sock.recv(&header);
        const OP op = parse<OP>((char *)header.data(), 0);                       // serverworker.cpp:39
        if (op == OP::TERM) break;
        Chunk chunk;
        memcpy(&chunk, (char *)header.data() + sizeof(OP), sizeof(Chunk));       // serverworker.cpp:43-44
        const unsigned featLayer = chunk.vertex ? chunk.layer : chunk.layer - 1; // serverworker.cpp:92
        if (op == OP::PULL) {                                                    // sendTensors, serverworker.cpp:84-113
            unsigned more = 1;
            sock.send(identity, ZMQ_SNDMORE);
            while (more) {
                zmq::message_t tensorHeader(TENSOR_HDR_SIZE);
                sock.recv(&tensorHeader);
                const std::string name = parseName((char *)tensorHeader.data());
                const unsigned reqOp = parse<unsigned>((char *)tensorHeader.data(), 0);   // = chunk.localId (message_service.cpp:52)
                size_t usize = sizeof(more);
                sock.getsockopt(ZMQ_RCVMORE, &more, &usize);
                snprintf(line, sizeof(line), "%s{\"op\": \"PULL\", \"layer\": %u, \"globalId\": %u, \"epoch\": %u, \"dir\": %d, \"name\": \"%s\", \"req_op\": %u, \"hdr_size\": %zu}",
                         log.empty() ? "" : ", ", chunk.layer, chunk.globalId, chunk.epoch, (int)chunk.dir, name.c_str(), reqOp, header.size());
                log += line;
                ++n_pull;
                auto &weights = store[featLayer];
                auto found = weights.find(name);
                if (found == weights.end()) {                                    // serverworker.cpp:101-106
                    zmq::message_t errorHeader(TENSOR_HDR_SIZE);
                    populateHeader(errorHeader.data(), ERR_HEADER_FIELD, name.c_str());
                    sock.send(errorHeader);
                    while (more) {                                               // (drain what is left of the request)
                        zmq::message_t rest;
                        sock.recv(&rest);
                        sock.getsockopt(ZMQ_RCVMORE, &more, &usize);
                    }
                    break;
                }
                Tensor &t = found->second;                                       // sendTensor, serverworker.cpp:138-154
                zmq::message_t responseHeader(TENSOR_HDR_SIZE);
                populateHeader(responseHeader.data(), OP::PULL, name.c_str(), t.rows, t.cols);
                zmq::message_t tensorData(t.data.size() * sizeof(float));
                memcpy(tensorData.data(), t.data.data(), t.data.size() * sizeof(float));
                sock.send(responseHeader, ZMQ_SNDMORE);
                if (!more) sock.send(tensorData); else sock.send(tensorData, ZMQ_SNDMORE);
            }
        } else if (op == OP::PUSH) {                                             // recvTensors, serverworker.cpp:115-125
            unsigned more = 1;
            while (more) {
                zmq::message_t tensorHeader(TENSOR_HDR_SIZE), tensorData;        // recvUpdateTensor, serverworker.cpp:156-160
                sock.recv(&tensorHeader);
                sock.recv(&tensorData);
                const std::string name = parseName((char *)tensorHeader.data());
                const unsigned hop = parse<unsigned>((char *)tensorHeader.data(), 0);
                const unsigned f1 = parse<unsigned>((char *)tensorHeader.data(), 3), f2 = parse<unsigned>((char *)tensorHeader.data(), 4),
                               f3 = parse<unsigned>((char *)tensorHeader.data(), 5);
                double sum = 0;
                const float *p = (const float *)tensorData.data();
                for (size_t i = 0; i < tensorData.size() / sizeof(float); ++i) sum += (double)p[i] * (double)(1 + i % 7);
                snprintf(line, sizeof(line), "%s{\"op\": \"PUSH\", \"layer\": %u, \"globalId\": %u, \"epoch\": %u, \"dir\": %d, \"name\": \"%s\", \"hdr_op\": %u, \"f1\": %u, \"f2\": %u, \"f3\": %u, \"payload_bytes\": %zu, \"weighted_sum\": %.17g}",
                         log.empty() ? "" : ", ", chunk.layer, chunk.globalId, chunk.epoch, (int)chunk.dir, name.c_str(), hop, f1, f2, f3, tensorData.size(), sum);
                log += line;
                ++n_push;
                // keep the pushed update where a later PULL of "<name>_upd" finds it (the client reads it back bit for bit)
                Tensor t{f2, f3, std::vector<float>(p, p + tensorData.size() / sizeof(float))};
                store[featLayer][(name + "_upd").substr(0, 8)] = t;
                size_t usize = sizeof(more);
                sock.getsockopt(ZMQ_RCVMORE, &more, &usize);
            }
        } else if (op == OP::EVAL) {                                             // recvEvalData, serverworker.cpp:127-136
            zmq::message_t evalMsg(2 * sizeof(float));
            sock.recv(&evalMsg);
            const float acc = *((float *)evalMsg.data()), loss = *(((float *)evalMsg.data()) + 1);
            snprintf(line, sizeof(line), "%s{\"op\": \"EVAL\", \"localId\": %u, \"globalId\": %u, \"upBound\": %u, \"layer\": %u, \"epoch\": %u, \"vertex\": %d, \"acc\": %.9g, \"loss\": %.9g}",
                     log.empty() ? "" : ", ", chunk.localId, chunk.globalId, chunk.upBound, chunk.layer, chunk.epoch, (int)chunk.vertex, acc, loss);
            log += line;
            ++n_eval;
        } else {
            snprintf(line, sizeof(line), "%s{\"op\": \"UNKNOWN\", \"value\": %u}", log.empty() ? "" : ", ", (unsigned)op);
            log += line;
            unsigned more = 1;
            size_t usize = sizeof(more);
            sock.getsockopt(ZMQ_RCVMORE, &more, &usize);
            while (more) { zmq::message_t rest; sock.recv(&rest); sock.getsockopt(ZMQ_RCVMORE, &more, &usize); }
        }
    }
    printf("{\"pulls\": %u, \"pushes\": %u, \"evals\": %u, \"messages\": [%s]}\n", n_pull, n_push, n_eval, log.c_str());
    return 0;
}
