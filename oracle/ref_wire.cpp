// ref_wire.cpp -- TEST INFRASTRUCTURE.  Emits golden bytes of the reference's message headers by calling the
// REFERENCE's own header-only serialisation code (common/utils.hpp: Chunk, OP, HEADER_SIZE, TENSOR_HDR_SIZE,
// serialize<>, the three populateHeader overloads, parseName) compiled from /root/reference in place (oracle/Makefile,
// target _ref/ref_wire).  No reference source is copied and no stand-in header is used.  The one packer that lives in a
// .cpp next to ZeroMQ code -- `populateHeader(void*, unsigned op, Chunk&)`, commmanager/message_service.cpp:2-6 and
// funcs/*/ops/network_ops.hpp: "memcpy(ptr, &op, 4); memcpy(ptr + 4, &chunk, sizeof(chunk));" -- is two memcpys of
// the reference's own struct, repeated here on the reference's Chunk type.
// Output: JSON on stdout (tests/golden/wire_headers.json via oracle/gen_golden.py).
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <string>

#include "common/utils.hpp"

static void hex(const char *label, const unsigned char *p, size_t n, bool comma = true) {
    printf("    \"%s\": \"", label);
    for (size_t i = 0; i < n; ++i) printf("%02x", p[i]);
    printf("\"%s\n", comma ? "," : "");
}

int main() {
    printf("{\n");
    printf("  \"sizes\": {\"HEADER_SIZE\": %zu, \"TENSOR_HDR_SIZE\": %zu, \"TENSOR_NAME_SIZE\": %zu, \"sizeof_Chunk\": %zu,\n",
           (size_t)HEADER_SIZE, (size_t)TENSOR_HDR_SIZE, (size_t)TENSOR_NAME_SIZE, sizeof(Chunk));
    printf("            \"off_localId\": %zu, \"off_globalId\": %zu, \"off_lowBound\": %zu, \"off_upBound\": %zu, \"off_layer\": %zu,\n",
           offsetof(Chunk, localId), offsetof(Chunk, globalId), offsetof(Chunk, lowBound), offsetof(Chunk, upBound), offsetof(Chunk, layer));
    printf("            \"off_dir\": %zu, \"off_epoch\": %zu, \"off_vertex\": %zu},\n", offsetof(Chunk, dir), offsetof(Chunk, epoch),
           offsetof(Chunk, vertex));
    printf("  \"ops\": {\"PUSH\": %d, \"PULL\": %d, \"PULLE\": %d, \"PUSHE\": %d, \"PULLEINFO\": %d, \"FIN\": %d, \"EVAL\": %d, \"RESP\": %d, \"INFO\": %d, \"TERM\": %d,\n",
           (int)OP::PUSH, (int)OP::PULL, (int)OP::PULLE, (int)OP::PUSHE, (int)OP::PULLEINFO, (int)OP::FIN, (int)OP::EVAL, (int)OP::RESP,
           (int)OP::INFO, (int)OP::TERM);
    printf("          \"REQ_VTX_FORWARD\": %d, \"PUSH_VTX_FORWARD\": %d, \"PULL_VTX_BACKWARD\": %d, \"PUSH_EDG_EVAL\": %d,\n",
           (int)OP::REQ_VTX_FORWARD, (int)OP::PUSH_VTX_FORWARD, (int)OP::PULL_VTX_BACKWARD, (int)OP::PUSH_EDG_EVAL);
    printf("          \"ERR_HEADER_FIELD\": %u, \"FORWARD\": %d, \"BACKWARD\": %d},\n", (unsigned)ERR_HEADER_FIELD, (int)PROP_TYPE::FORWARD,
           (int)PROP_TYPE::BACKWARD);
    printf("  \"cases\": [\n");
    struct C { unsigned id, gid, lo, up, layer; PROP_TYPE dir; unsigned ep; bool vtx; unsigned op; const char *name; unsigned f1, f2, f3, f4; };
    const C cs[] = {
        {0, 0, 0, 232965, 0, PROP_TYPE::FORWARD, 1, true, OP::PULL, "w", 0, 602, 128, 0},
        {3, 7, 100, 200, 1, PROP_TYPE::BACKWARD, 42, false, OP::PUSH, "w", 1, 128, 41, 0},
        {255, 4000000000u, 0, 4294967295u, 2, PROP_TYPE::BACKWARD, 0, true, OP::EVAL, "a_i", 2, 64, 1, 9},
        {1, 1, 0, 9430088, 2, PROP_TYPE::FORWARD, 7, true, OP::PUSH, "grad_long", 4294967295u, 25, 64, 3},   // 9-char name: truncated to 8 without NUL
    };
    const int n = (int)(sizeof(cs) / sizeof(cs[0]));
    for (int i = 0; i < n; ++i) {
        const C &c = cs[i];
        Chunk ch;                                   // the aggregate the reference builds (common/utils.cpp:15)
        std::memset(&ch, 0, sizeof(ch));            // (its padding bytes travel uninitialised; zeroed here)
        ch.localId = c.id; ch.globalId = c.gid; ch.lowBound = c.lo; ch.upBound = c.up; ch.layer = c.layer; ch.dir = c.dir;
        ch.epoch = c.ep; ch.vertex = c.vtx;
        unsigned char hdr[HEADER_SIZE];
        std::memset(hdr, 0, sizeof(hdr));
        { char *ptr = (char *)hdr; unsigned op = c.op; memcpy(ptr, &op, sizeof(unsigned)); memcpy(ptr + sizeof(unsigned), &ch, sizeof(ch)); }
        unsigned char th[TENSOR_HDR_SIZE], f5[5 * sizeof(unsigned)];
        std::memset(th, 0, sizeof(th));
        char name8[TENSOR_NAME_SIZE + 1];
        std::memset(name8, 0, sizeof(name8));
        std::strncpy(name8, c.name, TENSOR_NAME_SIZE);   // the reference memcpy's 8 bytes from a std::string buffer
        populateHeader((void *)th, c.op, (const char *)name8, c.f1, c.f2, c.f3, c.f4);        // utils.hpp:230-239
        populateHeader((void *)f5, c.op, c.f1, c.f2, c.f3, c.f4);                              // utils.hpp:220-227
        printf("   {\"localId\": %u, \"globalId\": %u, \"lowBound\": %u, \"upBound\": %u, \"layer\": %u, \"dir\": %d, \"epoch\": %u, \"vertex\": %d,\n",
               ch.localId, ch.globalId, ch.lowBound, ch.upBound, ch.layer, (int)ch.dir, ch.epoch, (int)ch.vertex);
        printf("    \"op\": %u, \"name\": \"%s\", \"f1\": %u, \"f2\": %u, \"f3\": %u, \"f4\": %u,\n", c.op, c.name, c.f1, c.f2, c.f3, c.f4);
        printf("    \"parsed_name\": \"%s\", \"parsed_f1\": %u, \"parsed_f2\": %u,\n", parseName((const char *)th).substr(0, 8).c_str(),
               parse<unsigned>((const char *)th, 3), parse<unsigned>((const char *)th, 4));
        hex("chunk_header", hdr, sizeof(hdr));
        hex("tensor_header", th, sizeof(th));
        hex("fields_header", f5, sizeof(f5), false);
        printf("   }%s\n", i + 1 < n ? "," : "");
    }
    printf("  ]\n}\n");
    return 0;
}
