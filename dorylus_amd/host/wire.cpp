// wire.cpp -- byte formats of the reference's weight-server / Lambda protocol (include/dorylus_wire.h).  Pinned byte
// for byte to the reference's own serialisation code through tests/golden/wire_headers.json (oracle/ref_wire.cpp).
#include <cstring>

#include "../../include/dorylus_wire.h"

namespace {
void put32(uint8_t *p, size_t word, uint32_t v) { std::memcpy(p + word * 4, &v, 4); }   // serialize<unsigned>(buf, offset, val)
uint32_t get32(const uint8_t *p, size_t word) { uint32_t v; std::memcpy(&v, p + word * 4, 4); return v; }

struct Frames {
    uint8_t *out; size_t cap; size_t *off; uint32_t max; uint32_t n = 0; size_t used = 0; bool ok = true;
    uint8_t *add(size_t bytes) {
        if (!ok || n >= max || used + bytes > cap) { ok = false; return nullptr; }
        uint8_t *p = out + used;
        off[n] = used;
        used += bytes;
        off[++n] = used;
        return p;
    }
};
}  // namespace

extern "C" {

void dory_wire_pack_chunk_header(void *buf, uint32_t op, const dory_wire_chunk *c) {
    uint8_t *p = static_cast<uint8_t *>(buf);
    std::memset(p, 0, DORY_WIRE_HEADER_SIZE);      // the reference leaves Chunk's 3 padding bytes as they were: zero here
    put32(p, 0, op);
    put32(p, 1, c->local_id); put32(p, 2, c->global_id); put32(p, 3, c->low_bound); put32(p, 4, c->up_bound);
    put32(p, 5, c->layer); put32(p, 6, (uint32_t)c->dir); put32(p, 7, c->epoch);
    p[32] = c->vertex ? 1 : 0;
}

int dory_wire_parse_chunk_header(const void *buf, uint32_t *op, dory_wire_chunk *c) {
    if (!buf || !op || !c) return -1;
    const uint8_t *p = static_cast<const uint8_t *>(buf);
    *op = get32(p, 0);
    c->local_id = get32(p, 1); c->global_id = get32(p, 2); c->low_bound = get32(p, 3); c->up_bound = get32(p, 4);
    c->layer = get32(p, 5); c->dir = (int32_t)get32(p, 6); c->epoch = get32(p, 7);
    c->vertex = p[32] ? 1 : 0;
    return 0;
}

void dory_wire_pack_tensor_header(void *buf, uint32_t op, const char *name, uint32_t f1, uint32_t f2, uint32_t f3, uint32_t f4) {
    uint8_t *p = static_cast<uint8_t *>(buf);
    std::memset(p, 0, DORY_WIRE_TENSOR_HDR_SIZE);
    put32(p, 0, op);
    if (name) std::strncpy(reinterpret_cast<char *>(p + 4), name, DORY_WIRE_TENSOR_NAME_SIZE);
    put32(p, 3, f1); put32(p, 4, f2); put32(p, 5, f3); put32(p, 6, f4);
}

int dory_wire_parse_tensor_header(const void *buf, uint32_t *op, char name9[9], uint32_t *f1, uint32_t *f2, uint32_t *f3,
                                  uint32_t *f4) {
    if (!buf) return -1;
    const uint8_t *p = static_cast<const uint8_t *>(buf);
    if (op) *op = get32(p, 0);
    if (name9) { std::memcpy(name9, p + 4, 8); name9[8] = 0; }
    if (f1) *f1 = get32(p, 3);
    if (f2) *f2 = get32(p, 4);
    if (f3) *f3 = get32(p, 5);
    if (f4) *f4 = get32(p, 6);
    return 0;
}

void dory_wire_pack_fields_header(void *buf, uint32_t op, uint32_t f1, uint32_t f2, uint32_t f3, uint32_t f4) {
    uint8_t *p = static_cast<uint8_t *>(buf);
    put32(p, 0, op); put32(p, 1, f1); put32(p, 2, f2); put32(p, 3, f3); put32(p, 4, f4);
}

int dory_wire_parse_fields_header(const void *buf, uint32_t *op, uint32_t *f1, uint32_t *f2, uint32_t *f3, uint32_t *f4) {
    if (!buf) return -1;
    const uint8_t *p = static_cast<const uint8_t *>(buf);
    if (op) *op = get32(p, 0);
    if (f1) *f1 = get32(p, 1);
    if (f2) *f2 = get32(p, 2);
    if (f3) *f3 = get32(p, 3);
    if (f4) *f4 = get32(p, 4);
    return 0;
}

// reqTensors (commmanager/message_service.cpp:40-60): header(PULL, chunk), then one tensor header per name whose op
// field carries chunk.localId
int dory_wire_build_pull(const dory_wire_chunk *chunk, const char *const *names, uint32_t n, uint8_t *out, size_t cap,
                         size_t *frame_off, uint32_t max_frames) {
    if (!chunk || !out || !frame_off || (n && !names)) return -1;
    Frames f{out, cap, frame_off, max_frames};
    if (uint8_t *p = f.add(DORY_WIRE_HEADER_SIZE)) dory_wire_pack_chunk_header(p, DORY_OP_PULL, chunk);
    for (uint32_t i = 0; i < n && f.ok; ++i)
        if (uint8_t *p = f.add(DORY_WIRE_TENSOR_HDR_SIZE)) dory_wire_pack_tensor_header(p, chunk->local_id, names[i], 0, 0, 0, 0);
    return f.ok ? (int)f.n : -1;
}

// sendTensors (message_service.cpp:83-108): header(PUSH, chunk), then per matrix a tensor header
// (PUSH, name, chunk.layer, rows, cols) and the raw fp32 data
int dory_wire_build_push(const dory_wire_chunk *chunk, const char *const *names, const uint32_t *rows, const uint32_t *cols,
                         const float *const *data, uint32_t n, uint8_t *out, size_t cap, size_t *frame_off,
                         uint32_t max_frames) {
    if (!chunk || !out || !frame_off || (n && (!names || !rows || !cols || !data))) return -1;
    Frames f{out, cap, frame_off, max_frames};
    if (uint8_t *p = f.add(DORY_WIRE_HEADER_SIZE)) dory_wire_pack_chunk_header(p, DORY_OP_PUSH, chunk);
    for (uint32_t i = 0; i < n && f.ok; ++i) {
        if (uint8_t *p = f.add(DORY_WIRE_TENSOR_HDR_SIZE))
            dory_wire_pack_tensor_header(p, DORY_OP_PUSH, names[i], chunk->layer, rows[i], cols[i], 0);
        const size_t bytes = (size_t)rows[i] * cols[i] * sizeof(float);
        if (uint8_t *p = f.add(bytes)) std::memcpy(p, data[i], bytes);
    }
    return f.ok ? (int)f.n : -1;
}

// MessageService::sendAccloss (message_service.cpp:225-245)
int dory_wire_build_accloss(uint32_t node_id, uint32_t epoch, uint32_t vtcs_cnt, float acc, float loss, uint8_t *out, size_t cap,
                            size_t *frame_off, uint32_t max_frames) {
    if (!out || !frame_off) return -1;
    Frames f{out, cap, frame_off, max_frames};
    const dory_wire_chunk c{node_id, node_id, 0, vtcs_cnt, 1, 0, epoch, 1};
    if (uint8_t *p = f.add(DORY_WIRE_HEADER_SIZE)) dory_wire_pack_chunk_header(p, DORY_OP_EVAL, &c);
    if (uint8_t *p = f.add(2 * sizeof(float))) { std::memcpy(p, &acc, 4); std::memcpy(p + 4, &loss, 4); }
    return f.ok ? (int)f.n : -1;
}

// recvTensor (message_service.cpp:17-38): resp code, name, rows at unsigned offset 3, cols at offset 4
int dory_wire_parse_pull_reply(const void *hdr, size_t payload_bytes, char name9[9], uint32_t *rows, uint32_t *cols) {
    if (!hdr || !rows || !cols) return -1;
    uint32_t op, r, c;
    dory_wire_parse_tensor_header(hdr, &op, name9, &r, &c, nullptr, nullptr);
    if (op == DORY_WIRE_ERR_HEADER_FIELD) return 1;
    if ((size_t)r * c * sizeof(float) != payload_bytes) return -1;
    *rows = r;
    *cols = c;
    return 0;
}

}  // extern "C"
