/* dorylus_wire.h -- the byte formats of the reference's weight-server / Lambda protocol:
 * what a HIP graph server has to put on a ZeroMQ socket to talk to an UNMODIFIED Dorylus weight server, so that it can
 * join a mixed deployment.  Formats only -- the socket stays the caller's (the product library does not link ZeroMQ);
 * together with dory_comm_set_host_transport / dory_weight_get / dory_weight_grad_get / dory_weight_set (dorylus_hip.h)
 * this is the whole seam.  All little-endian, packed exactly as the reference writes them.  Exercised over real ZeroMQ
 * sockets against a peer that reads the frames with the reference's own parse code: tests/test_wire_loopback.py.
 *
 *   chunk header   36 B  = u32 op | Chunk                (populateHeader(void*, op, Chunk&): commmanager/
 *                                                          message_service.cpp:2-6; HEADER_SIZE, common/utils.hpp:31)
 *   Chunk          32 B  = u32 localId, globalId, lowBound, upBound, layer | i32 dir | u32 epoch | u8 vertex | 3 pad
 *                                                         (common/utils.hpp:64-75)
 *   tensor header  28 B  = u32 op | char name[8] | u32 f1 | u32 f2 | u32 f3 | u32 f4
 *                                                         (populateHeader(void*, op, name, ...): common/utils.hpp:230-239;
 *                                                          the fields sit at unsigned offsets 3..6)
 *   fields header  20 B  = u32 op | u32 f1..f4            (common/utils.hpp:211-227)
 *
 * Messages on the graph-server <-> weight-server path (commmanager/message_service.cpp:17-110, weight-server/
 * serverworker.cpp:30-208), each a ZeroMQ multi-part message whose frames are:
 *   pull request :  chunk header (op PULL)  |  per tensor: tensor header (op = chunk.localId, name)
 *   pull reply   :  per tensor: tensor header (op = response code or ERR_HEADER_FIELD, name, f1 = rows, f2 = cols) | rows*cols f32
 *   push         :  chunk header (op PUSH)  |  per tensor: tensor header (op PUSH, name, f1 = layer, f2 = rows, f3 = cols) | rows*cols f32
 *   acc / loss   :  chunk header (op EVAL, Chunk{nodeId, nodeId, 0, vtcsCnt, 1, FORWARD, epoch, true}) | f32 acc, f32 loss
 */
#ifndef DORYLUS_WIRE_H
#define DORYLUS_WIRE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DORY_WIRE_HEADER_SIZE 36u       /* HEADER_SIZE */
#define DORY_WIRE_TENSOR_HDR_SIZE 28u   /* TENSOR_HDR_SIZE */
#define DORY_WIRE_FIELDS_HDR_SIZE 20u
#define DORY_WIRE_TENSOR_NAME_SIZE 8u
#define DORY_WIRE_ERR_HEADER_FIELD 0xFFFFFFFFu

/* enum OP (common/utils.hpp:35-45), the values on the wire */
enum dory_wire_op {
    DORY_OP_REQ_VTX_FORWARD = 0, DORY_OP_PUSH_VTX_FORWARD = 1, DORY_OP_PULL_VTX_FORWARD = 2, DORY_OP_REQ_VTX_BACKWARD = 3,
    DORY_OP_PUSH_VTX_BACKWARD = 4, DORY_OP_PULL_VTX_BACKWARD = 5, DORY_OP_PULL_VTX_EVAL = 6, DORY_OP_PUSH_VTX_EVAL = 7,
    DORY_OP_REQ_EDG_FORWARD = 8, DORY_OP_PUSH_EDG_FORWARD = 9, DORY_OP_PULL_EDG_FORWARD = 10, DORY_OP_REQ_EDG_BACKWARD = 11,
    DORY_OP_PUSH_EDG_BACKWARD = 12, DORY_OP_PULL_EDG_BACKWARD = 13, DORY_OP_PULL_EDG_EVAL = 14, DORY_OP_PUSH_EDG_EVAL = 15,
    DORY_OP_PUSH = 16, DORY_OP_PULL = 17, DORY_OP_PULLE = 18, DORY_OP_PUSHE = 19, DORY_OP_PULLEINFO = 20, DORY_OP_FIN = 21,
    DORY_OP_EVAL = 22, DORY_OP_RESP = 23, DORY_OP_INFO = 24, DORY_OP_TERM = 25
};

/* struct Chunk (common/utils.hpp:64-75), field for field; dir: 0 = FORWARD, 1 = BACKWARD */
struct dory_wire_chunk {
    uint32_t local_id, global_id, low_bound, up_bound, layer;
    int32_t dir;
    uint32_t epoch;
    uint8_t vertex;
};

/* pack: `buf` must hold the header size; parse: returns 0, or -1 on a null argument */
void dory_wire_pack_chunk_header(void *buf36, uint32_t op, const struct dory_wire_chunk *chunk);
int dory_wire_parse_chunk_header(const void *buf36, uint32_t *op, struct dory_wire_chunk *chunk);
/* `name` is copied up to 8 bytes and NUL-padded (the reference memcpy's 8 bytes of a std::string) */
void dory_wire_pack_tensor_header(void *buf28, uint32_t op, const char *name, uint32_t f1, uint32_t f2, uint32_t f3, uint32_t f4);
int dory_wire_parse_tensor_header(const void *buf28, uint32_t *op, char name9[9], uint32_t *f1, uint32_t *f2, uint32_t *f3,
                                  uint32_t *f4);
void dory_wire_pack_fields_header(void *buf20, uint32_t op, uint32_t f1, uint32_t f2, uint32_t f3, uint32_t f4);
int dory_wire_parse_fields_header(const void *buf20, uint32_t *op, uint32_t *f1, uint32_t *f2, uint32_t *f3, uint32_t *f4);

/* Whole messages as frame lists: the frames are written back to back into `out` (capacity `cap` bytes), frame i is
 * out[frame_off[i] .. frame_off[i+1]); `frame_off` needs max_frames + 1 entries.  Return the number of frames, or -1
 * when `out` / `frame_off` are too small or an argument is null.  Tensor payloads are dense row-major f32 (host
 * layout, what dory_weight_get / dory_weight_grad_get return). */
int dory_wire_build_pull(const struct dory_wire_chunk *chunk, const char *const *names, uint32_t n_tensors, uint8_t *out,
                         size_t cap, size_t *frame_off, uint32_t max_frames);
int dory_wire_build_push(const struct dory_wire_chunk *chunk, const char *const *names, const uint32_t *rows,
                         const uint32_t *cols, const float *const *data, uint32_t n_tensors, uint8_t *out, size_t cap,
                         size_t *frame_off, uint32_t max_frames);
int dory_wire_build_accloss(uint32_t node_id, uint32_t epoch, uint32_t vtcs_cnt, float acc, float loss, uint8_t *out,
                            size_t cap, size_t *frame_off, uint32_t max_frames);
/* one (tensor header, payload) pair of a pull reply: 0 = ok (rows/cols/name filled, payload size checked),
 * 1 = the server answered ERR_HEADER_FIELD, -1 = malformed */
int dory_wire_parse_pull_reply(const void *hdr28, size_t payload_bytes, char name9[9], uint32_t *rows, uint32_t *cols);

#ifdef __cplusplus
}
#endif
#endif
