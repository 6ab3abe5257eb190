"""the byte formats of the reference's weight-server / Lambda protocol (include/dorylus_wire.h,
dorylus_amd/host/wire.cpp) against bytes written by the reference's OWN serialisation code -- tests/golden/
wire_headers.json comes from oracle/_ref/ref_wire, which includes /root/reference/src/common/utils.hpp (Chunk, OP,
HEADER_SIZE, TENSOR_HDR_SIZE, serialize<>, populateHeader, parseName) unmodified.  CPU only."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Chunk(C.Structure):
    _fields_ = [("local_id", C.c_uint32), ("global_id", C.c_uint32), ("low_bound", C.c_uint32), ("up_bound", C.c_uint32),
                ("layer", C.c_uint32), ("dir", C.c_int32), ("epoch", C.c_uint32), ("vertex", C.c_uint8)]


@pytest.fixture(scope="module")
def lib():
    import dorylus_amd
    if not os.path.exists(dorylus_amd.LIB_PATH):
        pytest.skip("library not built")
    return C.CDLL(dorylus_amd.LIB_PATH)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "wire_headers.json")))


def _chunk(c):
    return Chunk(c["localId"], c["globalId"], c["lowBound"], c["upBound"], c["layer"], c["dir"], c["epoch"], c["vertex"])


def test_sizes_and_op_codes_match_the_reference(gold):
    hdr = open(os.path.join(ROOT, "include", "dorylus_wire.h")).read()
    s = gold["sizes"]
    assert s["HEADER_SIZE"] == 36 and s["TENSOR_HDR_SIZE"] == 28 and s["sizeof_Chunk"] == 32 and s["TENSOR_NAME_SIZE"] == 8
    assert "#define DORY_WIRE_HEADER_SIZE 36u" in hdr and "#define DORY_WIRE_TENSOR_HDR_SIZE 28u" in hdr
    assert [s[k] for k in ("off_localId", "off_globalId", "off_lowBound", "off_upBound", "off_layer", "off_dir", "off_epoch",
                           "off_vertex")] == [0, 4, 8, 12, 16, 20, 24, 28]
    for name, val in gold["ops"].items():
        if name in ("ERR_HEADER_FIELD", "FORWARD", "BACKWARD"):
            continue
        assert f"DORY_OP_{name} = {val}" in hdr, name
    assert gold["ops"]["ERR_HEADER_FIELD"] == 0xFFFFFFFF and gold["ops"]["FORWARD"] == 0 and gold["ops"]["BACKWARD"] == 1


def test_headers_are_byte_identical_to_the_reference(lib, gold):
    for c in gold["cases"]:
        ch = _chunk(c)
        buf = C.create_string_buffer(36)
        lib.dory_wire_pack_chunk_header(buf, C.c_uint32(c["op"]), C.byref(ch))
        assert buf.raw.hex() == c["chunk_header"], c
        th = C.create_string_buffer(28)
        lib.dory_wire_pack_tensor_header(th, C.c_uint32(c["op"]), c["name"].encode(), C.c_uint32(c["f1"]), C.c_uint32(c["f2"]),
                                         C.c_uint32(c["f3"]), C.c_uint32(c["f4"]))
        assert th.raw.hex() == c["tensor_header"], c
        fh = C.create_string_buffer(20)
        lib.dory_wire_pack_fields_header(fh, C.c_uint32(c["op"]), C.c_uint32(c["f1"]), C.c_uint32(c["f2"]), C.c_uint32(c["f3"]),
                                         C.c_uint32(c["f4"]))
        assert fh.raw.hex() == c["fields_header"], c
        # parse the reference's bytes back
        op = C.c_uint32()
        out = Chunk()
        assert lib.dory_wire_parse_chunk_header(bytes.fromhex(c["chunk_header"]), C.byref(op), C.byref(out)) == 0
        assert op.value == c["op"] and [getattr(out, f) for f, _ in Chunk._fields_] == [getattr(ch, f) for f, _ in Chunk._fields_]
        name = C.create_string_buffer(9)
        f = [C.c_uint32() for _ in range(4)]
        assert lib.dory_wire_parse_tensor_header(bytes.fromhex(c["tensor_header"]), C.byref(op), name, *[C.byref(x) for x in f]) == 0
        assert name.value.decode() == c["parsed_name"] and f[0].value == c["parsed_f1"] and f[1].value == c["parsed_f2"]
        assert [x.value for x in f] == [c["f1"], c["f2"], c["f3"], c["f4"]]


def test_whole_messages_are_the_reference_frames(lib, gold):
    """pull request / push / acc-loss as multi-part frame lists: every header frame equals what the reference's code
    writes for the same fields (golden cases), payload frames are the dense fp32 tensors."""
    c = gold["cases"][1]            # op PUSH, name "w", f1 = layer 1, rows 128, cols 41
    ch = _chunk(c)
    names = (C.c_char_p * 2)(b"w", b"a_i")
    out = (C.c_uint8 * 65536)()
    off = (C.c_size_t * 8)()
    n = lib.dory_wire_build_pull(C.byref(ch), names, 2, out, 65536, off, 7)
    assert n == 3 and list(off[:4]) == [0, 36, 64, 92]
    raw = bytes(out)
    hdr = C.create_string_buffer(36)
    lib.dory_wire_pack_chunk_header(hdr, 17, C.byref(ch))                     # OP::PULL
    assert raw[:36] == hdr.raw
    assert struct.unpack_from("<I", raw, 36)[0] == ch.local_id and raw[40:48] == b"w" + b"\0" * 7   # tensor header: op = chunk.localId
    assert raw[64 + 4:64 + 12] == b"a_i" + b"\0" * 5
    # push: the tensor header is the golden one (op PUSH, name, layer, rows, cols)
    W = np.arange(128 * 41, dtype=np.float32).reshape(128, 41)
    rows, cols = (C.c_uint32 * 1)(128), (C.c_uint32 * 1)(41)
    data = (C.POINTER(C.c_float) * 1)(W.ctypes.data_as(C.POINTER(C.c_float)))
    n = lib.dory_wire_build_push(C.byref(ch), (C.c_char_p * 1)(b"w"), rows, cols, data, 1, out, 65536, off, 7)
    assert n == 3 and list(off[:4]) == [0, 36, 64, 64 + 128 * 41 * 4]
    raw = bytes(out)
    assert raw[:36].hex() == c["chunk_header"] and raw[36:64].hex() == c["tensor_header"]
    assert np.array_equal(np.frombuffer(raw[64:64 + W.nbytes], np.float32).reshape(128, 41), W)
    # too small an output buffer / too few frame slots: refused, nothing half-built is reported
    assert lib.dory_wire_build_push(C.byref(ch), (C.c_char_p * 1)(b"w"), rows, cols, data, 1, out, 1000, off, 7) == -1
    assert lib.dory_wire_build_push(C.byref(ch), (C.c_char_p * 1)(b"w"), rows, cols, data, 1, out, 65536, off, 2) == -1
    # acc / loss (MessageService::sendAccloss): Chunk{nodeId, nodeId, 0, vtcsCnt, 1, FORWARD, epoch, true}
    n = lib.dory_wire_build_accloss(3, 9, 1000, C.c_float(0.75), C.c_float(1.25), out, 65536, off, 7)
    assert n == 2 and list(off[:3]) == [0, 36, 44]
    raw = bytes(out)
    assert struct.unpack_from("<I8I", raw, 0)[:8] == (22, 3, 3, 0, 1000, 1, 0, 9) and raw[32] == 1
    assert struct.unpack_from("<2f", raw, 36) == (0.75, 1.25)
    # pull reply: (header with rows, cols at unsigned offsets 3, 4; payload size must agree), server error code
    rep = C.create_string_buffer(28)
    lib.dory_wire_pack_tensor_header(rep, 23, b"w", 602, 128, 0, 0)
    name = C.create_string_buffer(9)
    r, cc = C.c_uint32(), C.c_uint32()
    assert lib.dory_wire_parse_pull_reply(rep, C.c_size_t(602 * 128 * 4), name, C.byref(r), C.byref(cc)) == 0
    assert (name.value, r.value, cc.value) == (b"w", 602, 128)
    assert lib.dory_wire_parse_pull_reply(rep, C.c_size_t(10), name, C.byref(r), C.byref(cc)) == -1
    lib.dory_wire_pack_tensor_header(rep, 0xFFFFFFFF, b"w", 0, 0, 0, 0)
    assert lib.dory_wire_parse_pull_reply(rep, C.c_size_t(0), name, C.byref(r), C.byref(cc)) == 1
