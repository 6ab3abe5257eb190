"""SURVEY 8 f-4 over a real socket: the frame lists include/dorylus_wire.h emits travel through ZeroMQ (DEALER -> ROUTER, tcp
loopback) to a peer that reads them with the REFERENCE's own header-only code (common/utils.hpp parse<> / parseName /
populateHeader and the vendored common/zmq.hpp; oracle/ref_wire_peer.cpp, built by oracle/Makefile from /root/reference in
place over the image's libzmq), and the peer's replies -- built by the reference's populateHeader -- are read by
dory_wire_parse_pull_reply.  The conversation is the graph server's: PULL "w" per layer (prefetchWeightsMatrix), an unknown
tensor (ERR_HEADER_FIELD), PUSH of a gradient (sendWeightUpdate) read back bit for bit, acc/loss (sendAccloss), TERM.
Not covered: the real weight server (needs boost + cblas.h, absent here) -- its averaging and Adam never run.
Skipped where the peer binary or libzmq is missing (the GPU box only has what travelled under oracle/_ref)."""
import json
import os
import socket
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEER = os.path.join(ROOT, "oracle", "_ref", "ref_wire_peer")
ZMQLIB = os.path.join(ROOT, "oracle", "_ref", "zmqlib")
ZMQ_H = "/opt/conda/include/zmq.h"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.skipif(not (os.path.exists(PEER) and os.path.exists(os.path.join(ZMQLIB, "libzmq.so.5")) and os.path.exists(ZMQ_H)),
                    reason="needs oracle/_ref/ref_wire_peer (reference tree + libzmq present at build time) and zmq.h")
def test_wire_frames_over_zeromq_to_reference_parse_code(tmp_path):
    client = str(tmp_path / "wire_client")
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-I/opt/conda/include", os.path.join(ROOT, "tests", "wire_loopback_client.cpp"),
                        os.path.join(ROOT, "dorylus_amd", "host", "wire.cpp"), os.path.join(ZMQLIB, "libzmq.so.5"),
                        "-Wl,-rpath," + ZMQLIB, "-lpthread", "-o", client], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    port = _free_port()
    peer = subprocess.Popen([PEER, str(port)], stdout=subprocess.PIPE, text=True)
    try:
        c = subprocess.run([client, str(port)], capture_output=True, text=True, timeout=60)
        assert c.returncode == 0, c.stdout + c.stderr
        out, _ = peer.communicate(timeout=30)          # the peer leaves after OP::TERM
    finally:
        if peer.poll() is None:
            peer.kill()
    mine, theirs = json.loads(c.stdout), json.loads(out)
    # product side: replies built by the reference's populateHeader parsed by dory_wire_parse_pull_reply
    assert mine["fails"] == 0 and mine["err_reply_rc"] == 1 and mine["readback_bit_identical"] is True
    assert [(p["rows"], p["cols"], p["name"], p["payload_ok"]) for p in mine["pulled"]] == [(602, 128, "w", True), (128, 41, "w", True)]
    # reference side: what its parse code saw in the product's frames
    assert (theirs["pulls"], theirs["pushes"], theirs["evals"]) == (4, 1, 1)
    m = theirs["messages"]
    assert [x["op"] for x in m] == ["PULL", "PULL", "PULL", "PUSH", "PULL", "EVAL"]
    assert [x["name"] for x in m[:5]] == ["w", "w", "nosuch", "w", "w_upd"]
    assert all(x["globalId"] == 3 and x["epoch"] == 5 for x in m)
    assert [x["layer"] for x in m[:2]] == [0, 1] and all(x["dir"] == 0 and x["hdr_size"] == 36 and x["req_op"] == 0 for x in (m[0], m[1], m[2], m[4]))
    push = m[3]
    assert push["dir"] == 1 and push["hdr_op"] == 16 and (push["f1"], push["f2"], push["f3"]) == (1, 128, 41)
    assert push["payload_bytes"] == 128 * 41 * 4 and push["weighted_sum"] == mine["pushed_weighted_sum"]
    ev = m[5]
    assert (ev["localId"], ev["upBound"], ev["layer"], ev["vertex"]) == (3, 153431, 1, 1) and ev["acc"] == 0.9375 and ev["loss"] == 1.25
