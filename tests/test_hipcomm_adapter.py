"""INTEGRATION.md's reference-side glue (tests/hipcomm_adapter.cpp) parses against the reference's own headers:
`HIPComm : ResourceComm` overrides what resource_comm.hpp:13-28 declares, every Engine / Graph / Chunk member the
adapter reads exists with a type the C-ABI accepts, and include/dorylus_hip.h is valid C++11 beside the reference's
unscoped enums.  Syntax check only (g++ -fsyntax-only); runs where /root/reference is present (the build container).
The one header the reference pulls in that this image lacks, cblas.h, is given as an EMPTY file in a temporary
directory: nothing of it is used by the declarations the adapter touches (no reference code is built or run)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"


def _zmq_include():
    for d in ("/usr/include", "/usr/local/include", "/opt/conda/include"):
        if os.path.exists(os.path.join(d, "zmq.h")):      # libzmq C header (the reference vendors zmq.hpp itself)
            return d
    return None


def test_hipcomm_adapter_parses_against_reference_headers(tmp_path):
    if not os.path.isdir(REF) or shutil.which("g++") is None:
        pytest.skip("reference sources or g++ not present (GPU box)")
    zmq = _zmq_include()
    if zmq is None:
        pytest.skip("zmq.h not installed")
    (tmp_path / "cblas.h").write_text("")
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Wno-unused", "-I", str(tmp_path), "-I", zmq,
           "-I", os.path.join(REF, "graph-server"), "-I", os.path.join(REF, "common"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "hipcomm_adapter.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    # the multi-node wiring is really there (round 2 computed the plan and threw it away): plan, communicator, and the
    # weight-server variant over the wire formats
    src = open(os.path.join(ROOT, "tests", "hipcomm_adapter.cpp")).read()
    for call in ("dory_halo_plan(", "dory_comm_unique_id(", "dory_comm_init(", "dory_weight_grad_get(", "dory_wire_build_push(",
                 "dory_wire_build_pull(", "dory_wire_parse_pull_reply(", "dory_weight_set(", "controlPushOut(", "controlPullIn("):
        assert call in src, call
    assert "(void)" not in src
    # the contract the adapter relies on is still what the reference declares
    rc = open(os.path.join(REF, "graph-server", "commmanager", "resource_comm.hpp")).read()
    assert "virtual void NNCompute(Chunk &chunk) = 0;" in rc and "void NNRecvCallback(Engine *engine, Chunk &chunk);" in rc
