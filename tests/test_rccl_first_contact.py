"""First-contact hardening of the in-library RCCL leg (BASELINE configs[4]).  mi355_init_multi dlopen()s librccl.so.1 and calls six nccl* entry points
through hand-written prototypes; no RCCL exchange between two DISTINCT devices has ever executed (one-GPU boxes).  What can be pinned without hardware, here: the installed
library exports every symbol the shim resolves, the prototypes and the one enum constant the shim hard-codes are the installed header's, and the two communicator owners (the
library's ncclCommInitAll, torch.distributed's process group) never share a process in bench.py."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORE = open(os.path.join(ROOT, "scroll-prover_amd", "csrc", "lib_core.hip")).read()
MSM = open(os.path.join(ROOT, "scroll-prover_amd", "csrc", "lib_msm.hip")).read()
HEADER = "/opt/rocm/include/rccl/rccl.h"


def shim_symbols():
    return sorted(set(re.findall(r'dlsym\(h, "(nccl\w+)"\)', CORE)))


def test_the_shim_resolves_exactly_the_calls_it_makes():
    assert shim_symbols() == ["ncclAllGather", "ncclCommDestroy", "ncclCommInitAll", "ncclGetErrorString", "ncclGroupEnd", "ncclGroupStart"]
    assert 'dlopen("librccl.so.1"' in CORE and 'dlopen("librccl.so"' in CORE                    # soname first, dev symlink second


def test_installed_librccl_exports_every_symbol_the_shim_needs():
    try:
        lib = ctypes.CDLL("librccl.so.1")
    except OSError as e:
        pytest.skip(f"librccl.so.1 not loadable here: {e}")
    for name in shim_symbols():
        assert hasattr(lib, name), name
    ver = ctypes.c_int(0)
    assert lib.ncclGetVersion(ctypes.byref(ver)) == 0 and ver.value >= 21800, ver.value          # grouped collectives over ncclCommInitAll communicators: NCCL 2.18+ semantics
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    assert lib.ncclGetErrorString(0) == b"no error"                                              # ncclSuccess == 0: the shim treats any non-zero result as a failure


@pytest.mark.skipif(not os.path.exists(HEADER), reason="rccl.h not installed")
def test_hand_written_prototypes_match_the_installed_header():
    hdr = open(HEADER).read()
    norm = lambda s: re.sub(r"\s+", " ", s).strip()
    proto = lambda name: norm(re.search(r"ncclResult_t\s+" + name + r"\(([^;]*)\);", hdr).group(1))
    assert proto("ncclCommInitAll") == "ncclComm_t* comm, int ndev, const int* devlist"
    assert "(int (*)(void **, int, const int *))dlsym(h, \"ncclCommInitAll\")" in CORE            # ncclComm_t is a pointer: void ** is its address
    assert proto("ncclAllGather") == "const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream"
    assert "(int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, \"ncclAllGather\")" in CORE
    assert proto("ncclCommDestroy") == "ncclComm_t comm" and proto("ncclGroupStart") in ("", "void") and proto("ncclGroupEnd") in ("", "void")
    # the one enum value the shim hard-codes: bytes are exchanged as ncclUint8
    assert re.search(r"ncclUint8\s*=\s*1\b", hdr) and "/* ncclUint8 */ 1" in MSM
    # sendcount is PER RANK (96-byte partials x M): the shim passes part_bytes, the receive buffer holds devices x part_bytes
    assert "g_rccl.AllGather(send[s], recv[s], part_bytes," in MSM


def test_library_communicator_and_torch_process_group_never_share_a_process():
    """bench.py --single-process drives N GPUs through mi355_init_multi (the library's own communicator) and must not create torch's; one rank per GPU uses torch's and
    initialises the library with ONE device (no ncclCommInitAll).  Checked on the source: the only init_process_group calls sit under `if world > 1`, and --single-process
    refuses to run under a launcher."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def main()"):]
    guard = body.index("if world > 1:\n        import torch.distributed as dist")
    for m in re.finditer(r"init_process_group\(", body):
        assert m.start() > guard and body.rfind("if world > 1:", 0, m.start()) == guard
    assert 'if single and world > 1:\n        raise SystemExit' in body
    assert "zk.init(dev_index)" in body and "zk.init([i % cnt" in body                            # one device per rank vs the device list of the single process
