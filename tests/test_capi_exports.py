"""CPU: the C-ABI library loads, exports every symbol include/gpann.h declares, and fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared(header="gpann.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpann_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import pygpann
    pygpann.build()
    L = ctypes.CDLL(pygpann.LIB_PATH)
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/gpann.h but not exported"
    assert set(pygpann.EXPORTS) == set(names)


def test_comm_library_exports_every_declared_symbol():
    """include/gpann_comm.h (the multi-GPU exchange layer) against libgpann_comm.so; loading it needs RCCL but no GPU"""
    import pygpann
    pygpann.build()
    from pygpann import comm
    L = comm.lib()
    names = _declared("gpann_comm.h")
    assert len(names) >= 17
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/gpann_comm.h but not exported"
    assert set(comm.EXPORTS) == set(names)


def test_host_mirror_library_loads():
    import pygpann
    pygpann.build()
    from pygpann import host
    L = host.lib()
    for n in ("gpann_host_router_train", "gpann_host_kmeans", "gpann_host_distance"):
        assert hasattr(L, n)


def test_no_device_is_an_error_not_a_fallback():
    import pygpann
    pygpann.build()
    if pygpann.device_count() > 0:
        pytest.skip("a GPU is present")
    pts = np.zeros((4, 4), np.float32)
    with pytest.raises(pygpann.GpannError) as e:
        pygpann.Shards(pts, [np.arange(4, dtype=np.uint32)], 0, 0)
    assert e.value.status == pygpann.ERR_NO_DEVICE
    flat = dict(d=4, S=1, node_shard=np.zeros(1, np.int32), node_cent_first=np.array([0, 1], np.uint32), node_child_first=np.zeros(1, np.uint32),
                node_child_count=np.zeros(1, np.uint32), shard_root=np.zeros(1, np.uint32), centroids=np.zeros((1, 4), np.float32))
    with pytest.raises(pygpann.GpannError) as e:
        pygpann.Router(flat, 0, 0)
    assert e.value.status == pygpann.ERR_NO_DEVICE


def test_merge_dedup_is_host_side(orc):
    """gpann_merge_dedup (DedupNeighbors, small_scale_queries.cpp:15-23) needs no device"""
    import pygpann
    rng = np.random.default_rng(1)
    nq, nprobe, k = 30, 4, 5
    ids = rng.integers(0, 40, (nq, nprobe, k)).astype(np.uint32)
    dist = (ids % 7).astype(np.float32)             # equal ids carry equal distances, as in the real path
    ids[3, 2, 3:] = 0xFFFFFFFF
    dist[3, 2, 3:] = np.inf
    for used in (1, 3, 4):
        md, mi, mc = pygpann.merge_dedup(dist, ids, used, k)
        flat_d, flat_i, offs = [], [], [0]
        for q in range(nq):
            v = ids[q, :used].reshape(-1) != 0xFFFFFFFF
            flat_d.extend(dist[q, :used].reshape(-1)[v])
            flat_i.extend(ids[q, :used].reshape(-1)[v])
            offs.append(len(flat_d))
        od, oi, oc = orc.dedup_neighbors(np.asarray(flat_d, np.float32), np.asarray(flat_i, np.uint32), np.asarray(offs, np.uint64), k)
        assert np.array_equal(mi, oi) and np.array_equal(md, od) and np.array_equal(mc, oc)


def test_options_table_and_no_getenv_on_the_call_paths():
    """every knob is a named option of the handle (include/gpann.h "Options"): the table enumerates, and the kernel translation units read the
    environment in ONE place — the snapshot a handle takes when it is created — plus the debug-output switches"""
    import pygpann
    pygpann.build()
    opts = pygpann.options()
    names = [n for n, _, _ in opts]
    assert len(names) == len(set(names)) >= 40 and all(n.startswith("GPANN_") for n in names) and all(h for _, _, h in opts)
    for must in ("GPANN_FILTER_HEAD_ROWS", "GPANN_ROUTER_BOUND_TABLE", "GPANN_ROUTER_HOST_REDO", "GPANN_FILTER_Q8_FROM_PASS", "GPANN_OVERLAP_REST_ITEMS", "GPANN_CELL_ROWS"):
        assert must in names
    assert dict((n, b) for n, b, _ in opts)["GPANN_CELL_ROWS"] is True and dict((n, b) for n, b, _ in opts)["GPANN_FILTER_HEAD_ROWS"] is False
    csrc = os.path.join(ROOT, "gp-ann_amd", "csrc")
    allowed = {"GPANN_CELLS_DEBUG", "GPANN_ROUTER_DEBUG", "GPANN_VERIFY_REPORT", "GPANN_SCAN_DEBUG_HOST"}     # debug output only
    for fn in sorted(os.listdir(csrc)):
        if fn == "comm.hip":        # (the communicator reads MASTER_ADDR / ports when it is CREATED: its own, creation-time, contract)
            continue
        text = open(os.path.join(csrc, fn)).read()
        for m in re.finditer(r"getenv\(([^)]*)\)", text):
            arg = m.group(1).strip()
            if arg.startswith('"'):
                assert arg.strip('"') in allowed, f"{fn}: getenv({arg}) — make it an option of the handle (shards.hip kKnobs)"
            else:
                assert fn == "shards.hip", f"{fn}: getenv({arg})"    # Options::SnapshotEnvironment
    # every option name a translation unit asks its handle for is in the table (a typo would silently read the default)
    for fn in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, fn)).read()
        for m in re.finditer(r"opt\.(?:Get|U32|IsSet|IsZero)\(\"(GPANN_[A-Z0-9_]+)\"", text):
            assert m.group(1) in names, f"{fn}: option {m.group(1)} is not in the table"


def test_matrix_core_kernels_claim_their_simds():
    """every kernel that issues matrix instructions calls TakeWholeRegisterFile() first (csrc/common.h): wavefronts of other launches that shared a SIMD
    with matrix-core wavefronts were measured to end with wrong registers (profiles/r05/overlap_race.md) — a new such kernel must not forget it"""
    csrc = os.path.join(ROOT, "gp-ann_amd", "csrc")
    assert 'asm volatile("" ::: "v255")' in open(os.path.join(csrc, "common.h")).read()
    found = 0
    for fn in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, fn)).read()
        # kernels: from a __global__ signature to the next one (or the end of the file)
        starts = [m.start() for m in re.finditer(r"__global__\s+void", text)]
        for a, b in zip(starts, starts[1:] + [len(text)]):
            body = text[a:b]
            if "__builtin_amdgcn_mfma_" in body:
                found += 1
                name = re.search(r"\)\s*(\w+)\s*\(|void\s+(?:__launch_bounds__\([^)]*\)\s*)?(\w+)\s*\(", body)
                assert "TakeWholeRegisterFile();" in body[: body.index("__builtin_amdgcn_mfma_")], f"{fn}: a matrix-core kernel without TakeWholeRegisterFile() ({name.group(0)[:60] if name else '?'})"
    assert found >= 2    # FilterKernelCells, RouterBoundKernel


def test_matrix_core_kernels_are_allocated_the_whole_register_file():
    """the BUILT library, not the source text: every symbol of every gfx950 code object in libgpann_hip.so whose disassembly holds a matrix instruction is
    a kernel with vgpr_count 256 and 512-thread workgroups (tools/kernel_resources.py --check-matrix-kernels) — the property GPANN_SCAN_OVERLAP's
    exactness rests on (profiles/r05/overlap_race.md), which a compiler change or a new template instance could silently take away"""
    import importlib.util
    import pygpann
    pygpann.build()
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    ok, bad = kr.check_matrix_kernels(pygpann.LIB_PATH)
    assert not bad, "\n".join(bad)
    mnemonics = {m for _, _, ms in ok for m in ms}
    assert "v_mfma_i32_32x32x32_i8" in mnemonics and "v_mfma_f32_32x32x16_bf16" in mnemonics   # (the check saw the filter and the router kernels)
    assert len(ok) >= 60
    # ... and no packed fp32 arithmetic anywhere (Makefile: -target-feature -packed-fp32-ops): beside another kernel's or process's int8 matrix
    # wavefronts it was measured to go wrong in lanes 48 - 63 — scalar fp32, the same IEEE operations, is not disturbed (profiles/r06/NOTES.md §2)
    packed = kr.check_no_packed_fp32(pygpann.LIB_PATH)
    assert not packed, "\n".join(packed[:10])
