"""CPU: the binary file formats the hot path's callers read and write (reference src/points_io.cpp:14-179) through the
host mirror: .fbin / .u8bin / .i8bin point files (bytes are widened to float; `size` overrides n for .fbin only) and the
big-ann-benchmarks ground-truth layout."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "gp-ann_amd"))
from pygpann import host


def _write(path, a):
    with open(path, "wb") as f:
        np.array(a.shape, np.uint32).tofile(f)
        a.tofile(f)


def test_fbin_write_read_and_size_override(tmp_path):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((1234, 17)).astype(np.float32)
    a[0, 0], a[1, 1] = np.float32(-0.0), np.float32(1e-42)          # signed zero and a denormal survive bit for bit
    path = str(tmp_path / "p.fbin")
    host.write_points(path, a)
    raw = np.fromfile(path, np.uint32)
    assert raw[0] == 1234 and raw[1] == 17 and os.path.getsize(path) == 8 + a.size * 4
    assert np.array_equal(raw[2:], a.reshape(-1).view(np.uint32))
    back = host.read_points(path)
    assert back.shape == a.shape and np.array_equal(back.view(np.uint32), a.view(np.uint32))
    head = host.read_points(path, size=100)                            # points_io.cpp:25-27
    assert head.shape == (100, 17) and np.array_equal(head.view(np.uint32), a[:100].view(np.uint32))


@pytest.mark.parametrize("ext,dtype", [(".u8bin", np.uint8), (".i8bin", np.int8)])
def test_byte_files_are_widened_to_float(tmp_path, ext, dtype):
    rng = np.random.default_rng(2)
    info = np.iinfo(dtype)
    a = rng.integers(info.min, info.max + 1, size=(777, 100), dtype=dtype)   # an odd shape: slab boundaries fall inside rows
    a[0, :4] = [info.min, info.max, 0, 1]
    path = str(tmp_path / ("p" + ext))
    _write(path, a)
    back = host.read_points(path)
    assert back.dtype == np.float32 and back.shape == a.shape and np.array_equal(back, a.astype(np.float32))
    # the reference ignores `size` for byte files (points_io.cpp:117-120)
    assert host.read_points(path, size=10).shape == a.shape


def test_unknown_extension_is_an_error(tmp_path):
    path = str(tmp_path / "p.bin")
    _write(path, np.zeros((2, 2), np.float32))
    with pytest.raises(RuntimeError, match="Invalid file ending"):
        host.read_points(path)


def test_ground_truth_layout_roundtrip(tmp_path):
    rng = np.random.default_rng(3)
    nq, k = 37, 10
    dist = np.sort(rng.random((nq, k)).astype(np.float32), axis=1)
    ids = rng.integers(0, 2 ** 32 - 1, size=(nq, k), dtype=np.uint32)
    path = str(tmp_path / "gt.bin")
    od, oi = host.ground_truth_roundtrip(path, dist, ids)
    assert np.array_equal(od.view(np.uint32), dist.view(np.uint32)) and np.array_equal(oi, ids)
    raw = np.fromfile(path, np.uint32)
    # u32 nq, u32 k, nq*k ids, then nq*k distances (points_io.cpp:136-179)
    assert raw[0] == nq and raw[1] == k and len(raw) == 2 + 2 * nq * k
    assert np.array_equal(raw[2:2 + nq * k].reshape(nq, k), ids)
    assert np.array_equal(raw[2 + nq * k:], dist.reshape(-1).view(np.uint32))
