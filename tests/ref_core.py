"""Driver for oracle/_ref/ref_core — the parts of the REFERENCE that compile from their own sources here (src/topn.h,
src/defs.cpp, src/metis_io.cpp; see oracle/ref_shim.cpp).  Test infrastructure."""
import os
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_core")


def available():
    return os.path.exists(BIN)


class RefCore:
    def __init__(self):
        if not available():
            raise RuntimeError("oracle/_ref/ref_core not built (needs /root/reference): make -C oracle ref")
        self.tmp = tempfile.mkdtemp(prefix="refcore")

    def run(self, cmd, inp, out):
        subprocess.check_call([BIN, cmd, inp, out])

    def topn(self, cases, descending=False):
        """cases: list of (dists f32, ids u32, n) -> list of (dists, ids) from the reference's TopN::Add / Take()
        (descending=True: ConvertTopKToNNVec)."""
        inp, out = os.path.join(self.tmp, "topn.in"), os.path.join(self.tmp, "topn.out")
        with open(inp, "wb") as f:
            f.write(struct.pack("<I", len(cases)))
            for d, i, n in cases:
                f.write(struct.pack("<QI", len(d), n))
                f.write(np.ascontiguousarray(d, np.float32).tobytes())
                f.write(np.ascontiguousarray(i, np.uint32).tobytes())
        self.run("topn_desc" if descending else "topn", inp, out)
        raw = open(out, "rb").read()
        pos, res = 0, []
        for _ in cases:
            (c,) = struct.unpack_from("<I", raw, pos)
            pos += 4
            d = np.frombuffer(raw, np.float32, c, pos).copy()
            pos += 4 * c
            i = np.frombuffer(raw, np.uint32, c, pos).copy()
            pos += 4 * c
            res.append((d, i))
        return res

    def partition(self, partition):
        inp, out = os.path.join(self.tmp, "part.in"), os.path.join(self.tmp, "part.out")
        p = np.ascontiguousarray(partition, np.int32)
        with open(inp, "wb") as f:
            f.write(struct.pack("<Q", len(p)))
            f.write(p.tobytes())
        self.run("partition", inp, out)
        raw = open(out, "rb").read()
        pos = 0
        (num_parts,) = struct.unpack_from("<i", raw, pos); pos += 4
        (S,) = struct.unpack_from("<I", raw, pos); pos += 4
        offsets = np.frombuffer(raw, np.uint64, S + 1, pos).copy(); pos += 8 * (S + 1)
        ids = np.frombuffer(raw, np.uint32, int(offsets[-1]), pos).copy(); pos += 4 * int(offsets[-1])
        (nc,) = struct.unpack_from("<Q", raw, pos); pos += 8
        sizes, flat = [], []
        for _ in range(nc):
            (c,) = struct.unpack_from("<I", raw, pos); pos += 4
            sizes.append(c)
            flat.extend(np.frombuffer(raw, np.int32, c, pos)); pos += 4 * c
        remapped = np.frombuffer(raw, np.int32, len(p), pos).copy()
        return dict(num_parts=num_parts, offsets=offsets, ids=ids, cover_sizes=np.asarray(sizes, np.uint32), cover_flat=np.asarray(flat, np.int32),
                    remapped=remapped)

    def extract(self, points, bucket):
        inp, out = os.path.join(self.tmp, "ext.in"), os.path.join(self.tmp, "ext.out")
        pts = np.ascontiguousarray(points, np.float32)
        b = np.ascontiguousarray(bucket, np.uint32)
        with open(inp, "wb") as f:
            f.write(struct.pack("<QI", pts.shape[0], pts.shape[1]))
            f.write(pts.tobytes())
            f.write(struct.pack("<Q", len(b)))
            f.write(b.tobytes())
        self.run("extract", inp, out)
        return np.fromfile(out, np.float32).reshape(len(b), pts.shape[1])
