#!/usr/bin/env python3
"""Registers / spills / LDS of the gfx950 kernels in a built object file or shared library, from the code objects' metadata notes.
    python tools/kernel_resources.py [gp-ann_amd/build/scan.o | gp-ann_amd/lib/libgpann_hip.so] [name fragment ...]
    python tools/kernel_resources.py --check-matrix-kernels [gp-ann_amd/lib/libgpann_hip.so]
--check-matrix-kernels disassembles every code object and fails (exit 1) if any symbol holds a packed fp32 instruction (v_pk_add/mul/fma_f32: the library
is built without them, profiles/r06/NOTES.md §2) or unless every symbol that holds a matrix instruction (v_mfma_*) is a
kernel that is allocated ALL 256 architectural vector registers: two such wavefronts fill a SIMD's register file, so no wavefront of another launch can
run beside them (csrc/common.h TakeWholeRegisterFile; profiles/r05/overlap_race.md — a wavefront that shared a SIMD with another kernel's
v_mfma_i32_32x32x32_i8 wavefront was measured to end with wrong registers).  tests/test_capi_exports.py runs the same check on the shipped library."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path, arch="gfx950"):
    """the device ELF images for `arch` inside an object file / shared library (one per translation unit): the .hip_fatbin section is a sequence of
    uncompressed clang offload bundles — magic, entry count, then (offset, size, id length, id) per entry"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", path, os.path.join(tmp, "unused.out")])
        data = open(fat, "rb").read()
    images = []
    for m in re.finditer(re.escape(BUNDLE_MAGIC), data):
        base = m.start()
        p = base + len(BUNDLE_MAGIC)
        (count,) = struct.unpack_from("<Q", data, p)
        p += 8
        for _ in range(count):
            off, size, idlen = struct.unpack_from("<QQQ", data, p)
            p += 24
            ident = data[p:p + idlen].decode()
            p += idlen
            if arch in ident and size:
                images.append(data[base + off: base + off + size])
    if not images:
        raise RuntimeError(f"{path}: no {arch} code object found (compressed bundles are not handled)")
    return images


def demangle(names):
    try:
        out = subprocess.check_output(["c++filt"], input="\n".join(names) + "\n", text=True)   # (binutils: the ROCm tree has no llvm-cxxfilt)
        return dict(zip(names, out.splitlines()))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


def kernel_table(image):
    """[(mangled name, {metadata key: value})] of one code object"""
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", f.name], text=True)
    rows = []
    for blk in notes.split("- .agpr_count:")[1:]:
        def g(key):
            m = re.search(rf"\.{key}:\s+(\S+)", blk)
            return m.group(1) if m else "?"
        rows.append((g("name"), {"agpr_count": blk.split()[0], **{k: g(k) for k in (
            "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size")}}))
    return rows


def matrix_symbols(image, pattern=r"\b(v_(?:mfma|smfmac)_\w+)"):
    """{symbol: set of matrix-instruction mnemonics in its body} of one code object (disassembly); `pattern`: another class of instruction"""
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        proc = subprocess.Popen([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f.name], stdout=subprocess.PIPE, text=True)
        found, sym = {}, None
        for line in proc.stdout:
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                sym = m.group(1)
                continue
            m = re.search(pattern, line)
            if m and sym:
                found.setdefault(sym, set()).add(m.group(1))
        proc.wait()
    return found


def check_matrix_kernels(path):
    """-> (list of (kernel, vgpr_count, mnemonics) that pass, list of failure messages)"""
    ok, bad = [], []
    for image in code_objects(path):
        table = dict(kernel_table(image))
        for sym, mnemonics in sorted(matrix_symbols(image).items()):
            meta = table.get(sym)
            if meta is None:
                bad.append(f"{sym}: matrix instructions in a symbol that is not a kernel (its callers' registers are unknown)")
            elif meta["vgpr_count"] != "256":
                bad.append(f"{sym}: {meta['vgpr_count']} vector registers — a matrix-core kernel must take all 256 (TakeWholeRegisterFile)")
            elif meta["max_flat_workgroup_size"] not in ("512", "1024"):
                # 256 registers = two wavefronts per SIMD = eight per CU: only a workgroup of eight (or sixteen) wavefronts is sure to fill its CU alone
                bad.append(f"{sym}: workgroups of {meta['max_flat_workgroup_size']} threads — a matrix-core kernel is expected to fill its CU (512)")
            else:
                ok.append((sym, meta["vgpr_count"], sorted(mnemonics)))
    return ok, bad


PACKED_FP32 = r"\b(v_pk_(?:add|mul|fma)_f32)\b"


def check_no_packed_fp32(path):
    """-> failure messages: symbols of the library that hold a packed fp32 instruction.  The library is built without them (Makefile HIPFLAGS:
    -target-feature -packed-fp32-ops) because beside another kernel's or process's int8 matrix wavefronts a wavefront's packed fp32 arithmetic was
    measured to go wrong in lanes 48 - 63 while scalar fp32 is not disturbed (profiles/r06/NOTES.md §2)."""
    bad = []
    for image in code_objects(path):
        for sym, mnemonics in sorted(matrix_symbols(image, PACKED_FP32).items()):
            bad.append(f"{sym}: packed fp32 instructions {sorted(mnemonics)} — the library is built with -target-feature -packed-fp32-ops")
    return bad


def main():
    args = sys.argv[1:]
    if args and args[0] == "--check-matrix-kernels":
        path = args[1] if len(args) > 1 else "gp-ann_amd/lib/libgpann_hip.so"
        ok, bad = check_matrix_kernels(path)
        names = demangle([s for s, _, _ in ok]) if ok else {}
        for sym, v, mn in ok:
            print(f"ok   {v:>4}  {','.join(mn):<60} {names[sym][:120]}")
        packed = check_no_packed_fp32(path)
        for msg in bad + packed:
            print("FAIL", msg)
        print(f"{len(ok)} matrix-core kernel instances with 256 vector registers, {len(bad)} failures; {len(packed)} symbols with packed fp32 instructions")
        bad = bad + packed
        sys.exit(1 if bad or not ok else 0)
    lib = args[0] if args and os.path.exists(args[0]) else "gp-ann_amd/build/scan.o"
    frags = [a for a in args if not os.path.exists(a)]
    rows = []
    for image in code_objects(lib):
        table = kernel_table(image)
        names = demangle([n for n, _ in table]) if table else {}
        for sym, meta in table:
            name = names[sym]
            if frags and not all(f in name for f in frags):
                continue
            rows.append((name, meta))
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>7}  kernel")
    for name, m in sorted(rows, key=lambda r: r[0]):
        print(f"{m['vgpr_count']:>5} {m['agpr_count']:>5} {m['sgpr_count']:>5} {m['vgpr_spill_count']:>6} {m['sgpr_spill_count']:>6} "
              f"{m['private_segment_fixed_size']:>7} {m['group_segment_fixed_size']:>7}  {name[:150]}")


if __name__ == "__main__":
    main()
