"""CPU: the build checks of tools/kernel_resources.py do fail when they should.  Two small kernels are compiled for gfx950 (hipcc cross-compiles without a GPU) in
the forms the checks exist to refuse — packed fp32 arithmetic, a matrix-core kernel that leaves room on its SIMDs — and in the forms the library ships."""
import importlib.util
import os
import subprocess

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
SRC = r'''
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void Pairs(const float* a, float* o, int n) {
    f32x2 acc = {0.f, 0.f};
    for (int i = 0; i < n; ++i) {
        const f32x2 p = {a[2 * i + threadIdx.x], a[2 * i + 1 + threadIdx.x]};
        const float q = a[i + 7];
        const f32x2 d = p - (f32x2){q, q};
        acc = acc + d * d;
    }
    o[threadIdx.x] = acc.x + acc.y;
}
template <bool CLAIM>
__global__ void __launch_bounds__(512, 1) Matrix(const int* a, int* o) {
    if (CLAIM) asm volatile("" ::: "v255");
    const v4i x = {a[threadIdx.x], a[threadIdx.x + 1], a[threadIdx.x + 2], a[threadIdx.x + 3]};
    v16i acc = {0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, x, acc, 0, 0, 0);
    o[threadIdx.x] = acc[0] + acc[7];
}
template __global__ void Matrix<true>(const int*, int*);
template __global__ void Matrix<false>(const int*, int*);
'''


def _kr():
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    return kr


def _compile(tmp_path, name, extra):
    src = tmp_path / "k.hip"
    src.write_text(SRC)
    obj = tmp_path / f"{name}.o"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", str(src), "-o", str(obj)] + extra,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return str(obj)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_the_packed_fp32_check_sees_packed_instructions_and_the_build_flag_removes_them(tmp_path):
    kr = _kr()
    packed = kr.check_no_packed_fp32(_compile(tmp_path, "packed", []))
    assert any("Pairs" in m and "v_pk_" in m for m in packed), packed            # plain -O3: v_pk_add_f32 / v_pk_mul_f32
    flags = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]        # the library's HIPFLAGS (gp-ann_amd/Makefile)
    assert kr.check_no_packed_fp32(_compile(tmp_path, "scalar", flags)) == []
    assert "-packed-fp32-ops" in open(os.path.join(ROOT, "gp-ann_amd", "Makefile")).read()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_the_matrix_kernel_check_refuses_a_kernel_that_leaves_room_on_its_simds(tmp_path):
    kr = _kr()
    ok, bad = kr.check_matrix_kernels(_compile(tmp_path, "matrix", []))
    names = kr.demangle([s for s, _, _ in ok])
    assert [n for n in names.values() if "Matrix<true>" in n], (ok, bad)          # 256 registers, 512-thread workgroups: accepted
    assert any("Matrix" in m and "vector registers" in m for m in bad), bad       # the one without the claim: a few registers — refused
    assert len(ok) == 1 and len(bad) == 1
