"""CPU: nerfacc_amd/csrc/march.h (exact O(#binades) marching) is bit-identical to the serial loop
it replaces, on random and adversarial inputs.  The header is compiled as host C++ here; the HIP
kernels include the same file."""
import ctypes as C
import os
import subprocess

import numpy as np

from conftest import ROOT

HARNESS = r'''
#include "march.h"
extern "C" void ff_batch(const float* t, const float* target, const float* dt, long n, float* a, float* b) {
    for (long i = 0; i < n; ++i) {
        a[i] = nfa::fast_forward_serial(t[i], target[i], dt[i]);
        b[i] = nfa::fast_forward_exact(t[i], target[i], dt[i]);
    }
}
'''


def _build(tmp_path):
    src = tmp_path / "harness.cpp"
    src.write_text(HARNESS)
    so = tmp_path / "libmarch_test.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-std=c++17",
                    "-I", os.path.join(ROOT, "nerfacc_amd", "csrc"), str(src), "-o", str(so)], check=True)
    return C.CDLL(str(so))


def _run(lib, t, target, dt):
    t, target, dt = (np.ascontiguousarray(a, np.float32) for a in (t, target, dt))
    a, b = np.empty_like(t), np.empty_like(t)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    lib.ff_batch(p(t), p(target), p(dt), C.c_long(t.size), p(a), p(b))
    return a, b


def test_fast_forward_exact_matches_serial(tmp_path):
    lib = _build(tmp_path)
    rng = np.random.default_rng(0)
    n = 200000
    # generic: start in [0, 4), steps ~1e-3..1e-2, targets up to 8 away
    t = (rng.random(n) * 4).astype(np.float32)
    t[: n // 10] = 0.0
    dt = (10 ** rng.uniform(-3.2, -1.5, n)).astype(np.float32)
    target = (t + rng.random(n) * 8).astype(np.float32)
    a, b = _run(lib, t, target, dt)
    assert (a.view(np.uint32) == b.view(np.uint32)).all()
    # adversarial: power-of-two steps (exact ties), starts just below binade edges, tiny steps
    t2 = np.concatenate([np.nextafter((2.0 ** rng.integers(-3, 4, 5000)).astype(np.float32), np.float32(0)),
                         (2.0 ** rng.integers(-3, 4, 5000)).astype(np.float32) * (1 + rng.integers(0, 5, 5000) * 2.0 ** -23),
                         rng.random(5000).astype(np.float32)]).astype(np.float32)
    dt2 = np.concatenate([(2.0 ** rng.integers(-12, -4, 5000)).astype(np.float32),
                          (2.0 ** rng.integers(-12, -4, 5000) * 1.5).astype(np.float32),      # k + 1/2 ulp ties
                          (rng.random(5000) * 2e-4 + 1e-5).astype(np.float32)])
    target2 = (t2 + rng.random(t2.size).astype(np.float32) * 3).astype(np.float32)
    a, b = _run(lib, t2, target2, dt2)
    assert (a.view(np.uint32) == b.view(np.uint32)).all()
    # no-progress guard, targets already reached, negative starts, huge values
    t3 = np.array([1e8, 5.0, -1.0, -0.5, 3.0, 0.0, 16777216.0], np.float32)
    tg3 = np.array([2e8, 1.0, 2.0, -0.25, 3.0, 1e-30, 16777300.0], np.float32)
    dt3 = np.array([1e-3, 0.1, 0.01, 0.125, 0.5, 1e-3, 1.0], np.float32)
    a, b = _run(lib, t3, tg3, dt3)
    assert (a.view(np.uint32) == b.view(np.uint32)).all() and a[0] == np.float32(2e8)
    # the headline case: 650 serial steps from the camera to the box
    a, b = _run(lib, np.zeros(1), np.array([2.2]), np.array([2 * 3 ** 0.5 / 1024]))
    assert a[0] == b[0] and abs(a[0] - 2.2) < 4e-3
