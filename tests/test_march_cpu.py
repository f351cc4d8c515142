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
#include <math.h>
#include <vector>
#include <stdint.h>
extern "C" void ff_batch(const float* t, const float* target, const float* dt, long n, float* a, float* b) {
    for (long i = 0; i < n; ++i) {
        a[i] = nfa::fast_forward_serial(t[i], target[i], dt[i]);
        b[i] = nfa::fast_forward_exact(t[i], target[i], dt[i]);
    }
}
extern "C" void ff_batch_stepper(const float* t, const float* target, const float* dt, long n, float* a, float* b) {
    for (long i = 0; i < n; ++i) {
        a[i] = nfa::fast_forward_serial(t[i], target[i], dt[i]);
        b[i] = nfa::fast_forward_stepper(t[i], target[i], dt[i]);
    }
}
// number of stepper_advance calls of one skipping march from t0 to thr, with / without the approach table
extern "C" int approach_calls(float t0, float dt, float thr, int use_table, float* t_out) {
    const float half = dt * 0.5f;
    nfa::Stepper s; nfa::stepper_init(s);
    nfa::ApproachTable tb; tb.n = 0;
    if (use_table) nfa::approach_table_build(tb, t0, dt);
    float t = t0;
    nfa::approach_table_apply(tb, s, t, half, thr);
    int calls = 0;
    while (t + half < thr) { float inc; if (nfa::stepper_advance(s, t, dt, half, thr, 0xFFFFFFFFu, &inc) == 0u) break; calls++; }
    *t_out = t;
    return calls;
}
// One ray: events (thr[k], emit[k]) consumed in order, as traverse2.hip's march() does (skip or emit while
// the step's mid-point is before thr; budget of `limit` samples).  Serial reference vs Stepper + run records
// expanded with one fused multiply-add per value (what expand_runs_kernel does).
// Returns the number of samples; out_* hold (ts, te) of both versions; t_last[2] the final positions.
extern "C" long ray_events(float t0, float dt, const float* thr, const int* emit, long n_ev, int limit, int use_table, int lean, long cap,
                           float* ts_a, float* te_a, float* ts_b, float* te_b, float* t_last, long* n_b_out, long* n_jumps) {
    const float half = dt * 0.5f;
    // ---- serial
    long na = 0;
    {
        float t = t0;
        for (long k = 0; k < n_ev; ++k) {
            if (limit > 0 && na >= limit) break;
            while (t + half < thr[k]) {
                if (emit[k] && limit > 0 && na >= limit) break;
                const float tn = t + dt;
                if (tn == t) { if (!emit[k]) t = thr[k]; break; }
                if (emit[k]) { if (na < cap) { ts_a[na] = t; te_a[na] = tn; } na++; }
                t = tn;
            }
        }
        t_last[0] = t;
    }
    // ---- stepper + runs
    struct Run { float t0, inc; long n; };
    std::vector<Run> runs;
    long nb = 0, jumps = 0;
    {
        nfa::Stepper s; nfa::stepper_init(s);
        nfa::ApproachTable tb; tb.n = 0;
        if (use_table) nfa::approach_table_build(tb, use_table == 2 ? t0 + 1.0f : t0, dt);  // 2: table of another near plane
        float t = t0;
        bool open = false, continuous = false, at_near = true;
        for (long k = 0; k < n_ev; ++k) {
            if (limit > 0 && nb >= limit) break;
            if (lean && limit <= 0 && !at_near) {   // the one-shot path of process_events (traverse2.hip)
                float tt = t; nfa::StepSeg segs[3];
                if (nfa::stepper_run_event(s, tt, dt, half, thr[k], segs[0], segs[1], segs[2])) {
                    for (int i = 0; i < 3; ++i) {
                        if (segs[i].n == 0) continue;
                        if (emit[k]) {
                            if (open && continuous && segs[i].inc == runs.back().inc) runs.back().n += segs[i].n;
                            else { runs.push_back({segs[i].t0, segs[i].inc, (long)segs[i].n}); open = true; }
                            nb += segs[i].n; continuous = true;
                        }
                        if (segs[i].n > 1) jumps++;
                    }
                    t = tt;
                    if (!emit[k]) continuous = false;
                    continue;
                }
            }
            if (at_near) { at_near = false; if (!emit[k]) nfa::approach_table_apply(tb, s, t, half, thr[k]); }
            for (;;) {
                if (!(t + half < thr[k])) break;
                uint32_t budget = 0xFFFFFFFFu;
                if (emit[k] && limit > 0) { if (nb >= limit) break; budget = (uint32_t)(limit - nb); }
                float tn = t, inc;
                const uint32_t n = nfa::stepper_advance(s, tn, dt, half, thr[k], budget, &inc);
                if (n == 0) { if (!emit[k]) { t = thr[k]; nfa::stepper_reset(s); } break; }
                if (n > 1) jumps++;
                if (emit[k]) {
                    if (open && continuous && inc == runs.back().inc) runs.back().n += n;
                    else { runs.push_back({t, inc, (long)n}); open = true; }
                    nb += n; continuous = true;
                }
                t = tn;
            }
            if (!emit[k]) continuous = false;
        }
        t_last[1] = t;
    }
    long w = 0;
    for (auto& r : runs)
        for (long i = 0; i < r.n; ++i, ++w)
            if (w < cap) { ts_b[w] = fmaf((float)i, r.inc, r.t0); te_b[w] = fmaf((float)(i + 1), r.inc, r.t0); }
    *n_b_out = nb; *n_jumps = jumps;
    return na;
}

// The lattice table (march.h): every tabulated point equals the serial accumulation, rows are contiguous, and J(thr)
// (fast form or exact search) equals the serial loop's step count for thresholds around row ends, around lattice
// points and at random.  Returns the number of thresholds checked; bad[0..2] = wrong points, wrong J, J served by the fast form.
extern "C" long lattice_check(float near, float dt, long max_steps, unsigned seed, long n_random, long* bad, unsigned* shape) {
    nfa::LatticeTable tb;
    nfa::lattice_table_build(tb, near, dt);
    bad[0] = bad[1] = bad[2] = 0;
    shape[0] = tb.n_rows; shape[1] = tb.j_end; shape[2] = tb.n_binades;
    if (tb.n_rows == 0) return 0;
    // rows contiguous
    for (unsigned r = 0; r + 1 < tb.n_rows; ++r) if (tb.jA[r] + tb.n[r] != tb.jA[r + 1]) bad[0]++;
    if (tb.jA[0] != 0 || tb.A[0] != nfa::f32_bits(near)) bad[0]++;
    // the serial sequence
    std::vector<float> L;
    {
        float t = near;
        for (long j = 0; j < max_steps; ++j) { L.push_back(t); const float tn = t + dt; if (tn == t) break; t = tn; }
    }
    const long n_pts = (long)L.size() < (long)tb.j_end ? (long)L.size() : (long)tb.j_end;
    if ((long)L.size() < max_steps && (long)tb.j_end != (long)L.size()) bad[0]++;   // the whole sequence fits: same length
    {
        unsigned r = 0;
        for (long j = 0; j < n_pts; ++j) {
            while (r + 1 < tb.n_rows && tb.jA[r + 1] <= (unsigned)j) ++r;
            if (nfa::f32_bits(nfa::lattice_point_in_row(tb, r, (unsigned)j)) != nfa::f32_bits(L[j])) { bad[0]++; if (bad[0] > 10) break; }
        }
    }
    const float half = dt * 0.5f;
    auto serial_J = [&](float thr) -> long {   // first j with !(L[j] + half < thr), -1 when beyond the points we have
        long lo = 0, hi = n_pts;               // the condition is monotone along the sequence
        while (lo < hi) { const long mid = (lo + hi) >> 1; if (L[mid] + half < thr) lo = mid + 1; else hi = mid; }
        return lo < n_pts ? lo : -1;
    };
    long checked = 0;
    auto check = [&](float thr) {
        const long want = serial_J(thr);
        if (want < 0) return;
        const uint32_t r = nfa::lattice_main_row(tb, thr, half);
        uint32_t wf = nfa::LATTICE_FAIL;
        if (r < tb.n_rows && tb.q[r] != 0u) {
            const float rcp = 1.0f / ldexpf((float)tb.q[r], (int)(tb.A[r] >> 23) - 150);
            wf = nfa::lattice_J_fast(tb.A[r], tb.jA[r], tb.q[r], tb.n[r], rcp * (1.0f + 1e-6f), r, thr, half);   // (a 1-ulp-ish reciprocal, as on the device)
        }
        const uint32_t ws = nfa::lattice_J_search(tb, thr, half);
        const uint32_t w = nfa::lattice_J(tb, thr, half);
        if (wf != nfa::LATTICE_FAIL) { bad[2]++; if ((long)(wf >> nfa::LATTICE_ROW_BITS) != want || wf != ws) bad[1]++; }
        if (ws == nfa::LATTICE_OFF_TABLE || (long)(ws >> nfa::LATTICE_ROW_BITS) != want) bad[1]++;
        else {
            const uint32_t row = ws & ((1u << nfa::LATTICE_ROW_BITS) - 1u), j = ws >> nfa::LATTICE_ROW_BITS;
            if (!(row < tb.n_rows && j >= tb.jA[row] && j < tb.jA[row] + tb.n[row])) bad[1]++;
        }
        if (w != ws) bad[1]++;
        checked++;
    };
    // around every row end and start, and around the lattice points themselves
    for (unsigned r = 0; r < tb.n_rows; ++r)
        for (long d = -4; d <= 4; ++d) {
            const long js[2] = {(long)tb.jA[r] + d, (long)tb.jA[r] + (long)tb.n[r] - 1 + d};
            for (long j : js) {
                if (j < 0 || j >= n_pts) continue;
                const float base = L[j] + half;
                float v = base;
                for (int u = 0; u < 3; ++u) { check(v); v = nextafterf(v, 3.0e38f); }
                v = base;
                for (int u = 0; u < 3; ++u) { v = nextafterf(v, -3.0e38f); check(v); }
                check(L[j]); check(L[j] + dt * 0.25f); check(L[j] + dt * 0.75f);
            }
        }
    uint64_t st = seed * 6364136223846793005ull + 1442695040888963407ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)(st >> 11) / 9007199254740992.0; };
    for (long i = 0; i < n_random; ++i) {
        const long j = (long)(rnd() * (double)n_pts);
        check(L[j < n_pts ? j : n_pts - 1] + (float)(rnd() * 2.0 - 0.5) * dt);
        check((float)(rnd() * (double)L[n_pts - 1]));
        check(near - dt * (float)rnd());
    }
    // beyond the table
    if ((long)L.size() < max_steps) {
        const float far = L.back() * 1.5f + dt;
        if (nfa::lattice_J(tb, far, half) != nfa::LATTICE_OFF_TABLE) bad[1]++;
    }
    return checked;
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


def a_serial(lib, t, thr, dt):
    a, _ = _run(lib, np.array([t]), np.array([thr]), np.array([dt]))
    return a[0]


def test_stepper_matches_serial(tmp_path):
    """march.h's Stepper (remembered stable increment, verified jump counts, no safety margin) gives the
    serial loop's result bit for bit: single marches, and whole rays of alternating skip / emit events with
    the run records expanded by fused multiply-adds, with and without a sample budget."""
    lib = _build(tmp_path)
    rng = np.random.default_rng(1)
    n = 200000
    t = (rng.random(n) * 4).astype(np.float32)
    t[: n // 10] = 0.0
    dt = (10 ** rng.uniform(-3.2, -1.5, n)).astype(np.float32)
    target = (t + rng.random(n) * 8).astype(np.float32)
    t2 = np.concatenate([np.nextafter((2.0 ** rng.integers(-3, 4, 5000)).astype(np.float32), np.float32(0)),
                         (2.0 ** rng.integers(-3, 4, 5000)).astype(np.float32) * (1 + rng.integers(0, 5, 5000) * 2.0 ** -23),
                         rng.random(5000).astype(np.float32)]).astype(np.float32)
    dt2 = np.concatenate([(2.0 ** rng.integers(-12, -4, 5000)).astype(np.float32),
                          (2.0 ** rng.integers(-12, -4, 5000) * 1.5).astype(np.float32),
                          (rng.random(5000) * 2e-4 + 1e-5).astype(np.float32)])
    target2 = (t2 + rng.random(t2.size).astype(np.float32) * 3).astype(np.float32)
    t3 = np.array([1e8, 5.0, -1.0, -0.5, 3.0, 0.0, 16777216.0, 0.0], np.float32)
    tg3 = np.array([2e8, 1.0, 2.0, -0.25, 3.0, 1e-30, 16777300.0, 2.2], np.float32)
    dt3 = np.array([1e-3, 0.1, 0.01, 0.125, 0.5, 1e-3, 1.0, 2 * 3 ** 0.5 / 1024], np.float32)
    for tt, tg, dd in ((t, target, dt), (t2, target2, dt2), (t3, tg3, dt3)):
        tt, tg, dd = (np.ascontiguousarray(a, np.float32) for a in (tt, tg, dd))
        a, b = np.empty_like(tt), np.empty_like(tt)
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        lib.ff_batch_stepper(p(tt), p(tg), p(dd), C.c_long(tt.size), p(a), p(b))
        assert (a.view(np.uint32) == b.view(np.uint32)).all()

    lib.ray_events.restype = C.c_long
    cap = 1 << 16
    bufs = [np.empty(cap, np.float32) for _ in range(4)]
    tl = np.empty(2, np.float32)
    nb, nj = C.c_long(), C.c_long()
    total_jumps = total_samples = 0
    for case in range(3000):
        kind = case % 6
        t0 = np.float32([0.0, rng.random() * 4, 2.0 - 1e-5 * rng.random(), rng.random() * 1e-3, 3.9, 0.7][kind])
        if kind == 2:
            dtv = np.float32(2.0 ** rng.integers(-11, -6) * [1.0, 1.5][case % 2])       # exact ties across the 2.0 edge
        elif kind == 4:
            dtv = np.float32(2 * 3 ** 0.5 / 1024)                                        # the bench's step across 4.0
        else:
            dtv = np.float32(10 ** rng.uniform(-3.3, -1.8))
        n_ev = int(rng.integers(1, 40))
        gaps = rng.random(n_ev) * [0.05, 0.5, 0.01, 2.0][case % 4]
        gaps[rng.random(n_ev) < 0.2] = 0.0                                               # repeated thresholds
        thr = (t0 + np.cumsum(gaps)).astype(np.float32)
        if case % 17 == 0:
            thr[-1] = np.float32(1e10) if dtv > 1e-2 else thr[-1]
        emit = (rng.random(n_ev) < 0.5).astype(np.int32)
        limit = int([0, 0, 7, 64][case % 4])
        if case % 5 == 1:
            thr += np.float32(rng.random() * 6)      # grid far from the near plane: the approach table's case
            emit[0] = 0
        na = lib.ray_events(C.c_float(float(t0)), C.c_float(float(dtv)), thr.ctypes.data_as(C.c_void_p),
                            emit.ctypes.data_as(C.c_void_p), C.c_long(n_ev), C.c_int(limit), C.c_int(case % 3), C.c_int((case // 3) % 2), C.c_long(cap),
                            *[b.ctypes.data_as(C.c_void_p) for b in bufs], tl.ctypes.data_as(C.c_void_p),
                            C.byref(nb), C.byref(nj))
        assert na == nb.value, (case, na, nb.value)
        assert tl.view(np.uint32)[0] == tl.view(np.uint32)[1], (case, tl)
        m = min(na, cap)
        assert (bufs[0][:m].view(np.uint32) == bufs[2][:m].view(np.uint32)).all(), case
        assert (bufs[1][:m].view(np.uint32) == bufs[3][:m].view(np.uint32)).all(), case
        total_jumps += nj.value; total_samples += na
    assert total_samples > 100000 and total_jumps > 3000     # the shortcut is actually exercised
    # the headline case (camera 2.2 .. 4.2 units from the box, bench step): ~30-40 calls without the table, a handful with
    lib.approach_calls.restype = C.c_int
    out = [C.c_float(), C.c_float()]
    for thr in (2.2, 2.9, 3.99, 4.01, 4.4, 0.9):
        c0 = lib.approach_calls(C.c_float(0.0), C.c_float(2 * 3 ** 0.5 / 1024), C.c_float(thr), 0, C.byref(out[0]))
        c1 = lib.approach_calls(C.c_float(0.0), C.c_float(2 * 3 ** 0.5 / 1024), C.c_float(thr), 1, C.byref(out[1]))
        assert out[0].value == out[1].value == float(a_serial(lib, 0.0, thr, 2 * 3 ** 0.5 / 1024))
        assert c0 >= 15 and c1 <= 4, (thr, c0, c1)


def test_lattice_table_and_J_match_the_serial_loop(tmp_path):
    """march.h's lattice table (rows of the sequence near, near + dt, ...) reproduces the serial accumulation point by
    point, and J(thr) -- the fast three-probe form and the exact search -- is the serial loop's step count, for steps
    with exact ties, power-of-two steps, near planes at 0 / inside a binade / at a binade edge, thresholds at row ends."""
    lib = _build(tmp_path)
    lib.lattice_check.restype = C.c_long
    rng = np.random.default_rng(5)
    cases = [(0.0, 2 * 3 ** 0.5 / 1024), (0.2, 1e-3), (0.0, 1e-3), (0.0, 0.01), (2.0, 0.01), (0.05, 5e-3), (0.0, 2.0 ** -8),
             (0.0, 1.5 * 2.0 ** -9), (1.999999, 2.0 ** -9), (4.0, 2 * 3 ** 0.5 / 1024), (0.0, 0.3), (1e-3, 1e-5), (7.3, 0.11),
             (0.0, 3.0), (100.0, 0.5)]
    for _ in range(40):
        cases.append((float(rng.choice([0.0, rng.random() * 3, 2.0 ** rng.integers(-3, 3)])), float(10 ** rng.uniform(-4.0, -0.5))))
    total = fast = 0
    for near, dt in cases:
        bad = (C.c_long * 3)()
        shape = (C.c_uint * 3)()
        n = lib.lattice_check(C.c_float(near), C.c_float(dt), C.c_long(3_000_000), C.c_uint(int(rng.integers(1, 1 << 30))), C.c_long(3000),
                              bad, shape)
        assert shape[0] >= 1 and bad[0] == 0 and bad[1] == 0, (near, dt, list(bad), list(shape))
        assert n > 3000
        total += n; fast += bad[2]
    assert fast > 0.5 * total      # the fast form serves most thresholds


LAYOUT_HARNESS = r'''
#define __host__
#define __device__
#include "walk_layout.h"
using namespace nfa;
extern "C" int layout_check()
{
    int bad = 0;
    for (int nb = 1; nb <= 9; ++nb) {
        // the three axes need the same number of index bits, not the same size
        int32_t res[3] = {1 << nb, (1 << nb) - (nb > 2 ? 3 : 0), 1 << nb};
        const WalkLayout L = walk_layout(res);
        if (!walk_layout_regular(L)) { bad += 1000; continue; }
        for (int ax = 0; ax < 3; ++ax)
            for (uint32_t v = 0; v < (1u << nb); ++v)
                if (bit_deposit(v, L.mask[ax]) != (spread_by_3(v) << ((ax + 1) % 3))) ++bad;
    }
    int32_t r2[3] = {128, 64, 128}, r3[3] = {1, 1, 1}, r4[3] = {100, 200, 100};
    if (walk_layout_regular(walk_layout(r2)) || walk_layout_regular(walk_layout(r3)) || walk_layout_regular(walk_layout(r4))) bad += 100000;
    return bad;
}
'''


def test_regular_layout_deposit_equals_the_generic_one(tmp_path):
    """walk.hip's span setup deposits coordinates with the 'one bit in three' spread when the layout is the plain rotation
    (walk_layout.h: walk_layout_regular, spread_by_3): the same bits as the generic deposit for every coordinate, and layouts
    with unequal bit counts are not taken for regular."""
    src = tmp_path / "layout.cpp"
    src.write_text(LAYOUT_HARNESS)
    so = tmp_path / "liblayout_test.so"
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "nerfacc_amd", "csrc"), str(src),
                    "-o", str(so)], check=True)
    assert C.CDLL(str(so)).layout_check() == 0
