"""Differential check of the one-launch PCG iteration (fv_tune key 63) against the pass + vector-update pair it replaces: random
tile-sized boxes (the symmetric form, chunk traversal), heterogeneous or uniform conductivity, random schedules of (dt, steps, rtol) mixing
one-iteration stretches, many-iteration stretches, loose steps (zero iterations) and chunked runs — with and without a trajectory being
recorded (no deferred flush then), bursts on and off.  Same Jacobi-PCG iteration: iteration counts within one, heads to 1e-10 of each other
(both runs are converged to their rtol; the polynomial beta makes the iterates differ inside that tolerance).
usage: python tools/ploop_fuzz.py [nseeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402


def run(fv, nseeds=12, verbose=True, first=0, trace=False):
    lib = fv.load()
    bad = 0
    for seed in range(first, first + nseeds):
        rng = np.random.default_rng(1000 + seed)
        ns = [int(rng.integers(36, 50)), int(rng.integers(184, 200)), 2 * int(rng.integers(93, 105))]
        mins, maxs = bench.spacing_box(ns)
        dn, src = bench.box_setup(ns)
        if seed % 4 == 0:
            src[:] = 0.0
        uniform = seed % 3 == 2
        record = seed % 5 == 1
        chain = 0 if seed % 7 == 3 else 8
        dts, rtols = [2.0**-8, 1.0, 20.0, 60.0, 600.0, 3600.0, 40000.0], [1e-3, 1e-6, 1e-8, 1e-10, 1e-12]
        schedule = [(float(rng.choice(dts)), int(rng.integers(1, 12)), float(rng.choice(rtols))) for _ in range(int(rng.integers(3, 8)))]
        out = {}
        for name, key in (("pair", 0), ("one launch", 1)):
            assert lib.fv_tune(63, key) == 0 and lib.fv_tune(13, chain) == 0
            p = fv.Problem.regulargrid(mins, maxs, ns, dn)
            K = np.array([1e-5 * (1.0 + seed)]) if uniform else 1e-5 * np.exp(np.random.default_rng(100 + seed).standard_normal(p.F))
            p.assemble(K, src, np.full(len(dn), 1e3))
            st = p.transient_begin(0.1, None, np.full(p.N, 1e3) + np.random.default_rng(200 + seed).standard_normal(p.N))
            tr = None
            if record:
                tr = p.new_trajectory()
                tr.push(st, 0.0)
                p.record(tr)
            its, forms = [], []
            for dt, k, rtol in schedule:
                if trace:
                    print("  seed %d %s: %d steps of dt %g at rtol %g ..." % (seed, name, k, dt, rtol), flush=True)
                it, info, _ = p.run_fixed(st, dt, k, rtol, maxiter=5000)
                assert info.converged
                its.append(it.copy())
                forms.append(p.loop_form())
            if tr is not None:
                p.record(None)
                last = tr.free_values(len(tr) - 1)
                assert np.array_equal(last, st.free_values()), "the recorded last state is not the state"
                tr.close()
            out[name] = (st.free_values(), np.concatenate(its), forms)
            p.close()
        a, b = out["pair"], out["one launch"]
        dits = np.abs(a[1].astype(int) - b[1].astype(int)).max()
        dh = np.abs(a[0] - b[0]).max() / np.abs(a[0]).max()
        used = sum(1 for f in b[2] if f in (89, 67))
        ok = dits <= 1 and dh <= 1e-10
        if not ok:
            bad += 1
            print("seed %d: MISMATCH iterations differ by %d, heads by %.3e; schedule %s" % (seed, dits, dh, schedule), flush=True)
        if verbose:
            print("seed %d ns %s %s%s chain %d: %d steps, iterations %s, one-launch loop in %d of %d stretches, heads %.1e apart" %
                  (seed, ns, "uniform" if uniform else "heterogeneous", ", recording" if record else "", chain, len(a[1]), np.bincount(b[1])[:8], used, len(schedule), dh), flush=True)
    lib.fv_tune(63, 1)
    lib.fv_tune(13, 8)
    return bad


if __name__ == "__main__":
    fv_ = load_package()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # usage: ploop_fuzz.py [nseeds [first seed [trace]]]
    print("mismatches:", run(fv_, n, first=first, trace=len(sys.argv) > 3))
