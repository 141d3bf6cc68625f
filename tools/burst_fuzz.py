"""Differential check of the burst machinery: random schedules of (dt, steps, rtol) on a heterogeneous box, run with
fv_tune(13, 0) (poll after every step) and with bursts (13 = 8, with and without the merged launches of key 22): the three
runs must agree bit for bit in state and iteration counts."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

def run(fv, nseeds=12, tight=False, verbose=True):
    """tight: small time steps and tight tolerances (long runs of one-iteration steps).  Returns the number of mismatches."""
    lib = fv.load()
    bad = 0
    for seed in range(nseeds):
        rng = np.random.default_rng(seed)
        ns = [int(rng.integers(12, 34)), int(rng.integers(12, 30)), int(rng.integers(8, 26))]
        if os.environ.get("FV_FUZZ_BIG"):  # > 2^20 rows with planes of > 4096 rows: sliced-DIA kernels, marching when forced (FV_TUNE 9=2)
            ns = [int(rng.integers(100, 131)), int(rng.integers(90, 120)), int(rng.integers(90, 120))]
        if os.environ.get("FV_FUZZ_TILE"):  # planes of > 32768 rows, lines of an even number of rows: the symmetric form, tiled traversal
            ns = [int(rng.integers(36, 50)), int(rng.integers(184, 200)), 2 * int(rng.integers(93, 105))]
        mins, maxs = bench.spacing_box(ns)
        dn, src = bench.box_setup(ns)
        if seed % 3 == 0:
            src[:] = 0.0
        dts, rtols = ([2.0**-10, 2.0**-8, 2.0**-6, 1.0, 3600.0], [1e-8, 1e-10, 1e-12, 1e-13]) if tight else ([2.0**-8, 1.0, 20.0, 60.0, 600.0, 3600.0], [1e-3, 1e-5, 3e-5, 1e-8, 1e-12])
        schedule = [(float(rng.choice(dts)), int(rng.integers(1, 46)), float(rng.choice(rtols))) for _ in range(int(rng.integers(3, 9)))]
        out = {}
        for name, chain, merged in (("polled", 0, 1), ("bursts", 8, 1), ("bursts, separate launches", 8, 0), ("bursts of 3", 3, 1)):
            lib.fv_tune(13, chain)
            lib.fv_tune(22, merged)
            p = fv.Problem.regulargrid(mins, maxs, ns, dn)
            K = 1e-5 * np.exp(rng.standard_normal(1)[0] * 0 + np.random.default_rng(100 + seed).standard_normal(p.F))
            if os.environ.get("FV_FUZZ_UNIFORM"):  # one conductivity: the matrix as codes, the fused step on chunks of a plane (fused_chunk_kernel)
                K = np.array([1e-5 * (1.0 + seed)])
            p.assemble(K, src, np.full(len(dn), 1e3))
            st = p.transient_begin(0.1, None, np.full(p.N, 1e3) + np.random.default_rng(200 + seed).standard_normal(p.N))
            its = []
            for dt, k, rtol in schedule:
                it, info, _ = p.run_fixed(st, dt, k, rtol, maxiter=5000)
                assert info.converged
                its.append(it.copy())
            out[name] = (st.free_values(), np.concatenate(its))
            p.close()
        ref = out["polled"]
        # where the fused step runs (tile-sized boxes) a burst does its steps by another kernel than a polled step: same
        # iteration counts, states to rounding (FV_FUZZ_RTOL, default 1e-11 of the heads); elsewhere bit for bit
        loose = float(os.environ.get("FV_FUZZ_RTOL", "1e-11")) if os.environ.get("FV_FUZZ_TILE") else 0.0
        for name, (state, its) in out.items():
            same = np.array_equal(state, ref[0]) if loose == 0.0 else np.abs(state - ref[0]).max() <= loose * np.abs(ref[0]).max()
            # (tile-sized boxes since round 5: steps of several iterations run as one launch per iteration, whose verdict uses r'.r' as a polynomial in
            # the step length — at rtol 1e-13, where the residual sits at the rounding of the heads, two runs whose states differ in the last bits may
            # stop one iteration apart)
            same_its = np.array_equal(its, ref[1]) if loose == 0.0 else (np.abs(its.astype(int) - ref[1].astype(int)) <= (ref[1] >= 2)).all()
            if not (same and same_its):
                bad += 1
                d = np.nonzero(its != ref[1])[0]
                print("seed %d %s: MISMATCH max |diff| %.3e, first differing step %s, schedule %s" % (seed, name, np.abs(state - ref[0]).max(), d[:3], schedule), flush=True)
        if verbose:
            print("seed %d ns %s: %d steps, iterations %s" % (seed, ns, len(ref[1]), np.bincount(ref[1])[:6]), flush=True)
    lib.fv_tune(13, 8)
    lib.fv_tune(22, 1)
    return bad


if __name__ == "__main__":
    fv_ = load_package()
    for kv in os.environ.get("FV_TUNE", "").split(","):
        if "=" in kv:
            fv_.load().fv_tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    print("mismatches:", run(fv_, n, len(sys.argv) > 2 and sys.argv[2] == "tight"))
