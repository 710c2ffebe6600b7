"""GPU: time of one forward chain (tnml_forward) and of predict at a BASELINE shape; checks f against the plain-FMA chain
(calibration pass uses it) through the oracle-free identity forward() == predict() and prints ms per chain."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tensornetworkforml_amd import _hip

def main():
    N, M, D, L, b = 784, int(os.environ.get('M', 20)), 2, int(os.environ.get('L', 2)), int(os.environ.get('B', 5000))
    rng = np.random.default_rng(0)
    p = rng.random((b, N)).astype(np.float32)
    X = np.stack([np.sin(np.pi * p / 2), np.cos(np.pi * p / 2)], -1).astype(np.float32)
    y = rng.integers(0, L, b)
    bond = [min(M, D ** min(i + 1, N - 1 - i)) for i in range(N - 1)]
    cores = []
    for i in range(N):
        ml = 1 if i == 0 else bond[i - 1]
        mr = 1 if i == N - 1 else bond[i]
        shp = (ml, D, mr, L) if i == 0 else (ml, D, mr)
        cores.append((rng.standard_normal(shp) * (0.9 / np.sqrt(max(ml, 1)))).astype(np.float32))
    ctx = _hip.Context(N, D, L, M, b)
    ctx.set_cores(cores, 0)
    ctx.set_input(X, y)
    f = ctx.forward()
    print('max|f|', np.abs(f).max(), 'finite', np.isfinite(f).all())
    for rep in range(3):
        ctx.synchronize(); ctx.timer_start()
        for _ in range(20):
            ctx.forward(want_f=False)
        ms = ctx.timer_stop() / 20
        print('forward: %.3f ms per chain  (%.2f TB/s of 4 b N (2M + D) bytes)' % (ms, 4 * b * N * (2 * M + D) / ms / 1e9))
    fp = ctx.predict(X)
    print('predict == forward bitwise:', np.array_equal(fp, f), 'max diff', np.abs(fp - f).max())
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.predict(X)
    print('predict incl. H2D: %.2f ms' % ((time.perf_counter() - t0) / 5 * 1e3))
    ctx.close()

if __name__ == '__main__':
    main()
