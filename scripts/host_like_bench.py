import numpy as np, time, ctypes, sys
if len(sys.argv) > 1:
    libc = ctypes.CDLL('libc.so.6'); print('mallopt', libc.mallopt(-3, 1 << 30), libc.mallopt(-1, 1 << 30))
n, D = 10000, 32
rng = np.random.default_rng(0)
x = rng.uniform(-10, 10, (n, D)); xf = np.asfortranarray(x)
def T(f, k=300):
    f(); t = time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter() - t) / k * 1e6
lo, hi = -10., 10.
def ros(x): return -np.sum(10.0 * (x[:, ::2] ** 2.0 - x[:, 1::2]) ** 2.0 + (x[:, ::2] - 1.0) ** 2.0, axis=1)
def ros_c(x):
    a = np.ascontiguousarray(x[:, ::2]); b = np.ascontiguousarray(x[:, 1::2])
    return -np.sum(10.0 * (a * a - b) ** 2 + (a - 1.0) ** 2, axis=1)
def ros_r(x):
    x3 = x.reshape(len(x), -1, 2)
    a = x3[:, :, 0]; b = x3[:, :, 1]
    return -np.sum(10.0 * (a * a - b) ** 2 + (a - 1.0) ** 2, axis=1)
def prior(x): return np.where(np.all((x >= lo) & (x <= hi), axis=1), 1.0, -np.inf)
def prior2(x): return np.where((np.abs(x) <= hi).all(axis=1), 1.0, -np.inf)
def prior3(x):
    m = np.abs(x).max(axis=1)
    return np.where(m <= hi, 1.0, -np.inf)
for name, f, arg in [("ros C", ros, x), ("ros F", ros, xf), ("ros_c C", ros_c, x), ("ros_r C", ros_r, x),
                     ("prior C", prior, x), ("prior F", prior, xf), ("prior2 C", prior2, x), ("prior3 C", prior3, x), ("prior3 F", prior3, xf),
                     ("copy x[m]", lambda a: a[np.ones(n, bool)], x), ("asfortran", np.asfortranarray, x), ("isfinite.all", lambda a: np.isfinite(a).all(), x)]:
    print(f"{name:12s} {T(lambda: f(arg)):9.1f} us")
