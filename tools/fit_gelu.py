import numpy as np
from scipy.special import erf, erfc
from scipy.optimize import minimize
x = np.linspace(-9, 9, 200001)
phi = 0.5 * erfc(-x / np.sqrt(2))
g = x * phi
def f(c, x=x):
    x2 = x * x
    p = np.zeros_like(x)
    for ci in c[::-1]: p = p * x2 + ci
    return x / (1 + np.exp(-x * p))
def err(c): return np.max(np.abs(f(c) - g))
for deg, c0 in [(2, [1.5958, 0.07135]), (3, [1.5958, 0.07135, 0.0]), (4, [1.5958, 0.07135, 0.0, 0.0])]:
    best = None
    c = np.array(c0)
    for it in range(6):
        r = minimize(err, c, method="Nelder-Mead", options=dict(xatol=1e-10, fatol=1e-12, maxiter=20000, maxfev=20000))
        c = r.x
    print(deg, repr(c), err(c))
    # float32 evaluation
    xs = x.astype(np.float32); c32 = c.astype(np.float32)
    x2 = xs * xs; p = np.zeros_like(xs)
    for ci in c32[::-1]: p = p * x2 + ci
    with np.errstate(over='ignore'):
        out = xs / (1 + np.exp2(-(xs * p) * np.float32(1.4426950408889634)))
    print("   fp32 max abs err", np.max(np.abs(out - g)), "rel (|g|>1e-3)", np.max(np.abs(out - g)[np.abs(g) > 1e-3] / np.abs(g)[np.abs(g) > 1e-3]))
# current A&S
t = 1 / (1 + 0.3275911 * np.abs(x) / np.sqrt(2)); ax = np.abs(x) / np.sqrt(2)
pe = ((((1.061405429 * t - 1.453152027) * t + 1.421413741) * t - 0.284496736) * t + 0.254829592) * t * np.exp(-ax * ax)
print("A&S 7.1.26 gelu err", np.max(np.abs(0.5 * x * (1 + np.sign(x) * (1 - pe)) - g)))
