"""Prototype (numpy) of the half-length real FFT used by k_afftdn: Stockham radix-4 N/2-point complex FFT + split post/pre-processing."""
import numpy as np

def stockham4(z, inverse=False):
    N = z.size; T = np.exp((2j if inverse else -2j) * np.pi * np.arange(N) / N)
    a = z.copy(); Ns = 1
    while Ns < N:
        b = np.empty_like(a)
        for j in range(N // 4):
            k = j % Ns
            v = [a[j + t * (N // 4)] * T[(k * t * (N // (4 * Ns)))] for t in range(4)]
            s = 1j if inverse else -1j
            y0 = v[0] + v[1] + v[2] + v[3]
            y1 = v[0] + s * v[1] - v[2] - s * v[3]
            y2 = v[0] - v[1] + v[2] - v[3]
            y3 = v[0] - s * v[1] - v[2] + s * v[3]
            o = (j // Ns) * 4 * Ns + k
            b[o], b[o + Ns], b[o + 2 * Ns], b[o + 3 * Ns] = y0, y1, y2, y3
        a = b; Ns *= 4
    return a

rng = np.random.default_rng(0)
N = 2048; H = N // 2
x = rng.standard_normal(N)
z = x[0::2] + 1j * x[1::2]
Z = stockham4(z)
assert np.allclose(Z, np.fft.fft(z))
# post: X[k], k = 0..H
k = np.arange(H + 1)
Zk = Z[k % H]; Zm = np.conj(Z[(H - k) % H])
W = np.exp(-2j * np.pi * k / N)
X = 0.5 * (Zk + Zm) - 0.5j * W * (Zk - Zm)
assert np.allclose(X, np.fft.rfft(x))
# inverse (unnormalised: x_un[m] = sum over all N bins X e^{+...} = N * irfft)
g = rng.uniform(0, 1, H + 1); Y = X * g
want = np.fft.irfft(Y, N) * N
kk = np.arange(H)
Yk = Y[kk]; Ym = np.conj(Y[H - kk])
Wi = np.exp(2j * np.pi * kk / N)
Zp = (Yk + Ym) + 1j * Wi * (Yk - Ym)
zz = stockham4(Zp, inverse=True)
got = np.empty(N); got[0::2] = zz.real; got[1::2] = zz.imag
print(np.abs(got - want).max(), np.abs(want).max())
assert np.allclose(got, want)
print("ok")
