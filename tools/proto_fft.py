"""Index-exact numpy prototype of the two-pass Hermitian four-step scheme used
by the HIP kernels (dev aid; mirrors xcorr_amd/csrc/*.hip).  Not product code."""
import numpy as np

def stockham(v, radices, inv):
    """v: [L, NB] complex; natural in, natural out. Mirrors the per-stage maps."""
    L = v.shape[0]
    sgn = 1.0 if inv else -1.0
    NS = 1
    for R in radices:
        out = np.empty_like(v)
        for j in range(L // R):
            k = j % NS
            x = np.stack([v[j + t * (L // R)] for t in range(R)])      # [R, NB]
            tw = np.exp(sgn * 2j * np.pi * k * np.arange(R) / (NS * R))
            x = x * tw[:, None]
            # radix-R DFT
            F = np.exp(sgn * 2j * np.pi * np.outer(np.arange(R), np.arange(R)) / R)
            y = F @ x
            j0 = (j - k) * R + k
            for t in range(R):
                out[j0 + t * NS] = y[t]
        v = out
        NS *= R
    return v

def fwd(x, n, N1, N2, r1, r2):
    """real x[n] -> spectrum layout Xmain[N1][N2h], Xnyq[N1] (unscaled)."""
    N = N1 * N2; N2h = N2 // 2
    xp = np.zeros(N); xp[:n] = x
    # F1: rows m1: u[m] = x[m1 + N1*2m] + i x[m1 + N1*(2m+1)]
    m1 = np.arange(N1)
    u = np.empty((N2h, N1), complex)
    for m in range(N2h):
        u[m] = xp[m1 + N1 * 2 * m] + 1j * xp[m1 + N1 * (2 * m + 1)]
    U = stockham(u, r2, False)                       # [N2h, N1rows]
    # untangle: H[k] = 0.5(U[k]+conj U[N2h-k]) - 0.5i e^{-2pi i k/N2}(U[k]-conj U[N2h-k])
    Tm = np.empty((N1, N2h), complex); Tn = np.empty(N1, complex)
    for k in range(N2h):
        a = U[k]; b = np.conj(U[(N2h - k) % N2h])
        H = 0.5 * (a + b) - 0.5j * np.exp(-2j * np.pi * k / N2) * (a - b)
        Tm[:, k] = H * np.exp(-2j * np.pi * m1 * k / N)
    a = U[0]; b = np.conj(U[0])
    Hn = 0.5 * (a + b) + 0.5j * (a - b)              # k = N2h: e^{-i pi} = -1
    Tn[:] = Hn * np.exp(-2j * np.pi * m1 * N2h / N)
    # F2: columns
    Xm = stockham(Tm, r1, False)                     # [N1(k1), N2h]
    Xn = stockham(Tn[:, None], r1, False)[:, 0]
    return Xm, Xn

def inv(Cm, Cn, n, N1, N2, r1, r2):
    """product spectrum layout -> circular correlation c[N] (unnormalised)."""
    N = N1 * N2; N2h = N2 // 2
    m1 = np.arange(N1)
    A = stockham(Cm, r1, True)                       # [m1, k2]
    k2 = np.arange(N2h)
    T = A * np.exp(2j * np.pi * np.outer(m1, k2) / N)
    An = stockham(Cn[:, None], r1, True)[:, 0]
    Tn = An * np.exp(2j * np.pi * m1 * N2h / N)
    # pack: T[m1][0] = (DC.re, Nyq.re)
    T[:, 0] = T[:, 0].real + 1j * Tn.real
    # I2 rows
    H = T.T.copy()                                   # [k, rows]
    U = np.empty_like(H)
    for k in range(N2h):
        if k == 0:
            dc = H[0].real; ny = H[0].imag
            U[0] = (dc + ny) + 1j * (dc - ny)
        else:
            a = H[k]; b = np.conj(H[N2h - k])
            U[k] = (a + b) + 1j * np.exp(2j * np.pi * k / N2) * (a - b)
    u = stockham(U, r2, True)                        # [m, rows]
    c = np.empty(N)
    for m in range(N2h):
        c[m1 + N1 * (2 * m)] = u[m].real
        c[m1 + N1 * (2 * m + 1)] = u[m].imag
    return c

if __name__ == '__main__':
    rng = np.random.default_rng(0)
    for (N1, r1, N2, r2, n) in [(8, [8], 16, [8], 60), (12, [2, 6], 20, [10], 100),
                                (24, [4, 6], 60, [6, 5], 700), (48, [8, 6], 96, [8, 6], 2000)]:
        N = N1 * N2
        x = rng.standard_normal(n); y = rng.standard_normal(n)
        Xm, Xn = fwd(x, n, N1, N2, r1, r2)
        X = np.fft.fft(np.pad(x, (0, N - n)))
        k1 = np.arange(N1)[:, None]; k2 = np.arange(N2 // 2)[None, :]
        print('fwd err', abs(Xm - X[(k1 * N2 + k2) % N]).max(), abs(Xn - X[(np.arange(N1) * N2 + N2 // 2) % N]).max())
        Ym, Yn = fwd(y, n, N1, N2, r1, r2)
        c = inv(Xm * np.conj(Ym), Xn * np.conj(Yn), n, N1, N2, r1, r2) / N
        ref = np.correlate(x, y, 'full')             # lags -(n-1)..(n-1)
        out = np.array([c[(j - (n - 1)) % N] for j in range(2 * n - 1)])
        print('cc err', abs(out - ref).max(), N, 2 * n - 1)
