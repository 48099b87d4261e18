"""NumPy model of the DEVICE algorithm (not the oracle).

This file restates, index for index, what the HIP kernels in
old-audiosync_amd/csrc/ do, in float64 NumPy, so the decomposition
(packed real FFT -> four-step complex FFT -> fused spectral combine -> inverse
four-step) can be checked on the CPU against numpy.fft before any GPU time is
spent.  It is test infrastructure only.

Notation (same as csrc/xcorr_kernels.hip):
  N   sample_len            F   real FFT length (even; F = 2N in the normal case)
  M   F/2 complex length    M = M1*M2   j = j1*M2 + j2   k = k1 + M1*k2
  layout of every intermediate: row-major [M1][M2]
"""
import numpy as np


def tw(F, p):
    """w_F^p = exp(-2*pi*i*p/F)"""
    return np.exp(-2j * np.pi * (np.asarray(p) % F) / F)


def fwd_cols(z, M1, M2):
    """kernel 1: DFT over j1 for every column j2, then twiddle w_M^(j2*k1).
    in : z[j1][j2]   out: A[k1][j2]"""
    M = M1 * M2
    a = np.fft.fft(z.reshape(M1, M2), axis=0)
    k1 = np.arange(M1)[:, None]
    j2 = np.arange(M2)[None, :]
    return a * tw(M, k1 * j2)


def fwd_rows(a):
    """first half of kernel 2: DFT over j2 of each row k1 -> Z[k1 + M1*k2] at [k1][k2]"""
    return np.fft.fft(a, axis=1)


def combine(Zx, Zy, M1, M2):
    """middle of kernel 2: from the permuted spectra of the two packed
    sequences build G (the packed spectrum of the inverse real transform),
    in the same permuted layout.  Mirrors the per-pair device code."""
    M = M1 * M2
    F = 2 * M
    G = np.zeros((M1, M2), dtype=complex)

    def pair(ax, bx, ay, by, k):
        # ax = Zx[k], bx = Zx[M-k], same for y.  returns G[k], G[M-k]
        w = tw(F, k)
        Ex = 0.5 * (ax + np.conj(bx)); Ox = -0.5j * (ax - np.conj(bx))
        Ey = 0.5 * (ay + np.conj(by)); Oy = -0.5j * (ay - np.conj(by))
        Xk = Ex + w * Ox; Xm = np.conj(Ex - w * Ox)
        Yk = Ey + w * Oy; Ym = np.conj(Ey - w * Oy)
        Pk = Xk * np.conj(Yk); Pm = Xm * np.conj(Ym)
        # inverse packing: G[k] = (P[k] + conj(P[M-k])) + i*conj(w^k)*(P[k] - conj(P[M-k]))
        Gk = (Pk + np.conj(Pm)) + 1j * np.conj(w) * (Pk - np.conj(Pm))
        # G[M-k]: w^(M-k) = -conj(w^k)  -> conj(w^(M-k)) = -w^k
        Gm = (Pm + np.conj(Pk)) + 1j * (-w) * (Pm - np.conj(Pk))
        return Gk, Gm

    for k1 in range(M1):
        m1 = (M1 - k1) % M1
        if k1 > m1 and m1 != 0:
            continue  # handled by the partner row
        for k2 in range(M2):
            k = k1 + M1 * k2
            if k == 0:
                z0x = Zx[0, 0]; z0y = Zx[0, 0] * 0 + Zy[0, 0]
                X0 = z0x.real + z0x.imag; XM = z0x.real - z0x.imag
                Y0 = z0y.real + z0y.imag; YM = z0y.real - z0y.imag
                P0 = X0 * Y0; PM = XM * YM
                G[0, 0] = (P0 + PM) + 1j * (P0 - PM)
                continue
            if k1 == 0:
                m2 = M2 - k2
            else:
                m2 = M2 - 1 - k2
            km = m1 + M1 * m2
            assert km == M - k
            if m1 == k1 and km < k:
                continue  # self-paired row: each pair once
            Gk, Gm = pair(Zx[k1, k2], Zx[m1, m2], Zy[k1, k2], Zy[m1, m2], k)
            G[k1, k2] = Gk
            if km != k:
                G[m1, m2] = Gm
    return G


def combine_pair_collapsed(ax, bx, ay, by, k, M):
    """the algebraically collapsed form k_rows uses (csrc/xcorr_kernels.hip::combine_pair):
    returns G[k], G[M-k] from Zx[k], Zx[M-k], Zy[k], Zy[M-k]"""
    Ex = ax + np.conj(bx); Ox = -1j * (ax - np.conj(bx))
    Ey = ay + np.conj(by); Oy = -1j * (ay - np.conj(by))
    W = Ex * np.conj(Ey) + Ox * np.conj(Oy)
    U = Ox * np.conj(Ey) + np.conj(tw(M, k)) * Ex * np.conj(Oy)
    return 0.5 * (W + 1j * U), 0.5 * (np.conj(W) + 1j * np.conj(U))


def inv_rows(G, M1, M2):
    """end of kernel 2: inverse DFT over k2 of each row k1, then twiddle conj(w_M^(k1*j2)).
    in: G[k1][k2]  out: B[k1][j2]"""
    M = M1 * M2
    b = np.fft.ifft(G, axis=1) * M2  # unnormalised
    k1 = np.arange(M1)[:, None]
    j2 = np.arange(M2)[None, :]
    return b * np.conj(tw(M, k1 * j2))


def inv_cols(B, M1, M2):
    """kernel 3: inverse DFT over k1 of each column -> g[j1*M2 + j2] at [j1][j2];
    r[2j] = Re g[j], r[2j+1] = Im g[j]"""
    g = np.fft.ifft(B, axis=0) * M1
    g = g.reshape(-1)
    r = np.empty(2 * g.size)
    r[0::2] = g.real
    r[1::2] = g.imag
    return r


def embed_params(N, smooth_even_at_least):
    """F selection: F = 2N when 2N is {2,3,5}-smooth, else the smallest
    smooth even F >= 3N-1 with the source extended periodically (no wrap)."""
    F = 2 * N
    if is_smooth(F):
        return F, 2 * N
    return smooth_even_at_least(3 * N - 1), 3 * N - 1


def is_smooth(n):
    for p in (2, 3, 5):
        while n % p == 0:
            n //= p
    return n == 1


def next_smooth_even(n):
    n += n & 1
    while not is_smooth(n):
        n += 2
    return n


def device_xcorr(source, sample, M1=None, M2=None):
    """whole device pipeline -> r[0..2N)"""
    N = len(sample)
    F, Ls = embed_params(N, next_smooth_even)
    M = F // 2
    if M1 is None:
        M1, M2 = split(M)
    assert M1 * M2 == M
    s = np.zeros(F); idx = np.arange(Ls); s[:Ls] = np.asarray(source)[idx % (2 * N)]
    t = np.zeros(F); t[:N] = sample
    zx = s[0::2] + 1j * s[1::2]
    zy = t[0::2] + 1j * t[1::2]
    Zx = fwd_rows(fwd_cols(zx, M1, M2))
    Zy = fwd_rows(fwd_cols(zy, M1, M2))
    G = combine(Zx, Zy, M1, M2)
    r = inv_cols(inv_rows(G, M1, M2), M1, M2)
    return r[: 2 * N]


# ---------------------------------------------------------------------------
# The real-column decomposition (csrc/rlayout.hip): the real sequence is the matrix x[j1][j2],
# j = j1*M2 + j2, with 2*M1 rows; k = k1 + 2*M1*k2.  Mirrors k_fwd_cols_r / k_rows_r / k_inv_cols_r
# step by step (packed rows, untangling inside the tile, row k1 independent, tangling inside the tile).
# ---------------------------------------------------------------------------

def rlayout_fwd_cols(x, M1, M2):
    """k_fwd_cols_r: x real, length 2*M1*M2 -> 2*C[k1][j2], k1 = 0..M1 (the factor 2 is taken back by the rows)"""
    X = np.asarray(x, dtype=float).reshape(2 * M1, M2)
    z = X[0::2] + 1j * X[1::2]                       # packed rows z[m] = x[2m] + i x[2m+1]
    Z = np.fft.fft(z, axis=0)                        # M1-point complex transform down the columns
    C2 = np.zeros((M1 + 1, M2), dtype=complex)
    for u in range(M1 // 2 + 1):
        za, zb = Z[u], Z[(M1 - u) % M1]
        S = za + np.conj(zb)
        Dm = -1j * (za - np.conj(zb))
        w = tw(2 * M1, u)
        C2[u] = S + w * Dm
        C2[M1 - u] = np.conj(S - w * Dm)
    return C2


def rlayout_rows(C2x, C2y, M1, M2):
    """k_rows_r: every row k1 on its own -- four-step twiddle (times 1/2), forward row transforms of both spectra,
    X conj(Y), inverse row transform, conjugate four-step twiddle -> Q[k1][j2]"""
    F = 2 * M1 * M2
    k1 = np.arange(M1 + 1)[:, None]
    j2 = np.arange(M2)[None, :]
    f = tw(F, k1 * j2)
    X = np.fft.fft(0.5 * C2x * f, axis=1)            # X[k1 + 2 M1 k2] at [k1][k2]
    Y = np.fft.fft(0.5 * C2y * f, axis=1)
    P = X * np.conj(Y)
    return np.fft.ifft(P, axis=1) * M2 * np.conj(f)  # unnormalised inverse


def rlayout_inv_cols(Q, M1, M2):
    """k_inv_cols_r: tangling of rows u and M1-u into the slots of the M1-point inverse transform, whose outputs are
    the packed rows z[m] = r[2m] + i r[2m+1]"""
    Zp = np.zeros((M1, M2), dtype=complex)
    for u in range(M1 // 2 + 1):
        qa, qb = Q[u], Q[M1 - u]
        S = qa + np.conj(qb)
        D = qa - np.conj(qb)
        t = 1j * np.conj(tw(2 * M1, u)) * D
        Zp[u] = S + t
        if (M1 - u) % M1 != u:
            Zp[M1 - u] = np.conj(S - t)
    z = np.fft.ifft(Zp, axis=0) * M1                 # unnormalised
    r = np.zeros((2 * M1, M2))
    r[0::2] = z.real
    r[1::2] = z.imag
    return r.reshape(-1)


def rlayout_xcorr(source, sample, M1, M2):
    """whole real-column pipeline -> r[0..2N), F = 2N = 2*M1*M2 (never embedded); r is F times the sum of products"""
    N = len(sample)
    assert 2 * N == 2 * M1 * M2 and M1 % 2 == 0
    t = np.zeros(2 * N); t[:N] = sample
    Q = rlayout_rows(rlayout_fwd_cols(source, M1, M2), rlayout_fwd_cols(t, M1, M2), M1, M2)
    return rlayout_inv_cols(Q, M1, M2)


def split(M):
    best = (1, M)
    for a in range(1, int(M ** 0.5) + 1):
        if M % a == 0:
            best = (a, M // a)
    return best


def reference_r(source, sample):
    """the reference recipe (src/cross_correlation.c:159-239) with numpy.fft"""
    N = len(sample)
    t = np.zeros(2 * N); t[:N] = sample
    X = np.fft.rfft(np.asarray(source, dtype=float)); Y = np.fft.rfft(t)
    return np.fft.irfft(X * np.conj(Y), 2 * N) * (2 * N)


# ---------------------------------------------------------------------------
# Spectral form of the Pearson coefficient (csrc/pearson_spectral.hip), step by step in float64:
#   window sums = whole bands of `band_rows` rows of the [rows][M2] sample matrix (per band x tile partial sums, as
#   k_fwd_cols_r leaves them) + the two band edges;  cross term = r[peak], minus -- for a negative lag -- the products
#   of the circular sum that did not wrap around.
# ---------------------------------------------------------------------------
def band_partials(x, M2, T, band_rows=8):
    """[nbands][ntiles][2]: sum and sum of squares of every (band, tile) cell; len(x) is a multiple of band_rows * M2"""
    x = np.asarray(x, dtype=float)
    rows = len(x) // M2
    m = x.reshape(rows // band_rows, band_rows, M2 // T, T)
    return np.stack([m.sum(axis=(1, 3)), (m * m).sum(axis=(1, 3))], axis=-1)


def window_sums(x, part, M2, lo, hi, band_rows=8):
    """sum and sum of squares of x[lo:hi] from whole bands of `part` plus the two edges (the whole range if it holds no band)"""
    gs = band_rows * M2
    ba, bb = -(-lo // gs), hi // gs
    x = np.asarray(x, dtype=float)
    if ba < bb:
        e = np.concatenate([x[lo: ba * gs], x[bb * gs: hi]])
        return e.sum() + part[ba:bb, :, 0].sum(), (e * e).sum() + part[ba:bb, :, 1].sum()
    e = x[lo:hi]
    return e.sum(), (e * e).sum()


def spectral_pearson(source, sample, peak, M2, T=16, band_rows=8):
    """(lag, coefficient, mode) as k_pearson_prep / k_pearson_final_spec form them, with the exact r[peak] and no error gate"""
    source = np.asarray(source, dtype=float); sample = np.asarray(sample, dtype=float)
    N = len(sample)
    r_peak = float(np.dot(np.roll(source, -peak)[:N], sample))      # the circular sum of src/cross_correlation.c:232-239
    if peak >= N:   # src/cross_correlation.c:256-263
        lag = (peak % N) - N
        so, mo, L = 0, -lag, N + lag
    else:           # :264-271
        lag, so, mo, L = peak, peak, 0, N
    px, py = band_partials(source, M2, T, band_rows), band_partials(sample, M2, T, band_rows)
    Sx, Sxx = window_sums(source, px, M2, so, so + L, band_rows)
    Sy, Syy = window_sums(sample, py, M2, mo, mo + L, band_rows)
    mode = "fast"
    sxy = r_peak
    if peak >= N:
        mode = "corr"
        c = N - L                                                   # = -lag products source[peak + n] * sample[n] did not wrap
        sxy -= float(np.dot(source[peak: peak + c], sample[:c]))
    if L < 2:
        return lag, float("nan"), mode
    coef = (sxy - Sx * Sy / L) / np.sqrt((Sxx - Sx * Sx / L) * (Syy - Sy * Sy / L))
    return lag, coef, mode


# ---------------------------------------------------------------------------
# In-place DIF / DIT stage model (what csrc/lds_fft.hip does inside LDS).
#   forward : natural order in  -> digit-reversed out   (Gentleman-Sande, DIF)
#   inverse : digit-reversed in -> natural order out    (Cooley-Tukey,   DIT)
# radices R_0..R_{s-1};  stage i works on sub-blocks of length ns_i = n / (R_0..R_{i-1})
# ---------------------------------------------------------------------------

def radix_list(n):
    out = []
    for r in (4, 2, 3, 5):
        while n % r == 0:
            out.append(r)
            n //= r
    assert n == 1, "length must be {2,3,5}-smooth"
    return out


def dif_forward_inplace(x, radices):
    x = np.array(x, dtype=complex)
    n = x.size
    ns = n
    for R in radices:
        q = ns // R
        y = x.copy()
        for b in range(n // ns):
            for j in range(q):
                v = np.array([x[b * ns + j + t * q] for t in range(R)])
                out = np.fft.fft(v)  # DFT_R
                for u in range(R):
                    y[b * ns + j + u * q] = out[u] * tw(ns, j * u)
        x = y
        ns = q
    return x


def dit_inverse_inplace(x, radices):
    x = np.array(x, dtype=complex)
    n = x.size
    # stage i of the forward run used ns_i; run them backwards
    ns_list = []
    ns = n
    for R in radices:
        ns_list.append(ns)
        ns //= R
    for R, ns in reversed(list(zip(radices, ns_list))):
        q = ns // R
        y = x.copy()
        for b in range(n // ns):
            for j in range(q):
                v = np.array([x[b * ns + j + u * q] * np.conj(tw(ns, j * u)) for u in range(R)])
                out = np.fft.ifft(v) * R
                for t in range(R):
                    y[b * ns + j + t * q] = out[t]
        x = y
    return x


def position_table(n, radices):
    """pos[k] = slot holding X[k] after dif_forward_inplace.
    k = u0 + R0*u1 + R0*R1*u2 + ... ;  pos = u0*(n/R0) + u1*(n/(R0*R1)) + ... """
    pos = np.zeros(n, dtype=np.int64)
    for k in range(n):
        rem, p, stride = k, 0, n
        for R in radices:
            stride //= R
            p += (rem % R) * stride
            rem //= R
        pos[k] = p
    return pos
