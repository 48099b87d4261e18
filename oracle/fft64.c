/* oracle/fft64.c — TEST INFRASTRUCTURE ONLY. See fft64.h for what is restated
 * and why (FFTW3 r2c/c2r as called at src/cross_correlation.c:34,237).
 *
 * Algorithm: iterative Stockham autosort, mixed radix {4,2,3,5,generic prime
 * <= 61}; a prime factor above 61 sends the whole length through Bluestein's
 * chirp-z with a power-of-two inner transform.  Everything is double.
 */
#include "fft64.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define OFFT_MAX_STAGES 64
#define OFFT_MAX_GENERIC 61

struct offt_plan {
    size_t n;
    int nstages;
    int radix[OFFT_MAX_STAGES];
    size_t ns[OFFT_MAX_STAGES];   /* product of the radices before this stage */
    ocpx *tw[OFFT_MAX_STAGES];    /* [ns][radix-1]: w_{ns*R}^(k*t), t = 1..R-1 */
    ocpx *wp[OFFT_MAX_STAGES];    /* generic radix only: w_R^q, q < R */
    /* Bluestein (only when n has a prime factor > OFFT_MAX_GENERIC) */
    size_t bm;                    /* inner power-of-two length, 0 if unused */
    offt_plan *inner;
    ocpx *chirp;                  /* exp(-i*pi*j^2/n), j < n */
    ocpx *chirp_spec;             /* forward DFT_bm of the wrapped conj chirp */
};

static ocpx unit(double num, double den)
{
    /* exp(-2*pi*i*num/den), argument reduced to the first octant so that the
     * libm calls see small arguments. */
    ocpx w;
    double x = fmod(num, den) / den; /* [0,1) */
    double a = 2.0 * M_PI * x;
    w.re = cos(a);
    w.im = -sin(a);
    /* exact values on the axes */
    if (x == 0.0) { w.re = 1; w.im = 0; }
    else if (x == 0.25) { w.re = 0; w.im = -1; }
    else if (x == 0.5) { w.re = -1; w.im = 0; }
    else if (x == 0.75) { w.re = 0; w.im = 1; }
    return w;
}

static inline ocpx cmul(ocpx a, ocpx b)
{
    ocpx r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
    return r;
}

static int factorize(size_t n, int *radix)
{
    int ns = 0;
    while (n % 4 == 0) { radix[ns++] = 4; n /= 4; }
    while (n % 2 == 0) { radix[ns++] = 2; n /= 2; }
    while (n % 3 == 0) { radix[ns++] = 3; n /= 3; }
    while (n % 5 == 0) { radix[ns++] = 5; n /= 5; }
    for (size_t p = 7; p <= OFFT_MAX_GENERIC && n > 1; p += 2)
        while (n % p == 0) { radix[ns++] = (int)p; n /= p; }
    if (n != 1) return -1; /* large prime factor left */
    return ns;
}

void offt_plan_destroy(offt_plan *p)
{
    if (!p) return;
    for (int s = 0; s < p->nstages; s++) { free(p->tw[s]); free(p->wp[s]); }
    if (p->inner) offt_plan_destroy(p->inner);
    free(p->chirp);
    free(p->chirp_spec);
    free(p);
}

static void stockham(const offt_plan *p, ocpx *a, ocpx *b);

offt_plan *offt_plan_create(size_t n)
{
    if (n == 0) return NULL;
    offt_plan *p = calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->n = n;
    int ns = factorize(n, p->radix);
    if (ns >= 0) {
        p->nstages = ns;
        size_t acc = 1;
        for (int s = 0; s < ns; s++) {
            int R = p->radix[s];
            p->ns[s] = acc;
            p->tw[s] = malloc(sizeof(ocpx) * acc * (size_t)(R - 1));
            if (!p->tw[s]) { offt_plan_destroy(p); return NULL; }
            for (size_t k = 0; k < acc; k++)
                for (int t = 1; t < R; t++)
                    p->tw[s][k * (size_t)(R - 1) + (size_t)(t - 1)] =
                        unit((double)k * t, (double)acc * R);
            if (R > 5) {
                p->wp[s] = malloc(sizeof(ocpx) * (size_t)R);
                if (!p->wp[s]) { offt_plan_destroy(p); return NULL; }
                for (int q = 0; q < R; q++) p->wp[s][q] = unit(q, R);
            }
            acc *= (size_t)R;
        }
        return p;
    }
    /* Bluestein: X[k] = c[k] * sum_j (x[j] c[j]) * conj(c[k-j]), c[j] = exp(-i*pi*j^2/n) */
    size_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    p->bm = m;
    p->inner = offt_plan_create(m);
    p->chirp = malloc(sizeof(ocpx) * n);
    p->chirp_spec = malloc(sizeof(ocpx) * m);
    ocpx *tmp = calloc(m, sizeof(ocpx));
    if (!p->inner || !p->chirp || !p->chirp_spec || !tmp) {
        free(tmp); offt_plan_destroy(p); return NULL;
    }
    for (size_t j = 0; j < n; j++) {
        /* j^2 mod 2n keeps the angle exact for large j */
        unsigned long long q = ((unsigned long long)j * j) % (2ull * n);
        p->chirp[j] = unit((double)q, 2.0 * (double)n);
    }
    tmp[0].re = p->chirp[0].re; tmp[0].im = -p->chirp[0].im;
    for (size_t j = 1; j < n; j++) {
        ocpx c = { p->chirp[j].re, -p->chirp[j].im };
        tmp[j] = c;
        tmp[m - j] = c;
    }
    offt_execute(p->inner, tmp, p->chirp_spec, -1);
    free(tmp);
    return p;
}

/* One forward Stockham pass sequence: input in a, scratch b; result ends in
 * whichever buffer the stage parity dictates; returns through *res. */
static ocpx *run_stages(const offt_plan *p, ocpx *a, ocpx *b)
{
    const size_t n = p->n;
    ocpx *in = a, *out = b;
    for (int s = 0; s < p->nstages; s++) {
        const int R = p->radix[s];
        const size_t Ns = p->ns[s];
        const size_t q = n / (size_t)R;
        const ocpx *tw = p->tw[s];
        for (size_t j = 0; j < q; j++) {
            const size_t k = j % Ns;
            const size_t base = (j - k) * (size_t)R + k;
            ocpx v[OFFT_MAX_GENERIC];
            v[0] = in[j];
            for (int t = 1; t < R; t++)
                v[t] = cmul(in[j + (size_t)t * q], tw[k * (size_t)(R - 1) + (size_t)(t - 1)]);
            if (R == 2) {
                ocpx y0 = { v[0].re + v[1].re, v[0].im + v[1].im };
                ocpx y1 = { v[0].re - v[1].re, v[0].im - v[1].im };
                out[base] = y0; out[base + Ns] = y1;
            } else if (R == 4) {
                ocpx s0 = { v[0].re + v[2].re, v[0].im + v[2].im };
                ocpx d0 = { v[0].re - v[2].re, v[0].im - v[2].im };
                ocpx s1 = { v[1].re + v[3].re, v[1].im + v[3].im };
                ocpx d1 = { v[1].re - v[3].re, v[1].im - v[3].im };
                /* -i * d1 */
                ocpx md1 = { d1.im, -d1.re };
                ocpx y0 = { s0.re + s1.re, s0.im + s1.im };
                ocpx y2 = { s0.re - s1.re, s0.im - s1.im };
                ocpx y1 = { d0.re + md1.re, d0.im + md1.im };
                ocpx y3 = { d0.re - md1.re, d0.im - md1.im };
                out[base] = y0; out[base + Ns] = y1;
                out[base + 2 * Ns] = y2; out[base + 3 * Ns] = y3;
            } else if (R == 3) {
                const double c = -0.5, sn = 0.86602540378443864676; /* sin(2pi/3) */
                ocpx t1 = { v[1].re + v[2].re, v[1].im + v[2].im };
                ocpx t2 = { v[1].re - v[2].re, v[1].im - v[2].im };
                ocpx y0 = { v[0].re + t1.re, v[0].im + t1.im };
                ocpx m = { v[0].re + c * t1.re, v[0].im + c * t1.im };
                /* -i*sn*t2 */
                ocpx r = { sn * t2.im, -sn * t2.re };
                ocpx y1 = { m.re + r.re, m.im + r.im };
                ocpx y2 = { m.re - r.re, m.im - r.im };
                out[base] = y0; out[base + Ns] = y1; out[base + 2 * Ns] = y2;
            } else if (R == 5) {
                const double c1 = 0.30901699437494742410, s1 = 0.95105651629515357212;
                const double c2 = -0.80901699437494742410, s2 = 0.58778525229247312917;
                ocpx a1 = { v[1].re + v[4].re, v[1].im + v[4].im };
                ocpx b1 = { v[1].re - v[4].re, v[1].im - v[4].im };
                ocpx a2 = { v[2].re + v[3].re, v[2].im + v[3].im };
                ocpx b2 = { v[2].re - v[3].re, v[2].im - v[3].im };
                ocpx y0 = { v[0].re + a1.re + a2.re, v[0].im + a1.im + a2.im };
                ocpx m1 = { v[0].re + c1 * a1.re + c2 * a2.re, v[0].im + c1 * a1.im + c2 * a2.im };
                ocpx m2 = { v[0].re + c2 * a1.re + c1 * a2.re, v[0].im + c2 * a1.im + c1 * a2.im };
                /* -i*(s1*b1 + s2*b2) and -i*(s2*b1 - s1*b2) */
                ocpx u1 = { s1 * b1.re + s2 * b2.re, s1 * b1.im + s2 * b2.im };
                ocpx u2 = { s2 * b1.re - s1 * b2.re, s2 * b1.im - s1 * b2.im };
                ocpx r1 = { u1.im, -u1.re };
                ocpx r2 = { u2.im, -u2.re };
                ocpx y1 = { m1.re + r1.re, m1.im + r1.im };
                ocpx y4 = { m1.re - r1.re, m1.im - r1.im };
                ocpx y2 = { m2.re + r2.re, m2.im + r2.im };
                ocpx y3 = { m2.re - r2.re, m2.im - r2.im };
                out[base] = y0; out[base + Ns] = y1; out[base + 2 * Ns] = y2;
                out[base + 3 * Ns] = y3; out[base + 4 * Ns] = y4;
            } else {
                const ocpx *wp = p->wp[s];
                for (int t = 0; t < R; t++) {
                    ocpx acc = v[0];
                    for (int u = 1; u < R; u++) {
                        ocpx w = wp[(u * t) % R];
                        acc.re += v[u].re * w.re - v[u].im * w.im;
                        acc.im += v[u].re * w.im + v[u].im * w.re;
                    }
                    out[base + (size_t)t * Ns] = acc;
                }
            }
        }
        ocpx *sw = in; in = out; out = sw;
    }
    return in;
}

static void stockham(const offt_plan *p, ocpx *a, ocpx *b)
{
    ocpx *res = run_stages(p, a, b);
    if (res != a) memcpy(a, res, sizeof(ocpx) * p->n);
}

/* smooth lengths only (no Bluestein): the same transform with the caller's scratch of n elements, for workers
 * that keep their buffers between calls (bench.py's node-throughput leg) */
int offt_execute_ws(const offt_plan *p, const ocpx *in, ocpx *out, int sign, ocpx *scratch)
{
    const size_t n = p->n;
    if (p->bm != 0) return -1;
    for (size_t j = 0; j < n; j++) {
        out[j].re = in[j].re;
        out[j].im = sign > 0 ? -in[j].im : in[j].im;
    }
    stockham(p, out, scratch);
    if (sign > 0)
        for (size_t j = 0; j < n; j++) out[j].im = -out[j].im;
    return 0;
}

void offt_execute(const offt_plan *p, const ocpx *in, ocpx *out, int sign)
{
    const size_t n = p->n;
    /* inverse = conj(forward(conj(.))) */
    if (p->bm == 0) {
        ocpx *scratch = malloc(sizeof(ocpx) * n);
        (void)offt_execute_ws(p, in, out, sign, scratch);
        free(scratch);
        return;
    }
    const size_t m = p->bm;
    ocpx *u = calloc(m, sizeof(ocpx));
    ocpx *v = malloc(sizeof(ocpx) * m);
    for (size_t j = 0; j < n; j++) {
        ocpx x = { in[j].re, sign > 0 ? -in[j].im : in[j].im };
        u[j] = cmul(x, p->chirp[j]);
    }
    offt_execute(p->inner, u, v, -1);
    for (size_t j = 0; j < m; j++) v[j] = cmul(v[j], p->chirp_spec[j]);
    offt_execute(p->inner, v, u, +1);
    const double inv = 1.0 / (double)m;
    for (size_t k = 0; k < n; k++) {
        ocpx y = { u[k].re * inv, u[k].im * inv };
        y = cmul(y, p->chirp[k]);
        out[k].re = y.re;
        out[k].im = sign > 0 ? -y.im : y.im;
    }
    free(u);
    free(v);
}

int offt_rfft(size_t L, const double *x, ocpx *X)
{
    if (L == 0) return -1;
    if (L % 2 == 0) {
        /* even length: one complex DFT of length M = L/2 on z[j] = x[2j] + i x[2j+1],
         * then X[k] = E[k] + w^k O[k],  E = (Z[k] + conj Z[M-k])/2, O = (Z[k] - conj Z[M-k])/(2i) */
        const size_t M = L / 2;
        offt_plan *p = offt_plan_create(M);
        ocpx *z = malloc(sizeof(ocpx) * M), *Z = malloc(sizeof(ocpx) * M);
        if (!p || !z || !Z) { offt_plan_destroy(p); free(z); free(Z); return -1; }
        for (size_t j = 0; j < M; j++) { z[j].re = x[2 * j]; z[j].im = x[2 * j + 1]; }
        offt_execute(p, z, Z, -1);
        X[0].re = Z[0].re + Z[0].im; X[0].im = 0.0;
        X[M].re = Z[0].re - Z[0].im; X[M].im = 0.0;
        for (size_t k = 1; k < M; k++) {
            ocpx a = Z[k], b = { Z[M - k].re, -Z[M - k].im };
            ocpx E = { 0.5 * (a.re + b.re), 0.5 * (a.im + b.im) };
            ocpx D = { 0.5 * (a.re - b.re), 0.5 * (a.im - b.im) };
            ocpx O = { D.im, -D.re }; /* D / i */
            ocpx w = unit((double)k, (double)L);
            ocpx wo = cmul(w, O);
            X[k].re = E.re + wo.re; X[k].im = E.im + wo.im;
        }
        offt_plan_destroy(p); free(z); free(Z);
        return 0;
    }
    offt_plan *p = offt_plan_create(L);
    ocpx *z = malloc(sizeof(ocpx) * L), *Z = malloc(sizeof(ocpx) * L);
    if (!p || !z || !Z) { offt_plan_destroy(p); free(z); free(Z); return -1; }
    for (size_t j = 0; j < L; j++) { z[j].re = x[j]; z[j].im = 0.0; }
    offt_execute(p, z, Z, -1);
    memcpy(X, Z, sizeof(ocpx) * (L / 2 + 1));
    offt_plan_destroy(p); free(z); free(Z);
    return 0;
}

int offt_irfft(size_t L, const ocpx *X, double *r)
{
    if (L == 0) return -1;
    if (L % 2 == 0) {
        /* G[k] = (X[k] + conj X[M-k]) + i conj(w^k) (X[k] - conj X[M-k]);  g = IDFT_M(G);
         * r[2j] = Re g[j], r[2j+1] = Im g[j].  Im X[0], Im X[M] are ignored (c2r contract). */
        const size_t M = L / 2;
        offt_plan *p = offt_plan_create(M);
        ocpx *G = malloc(sizeof(ocpx) * M), *g = malloc(sizeof(ocpx) * M);
        if (!p || !G || !g) { offt_plan_destroy(p); free(G); free(g); return -1; }
        G[0].re = X[0].re + X[M].re;
        G[0].im = X[0].re - X[M].re;
        for (size_t k = 1; k < M; k++) {
            ocpx a = X[k], b = { X[M - k].re, -X[M - k].im };
            ocpx S = { a.re + b.re, a.im + b.im };
            ocpx D = { a.re - b.re, a.im - b.im };
            ocpx w = unit((double)k, (double)L);
            ocpx wc = { w.re, -w.im };
            ocpx t = cmul(wc, D);
            /* + i*t */
            G[k].re = S.re - t.im;
            G[k].im = S.im + t.re;
        }
        offt_execute(p, G, g, +1);
        for (size_t j = 0; j < M; j++) { r[2 * j] = g[j].re; r[2 * j + 1] = g[j].im; }
        offt_plan_destroy(p); free(G); free(g);
        return 0;
    }
    offt_plan *p = offt_plan_create(L);
    ocpx *G = malloc(sizeof(ocpx) * L), *g = malloc(sizeof(ocpx) * L);
    if (!p || !G || !g) { offt_plan_destroy(p); free(G); free(g); return -1; }
    G[0].re = X[0].re; G[0].im = 0.0;
    for (size_t k = 1; k <= L / 2; k++) {
        G[k] = X[k];
        G[L - k].re = X[k].re; G[L - k].im = -X[k].im;
    }
    offt_execute(p, G, g, +1);
    for (size_t j = 0; j < L; j++) r[j] = g[j].re;
    offt_plan_destroy(p); free(G); free(g);
    return 0;
}
