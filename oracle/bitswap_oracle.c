/*
 * bitswap_oracle.c -- CPU restatement of the Bit-Swap entropy-coding hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bitswap_amd/ may import, link or call
 * this file.  It exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check / time the HIP path against a plain-C statement of
 * what the reference (fhkingma/bitswap, pure Python) computes.
 *
 * Parity status
 *   integer half (tables from pmfs, rANS push/pop words): PINNED -- checked
 *     bit-for-bit against the reference's own ANS class run in-process
 *     (tests/golden/make_golden.py -> tests/golden/ fixtures, tests/test_oracle.py).
 *   float half (logistic CDF, float64 sigmoid): parity UNPINNED by any reference
 *     test; torch.sigmoid(float64) is a third-party routine (PyTorch 1.0.0 pinned in
 *     the reference README.md:91) whose last bit differs between builds/ISAs.
 *     mode 0 below restates the reference formula with libm exp() and is checked
 *     against torch within a few ulp; mode 1 restates the deterministic routine
 *     the HIP kernels implement (spec: DESIGN.md "Deterministic logistic CDF")
 *     and is checked bit-for-bit against the GPU.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define ORC_OK 0
#define ORC_UNDERFLOW 1   /* "too few initial bits": list.pop on an empty stack, mnist_compress.py:66 */
#define ORC_OVERFLOW 2    /* stack capacity exhausted (the reference list is unbounded)               */
#define ORC_BAD_TABLE 3   /* cdf[K] != 2^bits, the assert at mnist_compress.py:47                     */

/* ------------------------------------------------------------------------- */
/* Deterministic sigmoid (mode 1).  IEEE-754 binary64 operations only: sub,    */
/* mul, min, max, rint (ties-to-even), fma, exact scaling by 2^n, add and a    */
/* correctly rounded reciprocal.  Identical bits on any conforming machine.    */
/* ------------------------------------------------------------------------- */
static const double DET_LOG2E  = 0x1.71547652b82fep+0;
static const double DET_LN2_HI = 0x1.62e42fee00000p-1;
static const double DET_LN2_LO = 0x1.a39ef35793c76p-33;
static const double DET_C[12] = {
    1.0, 1.0,
    0x1.0000000000011p-1,  0x1.555555555555ap-3,  0x1.555555554f0ccp-5,
    0x1.111111110f224p-7,  0x1.6c16c187fc4dep-10, 0x1.a01a01b143bc8p-13,
    0x1.a01991ab61789p-16, 0x1.71ddf573e8618p-19, 0x1.28b4068ef93d2p-22,
    0x1.af631e4ea6521p-26,
};

static double det_pow2(int n) { /* exact 2^n, -1022 <= n <= 1023 */
    uint64_t b = (uint64_t)(n + 1023) << 52;
    double d;
    memcpy(&d, &b, 8);
    return d;
}

double orc_det_sigmoid(double t) {
    double a = -t;
    a = fmin(fmax(a, -700.0), 700.0);
    double kd = rint(a * DET_LOG2E);
    double r = fma(-kd, DET_LN2_HI, a);
    r = fma(-kd, DET_LN2_LO, r);
    double p = DET_C[11];
    for (int i = 10; i >= 0; --i) p = fma(p, r, DET_C[i]);
    double e = p * det_pow2((int)kd);
    return 1.0 / (1.0 + e);
}

/* exp part of the deterministic sigmoid: exp(a) for the clamped argument, same operations as above */
static double det_exp(double a) {
    a = fmin(fmax(a, -700.0), 700.0);
    double kd = rint(a * DET_LOG2E);
    double r = fma(-kd, DET_LN2_HI, a);
    r = fma(-kd, DET_LN2_LO, r);
    double p = DET_C[11];
    for (int i = 10; i >= 0; --i) p = fma(p, r, DET_C[i]);
    return p * det_pow2((int)kd);
}

/*
 * BS_CDF_SPEC 2 (mode 2) -- the logistic CDF of one row of UNIFORM-width bins (every latent layer but the top
 * one: discretization.py:81-83,105-118 -> numpy.linspace edges), evaluated at the same K-1 stored endpoints as
 * utils/torch/rand.py:67-68 but with ONE exponential per group of N = K/64 consecutive bins instead of one per bin:
 *   rs = 1/scale;  hr = h * rs                 (h = bin width of the row, supplied by the caller)
 *   Q_b = exp(-(b * hr)),  b = 1..N-1          (the geometric factor between a group's anchor and its b-th bin)
 *   anchor of group g (bin j0 = g*N):  t_a = (e[j0] - mu) * rs;  A = exp(-t_a);  E = A
 *   bin j = j0 + b:  r = e[j] - fma(b, h, e[j0])   (how far the stored endpoint is from the ideal progression:
 *                                                   a few 1e-16, independent of the chain)
 *                    eps = r * rs;  u = fma(-A, eps, A)   ( = A (1 - eps) );   x = fma(Q_b, u, 1)   ( = 1 + E )
 *   cdf_j = 1 / x                                 (anchor: x = 1 + A)
 * exp() is det_exp above.  The argument of the exponential is t_a + b*hr + eps = (e[j] - mu)/scale up to three
 * roundings of relative size 2^-53, so x - 1 equals exp(-t_j) to a few ulp: the integer tables agree with spec 1 /
 * torch at the same sub-ppm level (tests/test_oracle.py::test_cdf_spec2_*).  Every operation is a single
 * IEEE-754 binary64 operation; the HIP kernel (k_logistic, uniform flavour) performs the same ones.
 */
/* Returns 0 when the row leaves the domain of spec 2: N h / scale < 650 (the geometric factors Q_b = exp(-b h/scale) must
 * stay clear of det_exp's clamp at -700; a clamped anchor alone only touches bins that truncate to f = 1 either way) --
 * such rows are not coded, callers report ORC_BAD_TABLE as the HIP kernels report BS_ST_BADTABLE
 * (bitswap_hip.hip::logistic_row). */
static int det2_row_cdf(const double* e, double h, double mu, double scale, int K, double* cdf /* K-1 */) {
    const int N = K >= 64 ? K / 64 : 1;
    const double rs = 1.0 / scale;
    const double hr = h * rs;
    int ok = 1;
    double Q[64];
    for (int b = 1; b < N && b < 64; ++b) Q[b] = det_exp(-((double)b * hr));
    if (!((double)N * fabs(hr) < 650.0)) ok = 0;
    for (int j0 = 0; j0 < K - 1; j0 += N) {
        const double ta = (e[j0] - mu) * rs;
        const double A = det_exp(-ta);
        cdf[j0] = 1.0 / (1.0 + A);
        for (int b = 1; b < N && j0 + b < K - 1; ++b) {
            const double r = e[j0 + b] - fma((double)b, h, e[j0]);
            const double eps = r * rs;
            const double u = fma(-A, eps, A);
            cdf[j0 + b] = 1.0 / fma(Q[b], u, 1.0);
        }
    }
    return ok;
}

/* det_exp with the upper clamp of CDF spec 3's anchors: exp(min(max(a, -700), hi)) */
static double det_exp_hi(double a, double hi) {
    a = fmin(fmax(a, -700.0), hi);
    double kd = rint(a * DET_LOG2E);
    double r = fma(-kd, DET_LN2_HI, a);
    r = fma(-kd, DET_LN2_LO, r);
    double p = DET_C[11];
    for (int i = 10; i >= 0; --i) p = fma(p, r, DET_C[i]);
    return p * det_pow2((int)kd);
}

/*
 * BS_CDF_SPEC 3 (mode 3) -- spec 2 with ONE reciprocal per block of up to 16 bins instead of one per bin (batch
 * inversion over a balanced product tree): the denominators x_b = 1 + E_b of a block are multiplied up pairwise, the
 * root is inverted once (correctly rounded) and the quotients come back down the tree, one multiplication per node.
 * Same endpoints and the same formula as utils/torch/rand.py:67-68; every operation below is ONE IEEE-754 binary64
 * operation:
 *   rs = 1/scale;  hr = h * rs;  N = K/64 bins per group
 *   rows with N |hr| >= 8 (a scale tiny against the bin width: peaked pixel rows only): the row is spec 2's, bit for bit
 *   Q_b = exp(-(b * hr)),  b = 1..N-1
 *   group g (bins j0 = g N .. j0 + N - 1):
 *     t_a = (e[j0] - mu) * rs;  A = exp(min(-t_a, 41));  Ars = A * rs            (the anchor, clamped: see below)
 *     x_0 = 1 + A;   x_b = fma(Q_b, fma(-Ars, r_b, A), 1),  r_b = e[j0+b] - fma(b, h, e[j0])   (r_b = 0 for the K-th,
 *                                                             virtual endpoint that closes the last group)
 *     blocks of n = min(N, 16) consecutive bins (K = 2048: two per group), per block a balanced binary tree:
 *       level 0: T_0[i] = x_i;   level k: T_k[j] = T_{k-1}[2j] * T_{k-1}[2j+1];   root T_{log2 n}[0]
 *       I_root = 1 / root;       I_{k-1}[2j] = I_k[j] * T_{k-1}[2j+1];   I_{k-1}[2j+1] = I_k[j] * T_{k-1}[2j]
 *       cdf_i = I_0[i]
 * (IEEE multiplication commutes, so a node's value does not depend on who computes it: a wavefront that holds one bin
 * per lane builds the same tree with a butterfly exchange, k_rans_pop_pivot; one that holds the block in registers walks
 * it in any order, k_logistic.)  The clamp keeps the root below 2^947 (x <= 1 + e^41 < 2^59.2).  With N |hr| < 8 a
 * clamped group has E_b >= e^33 in every bin, i.e. cdf <= 2^-47: its bins -- and the first bin of the group behind it --
 * truncate to f = 1 exactly as with the unclamped anchor.  Rounding: a quotient carries 2 log2 n + 1 roundings instead
 * of one; against torch.sigmoid the integer tables differ in ~0.1 ppm of the entries, |df| = 1 (tests/test_oracle.py),
 * the same level as specs 1 and 2.
 */
/*
 * BS_CDF_SPEC 4 (mode 4, round 6) -- spec 3 with blocks of at most 8 bins (K >= 1024: two or four trees per group of K/64) and
 * ONE residual correction per quotient before it is used:   c <- fma(fma(-x_b, c, 1), c, c)   (x_b: the denominator the tree
 * was built from).  The product tree leaves a quotient with 2 log2 n + 1 roundings (relative error up to ~4 ulp); the Newton
 * step squares that error away and rounds once, so c is RN(1 / x_b) in all but ~2^-50 of the cases -- spec 2's accuracy against
 * torch.sigmoid (tests/test_oracle.py: ppm of table entries, divergence horizon on the reference's own 100-block chains) for
 * 5.75 instead of 9 issue slots per bin.  Everything else (anchor clamp, the N |hr| >= 8 rows that take spec 2's arithmetic bit
 * for bit, the virtual K-th endpoint) is spec 3's.  nmax / correct parametrise the one routine: spec 3 = (16, 0), spec 4 = (8, 1).
 */
static int det34_row_cdf(const double* e, double h, double mu, double scale, int K, double* cdf, int nmax, int correct);
static int det3_row_cdf(const double* e, double h, double mu, double scale, int K, double* cdf /* K-1 */) {
    return det34_row_cdf(e, h, mu, scale, K, cdf, 16, 0);
}
static int det4_row_cdf(const double* e, double h, double mu, double scale, int K, double* cdf /* K-1 */) {
    return det34_row_cdf(e, h, mu, scale, K, cdf, 8, 1);
}
static int det34_row_cdf(const double* e, double h, double mu, double scale, int K, double* cdf /* K-1 */, int nmax, int correct) {
    const int N = K >= 64 ? K / 64 : 1;
    const double rs = 1.0 / scale;
    const double hr = h * rs;
    if (N < 4 || !((double)N * fabs(hr) < 8.0)) return det2_row_cdf(e, h, mu, scale, K, cdf);
    double Q[64], x[64], T[5][16], I[5][16];
    for (int b = 1; b < N && b < 64; ++b) Q[b] = det_exp(-((double)b * hr));
    const int n = N < nmax ? N : nmax;
    int levels = 0;
    while ((1 << levels) < n) ++levels;
    for (int j0 = 0; j0 < K - 1; j0 += N) {
        const double ta = (e[j0] - mu) * rs;
        const double A = det_exp_hi(-ta, 41.0);
        const double Ars = A * rs;
        x[0] = 1.0 + A;
        for (int b = 1; b < N; ++b) {
            const double r = (j0 + b < K - 1) ? e[j0 + b] - fma((double)b, h, e[j0]) : 0.0;
            x[b] = fma(Q[b], fma(-Ars, r, A), 1.0);
        }
        for (int i0 = 0; i0 < N; i0 += n) {
            for (int i = 0; i < n; ++i) T[0][i] = x[i0 + i];
            for (int k = 1; k <= levels; ++k)
                for (int j = 0; j < (n >> k); ++j) T[k][j] = T[k - 1][2 * j] * T[k - 1][2 * j + 1];
            I[levels][0] = 1.0 / T[levels][0];
            for (int k = levels; k >= 1; --k)
                for (int j = 0; j < (n >> k); ++j) {
                    I[k - 1][2 * j] = I[k][j] * T[k - 1][2 * j + 1];
                    I[k - 1][2 * j + 1] = I[k][j] * T[k - 1][2 * j];
                }
            for (int i = 0; i < n; ++i)
                if (j0 + i0 + i < K - 1) {
                    double c = I[0][i];
                    if (correct) c = fma(fma(-x[i0 + i], c, 1.0), c, c);
                    cdf[j0 + i0 + i] = c;
                }
        }
    }
    return 1;
}

/* reference formula: torch.sigmoid((x - mu) / scale), utils/torch/rand.py:67-68 */
static double ref_sigmoid(double x, double mu, double scale) {
    double t = (x - mu) / scale;
    return 1.0 / (1.0 + exp(-t));
}

/*
 * orc_logistic_pmf: CDF at the K-1 interior endpoints + pmf assembly.
 *   reference: utils/torch/rand.py:67-68 (logistic_cdf) and
 *              mnist_compress.py:183-185 (adjacent difference, pad with
 *              cdf[0] and 1 - cdf[-1]).
 *   endpoints [D,K-1] f64 row-major (row d = zendpoints[zi][d, :]); mu, scale [D].
 *   mode 0: libm restatement of the reference formula.
 *   mode 1: the deterministic spec of the HIP kernels:
 *           rs = 1/scale (correctly rounded), t = (e - mu) * rs, det sigmoid.
 */
static int detN_row_cdf(int spec, const double* e, double h, double mu, double scale, int K, double* cdf) {
    return spec == 4 ? det4_row_cdf(e, h, mu, scale, K, cdf) : spec == 3 ? det3_row_cdf(e, h, mu, scale, K, cdf)
                                                                          : det2_row_cdf(e, h, mu, scale, K, cdf);
}

void orc_logistic_pmf_uniform(const double* endpoints, const double* step, const double* mu, const double* scale,
                              int64_t D, int K, int spec, double* pmf) {
    double* c = (double*)malloc(sizeof(double) * (size_t)K);
    for (int64_t d = 0; d < D; ++d) {
        double* p = pmf + d * (int64_t)K;
        detN_row_cdf(spec, endpoints + d * (int64_t)(K - 1), step[d], mu[d], scale[d], K, c);
        for (int j = 0; j < K - 1; ++j) p[j] = (j == 0) ? c[0] : c[j] - c[j - 1];
        p[K - 1] = 1.0 - c[K - 2];
    }
    free(c);
}

void orc_logistic_pmf(const double* endpoints, const double* mu, const double* scale,
                      int64_t D, int K, int mode, double* pmf) {
    for (int64_t d = 0; d < D; ++d) {
        const double* e = endpoints + d * (int64_t)(K - 1);
        double* p = pmf + d * (int64_t)K;
        double prev = 0.0;
        double rs = 1.0 / scale[d];
        for (int j = 0; j < K - 1; ++j) {
            double c = mode ? orc_det_sigmoid((e[j] - mu[d]) * rs)
                            : ref_sigmoid(e[j], mu[d], scale[d]);
            p[j] = (j == 0) ? c : c - prev;
            prev = c;
        }
        p[K - 1] = 1.0 - prev;
    }
}

/*
 * orc_tables: integer pmf/cdf tables, ANS.__init__ (mnist_compress.py:14-47).
 *   multiplier = 2^bits - 2^quantbits            (:29)
 *   f = trunc(pmf * multiplier) + 1              (:30,:33; .long() truncates toward zero)
 *   f[first argmax f] += 2^bits - sum f          (:36; torch.argmax -> first maximal index)
 *   cdf = [0, inclusive cumsum f]                (:39-40)
 *   assert cdf[K] == 2^bits                      (:47)
 * f_out [D,K], cdf_out [D,K+1] as uint32 (every value <= 2^31 for bits = 31).
 */
int orc_tables(const double* pmf, int64_t D, int K, int bits, int quantbits,
               uint32_t* f_out, uint32_t* cdf_out) {
    const double mult = (double)(((int64_t)1 << bits) - ((int64_t)1 << quantbits));
    int rc = ORC_OK;
    int64_t* f = (int64_t*)malloc(sizeof(int64_t) * (size_t)K);
    for (int64_t d = 0; d < D; ++d) {
        const double* p = pmf + d * (int64_t)K;
        int64_t sum = 0, best = INT64_MIN;
        int arg = 0;
        for (int j = 0; j < K; ++j) {
            f[j] = (int64_t)(p[j] * mult) + 1;
            sum += f[j];
            if (f[j] > best) { best = f[j]; arg = j; }
        }
        f[arg] += ((int64_t)1 << bits) - sum;
        int64_t acc = 0;
        uint32_t* fo = f_out ? f_out + d * (int64_t)K : 0;
        uint32_t* co = cdf_out + d * (int64_t)(K + 1);
        co[0] = 0;
        for (int j = 0; j < K; ++j) {
            if (fo) fo[j] = (uint32_t)f[j];
            if (f[j] < 1) rc = ORC_BAD_TABLE;
            acc += f[j];
            co[j + 1] = (uint32_t)acc;
        }
        if (acc != ((int64_t)1 << bits)) rc = ORC_BAD_TABLE;
    }
    free(f);
    return rc;
}

/*
 * orc_push: ANS.encode (mnist_compress.py:49-56).  Symbols are pushed in
 * order i = 0..D-1.  State = 64-bit head + stack of 32-bit words (the
 * reference keeps both in one Python list, head last).
 *   if head >= 2^33 * f: push(head & 0xffffffff); head >>= 32     (:52-54)
 *   head = (head // f) << bits + head % f + c                      (:55)
 * cdf rows have stride ld (>= K+1); f = cdf[s+1] - cdf[s].
 */
int orc_push(uint64_t* head, uint32_t* stack, int64_t* len, int64_t cap,
             const uint32_t* cdf, int64_t ld, int64_t D, int bits, const int32_t* sym) {
    uint64_t h = *head;
    int64_t n = *len;
    for (int64_t i = 0; i < D; ++i) {
        const uint32_t* row = cdf + i * ld;
        uint64_t c = row[sym[i]];
        uint64_t f = (uint64_t)row[sym[i] + 1] - c;
        /* ((lbound >> bits) << 32) * pmf with lbound = 2^32 (:22,:52) */
        if (h >= ((((uint64_t)1 << 32) >> bits) << 32) * f) {
            if (n >= cap) { *head = h; *len = n; return ORC_OVERFLOW; }
            stack[n++] = (uint32_t)(h & 0xffffffffu);
            h >>= 32;
        }
        h = ((h / f) << bits) + (h % f) + c;
    }
    *head = h;
    *len = n;
    return ORC_OK;
}

/* same, from per-symbol (f, c) pairs -- the layout the HIP encode-flavour kernel emits */
int orc_push_fc(uint64_t* head, uint32_t* stack, int64_t* len, int64_t cap,
                const uint32_t* fs, const uint32_t* cs, int64_t D, int bits) {
    uint64_t h = *head;
    int64_t n = *len;
    for (int64_t i = 0; i < D; ++i) {
        uint64_t f = fs[i], c = cs[i];
        if (h >= ((((uint64_t)1 << 32) >> bits) << 32) * f) {
            if (n >= cap) { *head = h; *len = n; return ORC_OVERFLOW; }
            stack[n++] = (uint32_t)(h & 0xffffffffu);
            h >>= 32;
        }
        h = ((h / f) << bits) + (h % f) + c;
    }
    *head = h;
    *len = n;
    return ORC_OK;
}

/*
 * orc_pop: ANS.decode (mnist_compress.py:58-68).  Symbols are popped in
 * order i = D-1..0.
 *   m = head & (2^bits - 1)                                        (:61)
 *   s = searchsorted(cdf[i, :-1], m, 'right') - 1                  (:62)
 *   head = f_s * (head >> bits) + m - c_s                          (:64)
 *   if head < 2^32: head = head << 32 | stack.pop()                (:65-66)
 */
int orc_pop(uint64_t* head, uint32_t* stack, int64_t* len,
            const uint32_t* cdf, int64_t ld, int64_t D, int K, int bits, int32_t* sym_out) {
    uint64_t h = *head;
    int64_t n = *len;
    const uint64_t mask = ((uint64_t)1 << bits) - 1;
    for (int64_t i = D - 1; i >= 0; --i) {
        const uint32_t* row = cdf + i * ld;
        uint64_t m = h & mask;
        int lo = 0, hi = K; /* number of entries in row[0..K-1] that are <= m */
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if ((uint64_t)row[mid] <= m) lo = mid + 1; else hi = mid;
        }
        int s = lo - 1;
        sym_out[i] = s;
        uint64_t c = row[s];
        uint64_t f = (uint64_t)row[s + 1] - c;
        h = f * (h >> bits) + m - c;
        if (h < ((uint64_t)1 << 32)) {
            if (n <= 0) { *head = h; *len = n; return ORC_UNDERFLOW; }
            h = (h << 32) | stack[--n];
        }
    }
    *head = h;
    *len = n;
    return ORC_OK;
}

/*
 * Fused layer operations used by the chain replay and the CPU baseline:
 * logistic rows -> integer table -> pop / push for ONE chain, one latent
 * layer, without materialising more than one row.  Same arithmetic as the
 * three functions above (mnist_compress.py:183-188 / :198-203).
 */
static int row_table(const double* e, double mu, double scale, int K, int bits, int quantbits,
                     int mode, double h, double* p, int64_t* f, uint32_t* c) {
    double prev = 0.0, rs = 1.0 / scale;
    int ok = 1;
    if (mode >= 2 && mode <= 4) {   /* CDF spec 2 / 3 / 4: uniform bins of width h */
        double* cd = (double*)malloc(sizeof(double) * (size_t)K);
        ok = detN_row_cdf(mode, e, h, mu, scale, K, cd);
        for (int j = 0; j < K - 1; ++j) p[j] = (j == 0) ? cd[0] : cd[j] - cd[j - 1];
        prev = cd[K - 2];
        free(cd);
    } else {
        for (int j = 0; j < K - 1; ++j) {
            double v = mode ? orc_det_sigmoid((e[j] - mu) * rs) : ref_sigmoid(e[j], mu, scale);
            p[j] = (j == 0) ? v : v - prev;
            prev = v;
        }
    }
    p[K - 1] = 1.0 - prev;
    const double mult = (double)(((int64_t)1 << bits) - ((int64_t)1 << quantbits));
    int64_t sum = 0, best = INT64_MIN;
    int arg = 0;
    for (int j = 0; j < K; ++j) {
        f[j] = (int64_t)(p[j] * mult) + 1;
        sum += f[j];
        if (f[j] > best) { best = f[j]; arg = j; }
    }
    f[arg] += ((int64_t)1 << bits) - sum;
    int64_t acc = 0;
    c[0] = 0;
    for (int j = 0; j < K; ++j) { acc += f[j]; c[j + 1] = (uint32_t)acc; }
    return ok;
}

/* Specs 2 and 3: every row of the layer inside the domain of det2_row_cdf?  Checked BEFORE anything is coded, like the HIP
 * table kernels, which flag the chain and leave its state untouched. */
static int layer_in_domain(const double* scale, int64_t D, int K, int mode, const double* step) {
    if (mode < 2 || mode > 4 || !step) return 1;
    const int N = K >= 64 ? K / 64 : 1;
    for (int64_t i = 0; i < D; ++i) {
        const double rs = 1.0 / scale[i], hr = step[i] * rs;
        if (!((double)N * fabs(hr) < 650.0)) return 0;
    }
    return 1;
}

/* step: bin width per row, used by mode 2 only (may be NULL otherwise) */
int orc_layer_pop(uint64_t* head, uint32_t* stack, int64_t* len,
                  const double* endpoints, const double* mu, const double* scale,
                  int64_t D, int K, int bits, int quantbits, int mode, const double* step, int32_t* sym_out) {
    double* p = (double*)malloc(sizeof(double) * (size_t)K);
    int64_t* f = (int64_t*)malloc(sizeof(int64_t) * (size_t)K);
    uint32_t* c = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(K + 1));
    int rc = layer_in_domain(scale, D, K, mode, step) ? ORC_OK : ORC_BAD_TABLE;
    for (int64_t i = D - 1; i >= 0 && rc == ORC_OK; --i) {
        row_table(endpoints + i * (int64_t)(K - 1), mu[i], scale[i], K, bits, quantbits, mode, step ? step[i] : 0.0, p, f, c);
        rc = orc_pop(head, stack, len, c, 0, 1, K, bits, sym_out + i);
    }
    free(p); free(f); free(c);
    return rc;
}

int orc_layer_push(uint64_t* head, uint32_t* stack, int64_t* len, int64_t cap,
                   const double* endpoints, const double* mu, const double* scale,
                   int64_t D, int K, int bits, int quantbits, int mode, const double* step, const int32_t* sym) {
    double* p = (double*)malloc(sizeof(double) * (size_t)K);
    int64_t* f = (int64_t*)malloc(sizeof(int64_t) * (size_t)K);
    uint32_t* c = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(K + 1));
    int rc = layer_in_domain(scale, D, K, mode, step) ? ORC_OK : ORC_BAD_TABLE;
    for (int64_t i = 0; i < D && rc == ORC_OK; ++i) {
        row_table(endpoints + i * (int64_t)(K - 1), mu[i], scale[i], K, bits, quantbits, mode, step ? step[i] : 0.0, p, f, c);
        rc = orc_push(head, stack, len, cap, c, 0, 1, bits, sym + i);
    }
    free(p); free(f); free(c);
    return rc;
}

int orc_version(void) { return 3; }
