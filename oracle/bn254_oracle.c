/*
 * bn254_oracle.c -- CPU restatement of the BN254 MSM / NTT hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the timed CPU
 * baseline ("port"). The product path (circuits_halo2_amd/csrc) never links or calls it.
 *
 * Where the algorithm comes from: the reference (summa-dev/circuits-halo2) only CALLS the
 * arithmetic (zk_prover/src/circuits/utils.rs:55,64,70,75,76,94-101,171-178); the code
 * lives in un-vendored crates halo2_proofs 0.2.0 @ summa-dev/halo2#8386d6e and
 * halo2curves 0.1.0 (zk_prover/Cargo.lock:2223-2276), absent from /root/reference.
 * This file restates their published algorithms:
 *   T1/T2  4x64-bit-limb Montgomery Fq/Fr, G1 affine 64 B / Jacobian
 *   M1     best_multiexp: per-thread contiguous chunks, each multiexp_serial with window
 *          c = 1 (n<4), 3 (n<32), ceil(ln n); 256/c+1 segments high->low, unsigned digits,
 *          zero digits skipped, running-sum bucket reduction; chunk results summed
 *   N1     best_fft: bit-reverse permutation, twiddle table, radix-2 DIT; threaded as a
 *          depth-first recursive split
 *   N2-N4  EvaluationDomain::{ifft, coeff_to_extended, extended_to_coeff,
 *          divide_by_vanishing_poly}
 * Parity pin: pinned against the reference's own artefacts -- SRS file
 * backend/ptau/hermez-raw-11 (K1,K3) and contracts/src/InclusionVerifier.sol:217-271
 * (K2 fixed_comms[4], K4 omega/omega_inv/n_inv) -- by tests/test_oracle_golden.py, and
 * against the independent big-integer twin oracle/pyref.py.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct { u64 l[4]; } fe;            /* field element, Montgomery form, LE limbs */
typedef struct { u64 p[4]; u64 inv; fe r1; fe r2; } field_t;

/* moduli, R mod p, R^2 mod p, -p^-1 mod 2^64 */
static const field_t FQ = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0x87d20782e4866389ULL,
    {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}},
    {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}}};
static const field_t FR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0xc2e1f593efffffffULL,
    {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}},
    {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}}};

/* ---------------------------------------------------------------- field arithmetic */
static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int geq_p(const u64 a[4], const u64 p[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > p[i]) return 1;
        if (a[i] < p[i]) return 0;
    }
    return 1;
}
static inline void sub_p(u64 a[4], const u64 p[4]) {
    u128 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - p[i] - (u64)b;
        a[i] = (u64)d;
        b = (d >> 64) & 1;
    }
}
static inline void fe_add(const field_t *F, fe *o, const fe *a, const fe *b) {
    u128 c = 0;
    u64 t[4];
    for (int i = 0; i < 4; i++) {
        c += (u128)a->l[i] + b->l[i];
        t[i] = (u64)c;
        c >>= 64;
    }
    /* p < 2^254 so no carry out of 256 bits */
    if (geq_p(t, F->p)) sub_p(t, F->p);
    memcpy(o->l, t, 32);
}
static inline void fe_sub(const field_t *F, fe *o, const fe *a, const fe *b) {
    u64 t[4];
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - (u64)br;
        t[i] = (u64)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t[i] + F->p[i];
            t[i] = (u64)c;
            c >>= 64;
        }
    }
    memcpy(o->l, t, 32);
}
static inline void fe_neg(const field_t *F, fe *o, const fe *a) {
    fe z = {{0, 0, 0, 0}};
    fe_sub(F, o, &z, a);
}
static inline void fe_dbl(const field_t *F, fe *o, const fe *a) { fe_add(F, o, a, a); }

/* Montgomery product a*b*2^-256 mod p, CIOS over 4 64-bit limbs */
static inline void fe_mul(const field_t *F, fe *o, const fe *a, const fe *b) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (u64)c;
        t[5] = (u64)(c >> 64);
        u64 m = t[0] * F->inv;
        c = (u128)m * F->p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * F->p[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (u64)c;
        t[4] = t[5] + (u64)(c >> 64);
    }
    if (t[4] || geq_p(t, F->p)) sub_p(t, F->p);
    memcpy(o->l, t, 32);
}
static inline void fe_sqr(const field_t *F, fe *o, const fe *a) { fe_mul(F, o, a, a); }
static void fe_to_mont(const field_t *F, fe *o, const fe *canon) { fe_mul(F, o, canon, &F->r2); }
static void fe_from_mont(const field_t *F, fe *o, const fe *m) {
    fe one = {{1, 0, 0, 0}};
    fe_mul(F, o, m, &one);
}
/* x^e, e given as 4 LE limbs (plain integer) */
static void fe_pow(const field_t *F, fe *o, const fe *x, const u64 e[4]) {
    fe acc = F->r1;
    for (int i = 255; i >= 0; i--) {
        fe_sqr(F, &acc, &acc);
        if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(F, &acc, &acc, x);
    }
    *o = acc;
}
static void fe_inv(const field_t *F, fe *o, const fe *x) {
    u64 e[4] = {F->p[0] - 2, F->p[1], F->p[2], F->p[3]}; /* p[0] >= 2, no borrow */
    fe_pow(F, o, x, e);
}

/* ---------------------------------------------------------------- G1 (y^2 = x^3 + 3) */
typedef struct { fe x, y; } g1a;        /* affine; identity = (0,0) */
typedef struct { fe x, y, z; } g1j;     /* Jacobian; identity: z == 0 */

static inline int g1a_is_id(const g1a *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline void g1j_set_id(g1j *p) { memset(p, 0, sizeof *p); p->x = FQ.r1; p->y = FQ.r1; }
static inline int g1j_is_id(const g1j *p) { return fe_is_zero(&p->z); }

static void g1j_double(g1j *o, const g1j *p) {
    if (g1j_is_id(p)) { *o = *p; return; }
    const field_t *F = &FQ;
    fe a, b, c, d, e, f, t, x3, y3, z3;
    fe_sqr(F, &a, &p->x);
    fe_sqr(F, &b, &p->y);
    fe_sqr(F, &c, &b);
    fe_add(F, &t, &p->x, &b);
    fe_sqr(F, &t, &t);
    fe_sub(F, &t, &t, &a);
    fe_sub(F, &t, &t, &c);
    fe_dbl(F, &d, &t);
    fe_dbl(F, &e, &a);
    fe_add(F, &e, &e, &a);
    fe_sqr(F, &f, &e);
    fe_dbl(F, &t, &d);
    fe_sub(F, &x3, &f, &t);
    fe_sub(F, &t, &d, &x3);
    fe_mul(F, &y3, &e, &t);
    fe_dbl(F, &t, &c);
    fe_dbl(F, &t, &t);
    fe_dbl(F, &t, &t);
    fe_sub(F, &y3, &y3, &t);
    fe_mul(F, &z3, &p->y, &p->z);
    fe_dbl(F, &z3, &z3);
    o->x = x3; o->y = y3; o->z = z3;
}
/* o = p + q (q affine), all corner cases */
static void g1j_add_affine(g1j *o, const g1j *p, const g1a *q) {
    const field_t *F = &FQ;
    if (g1a_is_id(q)) { *o = *p; return; }
    if (g1j_is_id(p)) { o->x = q->x; o->y = q->y; o->z = F->r1; return; }
    fe z1z1, u2, s2, h, hh, hhh, r, v, t, x3, y3, z3;
    fe_sqr(F, &z1z1, &p->z);
    fe_mul(F, &u2, &q->x, &z1z1);
    fe_mul(F, &s2, &q->y, &p->z);
    fe_mul(F, &s2, &s2, &z1z1);
    if (fe_eq(&u2, &p->x)) {
        if (fe_eq(&s2, &p->y)) { g1j_double(o, p); return; }
        g1j_set_id(o);
        return;
    }
    fe_sub(F, &h, &u2, &p->x);
    fe_sqr(F, &hh, &h);
    fe_mul(F, &hhh, &h, &hh);
    fe_sub(F, &r, &s2, &p->y);
    fe_mul(F, &v, &p->x, &hh);
    fe_sqr(F, &x3, &r);
    fe_sub(F, &x3, &x3, &hhh);
    fe_dbl(F, &t, &v);
    fe_sub(F, &x3, &x3, &t);
    fe_sub(F, &t, &v, &x3);
    fe_mul(F, &y3, &r, &t);
    fe_mul(F, &t, &p->y, &hhh);
    fe_sub(F, &y3, &y3, &t);
    fe_mul(F, &z3, &p->z, &h);
    o->x = x3; o->y = y3; o->z = z3;
}
/* o = p + q, both Jacobian */
static void g1j_add(g1j *o, const g1j *p, const g1j *q) {
    const field_t *F = &FQ;
    if (g1j_is_id(q)) { *o = *p; return; }
    if (g1j_is_id(p)) { *o = *q; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, hh, hhh, r, v, t, x3, y3, z3;
    fe_sqr(F, &z1z1, &p->z);
    fe_sqr(F, &z2z2, &q->z);
    fe_mul(F, &u1, &p->x, &z2z2);
    fe_mul(F, &u2, &q->x, &z1z1);
    fe_mul(F, &s1, &p->y, &q->z);
    fe_mul(F, &s1, &s1, &z2z2);
    fe_mul(F, &s2, &q->y, &p->z);
    fe_mul(F, &s2, &s2, &z1z1);
    if (fe_eq(&u1, &u2)) {
        if (fe_eq(&s1, &s2)) { g1j_double(o, p); return; }
        g1j_set_id(o);
        return;
    }
    fe_sub(F, &h, &u2, &u1);
    fe_sqr(F, &hh, &h);
    fe_mul(F, &hhh, &h, &hh);
    fe_sub(F, &r, &s2, &s1);
    fe_mul(F, &v, &u1, &hh);
    fe_sqr(F, &x3, &r);
    fe_sub(F, &x3, &x3, &hhh);
    fe_dbl(F, &t, &v);
    fe_sub(F, &x3, &x3, &t);
    fe_sub(F, &t, &v, &x3);
    fe_mul(F, &y3, &r, &t);
    fe_mul(F, &t, &s1, &hhh);
    fe_sub(F, &y3, &y3, &t);
    fe_mul(F, &z3, &p->z, &q->z);
    fe_mul(F, &z3, &z3, &h);
    o->x = x3; o->y = y3; o->z = z3;
}
static void g1j_to_affine(g1a *o, const g1j *p) {
    if (g1j_is_id(p)) { memset(o, 0, sizeof *o); return; }
    const field_t *F = &FQ;
    fe zi, zi2, zi3;
    fe_inv(F, &zi, &p->z);
    fe_sqr(F, &zi2, &zi);
    fe_mul(F, &zi3, &zi2, &zi);
    fe_mul(F, &o->x, &p->x, &zi2);
    fe_mul(F, &o->y, &p->y, &zi3);
}

/* ---------------------------------------------------------------- M1: best_multiexp */
static inline unsigned get_digit(const u64 canon[4], unsigned seg, unsigned c) {
    unsigned skip = seg * c;
    if (skip >= 256) return 0;
    unsigned limb = skip >> 6, off = skip & 63;
    u64 v = canon[limb] >> off;
    if (off + c > 64 && limb < 3) v |= canon[limb + 1] << (64 - off);
    return (unsigned)(v & ((1ULL << c) - 1));
}
static unsigned window_for(size_t n) {
    if (n < 4) return 1;
    if (n < 32) return 3;
    return (unsigned)ceil(log((double)n));
}
static void multiexp_serial(const fe *canon, const g1a *bases, size_t n, g1j *acc) {
    unsigned c = window_for(n);
    unsigned segments = 256 / c + 1;
    size_t nb = ((size_t)1 << c) - 1;
    g1j *buckets = (g1j *)malloc(nb * sizeof(g1j));
    for (int seg = (int)segments - 1; seg >= 0; seg--) {
        for (unsigned i = 0; i < c; i++) g1j_double(acc, acc);
        for (size_t b = 0; b < nb; b++) g1j_set_id(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            unsigned d = get_digit(canon[i].l, (unsigned)seg, c);
            if (d) g1j_add_affine(&buckets[d - 1], &buckets[d - 1], &bases[i]);
        }
        g1j running;
        g1j_set_id(&running);
        for (size_t b = nb; b-- > 0;) {
            g1j_add(&running, &running, &buckets[b]);
            g1j_add(acc, acc, &running);
        }
    }
    free(buckets);
}
typedef struct { const fe *canon; const g1a *bases; size_t n; g1j acc; } msm_job;
static void *msm_worker(void *arg) {
    msm_job *j = (msm_job *)arg;
    multiexp_serial(j->canon, j->bases, j->n, &j->acc);
    return NULL;
}
/* scalars: n x 32 B Fr Montgomery; bases: n x 64 B affine Montgomery; out: affine 64 B */
int orc_best_multiexp(const uint8_t *scalars, const uint8_t *bases, size_t n, int threads,
                      uint8_t out_affine[64]) {
    if (threads < 1) threads = 1;
    fe *canon = (fe *)malloc((n ? n : 1) * sizeof(fe));
    for (size_t i = 0; i < n; i++) fe_from_mont(&FR, &canon[i], (const fe *)(scalars + 32 * i));
    const g1a *pts = (const g1a *)bases;
    g1j total;
    g1j_set_id(&total);
    if (n > (size_t)threads && threads > 1) {
        size_t chunk = n / (size_t)threads;
        size_t nchunks = (n + chunk - 1) / chunk;
        msm_job *jobs = (msm_job *)calloc(nchunks, sizeof(msm_job));
        pthread_t *tid = (pthread_t *)calloc(nchunks, sizeof(pthread_t));
        for (size_t k = 0; k < nchunks; k++) {
            size_t s = k * chunk, e = s + chunk > n ? n : s + chunk;
            jobs[k].canon = canon + s; jobs[k].bases = pts + s; jobs[k].n = e - s;
            g1j_set_id(&jobs[k].acc);
            pthread_create(&tid[k], NULL, msm_worker, &jobs[k]);
        }
        for (size_t k = 0; k < nchunks; k++) {
            pthread_join(tid[k], NULL);
            g1j_add(&total, &total, &jobs[k].acc);
        }
        free(jobs); free(tid);
    } else {
        multiexp_serial(canon, pts, n, &total);
    }
    free(canon);
    g1j_to_affine((g1a *)out_affine, &total);
    return 0;
}

/* ---------------------------------------------------------------- N1: best_fft */
static inline u64 bitrev64(u64 x, unsigned bits) {
    u64 r = 0;
    for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
typedef struct { fe *a; size_t n; size_t tchunk; const fe *tw; int depth; } fft_job;
static void butterfly_combine(fe *a, size_t n, size_t tchunk, const fe *tw) {
    size_t half = n / 2;
    fe *l = a, *r = a + half;
    fe t = r[0];
    fe_sub(&FR, &r[0], &l[0], &t);
    fe_add(&FR, &l[0], &l[0], &t);
    for (size_t i = 1; i < half; i++) {
        fe_mul(&FR, &t, &r[i], &tw[i * tchunk]);
        fe_sub(&FR, &r[i], &l[i], &t);
        fe_add(&FR, &l[i], &l[i], &t);
    }
}
static void *fft_rec(void *arg) {
    fft_job *j = (fft_job *)arg;
    if (j->n == 2) {
        fe t = j->a[1];
        fe_sub(&FR, &j->a[1], &j->a[0], &t);
        fe_add(&FR, &j->a[0], &j->a[0], &t);
        return NULL;
    }
    fft_job L = {j->a, j->n / 2, j->tchunk * 2, j->tw, j->depth - 1};
    fft_job Rj = {j->a + j->n / 2, j->n / 2, j->tchunk * 2, j->tw, j->depth - 1};
    if (j->depth > 0) {
        pthread_t t;
        pthread_create(&t, NULL, fft_rec, &L);
        fft_rec(&Rj);
        pthread_join(t, NULL);
    } else {
        fft_rec(&L);
        fft_rec(&Rj);
    }
    butterfly_combine(j->a, j->n, j->tchunk, j->tw);
    return NULL;
}
static void best_fft_fe(fe *a, const fe *omega, unsigned log_n, int threads) {
    size_t n = (size_t)1 << log_n;
    if (log_n == 0) return;
    for (size_t k = 0; k < n; k++) {
        size_t rk = bitrev64(k, log_n);
        if (k < rk) { fe t = a[k]; a[k] = a[rk]; a[rk] = t; }
    }
    fe *tw = (fe *)malloc((n / 2 ? n / 2 : 1) * sizeof(fe));
    tw[0] = FR.r1;
    for (size_t i = 1; i < n / 2; i++) fe_mul(&FR, &tw[i], &tw[i - 1], omega);
    int depth = 0;
    while ((1 << (depth + 1)) <= threads) depth++;
    if ((unsigned)depth > log_n - 1) depth = (int)log_n - 1;
    fft_job j = {a, n, 1, tw, depth};
    fft_rec(&j);
    free(tw);
}
void orc_best_fft(uint8_t *a, const uint8_t omega[32], uint32_t log_n, int threads) {
    fe w;
    memcpy(&w, omega, 32);
    best_fft_fe((fe *)a, &w, log_n, threads);
}

/* ---------------------------------------------------------------- domain constants */
static const u64 ROOT_CANON[4] = {0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL, 0x98865ea93dd31f74ULL,
                                  0x03ddb9f5166d18b7ULL}; /* 2^28-th root of unity */
static const u64 ZETA_CANON[4] = {0xb8ca0b2d36636f23ULL, 0xcc37a73fec2bc5e9ULL, 0x048b6e193fd84104ULL,
                                  0x30644e72e131a029ULL}; /* Fr::ZETA */
static void fr_omega(unsigned k, fe *o) {
    fe root;
    fe c;
    memcpy(c.l, ROOT_CANON, 32);
    fe_to_mont(&FR, &root, &c);
    for (unsigned i = k; i < 28; i++) fe_sqr(&FR, &root, &root);
    *o = root;
}
void orc_omega(uint32_t k, uint8_t out[32]) { fe w; fr_omega(k, &w); memcpy(out, &w, 32); }
void orc_omega_inv(uint32_t k, uint8_t out[32]) { fe w; fr_omega(k, &w); fe_inv(&FR, &w, &w); memcpy(out, &w, 32); }
void orc_n_inv(uint32_t k, uint8_t out[32]) {
    fe c = {{0, 0, 0, 0}}, m;
    c.l[0] = 1ULL << k;
    fe_to_mont(&FR, &m, &c);
    fe_inv(&FR, &m, &m);
    memcpy(out, &m, 32);
}
void orc_zeta(uint8_t out[32]) {
    fe c, m;
    memcpy(c.l, ZETA_CANON, 32);
    fe_to_mont(&FR, &m, &c);
    memcpy(out, &m, 32);
}

/* ---------------------------------------------------------------- N2-N4 */
static void scale_all(fe *a, size_t n, const fe *f) {
    for (size_t i = 0; i < n; i++) fe_mul(&FR, &a[i], &a[i], f);
}
/* EvaluationDomain::ifft: best_fft(omega_inv) then multiply by divisor */
void orc_ifft(uint8_t *a, const uint8_t omega_inv[32], const uint8_t divisor[32], uint32_t log_n,
              int threads) {
    fe w, d;
    memcpy(&w, omega_inv, 32);
    memcpy(&d, divisor, 32);
    best_fft_fe((fe *)a, &w, log_n, threads);
    scale_all((fe *)a, (size_t)1 << log_n, &d);
}
/* EvaluationDomain::lagrange_to_coeff with the domain's own constants */
void orc_lagrange_to_coeff(uint8_t *a, uint32_t k, int threads) {
    uint8_t wi[32], ni[32];
    orc_omega_inv(k, wi);
    orc_n_inv(k, ni);
    orc_ifft(a, wi, ni, k, threads);
}
static void distribute_powers_zeta(fe *a, size_t n, int into_coset) {
    fe z, z2, zc;
    memcpy(zc.l, ZETA_CANON, 32);
    fe_to_mont(&FR, &z, &zc);
    fe_sqr(&FR, &z2, &z); /* zeta^2 = zeta^-1 */
    const fe *p1 = into_coset ? &z : &z2, *p2 = into_coset ? &z2 : &z;
    for (size_t i = 0; i < n; i++) {
        size_t m = i % 3;
        if (m == 1) fe_mul(&FR, &a[i], &a[i], p1);
        else if (m == 2) fe_mul(&FR, &a[i], &a[i], p2);
    }
}
/* coeffs: 2^k x 32 B -> out: 2^ext_k x 32 B */
void orc_coeff_to_extended(const uint8_t *coeffs, uint32_t k, uint32_t ext_k, uint8_t *out, int threads) {
    size_t n = (size_t)1 << k, en = (size_t)1 << ext_k;
    memcpy(out, coeffs, n * 32);
    memset(out + n * 32, 0, (en - n) * 32);
    distribute_powers_zeta((fe *)out, n, 1);
    fe w;
    fr_omega(ext_k, &w);
    best_fft_fe((fe *)out, &w, ext_k, threads);
}
/* 1/(X^n - 1) on the coset zeta*<omega_ext>: table of 2^(ext_k-k) entries */
void orc_t_evaluations(uint32_t k, uint32_t ext_k, uint8_t *out) {
    size_t t = (size_t)1 << (ext_k - k);
    fe w, z, zc, cur;
    fr_omega(ext_k, &w);
    memcpy(zc.l, ZETA_CANON, 32);
    fe_to_mont(&FR, &z, &zc);
    cur = z;
    for (size_t i = 0; i < t; i++) {
        fe v = cur;
        for (uint32_t s = 0; s < k; s++) fe_sqr(&FR, &v, &v); /* x^(2^k) */
        fe_sub(&FR, &v, &v, &FR.r1);
        fe_inv(&FR, &v, &v);
        memcpy(out + 32 * i, &v, 32);
        fe_mul(&FR, &cur, &cur, &w);
    }
}
void orc_divide_by_vanishing_poly(uint8_t *ext, uint32_t k, uint32_t ext_k) {
    size_t t = (size_t)1 << (ext_k - k), en = (size_t)1 << ext_k;
    fe *tab = (fe *)malloc(t * sizeof(fe));
    orc_t_evaluations(k, ext_k, (uint8_t *)tab);
    fe *a = (fe *)ext;
    for (size_t i = 0; i < en; i++) fe_mul(&FR, &a[i], &a[i], &tab[i % t]);
    free(tab);
}
/* in place over 2^ext_k elements; caller truncates to n*quotient_degree */
void orc_extended_to_coeff(uint8_t *ext, uint32_t k, uint32_t ext_k, int threads) {
    (void)k;
    orc_lagrange_to_coeff(ext, ext_k, threads);
    distribute_powers_zeta((fe *)ext, (size_t)1 << ext_k, 0);
}

/* ---------------------------------------------------------------- helpers for tests */
void orc_fr_mul(const uint8_t a[32], const uint8_t b[32], uint8_t o[32]) { fe r; fe_mul(&FR, &r, (const fe *)a, (const fe *)b); memcpy(o, &r, 32); }
void orc_fq_mul(const uint8_t a[32], const uint8_t b[32], uint8_t o[32]) { fe r; fe_mul(&FQ, &r, (const fe *)a, (const fe *)b); memcpy(o, &r, 32); }
void orc_fr_add(const uint8_t a[32], const uint8_t b[32], uint8_t o[32]) { fe r; fe_add(&FR, &r, (const fe *)a, (const fe *)b); memcpy(o, &r, 32); }
void orc_fr_sub(const uint8_t a[32], const uint8_t b[32], uint8_t o[32]) { fe r; fe_sub(&FR, &r, (const fe *)a, (const fe *)b); memcpy(o, &r, 32); }
void orc_fr_inv(const uint8_t a[32], uint8_t o[32]) { fe r; fe_inv(&FR, &r, (const fe *)a); memcpy(o, &r, 32); }
void orc_fr_to_mont(const uint8_t canon[32], uint8_t o[32]) { fe r; fe_to_mont(&FR, &r, (const fe *)canon); memcpy(o, &r, 32); }
void orc_fr_from_mont(const uint8_t m[32], uint8_t o[32]) { fe r; fe_from_mont(&FR, &r, (const fe *)m); memcpy(o, &r, 32); }
void orc_fr_to_mont_n(const uint8_t *canon, size_t n, uint8_t *o) { for (size_t i = 0; i < n; i++) orc_fr_to_mont(canon + 32 * i, o + 32 * i); }
void orc_fr_from_mont_n(const uint8_t *m, size_t n, uint8_t *o) { for (size_t i = 0; i < n; i++) orc_fr_from_mont(m + 32 * i, o + 32 * i); }

/* pointwise a[i]*b[i] and dot product (known-answer checks for MSM on s_i*G bases) */
void orc_fr_dot(const uint8_t *a, const uint8_t *b, size_t n, uint8_t o[32]) {
    fe acc = {{0, 0, 0, 0}}, t;
    for (size_t i = 0; i < n; i++) {
        fe_mul(&FR, &t, (const fe *)(a + 32 * i), (const fe *)(b + 32 * i));
        fe_add(&FR, &acc, &acc, &t);
    }
    memcpy(o, &acc, 32);
}
/* out[i] = tau^i (Montgomery) */
void orc_fr_powers(const uint8_t tau[32], size_t n, uint8_t *out) {
    fe cur = FR.r1, t;
    memcpy(&t, tau, 32);
    for (size_t i = 0; i < n; i++) { memcpy(out + 32 * i, &cur, 32); fe_mul(&FR, &cur, &cur, &t); }
}
/* Horner evaluation of sum coeffs[i] x^i */
void orc_fr_eval_poly(const uint8_t *coeffs, size_t n, const uint8_t x[32], uint8_t o[32]) {
    fe acc = {{0, 0, 0, 0}}, xx;
    memcpy(&xx, x, 32);
    for (size_t i = n; i-- > 0;) {
        fe_mul(&FR, &acc, &acc, &xx);
        fe_add(&FR, &acc, &acc, (const fe *)(coeffs + 32 * i));
    }
    memcpy(o, &acc, 32);
}

/* ff::BatchInvert::batch_invert: in place, zeros stay zero */
void orc_fr_batch_invert(uint8_t *a, size_t n) {
    fe *v = (fe *)a;
    fe *pre = (fe *)malloc((n ? n : 1) * sizeof(fe));
    fe acc = FR.r1;
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        if (!fe_is_zero(&v[i])) fe_mul(&FR, &acc, &acc, &v[i]);
    }
    fe_inv(&FR, &acc, &acc);
    for (size_t i = n; i-- > 0;) {
        if (fe_is_zero(&v[i])) continue;
        fe t;
        fe_mul(&FR, &t, &acc, &pre[i]);
        fe_mul(&FR, &acc, &acc, &v[i]);
        v[i] = t;
    }
    free(pre);
}
/* out[0] = 1, out[i] = a[0] * ... * a[i-1], i <= n (the z of a grand product) */
void orc_fr_prefix_product(const uint8_t *a, size_t n, uint8_t *out) {
    fe acc = FR.r1;
    for (size_t i = 0; i <= n; i++) {
        memcpy(out + 32 * i, &acc, 32);
        if (i < n) fe_mul(&FR, &acc, &acc, (const fe *)(a + 32 * i));
    }
}
/* halo2 arithmetic::kate_division: q(X) = (a(X) - a(b)) / (X - b); q has n - 1 coefficients */
void orc_fr_kate_division(const uint8_t *a, size_t n, const uint8_t b[32], uint8_t *q, uint8_t rem[32]) {
    fe bb, s = {{0, 0, 0, 0}};
    memcpy(&bb, b, 32);
    for (size_t i = n; i-- > 0;) {          /* s_i = a_i + b s_{i+1};  q_{i-1} = s_i */
        fe_mul(&FR, &s, &s, &bb);
        fe_add(&FR, &s, &s, (const fe *)(a + 32 * i));
        if (i >= 1) memcpy(q + 32 * (i - 1), &s, 32);
        else if (rem) memcpy(rem, &s, 32);
    }
}
void orc_fr_lincomb(const uint8_t *const *polys, const uint8_t *coeffs, uint32_t m, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        fe acc = {{0, 0, 0, 0}}, t;
        for (uint32_t j = 0; j < m; j++) {
            fe_mul(&FR, &t, (const fe *)(polys[j] + 32 * i), (const fe *)(coeffs + 32 * (size_t)j));
            fe_add(&FR, &acc, &acc, &t);
        }
        memcpy(out + 32 * i, &acc, 32);
    }
}
void orc_fr_mul_n(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) fe_mul(&FR, (fe *)(out + 32 * i), (const fe *)(a + 32 * i), (const fe *)(b + 32 * i));
}

/* halo2 permutation::prover::commit, one chunk:
 * z[0] = z0, z[i+1] = z[i] * prod_c (v_c[i] + delta_start*delta^c*omega^i*beta + gamma)
 *                           / prod_c (v_c[i] + beta*sigma_c[i] + gamma); n = 2^k values */
static const u64 DELTA_CANON[4] = {0x870e56bbe533e9a2ULL, 0x5b5f898e5e963f25ULL, 0x64ec26aad4c86e71ULL,
                                   0x09226b6e22c6f0caULL};
void orc_permutation_product(const uint8_t *const *values, const uint8_t *const *sigma, uint32_t ncols,
                             const uint8_t beta[32], const uint8_t gamma[32], const uint8_t delta_start[32],
                             uint32_t k, const uint8_t *z0, uint8_t *z_out) {
    size_t n = (size_t)1 << k;
    fe b, g, ds, delta, dc, omega;
    memcpy(&b, beta, 32); memcpy(&g, gamma, 32); memcpy(&ds, delta_start, 32);
    memcpy(dc.l, DELTA_CANON, 32);
    fe_to_mont(&FR, &delta, &dc);
    fr_omega(k, &omega);
    fe *mod = (fe *)malloc(n * sizeof(fe));
    for (size_t i = 0; i < n; i++) {
        fe acc = FR.r1, t;
        for (uint32_t c = 0; c < ncols; c++) {
            fe_mul(&FR, &t, &b, (const fe *)(sigma[c] + 32 * i));
            fe_add(&FR, &t, &t, &g);
            fe_add(&FR, &t, &t, (const fe *)(values[c] + 32 * i));
            fe_mul(&FR, &acc, &acc, &t);
        }
        mod[i] = acc;
    }
    orc_fr_batch_invert((uint8_t *)mod, n);
    fe wi = FR.r1;
    for (size_t i = 0; i < n; i++) {
        fe dw, t;
        fe_mul(&FR, &dw, &ds, &wi);
        for (uint32_t c = 0; c < ncols; c++) {
            fe_mul(&FR, &t, &dw, &b);
            fe_add(&FR, &t, &t, &g);
            fe_add(&FR, &t, &t, (const fe *)(values[c] + 32 * i));
            fe_mul(&FR, &mod[i], &mod[i], &t);
            fe_mul(&FR, &dw, &dw, &delta);
        }
        fe_mul(&FR, &wi, &wi, &omega);
    }
    fe z = FR.r1;
    if (z0) memcpy(&z, z0, 32);
    for (size_t i = 0; i < n; i++) {
        memcpy(z_out + 32 * i, &z, 32);
        fe_mul(&FR, &z, &z, &mod[i]);
    }
    free(mod);
}
/* halo2 lookup::prover::commit_product (step 6) on compressed / permuted columns */
void orc_lookup_product(const uint8_t *a, const uint8_t *s, const uint8_t *ap, const uint8_t *sp,
                        const uint8_t beta[32], const uint8_t gamma[32], size_t n, uint8_t *z_out) {
    fe b, g;
    memcpy(&b, beta, 32); memcpy(&g, gamma, 32);
    fe *mod = (fe *)malloc(n * sizeof(fe));
    for (size_t i = 0; i < n; i++) {
        fe x, y;
        fe_add(&FR, &x, (const fe *)(ap + 32 * i), &b);
        fe_add(&FR, &y, (const fe *)(sp + 32 * i), &g);
        fe_mul(&FR, &mod[i], &x, &y);
    }
    orc_fr_batch_invert((uint8_t *)mod, n);
    fe z = FR.r1;
    for (size_t i = 0; i < n; i++) {
        fe x, y;
        fe_add(&FR, &x, (const fe *)(a + 32 * i), &b);
        fe_add(&FR, &y, (const fe *)(s + 32 * i), &g);
        fe_mul(&FR, &x, &x, &y);
        fe_mul(&FR, &mod[i], &mod[i], &x);
        memcpy(z_out + 32 * i, &z, 32);
        fe_mul(&FR, &z, &z, &mod[i]);
    }
    free(mod);
}

/* ---------------------------------------------------------------- 8f-1: quotient numerator
 * The permutation and lookup blocks of halo2's Evaluator::evaluate_h (halo2_proofs
 * plonk/evaluation.rs in the summa-dev/halo2 fork pinned by zk_prover/Cargo.toml:13 -- a git
 * dependency, not vendored under /root/reference).  Formulas and fold order are anchored on the
 * reference's generated verifier, which evaluates the same expressions at the challenge point:
 * contracts/src/InclusionVerifier.sol:903-997.  Row i of the extended domain is zeta*omega_ext^i;
 * r_next / r_prev / r_last are index shifts by 2^(ext_k-k) * {+1, -1, -last_rotation_abs}. */
static inline const fe *row(const uint8_t *a, size_t i) { return (const fe *)(a + 32 * i); }
/* rows of the extended domain are independent (upstream's `parallelize` splits them over rayon threads): the three
 * quotient blocks below run their row loop on orc_set_quotient_threads() pthreads (default 1) */
static int g_quotient_threads = 1;
void orc_set_quotient_threads(int threads) { g_quotient_threads = threads < 1 ? 1 : threads > 256 ? 256 : threads; }
typedef void (*row_range_fn)(void *ctx, size_t lo, size_t hi);
typedef struct { row_range_fn fn; void *ctx; size_t lo, hi; } row_job;
static void *row_worker(void *p) { row_job *j = (row_job *)p; j->fn(j->ctx, j->lo, j->hi); return NULL; }
static void par_rows(row_range_fn fn, void *ctx, size_t n) {
    int t = g_quotient_threads;
    if (t <= 1 || n < 1024) { fn(ctx, 0, n); return; }
    row_job *jobs = (row_job *)calloc((size_t)t, sizeof(row_job));
    pthread_t *tid = (pthread_t *)calloc((size_t)t, sizeof(pthread_t));
    size_t per = (n + (size_t)t - 1) / (size_t)t;
    for (int k = 0; k < t; k++) {
        jobs[k].fn = fn; jobs[k].ctx = ctx; jobs[k].lo = (size_t)k * per < n ? (size_t)k * per : n;
        jobs[k].hi = jobs[k].lo + per < n ? jobs[k].lo + per : n;
        pthread_create(&tid[k], NULL, row_worker, &jobs[k]);
    }
    for (int k = 0; k < t; k++) pthread_join(tid[k], NULL);
    free(jobs); free(tid);
}
static void fold_term(fe *acc, const fe *y, const fe *term) {
    fe_mul(&FR, acc, acc, y);
    fe_add(&FR, acc, acc, term);
}
typedef struct {
    uint8_t *values; const uint8_t *const *z; uint32_t nsets; const uint8_t *const *cols; const uint8_t *const *sigma;
    uint32_t ncols, chunk_len; const uint8_t *l0, *l_last, *l_active; fe b, g, y, delta, beta_zeta, w_ext;
    size_t n_ext, rot, last_rotation_abs;
} perm_ctx;
static void quotient_permutation_rows(void *p, size_t lo, size_t hi) {
    const perm_ctx *c = (const perm_ctx *)p;
    const size_t n_ext = c->n_ext, rot = c->rot, mask = n_ext - 1;
    const uint8_t *const *z = c->z, *const *cols = c->cols, *const *sigma = c->sigma;
    const uint32_t nsets = c->nsets;
    fe beta_term = c->beta_zeta, wl = c->w_ext, acc_w = FR.r1;  /* beta * zeta * omega_ext^lo */
    for (size_t e = lo; e; e >>= 1) {
        if (e & 1) fe_mul(&FR, &acc_w, &acc_w, &wl);
        fe_sqr(&FR, &wl, &wl);
    }
    fe_mul(&FR, &beta_term, &beta_term, &acc_w);
    for (size_t i = lo; i < hi; i++) {
        const size_t i_next = (i + rot) & mask, i_last = (i + n_ext - c->last_rotation_abs * rot) & mask;
        fe acc = *row(c->values, i), t, u;
        /* l0 (1 - z_0) */
        fe_sub(&FR, &t, &FR.r1, row(z[0], i));
        fe_mul(&FR, &t, &t, row(c->l0, i));
        fold_term(&acc, &c->y, &t);
        /* l_last (z_l^2 - z_l) */
        fe_sqr(&FR, &t, row(z[nsets - 1], i));
        fe_sub(&FR, &t, &t, row(z[nsets - 1], i));
        fe_mul(&FR, &t, &t, row(c->l_last, i));
        fold_term(&acc, &c->y, &t);
        /* l0 (z_s - z_{s-1}(omega^-last X)) */
        for (uint32_t s = 1; s < nsets; s++) {
            fe_sub(&FR, &t, row(z[s], i), row(z[s - 1], i_last));
            fe_mul(&FR, &t, &t, row(c->l0, i));
            fold_term(&acc, &c->y, &t);
        }
        fe cur = beta_term;
        uint32_t col = 0;
        for (uint32_t s = 0; s < nsets; s++) {
            fe left = *row(z[s], i_next), right = *row(z[s], i);
            for (uint32_t j = 0; j < c->chunk_len && col < c->ncols; j++, col++) {
                fe_mul(&FR, &t, &c->b, row(sigma[col], i));
                fe_add(&FR, &t, &t, row(cols[col], i));
                fe_add(&FR, &t, &t, &c->g);
                fe_mul(&FR, &left, &left, &t);
                fe_add(&FR, &u, row(cols[col], i), &cur);
                fe_add(&FR, &u, &u, &c->g);
                fe_mul(&FR, &right, &right, &u);
                fe_mul(&FR, &cur, &cur, &c->delta);
            }
            fe_sub(&FR, &t, &left, &right);
            fe_mul(&FR, &t, &t, row(c->l_active, i));
            fold_term(&acc, &c->y, &t);
        }
        memcpy(c->values + 32 * i, &acc, 32);
        fe_mul(&FR, &beta_term, &beta_term, &c->w_ext);
    }
}
void orc_quotient_permutation(uint8_t *values, const uint8_t *const *z, uint32_t nsets, const uint8_t *const *cols,
                              const uint8_t *const *sigma, uint32_t ncols, uint32_t chunk_len, const uint8_t *l0,
                              const uint8_t *l_last, const uint8_t *l_active, const uint8_t beta[32],
                              const uint8_t gamma[32], const uint8_t y_[32], uint32_t k, uint32_t ext_k,
                              uint32_t last_rotation_abs) {
    perm_ctx c;
    fe dc, zeta, zc;
    c.values = values; c.z = z; c.nsets = nsets; c.cols = cols; c.sigma = sigma; c.ncols = ncols; c.chunk_len = chunk_len;
    c.l0 = l0; c.l_last = l_last; c.l_active = l_active;
    c.n_ext = (size_t)1 << ext_k; c.rot = (size_t)1 << (ext_k - k); c.last_rotation_abs = last_rotation_abs;
    memcpy(&c.b, beta, 32); memcpy(&c.g, gamma, 32); memcpy(&c.y, y_, 32);
    memcpy(dc.l, DELTA_CANON, 32); fe_to_mont(&FR, &c.delta, &dc);
    memcpy(zc.l, ZETA_CANON, 32); fe_to_mont(&FR, &zeta, &zc);
    fr_omega(ext_k, &c.w_ext);
    fe_mul(&FR, &c.beta_zeta, &c.b, &zeta);               /* beta * zeta * omega_ext^i, advanced per row */
    par_rows(quotient_permutation_rows, &c, c.n_ext);
}
typedef struct {
    uint8_t *values; const uint8_t *z, *ap, *sp, *a, *s, *l0, *l_last, *l_active; fe b, g, y; size_t n_ext, rot;
} lookup_ctx;
static void quotient_lookup_rows(void *p, size_t lo, size_t hi) {
    const lookup_ctx *c = (const lookup_ctx *)p;
    const size_t n_ext = c->n_ext, rot = c->rot, mask = n_ext - 1;
    uint8_t *values = c->values;
    const uint8_t *z = c->z, *ap = c->ap, *sp = c->sp, *a = c->a, *s = c->s, *l0 = c->l0, *l_last = c->l_last, *l_active = c->l_active;
    const fe b = c->b, g = c->g, y = c->y;
    for (size_t i = lo; i < hi; i++) {
        const size_t i_next = (i + rot) & mask, i_prev = (i + n_ext - rot) & mask;
        fe acc = *row(values, i), t, u, v, d;
        fe_sub(&FR, &t, &FR.r1, row(z, i));                       /* l0 (1 - z) */
        fe_mul(&FR, &t, &t, row(l0, i));
        fold_term(&acc, &y, &t);
        fe_sqr(&FR, &t, row(z, i));                               /* l_last (z^2 - z) */
        fe_sub(&FR, &t, &t, row(z, i));
        fe_mul(&FR, &t, &t, row(l_last, i));
        fold_term(&acc, &y, &t);
        fe_add(&FR, &u, row(ap, i), &b);                          /* z(wX)(a'+beta)(s'+gamma) */
        fe_add(&FR, &v, row(sp, i), &g);
        fe_mul(&FR, &t, &u, &v);
        fe_mul(&FR, &t, &t, row(z, i_next));
        fe_add(&FR, &u, row(a, i), &b);                           /* z(X)(a+beta)(s+gamma) */
        fe_add(&FR, &v, row(s, i), &g);
        fe_mul(&FR, &u, &u, &v);
        fe_mul(&FR, &u, &u, row(z, i));
        fe_sub(&FR, &t, &t, &u);
        fe_mul(&FR, &t, &t, row(l_active, i));
        fold_term(&acc, &y, &t);
        fe_sub(&FR, &d, row(ap, i), row(sp, i));                  /* l0 (a' - s') */
        fe_mul(&FR, &t, &d, row(l0, i));
        fold_term(&acc, &y, &t);
        fe_sub(&FR, &u, row(ap, i), row(ap, i_prev));             /* l_active (a'-s')(a'-a'(w^-1 X)) */
        fe_mul(&FR, &t, &d, &u);
        fe_mul(&FR, &t, &t, row(l_active, i));
        fold_term(&acc, &y, &t);
        memcpy(values + 32 * i, &acc, 32);
    }
}
void orc_quotient_lookup(uint8_t *values, const uint8_t *z, const uint8_t *ap, const uint8_t *sp, const uint8_t *a,
                         const uint8_t *s, const uint8_t *l0, const uint8_t *l_last, const uint8_t *l_active,
                         const uint8_t beta[32], const uint8_t gamma[32], const uint8_t y_[32], uint32_t k,
                         uint32_t ext_k) {
    lookup_ctx c;
    c.values = values; c.z = z; c.ap = ap; c.sp = sp; c.a = a; c.s = s; c.l0 = l0; c.l_last = l_last; c.l_active = l_active;
    c.n_ext = (size_t)1 << ext_k; c.rot = (size_t)1 << (ext_k - k);
    memcpy(&c.b, beta, 32); memcpy(&c.g, gamma, 32); memcpy(&c.y, y_, 32);
    par_rows(quotient_lookup_rows, &c, c.n_ext);
}

/* ---------------------------------------------------------------- 8f-1: custom gates
 * Interpreter for halo2's GraphEvaluator program (plonk/evaluation.rs in the pinned summa-dev/halo2
 * fork; not vendored): calculations[i] defines intermediate i from two value sources; the value of
 * the LAST calculation is the new values[row] (the running numerator enters as PreviousValue).
 * Same plain-struct ABI as include/summa_gpu.h (sg_graph). */
typedef struct { uint32_t kind, index, rotation; } orc_value_source;
typedef struct { uint32_t op; orc_value_source a, b; uint32_t parts_offset, parts_len; } orc_calculation;
typedef struct {
    const uint8_t *constants; uint32_t n_constants;
    const int32_t *rotations; uint32_t n_rotations;
    const orc_calculation *calculations; uint32_t n_calculations;
    const orc_value_source *horner_parts; uint32_t n_horner_parts;
} orc_graph;
enum { VS_CONSTANT, VS_INTERMEDIATE, VS_FIXED, VS_ADVICE, VS_INSTANCE, VS_CHALLENGE, VS_BETA, VS_GAMMA, VS_THETA, VS_Y,
       VS_PREVIOUS };
enum { OP_ADD, OP_SUB, OP_MUL, OP_SQUARE, OP_DOUBLE, OP_NEGATE, OP_HORNER, OP_STORE };
typedef struct {
    const orc_graph *g;
    const uint8_t *const *fixed, *const *advice, *const *instance;
    const uint8_t *challenges;
    fe beta, gamma, theta, y, prev;
    const fe *inter;
    size_t row, n_ext, rot_scale;
} gate_ctx;
static fe gate_value(const gate_ctx *c, const orc_value_source *v) {
    fe r;
    switch (v->kind) {
    case VS_CONSTANT: memcpy(&r, c->g->constants + 32 * (size_t)v->index, 32); return r;
    case VS_INTERMEDIATE: return c->inter[v->index];
    case VS_FIXED: case VS_ADVICE: case VS_INSTANCE: {
        const uint8_t *const *cols = v->kind == VS_FIXED ? c->fixed : v->kind == VS_ADVICE ? c->advice : c->instance;
        long long rot = c->g->rotations[v->rotation];
        size_t i = (size_t)(((long long)c->row + rot * (long long)c->rot_scale) & (long long)(c->n_ext - 1));
        return *row(cols[v->index], i);
    }
    case VS_CHALLENGE: memcpy(&r, c->challenges + 32 * (size_t)v->index, 32); return r;
    case VS_BETA: return c->beta;
    case VS_GAMMA: return c->gamma;
    case VS_THETA: return c->theta;
    case VS_Y: return c->y;
    default: return c->prev;
    }
}
typedef struct { gate_ctx base; uint8_t *values; } gates_job;
static void quotient_gates_rows(void *p, size_t lo, size_t hi) {
    const gates_job *job = (const gates_job *)p;
    gate_ctx c = job->base;
    const orc_graph *g = c.g;
    uint8_t *values = job->values;
    fe *inter = (fe *)malloc(sizeof(fe) * (g->n_calculations ? g->n_calculations : 1));
    c.inter = inter;
    for (size_t i = lo; i < hi; i++) {
        c.row = i;
        c.prev = *row(values, i);
        for (uint32_t q = 0; q < g->n_calculations; q++) {
            const orc_calculation *cal = &g->calculations[q];
            fe a = gate_value(&c, &cal->a), b, r;
            switch (cal->op) {
            case OP_ADD: b = gate_value(&c, &cal->b); fe_add(&FR, &r, &a, &b); break;
            case OP_SUB: b = gate_value(&c, &cal->b); fe_sub(&FR, &r, &a, &b); break;
            case OP_MUL: b = gate_value(&c, &cal->b); fe_mul(&FR, &r, &a, &b); break;
            case OP_SQUARE: fe_sqr(&FR, &r, &a); break;
            case OP_DOUBLE: fe_dbl(&FR, &r, &a); break;
            case OP_NEGATE: fe_neg(&FR, &r, &a); break;
            case OP_HORNER: {   /* start * factor^len + parts[0] * factor^(len-1) + ... + parts[len-1] */
                b = gate_value(&c, &cal->b);
                r = a;
                for (uint32_t t = 0; t < cal->parts_len; t++) {
                    fe part = gate_value(&c, &g->horner_parts[cal->parts_offset + t]);
                    fe_mul(&FR, &r, &r, &b);
                    fe_add(&FR, &r, &r, &part);
                }
                break;
            }
            default: r = a; break;  /* OP_STORE */
            }
            inter[q] = r;
        }
        if (g->n_calculations) memcpy(values + 32 * i, &inter[g->n_calculations - 1], 32);
    }
    free(inter);
}
void orc_quotient_gates(uint8_t *values, const orc_graph *g, const uint8_t *const *fixed, const uint8_t *const *advice,
                        const uint8_t *const *instance, const uint8_t *challenges, const uint8_t beta[32],
                        const uint8_t gamma[32], const uint8_t theta[32], const uint8_t y[32], uint32_t k,
                        uint32_t ext_k) {
    gates_job job;
    gate_ctx *c = &job.base;
    c->g = g; c->fixed = fixed; c->advice = advice; c->instance = instance; c->challenges = challenges;
    memcpy(&c->beta, beta, 32); memcpy(&c->gamma, gamma, 32); memcpy(&c->theta, theta, 32); memcpy(&c->y, y, 32);
    c->n_ext = (size_t)1 << ext_k; c->rot_scale = (size_t)1 << (ext_k - k);
    c->inter = NULL; c->row = 0;
    job.values = values;
    par_rows(quotient_gates_rows, &job, c->n_ext);
}

/* ---------------------------------------------------------------- witness side (row W)
 * Poseidon(t = 2, rate 1, R_F = 8, R_P = 56, x^5) sponge of halo2_gadgets as used by
 * zk_prover/src/merkle_sum_tree/node.rs:57-84; parameters are supplied by the caller
 * (oracle/poseidon_params.py regenerates them from the published Grain-LFSR procedure). */
static fe PSD_RC[64][2], PSD_MDS[2][2];
void orc_poseidon_set_params(const uint8_t *rc_mont /*128 x 32 B*/, const uint8_t *mds_mont /*4 x 32 B*/) {
    memcpy(PSD_RC, rc_mont, sizeof PSD_RC);
    memcpy(PSD_MDS, mds_mont, sizeof PSD_MDS);
}
static void psd_pow5(fe *x) {
    fe x2, x4;
    fe_sqr(&FR, &x2, x);
    fe_sqr(&FR, &x4, &x2);
    fe_mul(&FR, x, &x4, x);
}
static void psd_mix(fe s[2]) {
    fe a, b, t;
    fe_mul(&FR, &a, &PSD_MDS[0][0], &s[0]); fe_mul(&FR, &t, &PSD_MDS[0][1], &s[1]); fe_add(&FR, &a, &a, &t);
    fe_mul(&FR, &b, &PSD_MDS[1][0], &s[0]); fe_mul(&FR, &t, &PSD_MDS[1][1], &s[1]); fe_add(&FR, &b, &b, &t);
    s[0] = a; s[1] = b;
}
static void psd_permute(fe s[2]) {
    int r = 0;
    for (int k = 0; k < 4; k++, r++) {
        for (int i = 0; i < 2; i++) { fe_add(&FR, &s[i], &s[i], &PSD_RC[r][i]); psd_pow5(&s[i]); }
        psd_mix(s);
    }
    for (int k = 0; k < 56; k++, r++) {
        for (int i = 0; i < 2; i++) fe_add(&FR, &s[i], &s[i], &PSD_RC[r][i]);
        psd_pow5(&s[0]);
        psd_mix(s);
    }
    for (int k = 0; k < 4; k++, r++) {
        for (int i = 0; i < 2; i++) { fe_add(&FR, &s[i], &s[i], &PSD_RC[r][i]); psd_pow5(&s[i]); }
        psd_mix(s);
    }
}
/* Hash<_, _, ConstantLength<L>, 2, 1>: state = [0, L * 2^64]; absorb one input per permutation */
void orc_poseidon_hash(const uint8_t *inputs, size_t L, uint8_t out[32]) {
    fe s[2], cap = {{0, (u64)L, 0, 0}};
    memset(&s[0], 0, sizeof(fe));
    fe_to_mont(&FR, &s[1], &cap);
    for (size_t i = 0; i < L; i++) {
        fe_add(&FR, &s[0], &s[0], (const fe *)(inputs + 32 * i));
        psd_permute(s);
    }
    memcpy(out, &s[0], 32);
}
/* leaves: hash_i = H(user_i, bal_i0 .. bal_i,nc-1) */
void orc_mst_leaves(const uint8_t *users, const uint8_t *balances, size_t n, uint32_t nc, uint8_t *hashes) {
    uint8_t *pre = (uint8_t *)malloc(32 * (nc + 1));
    for (size_t i = 0; i < n; i++) {
        memcpy(pre, users + 32 * i, 32);
        memcpy(pre + 32, balances + 32 * (size_t)nc * i, 32 * (size_t)nc);
        orc_poseidon_hash(pre, nc + 1, hashes + 32 * i);
    }
    free(pre);
}
/* one level: parent p of children 2p, 2p+1 (build_tree.rs:54-78) */
void orc_mst_level(const uint8_t *child_hash, const uint8_t *child_bal, size_t m, uint32_t nc, uint8_t *hashes,
                   uint8_t *bal) {
    uint8_t *pre = (uint8_t *)malloc(32 * (nc + 2));
    for (size_t p = 0; p < m; p++) {
        for (uint32_t c = 0; c < nc; c++) {
            fe s;
            fe_add(&FR, &s, (const fe *)(child_bal + 32 * ((2 * p) * nc + c)), (const fe *)(child_bal + 32 * ((2 * p + 1) * nc + c)));
            memcpy(bal + 32 * (p * nc + c), &s, 32);
            memcpy(pre + 32 * c, &s, 32);
        }
        memcpy(pre + 32 * nc, child_hash + 32 * (2 * p), 32);
        memcpy(pre + 32 * (nc + 1), child_hash + 32 * (2 * p + 1), 32);
        orc_poseidon_hash(pre, nc + 2, hashes + 32 * p);
    }
    free(pre);
}

int orc_g1_is_on_curve(const uint8_t p[64]) {
    const g1a *a = (const g1a *)p;
    if (g1a_is_id(a)) return 1;
    fe y2, x3, three, t;
    fe c3 = {{3, 0, 0, 0}};
    fe_to_mont(&FQ, &three, &c3);
    fe_sqr(&FQ, &y2, &a->y);
    fe_sqr(&FQ, &t, &a->x);
    fe_mul(&FQ, &x3, &t, &a->x);
    fe_add(&FQ, &x3, &x3, &three);
    return fe_eq(&y2, &x3);
}
/* out = scalar * p (scalar Fr Montgomery), double-and-add */
static void g1_mul_canon(g1j *o, const g1a *p, const fe *canon) {
    g1j acc;
    g1j_set_id(&acc);
    for (int i = 255; i >= 0; i--) {
        g1j_double(&acc, &acc);
        if ((canon->l[i >> 6] >> (i & 63)) & 1) g1j_add_affine(&acc, &acc, p);
    }
    *o = acc;
}
void orc_g1_mul(const uint8_t p[64], const uint8_t scalar[32], uint8_t out[64]) {
    fe c;
    fe_from_mont(&FR, &c, (const fe *)scalar);
    g1j r;
    g1_mul_canon(&r, (const g1a *)p, &c);
    g1j_to_affine((g1a *)out, &r);
}
void orc_g1_add(const uint8_t p[64], const uint8_t q[64], uint8_t out[64]) {
    g1j r;
    const g1a *pa = (const g1a *)p;
    g1j_set_id(&r);
    g1j_add_affine(&r, &r, pa);
    g1j_add_affine(&r, &r, (const g1a *)q);
    g1j_to_affine((g1a *)out, &r);
}
void orc_g1_generator(uint8_t out[64]) {
    fe one = {{1, 0, 0, 0}}, two = {{2, 0, 0, 0}};
    g1a g;
    fe_to_mont(&FQ, &g.x, &one);
    fe_to_mont(&FQ, &g.y, &two);
    memcpy(out, &g, 64);
}
/* out[i] = scalars[i] * G (affine), 8-bit fixed-base windows, threaded; what
 * ParamsKZG::setup does for g[] */
typedef struct { const uint8_t *sc; uint8_t *out; size_t s, e; const g1a *table; } fb_job;
static void *fb_worker(void *arg) {
    fb_job *j = (fb_job *)arg;
    for (size_t i = j->s; i < j->e; i++) {
        fe c;
        fe_from_mont(&FR, &c, (const fe *)(j->sc + 32 * i));
        g1j acc;
        g1j_set_id(&acc);
        for (int w = 0; w < 32; w++) {
            unsigned d = (unsigned)((c.l[w >> 3] >> ((w & 7) * 8)) & 0xff);
            if (d) g1j_add_affine(&acc, &acc, &j->table[w * 256 + d]);
        }
        g1j_to_affine((g1a *)(j->out + 64 * i), &acc);
    }
    return NULL;
}
void orc_fixed_base_mul(const uint8_t *scalars, size_t n, int threads, uint8_t *out) {
    g1a *table = (g1a *)malloc(32 * 256 * sizeof(g1a));
    g1a gen;
    orc_g1_generator((uint8_t *)&gen);
    g1j base;
    base.x = gen.x; base.y = gen.y; base.z = FQ.r1;
    for (int w = 0; w < 32; w++) {
        g1a ba;
        g1j_to_affine(&ba, &base);
        g1j acc;
        g1j_set_id(&acc);
        memset(&table[w * 256], 0, sizeof(g1a));
        for (int d = 1; d < 256; d++) {
            g1j_add_affine(&acc, &acc, &ba);
            g1j_to_affine(&table[w * 256 + d], &acc);
        }
        for (int b = 0; b < 8; b++) g1j_double(&base, &base);
    }
    if (threads < 1) threads = 1;
    fb_job *jobs = (fb_job *)calloc((size_t)threads, sizeof(fb_job));
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; t++) {
        size_t s = (size_t)t * per, e = s + per > n ? n : s + per;
        if (s > n) s = n;
        jobs[t] = (fb_job){scalars, out, s, e, table};
        pthread_create(&tid[t], NULL, fb_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    free(jobs); free(tid); free(table);
}

/* ---------------------------------------------------------------- seeded inputs */
static inline u64 splitmix_next(u64 *s) {
    u64 z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
/* same rule as oracle/pyref.py::random_fr; output canonical -> Montgomery */
void orc_random_fr(uint64_t seed, size_t n, uint8_t *out) {
    u64 s = seed;
    for (size_t i = 0; i < n; i++) {
        fe c;
        for (int j = 0; j < 4; j++) c.l[j] = splitmix_next(&s);
        c.l[3] &= (1ULL << 62) - 1;
        u64 jj = 0;
        while (geq_p(c.l, FR.p)) {
            u64 s2 = (seed ^ (u64)(i + 1)) + (jj << 32);
            for (int j = 0; j < 4; j++) c.l[j] = splitmix_next(&s2);
            c.l[3] &= (1ULL << 62) - 1;
            jj++;
        }
        fe m;
        fe_to_mont(&FR, &m, &c);
        memcpy(out + 32 * i, &m, 32);
    }
}
