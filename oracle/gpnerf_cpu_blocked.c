/*
 * gpnerf_cpu_blocked.c -- THROUGHPUT twin of gpnerf_oracle.c: the same per-ray render path (libs/renders/BaseRender.py:110-184,
 * libs/nerfheads/trainhead.py:43-59,118-163), written the way a CPU wants it, for bench.py's `cpu_baseline` (kind
 * "port-blocked").
 *
 * TEST / BENCH INFRASTRUCTURE ONLY, like everything under oracle/: the product never links, imports or calls it.  It is NOT the
 * checker -- gpnerf_oracle.c (scalar, op-for-op, built -O2 without fast-math, pinned to the reference's golden vectors) stays
 * that; this file is checked AGAINST the oracle (tests/test_cpu_blocked.py, <= 1e-5) and exists so that the CPU number beside the
 * GPU one is a fair one: the scalar oracle does 57 rays/s per thread, the reference's own chunked torch path 282.
 *
 * How it differs from the oracle (none of it changes the algorithm):
 *   - a block of B = G rays x S samples (64..128 points) goes through every layer together, activations feature-major
 *     [feature][point], so a layer is a [out x in] x [in x B] product whose inner loop runs over 16-point vectors with the weight
 *     broadcast: 4 output rows x 4 point vectors = 16 accumulators per micro-kernel (AVX-512: all in zmm registers; on AVX2
 *     hosts the compiler splits each vector in two);
 *   - per-frame tensors are channels-last (volumes [D][H][W][32], feature maps [V][h][w][32], images [V][H][W][4]) so a
 *     trilinear / bilinear tap is one or two vector loads; the re-layout is per-frame preparation (oracle/blocked.py `Frame`),
 *     outside the timed call exactly as the GPU path's `Frame` construction is outside its timed step;
 *   - exp() in ELU / sigmoid / alpha is a vectorised Cephes-style polynomial (~2 ulp);
 *   - OpenMP over blocks of rays, dynamic schedule.
 * Built by oracle/blocked.py on the host it runs on: gcc -O3 -march=native -fopenmp -ffp-contract=off, no -ffast-math.  The dense
 * layers (`layer2`) are compiled with FMA contraction; the geometry is not: sample positions, projections and the in-bounds tests
 * are the reference's unfused fp32 operations (a fused projection flips the view mask of samples that land on an image edge).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NV 3
#define NC 32
#define NL 4
#define XF (NC + 3)
#define VL 16                      /* floats per vector */
#define MAXP 512                   /* points per block, upper bound (S <= 512) */

typedef float vf __attribute__((vector_size(4 * VL), aligned(4), may_alias));
typedef int32_t vi __attribute__((vector_size(4 * VL), aligned(4), may_alias));

typedef struct {
    /* channels-last per-frame tensors (blocked_prepare) */
    const float *imgs;       /* [V][H][W][4]: r g b 0, de-normalised to [0,1] (BaseRender.py:231) */
    const float *featmaps;   /* [V][fh][fw][32] */
    const float *vol[NL];    /* [Dk][Hk][Wk][32] */
    int32_t vol_dhw[NL][3];
    int32_t img_h, img_w, feat_h, feat_w;
    float K4P4[NV][16];
    float Rh[9], Th[3], bounds_min[3], voxel[3];
    int32_t out_sh[3];
    const float *geo_w, *geo_b, *b1_w, *b1_b, *b2_w, *b2_b, *v1_w, *v1_b, *v2_w, *v2_b, *r1_w, *r1_b, *r2_w, *r2_b, *r3_w, *r3_b;
    const float *d1_w, *d1_b, *d2_w, *d2_b, *d3_w, *d3_b, *d4_w, *d4_b;
} BlockedFrame;

typedef struct {
    float *rgb, *depth, *acc, *disp;   /* [N,3], [N], [N], [N] */
    float *weights;                    /* [N,S] or NULL */
    float *rgb_in;                     /* [N,9] or NULL */
    uint8_t *ray_mask;                 /* [N] or NULL */
} BlockedOut;

static inline vf splat(float x) { return (vf){x, x, x, x, x, x, x, x, x, x, x, x, x, x, x, x}; }
static inline vf vsel(vi m, vf a, vf b) { return (vf)(((vi)a & m) | ((vi)b & ~m)); }
static inline vf vmaxf(vf a, vf b) { return vsel(a > b, a, b); }
static inline vf vminf(vf a, vf b) { return vsel(a < b, a, b); }

/* exp(x), |rel err| ~ 2e-7: n = round(x log2 e), r = x - n ln 2 (two-piece), degree-6 polynomial, 2^n through the exponent */
static inline vf vexp(vf x) {
    x = vminf(vmaxf(x, splat(-87.f)), splat(88.f));
    const vf magic = splat(12582912.f);                       /* 1.5 * 2^23: adding it rounds to nearest integer */
    vf n = (x * splat(1.44269504088896341f) + magic) - magic;
    vf r = (x - n * splat(0.693359375f)) - n * splat(-2.12194440e-4f);
    vf p = splat(1.9875691500e-4f);
    p = p * r + splat(1.3981999507e-3f);
    p = p * r + splat(8.3334519073e-3f);
    p = p * r + splat(4.1665795894e-2f);
    p = p * r + splat(1.6666665459e-1f);
    p = p * r + splat(5.0000001201e-1f);
    p = p * (r * r) + r + splat(1.f);
    vi e = (__builtin_convertvector(n, vi) + 127) << 23;
    return p * (vf)e;
}
static inline vf velu(vf x) { return vsel(x > splat(0.f), x, vexp(x) - splat(1.f)); }

enum { ACT_NONE = 0, ACT_ELU = 1 };

/* Y[o][p] = act(b[o] + sum_i W[o][i0 + i] X1[i][p] + sum_i W[o][i0 + n1 + i] X2[i][p]);  X, Y feature-major with row stride P
 * (a multiple of VL).  W row-major [n_out][ldw] (nn.Linear).  Micro-kernel: 4 rows x 4 vectors. */
__attribute__((noinline, optimize("fp-contract=fast")))
static void layer2(const float *restrict W, const float *restrict b, int ldw, const float *restrict X1, int n1,
                   const float *restrict X2, int n2, float *restrict Y, int n_out, int P, int act) {
    for (int o = 0; o < n_out; o += 4) {
        const int no = n_out - o < 4 ? n_out - o : 4;
        const float *w0 = W + (size_t)o * ldw, *w1 = W + (size_t)(o + (no > 1 ? 1 : 0)) * ldw,
                    *w2 = W + (size_t)(o + (no > 2 ? 2 : 0)) * ldw, *w3 = W + (size_t)(o + (no > 3 ? 3 : 0)) * ldw;
        for (int p = 0; p < P; p += 4 * VL) {
            const int nvec = (P - p) / VL < 4 ? (P - p) / VL : 4;
            vf a[4][4];
            for (int r = 0; r < 4; ++r)
                for (int v = 0; v < 4; ++v) a[r][v] = splat(b[o + (r < no ? r : 0)]);
            if (nvec == 4) {
                for (int seg = 0; seg < 2; ++seg) {
                    const float *X = seg ? X2 : X1;
                    const int n = seg ? n2 : n1, off = seg ? n1 : 0;
                    for (int i = 0; i < n; ++i) {
                        const vf *x = (const vf *)(X + (size_t)i * P + p);
                        const vf x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
                        const vf c0 = splat(w0[off + i]), c1 = splat(w1[off + i]), c2 = splat(w2[off + i]), c3 = splat(w3[off + i]);
                        a[0][0] += c0 * x0; a[0][1] += c0 * x1; a[0][2] += c0 * x2; a[0][3] += c0 * x3;
                        a[1][0] += c1 * x0; a[1][1] += c1 * x1; a[1][2] += c1 * x2; a[1][3] += c1 * x3;
                        a[2][0] += c2 * x0; a[2][1] += c2 * x1; a[2][2] += c2 * x2; a[2][3] += c2 * x3;
                        a[3][0] += c3 * x0; a[3][1] += c3 * x1; a[3][2] += c3 * x2; a[3][3] += c3 * x3;
                    }
                }
            } else {
                for (int seg = 0; seg < 2; ++seg) {
                    const float *X = seg ? X2 : X1;
                    const int n = seg ? n2 : n1, off = seg ? n1 : 0;
                    for (int i = 0; i < n; ++i) {
                        const vf c0 = splat(w0[off + i]), c1 = splat(w1[off + i]), c2 = splat(w2[off + i]), c3 = splat(w3[off + i]);
                        for (int v = 0; v < nvec; ++v) {
                            const vf xv = *(const vf *)(X + (size_t)i * P + p + v * VL);
                            a[0][v] += c0 * xv; a[1][v] += c1 * xv; a[2][v] += c2 * xv; a[3][v] += c3 * xv;
                        }
                    }
                }
            }
            for (int r = 0; r < no; ++r)
                for (int v = 0; v < nvec; ++v)
                    *(vf *)(Y + (size_t)(o + r) * P + p + v * VL) = act == ACT_ELU ? velu(a[r][v]) : a[r][v];
        }
    }
}

/* bilinear tap set of F.grid_sample (zeros padding, align_corners=True): indices + weights (weight 0 for taps outside) */
static inline void taps1d(float g, int n, int *i0, int *i1, float *w0, float *w1) {
    const float x = ((g + 1.f) / 2.f) * (float)(n - 1), f = floorf(x), t = x - f;
    const int v0 = (f >= 0.f && f <= (float)(n - 1)), v1 = (f + 1.f >= 0.f && f + 1.f <= (float)(n - 1));
    *i0 = v0 ? (int)f : 0; *i1 = v1 ? (int)(f + 1.f) : 0;
    *w0 = v0 ? 1.f - t : 0.f; *w1 = v1 ? t : 0.f;
}

typedef struct {
    float *vol;    /* [128][P] */
    float *xv;     /* [V][35][P] */
    float *mv;     /* [70][P]: mean, var */
    float *s;      /* [64][P] sigma features (followed by mv in memory: the density net's input is [s | mean | var]) */
    float *h1, *h2, *h3, *y, *t1, *t2, *xs;
    float *raw;    /* [4][P] */
    float *mask;   /* [V][P] */
    float *z;      /* [P] */
    float *base;
} Scratch;

static Scratch scratch_new(void) {
    Scratch s;
    const size_t P = MAXP;
    /* s and mv adjacent: rows 0..63 = s, 64..133 = mean | var */
    const size_t n = (128 + NV * XF + 134 + 64 + 32 + 16 + NV * 32 + 32 + 32 + 32 + 4 + NV + 1) * P;
    s.base = (float *)aligned_alloc(64, n * sizeof(float));
    float *q = s.base;
    s.vol = q; q += 128 * P;
    s.xv = q; q += NV * XF * P;
    s.s = q; q += 64 * P;
    s.mv = q; q += 70 * P;
    s.h1 = q; q += 64 * P;
    s.h2 = q; q += 32 * P;
    s.h3 = q; q += 16 * P;
    s.y = q; q += NV * 32 * P;
    s.t1 = q; q += 32 * P;
    s.t2 = q; q += 32 * P;
    s.xs = q; q += 32 * P;
    s.raw = q; q += 4 * P;
    s.mask = q; q += NV * P;
    s.z = q;
    return s;
}

/* G rays x S samples = np points (P = np rounded up to VL; the padding lanes compute on zeros and are never read) */
static void render_block(const BlockedFrame *f, const float *rays, int64_t r0, int G, int S, int flags, const BlockedOut *o, Scratch *sc) {
    const int neg_ray = flags & 1, flip = (flags & 2) != 0;
    const int np = G * S, P = (np + VL - 1) / VL * VL;
    float *vol = sc->vol, *xv = sc->xv;
    /* ---- geometry + gathers, point by point (vector loads over the 32 channels) ---- */
    for (int pt = 0; pt < P; ++pt) {
        if (pt >= np) {
            for (int c = 0; c < 128; ++c) vol[(size_t)c * P + pt] = 0.f;
            for (int c = 0; c < NV * XF; ++c) xv[(size_t)c * P + pt] = 0.f;
            for (int v = 0; v < NV; ++v) sc->mask[(size_t)v * P + pt] = 0.f;
            sc->z[pt] = 0.f;
            continue;
        }
        const int g = pt / S, k = pt - g * S;
        const float *ray = rays + 8 * (r0 + g);
        const float near = ray[6], far = ray[7];
        const float step = (S > 1) ? 1.f / (float)(S - 1) : 0.f;
        float t = (k < S / 2) ? fmaf(step, (float)k, 0.f) : fmaf(-step, (float)(S - 1 - k), 1.f);      /* torch.linspace: ONE rounding per element, see gpnerf_oracle.c */
        if (S == 1) t = 0.f;
        const float z = near * (1.f - t) + far * t;
        sc->z[pt] = z;
        const float p[3] = {ray[0] + ray[3] * z, ray[1] + ray[4] * z, ray[2] + ray[5] * z};
        const float q0[3] = {p[0] - f->Th[0], p[1] - f->Th[1], p[2] - f->Th[2]};
        float q[3], gc[3];
        for (int j = 0; j < 3; ++j) q[j] = fmaf(q0[2], f->Rh[6 + j], fmaf(q0[1], f->Rh[3 + j], q0[0] * f->Rh[j]));
        for (int a = 0; a < 3; ++a) gc[2 - a] = (q[2 - a] - f->bounds_min[2 - a]) / f->voxel[a] / (float)f->out_sh[a] * 2.f - 1.f;
        for (int l = 0; l < NL; ++l) {
            const int D = f->vol_dhw[l][0], H = f->vol_dhw[l][1], W = f->vol_dhw[l][2];
            int xi[2], yi[2], zi[2];
            float wx[2], wy[2], wz[2];
            taps1d(gc[0], W, &xi[0], &xi[1], &wx[0], &wx[1]);
            taps1d(gc[1], H, &yi[0], &yi[1], &wy[0], &wy[1]);
            taps1d(gc[2], D, &zi[0], &zi[1], &wz[0], &wz[1]);
            vf a0 = splat(0.f), a1 = splat(0.f);
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    for (int e = 0; e < 2; ++e) {
                        const float w = wx[e] * wy[b] * wz[a];
                        if (w == 0.f) continue;
                        const float *src = f->vol[l] + (((size_t)zi[a] * H + yi[b]) * W + xi[e]) * NC;
                        a0 += splat(w) * *(const vf *)src;
                        a1 += splat(w) * *(const vf *)(src + VL);
                    }
            float tmp[NC];
            *(vf *)tmp = a0; *(vf *)(tmp + VL) = a1;
            for (int c = 0; c < NC; ++c) vol[(size_t)(l * NC + c) * P + pt] = tmp[c];
        }
        for (int v = 0; v < NV; ++v) {
            const float *M = f->K4P4[v];
            float h[3];
            for (int a = 0; a < 3; ++a) h[a] = fmaf(M[a * 4 + 2], p[2], fmaf(M[a * 4 + 1], p[1], M[a * 4 + 0] * p[0])) + M[a * 4 + 3];   /* the reference's sgemm order (gpnerf_oracle.c) */
            float u = h[0] / h[2], w = h[1] / h[2];
            u = fminf(fmaxf(u, -1e6f), 1e6f);
            w = fminf(fmaxf(w, -1e6f), 1e6f);
            const int front = neg_ray ? (h[2] < 0.f) : (h[2] > 0.f);
            const int inb = (u <= (float)f->img_w - 1.f) && (u >= 0.f) && (w <= (float)f->img_h - 1.f) && (w >= 0.f);
            sc->mask[(size_t)v * P + pt] = (front && inb) ? 1.f : 0.f;
            const float nx = 2.f * u / ((float)f->img_w - 1.f) - 1.f, ny = 2.f * w / ((float)f->img_h - 1.f) - 1.f;
            int xi[2], yi[2];
            float wx[2], wy[2];
            float rgb[4] = {0, 0, 0, 0};
            taps1d(nx, f->img_w, &xi[0], &xi[1], &wx[0], &wx[1]);
            taps1d(ny, f->img_h, &yi[0], &yi[1], &wy[0], &wy[1]);
            for (int b = 0; b < 2; ++b)
                for (int e = 0; e < 2; ++e) {
                    const float ww = wx[e] * wy[b];
                    const float *src = f->imgs + (((size_t)v * f->img_h + yi[b]) * f->img_w + xi[e]) * 4;
                    rgb[0] += ww * src[0]; rgb[1] += ww * src[1]; rgb[2] += ww * src[2];
                }
            taps1d(nx, f->feat_w, &xi[0], &xi[1], &wx[0], &wx[1]);
            taps1d(ny, f->feat_h, &yi[0], &yi[1], &wy[0], &wy[1]);
            vf a0 = splat(0.f), a1 = splat(0.f);
            for (int b = 0; b < 2; ++b)
                for (int e = 0; e < 2; ++e) {
                    const float ww = wx[e] * wy[b];
                    const float *src = f->featmaps + (((size_t)v * f->feat_h + yi[b]) * f->feat_w + xi[e]) * NC;
                    a0 += splat(ww) * *(const vf *)src;
                    a1 += splat(ww) * *(const vf *)(src + VL);
                }
            float tmp[NC];
            *(vf *)tmp = a0; *(vf *)(tmp + VL) = a1;
            float *dst = xv + (size_t)v * XF * P + pt;
            dst[0] = rgb[0]; dst[(size_t)P] = rgb[1]; dst[(size_t)2 * P] = rgb[2];
            for (int c = 0; c < NC; ++c) dst[(size_t)(3 + c) * P] = tmp[c];
        }
    }
    /* ---- mean / variance over the views (trainhead.py:20-24), vectors over points ---- */
    for (int c = 0; c < XF; ++c)
        for (int p = 0; p < P; p += VL) {
            const vf a = *(vf *)(xv + (size_t)c * P + p), b = *(vf *)(xv + (size_t)(XF + c) * P + p), d = *(vf *)(xv + (size_t)(2 * XF + c) * P + p);
            const vf m = (a + b + d) / splat((float)NV);
            const vf da = a - m, db = b - m, dd = d - m;
            *(vf *)(sc->mv + (size_t)c * P + p) = m;
            *(vf *)(sc->mv + (size_t)(XF + c) * P + p) = (da * da + db * db + dd * dd) / splat((float)NV);
        }
    /* ---- the dense layers ---- */
    layer2(f->geo_w, f->geo_b, 128, vol, 128, NULL, 0, sc->s, 64, P, ACT_ELU);                       /* trainhead.py:39-40,58 */
    /* density branch :127-137; its input [s | mean | var] is contiguous in the scratch (s then mv), re-strided below */
    layer2(f->d1_w, f->d1_b, 134, sc->s, 64, sc->mv, 70, sc->h1, 64, P, ACT_ELU);
    layer2(f->d2_w, f->d2_b, 64, sc->h1, 64, NULL, 0, sc->h2, 32, P, ACT_ELU);
    layer2(f->d3_w, f->d3_b, 32, sc->h2, 32, NULL, 0, sc->h3, 16, P, ACT_ELU);
    layer2(f->d4_w, f->d4_b, 16, sc->h3, 16, NULL, 0, sc->raw + (size_t)3 * P, 1, P, ACT_NONE);
    for (int p = 0; p < P; p += VL) {
        vf sg = *(vf *)(sc->raw + (size_t)3 * P + p);
        sg = vmaxf(sg, splat(0.f));                                                                  /* nn.ReLU :110 */
        const vf nvalid = *(vf *)(sc->mask + p) + *(vf *)(sc->mask + (size_t)P + p) + *(vf *)(sc->mask + (size_t)2 * P + p);
        *(vf *)(sc->raw + (size_t)3 * P + p) = vsel(nvalid < splat(1.f), splat(0.f), sg);            /* masked_fill :136-137 */
    }
    /* colour branch :131,139-143 */
    for (int v = 0; v < NV; ++v) {
        float *yv = sc->y + (size_t)v * 32 * P;
        layer2(f->b1_w, f->b1_b, 105, sc->mv, 70, xv + (size_t)v * XF * P, XF, sc->h1, 64, P, ACT_ELU);
        layer2(f->b2_w, f->b2_b, 64, sc->h1, 64, NULL, 0, yv, 32, P, ACT_ELU);
        for (size_t i = 0; i < (size_t)32 * P; i += VL) *(vf *)(sc->xs + i) = *(vf *)(yv + i) * splat(1.0f) / splat((float)NV);
        layer2(f->v1_w, f->v1_b, 32, sc->xs, 32, NULL, 0, sc->t1, 32, P, ACT_ELU);
        layer2(f->v2_w, f->v2_b, 32, sc->t1, 32, NULL, 0, sc->t2, 32, P, ACT_ELU);
        for (size_t i = 0; i < (size_t)32 * P; i += VL) *(vf *)(yv + i) += *(vf *)(sc->t2 + i);
    }
    layer2(f->r1_w, f->r1_b, 96, sc->y, 96, NULL, 0, sc->h2, 32, P, ACT_ELU);
    layer2(f->r2_w, f->r2_b, 32, sc->h2, 32, NULL, 0, sc->h3, 16, P, ACT_ELU);
    layer2(f->r3_w, f->r3_b, 16, sc->h3, 16, NULL, 0, sc->raw, 3, P, ACT_NONE);
    for (size_t i = 0; i < (size_t)3 * P; i += VL) *(vf *)(sc->raw + i) = splat(1.f) / (splat(1.f) + vexp(-*(vf *)(sc->raw + i)));
    /* alpha = 1 - exp(-sigma), vectorised; kept in the sigma row */
    for (int p = 0; p < P; p += VL) *(vf *)(sc->raw + (size_t)3 * P + p) = splat(1.f) - vexp(-*(vf *)(sc->raw + (size_t)3 * P + p));
    /* ---- raw2outputs (BaseRender.py:75-107), per ray ---- */
    for (int g = 0; g < G; ++g) {
        const int64_t r = r0 + g;
        const int b0 = g * S;
        float T = 1.f, rgb[3] = {0, 0, 0}, depth = 0.f, acc = 0.f, rin[9] = {0};
        int n_two = 0;
        for (int k = 0; k < S; ++k) {
            const int src = b0 + (flip ? S - 1 - k : k), here = b0 + k;
            const float alpha = sc->raw[(size_t)3 * P + src];
            const float w = alpha * T;
            T = T * (1.f - alpha + 1e-10f);
            n_two += (sc->mask[src] + sc->mask[(size_t)P + src] + sc->mask[(size_t)2 * P + src]) > 1.f;
            for (int c = 0; c < 3; ++c) rgb[c] += w * sc->raw[(size_t)c * P + src];
            depth += w * sc->z[here];
            acc += w;
            if (o->rgb_in)
                for (int v = 0; v < NV; ++v)
                    for (int c = 0; c < 3; ++c) rin[v * 3 + c] += w * xv[((size_t)v * XF + c) * P + here];
            if (o->weights) o->weights[(size_t)r * S + k] = w;
        }
        memcpy(o->rgb + 3 * r, rgb, sizeof(rgb));
        o->depth[r] = depth;
        o->acc[r] = acc;
        const float qd = depth / acc;
        o->disp[r] = 1.f / ((qd != qd) ? qd : fmaxf(1e-10f, qd));
        if (o->rgb_in) memcpy(o->rgb_in + 9 * r, rin, sizeof(rin));
        if (o->ray_mask) o->ray_mask[r] = (uint8_t)(n_two > 8);
    }
}

int blocked_render(const BlockedFrame *f, const float *rays, int64_t N, int S, int flags, const BlockedOut *out, int n_threads) {
    if (S < 1 || S > MAXP) return -1;
    int G = 64 / S;
    if (G < 1) G = 1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    const int64_t nblk = (N + G - 1) / G;
#pragma omp parallel
    {
        Scratch sc = scratch_new();
#pragma omp for schedule(dynamic, 8)
        for (int64_t b = 0; b < nblk; ++b) {
            const int64_t r0 = b * G;
            const int g = (int)(N - r0 < G ? N - r0 : G);
            render_block(f, rays, r0, g, S, flags, out, &sc);
        }
        free(sc.base);
    }
    return 0;
}

int blocked_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
