/*
 * C restatement of the mask-based multi-style injection path (CPU, float32, OpenMP).
 *
 * TEST INFRASTRUCTURE ONLY: the checker of tests/ and the timed `cpu_baseline` leg of
 * bench.py.  Nothing in the product path may call it.  Pinned, like oracle/sid_oracle.py,
 * by tests/golden/ (outputs of the reference's own modules).
 *
 * Row-wise formulation: every query row is independent, so a caller can compute (and time) any
 * sample of rows with all keys.  Citations: paths relative to the reference repository root.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static float dotf(const float* a, const float* b, int d) {
    float s = 0.f;
    for (int i = 0; i < d; ++i) s += a[i] * b[i];
    return s;
}

int sid_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/*
 * Injected three-block attention for query rows [row0, row0+nrows), all heads.
 *   xq (N, H*d); q_cnt,k1,v1,k2,v2,k_cnt,v_cnt (H, N, d); mask (N) signed (+1 char, -1 back)
 *   out (nrows, H*d)
 * ldm/modules/attention.py:381 (blend), :206 (cc_sim), :218-221 (style logits), :236-241 (raw
 * max, min cc), :273,:285-288 (bias), :538-546 (concat softmax, PV, head merge).
 */
void sid_oracle_injected_attn_rows(const float* xq, const float* q_cnt, const float* k1, const float* v1,
                                   const float* k2, const float* v2, const float* k_cnt, const float* v_cnt,
                                   const float* mask, float* out, int H, int N, int d, float gamma,
                                   float q_self_coef, float T, float scale, int row0, int nrows) {
#pragma omp parallel
    {
        float* logit = (float*)malloc(sizeof(float) * 3 * (size_t)N);
        float* q = (float*)malloc(sizeof(float) * d);
        float* acc = (float*)malloc(sizeof(float) * d);
#pragma omp for collapse(2) schedule(dynamic, 4)
        for (int h = 0; h < H; ++h)
            for (int r = 0; r < nrows; ++r) {
                const int i = row0 + r;
                const float* qc = q_cnt + ((size_t)h * N + i) * d;
                const float* xr = xq + (size_t)i * H * d + (size_t)h * d;
                for (int t = 0; t < d; ++t) q[t] = qc[t] * gamma + xr[t] * q_self_coef;
                float min_cc = INFINITY, max1 = -INFINITY, max2 = -INFINITY;
                float* l1 = logit; float* l2 = logit + N; float* lc = logit + 2 * (size_t)N;
                for (int j = 0; j < N; ++j) {
                    const float c = dotf(qc, k_cnt + ((size_t)h * N + j) * d, d) * scale;
                    const float a = dotf(q, k1 + ((size_t)h * N + j) * d, d);
                    const float b = dotf(q, k2 + ((size_t)h * N + j) * d, d);
                    lc[j] = c; l1[j] = a; l2[j] = b;
                    min_cc = c < min_cc ? c : min_cc;
                    max1 = a > max1 ? a : max1;
                    max2 = b > max2 ? b : max2;
                }
                const float b1 = (min_cc - max1) * (mask[i] * -1.0f);
                const float b2 = (min_cc - max2) * (mask[i] * 1.0f);
                float mx = -INFINITY;
                for (int j = 0; j < N; ++j) {
                    l1[j] = (l1[j] * T) * scale + b1;
                    l2[j] = (l2[j] * T) * scale + b2;
                    mx = fmaxf(mx, fmaxf(l1[j], fmaxf(l2[j], lc[j])));
                }
                float den = 0.f;
                for (int t = 0; t < d; ++t) acc[t] = 0.f;
                const float* vs[3] = {v1 + (size_t)h * N * d, v2 + (size_t)h * N * d, v_cnt + (size_t)h * N * d};
                for (int blk = 0; blk < 3; ++blk) {
                    const float* l = logit + (size_t)blk * N;
                    for (int j = 0; j < N; ++j) {
                        const float p = expf(l[j] - mx);
                        den += p;
                        const float* vr = vs[blk] + (size_t)j * d;
                        for (int t = 0; t < d; ++t) acc[t] += p * vr[t];
                    }
                }
                float* o = out + (size_t)r * H * d + (size_t)h * d;
                for (int t = 0; t < d; ++t) o[t] = acc[t] / den;
            }
        free(logit); free(q); free(acc);
    }
}

/*
 * Single-block attention for query rows [row0, row0+nrows): softmax(((q k^T)*T)*scale) v.
 *   q (H, Nq, d), k,v (H, Nk, d), out (nrows, H*d).  ldm/modules/attention.py:557-566.
 */
void sid_oracle_plain_attn_rows(const float* q, const float* k, const float* v, float* out, int H, int Nq, int Nk,
                                int d, float T, float scale, int row0, int nrows) {
#pragma omp parallel
    {
        float* logit = (float*)malloc(sizeof(float) * (size_t)Nk);
        float* acc = (float*)malloc(sizeof(float) * d);
#pragma omp for collapse(2) schedule(dynamic, 4)
        for (int h = 0; h < H; ++h)
            for (int r = 0; r < nrows; ++r) {
                const int i = row0 + r;
                const float* qr = q + ((size_t)h * Nq + i) * d;
                float mx = -INFINITY;
                for (int j = 0; j < Nk; ++j) {
                    logit[j] = (dotf(qr, k + ((size_t)h * Nk + j) * d, d) * T) * scale;
                    mx = fmaxf(mx, logit[j]);
                }
                float den = 0.f;
                for (int t = 0; t < d; ++t) acc[t] = 0.f;
                for (int j = 0; j < Nk; ++j) {
                    const float p = expf(logit[j] - mx);
                    den += p;
                    const float* vr = v + ((size_t)h * Nk + j) * d;
                    for (int t = 0; t < d; ++t) acc[t] += p * vr[t];
                }
                float* o = out + (size_t)r * H * d + (size_t)h * d;
                for (int t = 0; t < d; ++t) o[t] = acc[t] / den;
            }
        free(logit); free(acc);
    }
}

/* ---- high-frequency filter: high_frequency_final.py:4-36 -------------------------------- */
static void fft1d(float* re, float* im, int n, int stride, int inverse) {
    /* iterative radix-2 for powers of two, O(n^2) DFT otherwise; unnormalised */
    if ((n & (n - 1)) == 0) {
        for (int i = 1, j = 0; i < n; ++i) {
            int bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) {
                float t = re[i * stride]; re[i * stride] = re[j * stride]; re[j * stride] = t;
                t = im[i * stride]; im[i * stride] = im[j * stride]; im[j * stride] = t;
            }
        }
        for (int len = 2; len <= n; len <<= 1) {
            const double ang = (inverse ? 2.0 : -2.0) * M_PI / len;
            for (int i = 0; i < n; i += len)
                for (int k = 0; k < len / 2; ++k) {
                    const float wr = (float)cos(ang * k), wi = (float)sin(ang * k);
                    const int a = (i + k) * stride, b = (i + k + len / 2) * stride;
                    const float xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                    re[b] = re[a] - xr; im[b] = im[a] - xi;
                    re[a] += xr; im[a] += xi;
                }
        }
    } else {
        double* tr = (double*)malloc(sizeof(double) * 2 * n);
        for (int k = 0; k < n; ++k) {
            double sr = 0, si = 0;
            for (int t = 0; t < n; ++t) {
                const double ang = (inverse ? 2.0 : -2.0) * M_PI * (double)((long long)k * t % n) / n;
                sr += re[t * stride] * cos(ang) - im[t * stride] * sin(ang);
                si += re[t * stride] * sin(ang) + im[t * stride] * cos(ang);
            }
            tr[2 * k] = sr; tr[2 * k + 1] = si;
        }
        for (int k = 0; k < n; ++k) { re[k * stride] = (float)tr[2 * k]; im[k * stride] = (float)tr[2 * k + 1]; }
        free(tr);
    }
}

/* out = [skip + h +] HF(h_cnt); tensors (maps, H, W) with maps = B*C; skip/h may be NULL. */
void sid_oracle_hf_filter(const float* h_cnt, const float* skip, const float* h, float* out, int maps, int H, int W,
                          int radius) {
#pragma omp parallel
    {
        float* re = (float*)malloc(sizeof(float) * (size_t)H * W);
        float* im = (float*)malloc(sizeof(float) * (size_t)H * W);
#pragma omp for schedule(dynamic, 1)
        for (int m = 0; m < maps; ++m) {
            const float* x = h_cnt + (size_t)m * H * W;
            for (int i = 0; i < H * W; ++i) { re[i] = x[i]; im[i] = 0.f; }
            for (int y = 0; y < H; ++y) fft1d(re + y * W, im + y * W, W, 1, 0);
            for (int xx = 0; xx < W; ++xx) fft1d(re + xx, im + xx, H, W, 0);
            /* shifted bin (y,x) holds signed frequency (y-H/2, x-W/2): the disc test in unshifted order */
            for (int ky = 0; ky < H; ++ky) {
                const int fy = ky < H - H / 2 ? ky : ky - H;
                for (int kx = 0; kx < W; ++kx) {
                    const int fx = kx < W - W / 2 ? kx : kx - W;
                    if (fy * fy + fx * fx < radius * radius) { re[ky * W + kx] = 0.f; im[ky * W + kx] = 0.f; }
                }
            }
            for (int xx = 0; xx < W; ++xx) fft1d(re + xx, im + xx, H, W, 1);
            for (int y = 0; y < H; ++y) fft1d(re + y * W, im + y * W, W, 1, 1);
            const float norm = 1.0f / (float)(H * W);   /* two 'ortho' factors */
            float* o = out + (size_t)m * H * W;
            for (int i = 0; i < H * W; ++i) {
                const float hf = re[i] * norm;
                o[i] = skip ? (skip[(size_t)m * H * W + i] + h[(size_t)m * H * W + i]) + hf : hf;
            }
        }
        free(re); free(im);
    }
}
