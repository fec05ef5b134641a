/*
 * urnn_oracle.c -- CPU restatement of the U-RNN rollout hot path (TEST INFRASTRUCTURE).
 *
 * This file is the parity oracle for the HIP kernels in u-rnn_amd/csrc.  It is NOT a
 * product path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  It restates, from the equations, what the reference's PyTorch modules compute
 * for one timestep; every function cites the reference file:line it
 * follows (paths relative to /root/reference/code/).
 *
 * Parity pinning: the reference ships no tests or golden vectors, so this
 * oracle is pinned against outputs of the reference itself, generated in the build container
 * by tests/golden/make_golden.py and committed as tests/golden/*.npz; tests/test_oracle.py
 * checks every entry point below against them.
 *
 * Numerics: storage is float32 like the reference; dot products and normalisation
 * statistics accumulate in double and are rounded once, so the oracle sits within ~1e-6
 * relative of the reference's fp32 arithmetic whatever its summation order.
 *
 * Layout: every tensor is NCHW contiguous float32; P = H*W is the plane size.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_BLK 256

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * 1x1 convolution over up to three channel-concatenated inputs (torch.cat + nn.Conv2d k=1):
 *   out[b][n][p] = bias[n] + sum_s sum_k w[n][koff_s + k] * in_s[b][k][p]
 * A NULL segment pointer with C>0 stands for an all-zero input (decoder stage 3, x == 0:
 * ConvRNN.py:143-146).  w is (Cout, Ktot) row-major as in nn.Conv2d.weight[:, :, 0, 0]
 * (ConvRNN.py:94-104, utils.py:109-115).  bias may be NULL (head BaseConv, network_blocks.py:78).
 * ---------------------------------------------------------------------------------------- */
void orc_conv1x1_cat(const float *in0, int C0, const float *in1, int C1, const float *in2, int C2,
                     const float *w, const float *bias, float *out, int B, int Cout, long P)
{
    const int Ktot = C0 + C1 + C2;
    const float *seg[3] = {in0, in1, in2};
    const int segC[3] = {C0, C1, C2};
    const long nblk = (P + ORC_BLK - 1) / ORC_BLK;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (long blk = 0; blk < nblk; ++blk) {
            const long p0 = blk * ORC_BLK;
            const int np = (int)((P - p0) < ORC_BLK ? (P - p0) : ORC_BLK);
            double acc[ORC_BLK];
            for (int n = 0; n < Cout; ++n) {
                const double b0 = bias ? (double)bias[n] : 0.0;
                for (int i = 0; i < np; ++i) acc[i] = b0;
                int koff = 0;
                for (int s = 0; s < 3; ++s) {
                    if (segC[s] > 0 && seg[s]) {
                        const float *src = seg[s] + ((size_t)b * segC[s]) * P + p0;
                        for (int k = 0; k < segC[s]; ++k) {
                            const double a = (double)w[(size_t)n * Ktot + koff + k];
                            const float *row = src + (size_t)k * P;
                            for (int i = 0; i < np; ++i) acc[i] += a * (double)row[i];
                        }
                    }
                    koff += segC[s];
                }
                float *dst = out + ((size_t)b * Cout + n) * P + p0;
                for (int i = 0; i < np; ++i) dst[i] = (float)acc[i];
            }
        }
    }
}

/* LeakyReLU(slope) in place (utils.py:63, network_blocks.py:38). */
void orc_leaky_relu(float *x, long n, float slope)
{
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) x[i] = x[i] >= 0.f ? x[i] : x[i] * slope;
}

/* AvgPool2d(kernel=2, stride=2, padding=0), floor mode (utils.py:92-94, net_params.py:82-88). */
void orc_avgpool2(const float *in, float *out, int BC, int H, int W)
{
    const int H2 = H / 2, W2 = W / 2;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < BC; ++c) {
        const float *src = in + (size_t)c * H * W;
        float *dst = out + (size_t)c * H2 * W2;
        for (int y = 0; y < H2; ++y)
            for (int x = 0; x < W2; ++x) {
                const double s = (double)src[(size_t)(2 * y) * W + 2 * x] + (double)src[(size_t)(2 * y) * W + 2 * x + 1] +
                                 (double)src[(size_t)(2 * y + 1) * W + 2 * x] + (double)src[(size_t)(2 * y + 1) * W + 2 * x + 1];
                dst[(size_t)y * W2 + x] = (float)(0.25 * s);
            }
    }
}

/* Encoder / decoder stage conv: [AvgPool2](LeakyReLU(W.x + b)) (encoder.py:140-151, utils.py:109-121,
 * net_params.py:80-88).  out is (B, Cout, H/2, W/2) when pool != 0, else (B, Cout, H, W).  scratch must
 * hold B*Cout*H*W floats when pool != 0 (ignored otherwise). */
void orc_stage_conv(const float *in, const float *w, const float *bias, float *out, float *scratch, int B, int Cin,
                    int Cout, int H, int W, int pool, float slope)
{
    const long P = (long)H * W;
    float *full = pool ? scratch : out;
    orc_conv1x1_cat(in, Cin, NULL, 0, NULL, 0, w, bias, full, B, Cout, P);
    orc_leaky_relu(full, (long)B * Cout * P, slope);
    if (pool) orc_avgpool2(full, out, B * Cout, H, W);
}

/* ConvTranspose2d(k=2, s=2, p=0) + LeakyReLU (utils.py:95-107, net_params.py:106-120):
 *   y[b][co][2i+a][2j+c] = bias[co] + sum_ci x[b][ci][i][j] * w[ci][co][a][c]
 * weight layout (Cin, Cout, 2, 2) as nn.ConvTranspose2d.weight. */
void orc_deconv2x2(const float *in, const float *w, const float *bias, float *out, int B, int Cin, int Cout, int H, int W,
                   float slope)
{
    const int H2 = 2 * H, W2 = 2 * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int co = 0; co < Cout; ++co) {
            float *dst = out + ((size_t)b * Cout + co) * H2 * W2;
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < W; ++j) {
                    double acc[4];
                    for (int q = 0; q < 4; ++q) acc[q] = bias ? (double)bias[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci) {
                        const double xv = (double)in[(((size_t)b * Cin + ci) * H + i) * W + j];
                        const float *wq = w + ((size_t)ci * Cout + co) * 4;
                        acc[0] += xv * wq[0];
                        acc[1] += xv * wq[1];
                        acc[2] += xv * wq[2];
                        acc[3] += xv * wq[3];
                    }
                    for (int a = 0; a < 2; ++a)
                        for (int c = 0; c < 2; ++c) {
                            float v = (float)acc[a * 2 + c];
                            v = v >= 0.f ? v : v * slope;
                            dst[(size_t)(2 * i + a) * W2 + 2 * j + c] = v;
                        }
                }
        }
    }
}

/* nn.GroupNorm(groups, C), eps, biased variance over (C/groups channels x P) per sample, per-channel
 * affine (ConvRNN.py:97,103).  In place.  Two-pass (mean, then centred squares) in double; parallel over
 * channels with a fixed-order combine per group. */
void orc_group_norm(float *x, const float *gamma, const float *beta, int B, int C, long P, int groups, float eps)
{
    const int cg = C / groups;
    double *part = (double *)malloc(sizeof(double) * (size_t)B * C);
    double *mean = (double *)malloc(sizeof(double) * (size_t)B * groups);
    double *rstd = (double *)malloc(sizeof(double) * (size_t)B * groups);
#pragma omp parallel for schedule(static)
    for (int bc = 0; bc < B * C; ++bc) {
        const float *row = x + (size_t)bc * P;
        double s = 0.0;
        for (long i = 0; i < P; ++i) s += row[i];
        part[bc] = s;
    }
    for (int bg = 0; bg < B * groups; ++bg) {
        double s = 0.0;
        for (int c = 0; c < cg; ++c) s += part[(size_t)bg * cg + c];
        mean[bg] = s / ((double)cg * (double)P);
    }
#pragma omp parallel for schedule(static)
    for (int bc = 0; bc < B * C; ++bc) {
        const float *row = x + (size_t)bc * P;
        const double m = mean[bc / cg];
        double v = 0.0;
        for (long i = 0; i < P; ++i) {
            const double d = row[i] - m;
            v += d * d;
        }
        part[bc] = v;
    }
    for (int bg = 0; bg < B * groups; ++bg) {
        double v = 0.0;
        for (int c = 0; c < cg; ++c) v += part[(size_t)bg * cg + c];
        rstd[bg] = 1.0 / sqrt(v / ((double)cg * (double)P) + (double)eps);
    }
#pragma omp parallel for schedule(static)
    for (int bc = 0; bc < B * C; ++bc) {
        float *row = x + (size_t)bc * P;
        const int c = bc % C;
        const double m = mean[bc / cg], r = rstd[bc / cg], ga = gamma[c], be = beta[c];
        for (long i = 0; i < P; ++i) row[i] = (float)(((double)row[i] - m) * r * ga + be);
    }
    free(part);
    free(mean);
    free(rstd);
}

/* nn.LayerNorm([C, H, W]): statistics over all N = C*H*W elements of a sample, element-wise affine of
 * shape (C, H, W) (network_blocks.py:90-91).  In place.  Same two-pass scheme over 4096-element chunks. */
void orc_layer_norm(float *x, const float *gamma, const float *beta, int B, long N, float eps)
{
    const long CH = 4096;
    const long nch = (N + CH - 1) / CH;
    double *part = (double *)malloc(sizeof(double) * (size_t)nch);
    for (int b = 0; b < B; ++b) {
        float *base = x + (size_t)b * N;
#pragma omp parallel for schedule(static)
        for (long c = 0; c < nch; ++c) {
            const long lo = c * CH, hi = (lo + CH) < N ? (lo + CH) : N;
            double s = 0.0;
            for (long i = lo; i < hi; ++i) s += base[i];
            part[c] = s;
        }
        double s = 0.0;
        for (long c = 0; c < nch; ++c) s += part[c];
        const double mean = s / (double)N;
#pragma omp parallel for schedule(static)
        for (long c = 0; c < nch; ++c) {
            const long lo = c * CH, hi = (lo + CH) < N ? (lo + CH) : N;
            double v = 0.0;
            for (long i = lo; i < hi; ++i) {
                const double d = base[i] - mean;
                v += d * d;
            }
            part[c] = v;
        }
        double v = 0.0;
        for (long c = 0; c < nch; ++c) v += part[c];
        const double rstd = 1.0 / sqrt(v / (double)N + (double)eps);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < N; ++i) base[i] = (float)(((double)base[i] - mean) * rstd * gamma[i] + beta[i]);
    }
    free(part);
}

static inline float orc_sigmoidf(float v) { return (float)(1.0 / (1.0 + exp(-(double)v))); }

/* ConvGRU / Skip-ConvGRU cell, one timestep (CGRU_cell.forward, ConvRNN.py:111-194; seq_len == 1):
 *   hidden = h                      (encoder)   |  cat(e, d)            (decoder; decoder.py:130-135)
 *   g  = GN_{2F/32}(W1 . cat(x, hidden) + b1);  z = sigma(g[:F]);  r = sigma(g[F:])      (:153-163)
 *   n  = tanh(GN_{F/32}(W2 . cat(x, [e,] r*h) + b2))                                    (:166-180)
 *   h' = (1 - z) * h + z * n                                                            (:183-189)
 * x == NULL means x == 0 with I channels (decoder stage 3, :143-146).  e == NULL selects the encoder cell.
 * W1 is (2F, I[+F]+F), W2 is (F, I[+F]+F), row-major.  scratch: >= B*(3F)*P floats... see below.
 * scratch layout: g (B*2F*P) | c (B*F*P) | rh (B*F*P)  => B*4F*P floats. */
void orc_gru_cell(const float *x, const float *e, const float *h, const float *W1, const float *b1, const float *gn1_w,
                  const float *gn1_b, const float *W2, const float *b2, const float *gn2_w, const float *gn2_b, float *h_out,
                  float *scratch, int B, int I, int F, long P, float eps)
{
    const int Fe = e ? F : 0;
    float *g = scratch;
    float *c = g + (size_t)B * 2 * F * P;
    float *rh = c + (size_t)B * F * P;

    /* conv1 over cat(x, e, h) followed by GroupNorm(2F/32 groups) */
    if (e)
        orc_conv1x1_cat(x, I, e, F, h, F, W1, b1, g, B, 2 * F, P);
    else
        orc_conv1x1_cat(x, I, h, F, NULL, 0, W1, b1, g, B, 2 * F, P);
    orc_group_norm(g, gn1_w, gn1_b, B, 2 * F, P, (2 * F) / 32, eps);

    /* r * h with r = sigmoid(g[:, F:2F]) */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f) {
            const float *gr = g + ((size_t)b * 2 * F + F + f) * P;
            const float *hh = h + ((size_t)b * F + f) * P;
            float *o = rh + ((size_t)b * F + f) * P;
            for (long i = 0; i < P; ++i) o[i] = orc_sigmoidf(gr[i]) * hh[i];
        }

    /* conv2 over cat(x, e, r*h) followed by GroupNorm(F/32 groups) */
    if (e)
        orc_conv1x1_cat(x, I, e, F, rh, F, W2, b2, c, B, F, P);
    else
        orc_conv1x1_cat(x, I, rh, F, NULL, 0, W2, b2, c, B, F, P);
    (void)Fe;
    orc_group_norm(c, gn2_w, gn2_b, B, F, P, F / 32, eps);

    /* blend */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f) {
            const float *gz = g + ((size_t)b * 2 * F + f) * P;
            const float *cc = c + ((size_t)b * F + f) * P;
            const float *hh = h + ((size_t)b * F + f) * P;
            float *o = h_out + ((size_t)b * F + f) * P;
            for (long i = 0; i < P; ++i) {
                const float z = orc_sigmoidf(gz[i]);
                const float n = (float)tanh((double)cc[i]);
                o[i] = (1.f - z) * hh[i] + z * n;
            }
        }
}

static inline float orc_siluf(float v) { return (float)((double)v / (1.0 + exp(-(double)v))); }

/* BaseConv: SiLU(LayerNorm[16,H,W](W . x)), conv without bias (network_blocks.py:74-101). */
static void orc_base_conv(const float *in, const float *w, const float *ln_w, const float *ln_b, float *out, int B, int C,
                          long P, float eps)
{
    orc_conv1x1_cat(in, C, NULL, 0, NULL, 0, w, NULL, out, B, C, P);
    orc_layer_norm(out, ln_w, ln_b, B, (long)C * P, eps);
    const long n = (long)B * C * P;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) out[i] = orc_siluf(out[i]);
}

/* YOLOXHead.forward + correction_depth (flood_head.py:131-202):
 *   t = BaseConv_stem(f); c = BaseConv_c2(BaseConv_c1(t)); q = BaseConv_q2(BaseConv_q1(t))
 *   cls = sigmoid(w_c . c + b_c); reg = LeakyReLU_0.2(w_r . q + b_r)       (network_blocks.py:129-171)
 *   out0 = reg * [cls >= cls_thred]; out1 = cls
 * conv_w: 5 x (C x C) in the order stem, cls0, cls1, reg0, reg1; ln_w / ln_b: 5 x (C*P) same order.
 * Outputs (each B*P floats): out_masked, out_cls, out_reg_raw (pre-mask).  scratch: 3*B*C*P floats. */
void orc_head(const float *f, const float *conv_w, const float *ln_w, const float *ln_b, const float *cls_w,
              const float *cls_b, const float *reg_w, const float *reg_b, float *out_masked, float *out_cls,
              float *out_reg_raw, float *scratch, int B, int C, long P, float cls_thred, float eps, float slope)
{
    const size_t CP = (size_t)C * P, CC = (size_t)C * C;
    float *t = scratch, *u = t + (size_t)B * CP, *v = u + (size_t)B * CP;
    orc_base_conv(f, conv_w + 0 * CC, ln_w + 0 * CP, ln_b + 0 * CP, t, B, C, P, eps);
    /* cls branch */
    orc_base_conv(t, conv_w + 1 * CC, ln_w + 1 * CP, ln_b + 1 * CP, u, B, C, P, eps);
    orc_base_conv(u, conv_w + 2 * CC, ln_w + 2 * CP, ln_b + 2 * CP, v, B, C, P, eps);
    orc_conv1x1_cat(v, C, NULL, 0, NULL, 0, cls_w, cls_b, out_cls, B, 1, P);
    {
        const long n = (long)B * P;
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) out_cls[i] = orc_sigmoidf(out_cls[i]);
    }
    /* reg branch */
    orc_base_conv(t, conv_w + 3 * CC, ln_w + 3 * CP, ln_b + 3 * CP, u, B, C, P, eps);
    orc_base_conv(u, conv_w + 4 * CC, ln_w + 4 * CP, ln_b + 4 * CP, v, B, C, P, eps);
    orc_conv1x1_cat(v, C, NULL, 0, NULL, 0, reg_w, reg_b, out_reg_raw, B, 1, P);
    orc_leaky_relu(out_reg_raw, (long)B * P, slope);
    {
        const long n = (long)B * P;
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) out_masked[i] = out_reg_raw[i] * (out_cls[i] >= cls_thred ? 1.f : 0.f);
    }
}

/* preprocess_inputs (Dynamic2DFlood.py:265-320) + get_past_rainfall (:323-366) + MinMaxScaler (:369-376).
 * Output (B, C = 2*nums+3, H, W): [rain(t-n+1..t)/rain_max, cumsum(...)/cumsum_max, (DEM-min)/(max-min),
 * (imp-0.05)/0.9, manhole], history left-zero-padded for t < nums.
 * rain / cumsum: (B, T) when spatial == 0 (scalar rain broadcast over the grid), else (B, T, H, W).
 * dem_min / dem_max: the reference indexes inputs["max_DEM"][0] (:305-306), i.e. sample 0's range is
 * applied to every sample of the batch; pass one value each. */
void orc_preprocess(int t, const float *rain, const float *cumsum, int T, int spatial, const float *dem, const float *imperv,
                    const float *manhole, float dem_min, float dem_max, float *out, int B, int nums, int H, int W,
                    float rain_max, float cumsum_max)
{
    const long P = (long)H * W;
    const int C = 2 * nums + 3;
    const int start = (t - nums + 1) > 0 ? (t - nums + 1) : 0;
    const int end = (t + 1) < T ? (t + 1) : T;
    const int nsteps = end - start;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            float *dst = out + ((size_t)b * C + c) * P;
            if (c < 2 * nums) {
                const int which = c / nums;         /* 0 rain, 1 cumsum */
                const int slot = c % nums;          /* position in the history window */
                const int k = slot - (nums - nsteps); /* index into [start, end) */
                const float *src = which ? cumsum : rain;
                const float mx = which ? cumsum_max : rain_max;
                if (k < 0) {
                    /* zero-padded history: MinMaxScaler(0, max, 0) = 0 */
                    for (long i = 0; i < P; ++i) dst[i] = (0.f - 0.f) / (mx - 0.f);
                } else if (!spatial) {
                    const float v = (src[(size_t)b * T + start + k] - 0.f) / (mx - 0.f);
                    for (long i = 0; i < P; ++i) dst[i] = v;
                } else {
                    const float *plane = src + ((size_t)b * T + start + k) * P;
                    for (long i = 0; i < P; ++i) dst[i] = (plane[i] - 0.f) / (mx - 0.f);
                }
            } else if (c == 2 * nums) {
                const float *plane = dem + (size_t)b * P;
                for (long i = 0; i < P; ++i) dst[i] = (plane[i] - dem_min) / (dem_max - dem_min);
            } else if (c == 2 * nums + 1) {
                const float *plane = imperv + (size_t)b * P;
                const float lo = (float)0.05, span = (float)(0.95 - 0.05); /* python-double 0.95-0.05 rounded once, as torch does */
                for (long i = 0; i < P; ++i) dst[i] = (plane[i] - lo) / span;
            } else {
                const float *plane = manhole + (size_t)b * P;
                for (long i = 0; i < P; ++i) dst[i] = (plane[i] - 0.f) / (1.f - 0.f);
            }
        }
}
