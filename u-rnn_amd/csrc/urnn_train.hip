// urnn_train.hip -- kernels of the training path: backward of the ConvGRU / Skip-ConvGRU cell, the
// stage convs, the transposed convs and the head; loss; clipped Adam.  Deterministic reductions (per-chunk fp32 partials ->
// fixed-order double); the two contraction shapes run the forward path's arithmetic (bf16 x 6 split, fp32 accumulate):
//   dX = W^T . dY   -- the forward GEMM kernel (urnn_gemm.hip) on gathered / transposed packed weights, identity epilogue;
//   dW = dY . X^T   -- wgrad_kernel below: contraction over PIXELS, both operands split into bf16 pieces on the way into LDS.
#include "urnn_common.h"
#include "urnn_kernels.h"

// ------------------------------------------------------------------------------------------------------------------
// Per-(sample, channel) sums  S1 = sum_p a,  S2 = sum_p a * xhat(v)  with xhat = (v - mean_g) * rstd_g  (v == nullptr: S2 = 0).
// grid (chunks, B*C); partial[(b*C + c)][chunk][2]
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chan_partial_kernel(const float *__restrict__ a, long a_bs, const float *__restrict__ v, long v_bs,
                                                           const float *__restrict__ stat, int C, int P, int nchunk,
                                                           float *__restrict__ partial)
{
    __shared__ float sh[2][4];
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    const float *ap = a + (size_t)b * a_bs + (size_t)c * P;
    const float *vp = v ? v + (size_t)b * v_bs + (size_t)c * P : nullptr;
    float mu = 0.f, rs = 0.f;
    if (v) {
        mu = stat[((size_t)b * (C / 32) + c / 32) * 2];
        rs = stat[((size_t)b * (C / 32) + c / 32) * 2 + 1];
    }
    const int per = (P + nchunk - 1) / nchunk;
    const int lo = blockIdx.x * per, hi = min(P, lo + per);
    float s1 = 0.f, s2 = 0.f;
    for (int p = lo + threadIdx.x; p < hi; p += 256) {
        const float av = ap[p];
        s1 += av;
        if (vp) s2 += av * ((vp[p] - mu) * rs);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wave] = s1; sh[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float *pp = partial + ((size_t)bc * nchunk + blockIdx.x) * 2;
        pp[0] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        pp[1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}

// one wave per (b, c): chunks summed in double, fixed order -> sums[(b*C + c)][2] (double)
__global__ __launch_bounds__(64) void chan_finalize_kernel(const float *__restrict__ partial, int nchunk, double *__restrict__ sums)
{
    const int bc = blockIdx.x, lane = threadIdx.x;
    const float *pp = partial + (size_t)bc * nchunk * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int t = lane; t < nchunk; t += 64) {
        s1 += (double)pp[2 * t];
        s2 += (double)pp[2 * t + 1];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s1 += __shfl_xor(s1, m, 64);
        s2 += __shfl_xor(s2, m, 64);
    }
    if (lane == 0) {
        sums[(size_t)bc * 2] = s1;
        sums[(size_t)bc * 2 + 1] = s2;
    }
}

// GroupNorm backward coefficients.  One thread per (b, group): m1 = mean(dxhat), m2 = mean(dxhat * xhat) over the group's
// 32*P values (dxhat = dy * gamma); coef[(b*G + g)][2].  Threads with b == 0 also emit dgamma / dbeta (summed over samples).
__global__ void gn_bwd_coef_kernel(const double *__restrict__ sums, const float *__restrict__ gamma, int B, int C, double count,
                                   float *__restrict__ coef, float *__restrict__ dgamma, float *__restrict__ dbeta, int accumulate)
{
    const int G = C / 32;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < B * G) {
        const int b = idx / G, g = idx - b * G;
        double m1 = 0.0, m2 = 0.0;
        for (int j = 0; j < 32; ++j) {
            const int c = g * 32 + j;
            m1 += (double)gamma[c] * sums[((size_t)b * C + c) * 2];
            m2 += (double)gamma[c] * sums[((size_t)b * C + c) * 2 + 1];
        }
        coef[(size_t)idx * 2] = (float)(m1 / count);
        coef[(size_t)idx * 2 + 1] = (float)(m2 / count);
    }
    if (idx < C) {
        double d1 = 0.0, d2 = 0.0;
        for (int b = 0; b < B; ++b) {
            d1 += sums[((size_t)b * C + idx) * 2];
            d2 += sums[((size_t)b * C + idx) * 2 + 1];
        }
        dbeta[idx] = (accumulate ? dbeta[idx] : 0.f) + (float)d1;
        dgamma[idx] = (accumulate ? dgamma[idx] : 0.f) + (float)d2;
    }
}

// The two kernels above in one launch, for partials that the producer of dy already wrote (blend / reset-gate backward): one
// wave per 32-channel group; lane (j, half) adds every second chunk of channel g*32 + j in double, the halves are joined, the
// group coefficients come from an xor butterfly over the 32 channels (fixed order: bit-reproducible).
__global__ __launch_bounds__(64) void gn_bwd_sums_coef_kernel(const float *__restrict__ partial, int nchunk, const float *__restrict__ gamma,
                                                              int B, int C, double count, float *__restrict__ coef,
                                                              float *__restrict__ dgamma, float *__restrict__ dbeta, int accumulate)
{
    const int g = blockIdx.x, lane = threadIdx.x, j = lane & 31, hf = lane >> 5, G = C / 32;
    const int c = g * 32 + j;
    const double gm = (double)gamma[c];
    double d1 = 0.0, d2 = 0.0;
    for (int b = 0; b < B; ++b) {
        const float *pp = partial + ((size_t)b * C + c) * nchunk * 2;
        double s1 = 0.0, s2 = 0.0;
        for (int t = hf; t < nchunk; t += 2) {
            const f32x2 v = *reinterpret_cast<const f32x2 *>(pp + 2 * t);
            s1 += (double)v.x;
            s2 += (double)v.y;
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        d1 += s1;
        d2 += s2;
        double m1 = gm * s1, m2 = gm * s2;
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) {
            m1 += __shfl_xor(m1, m, 64);
            m2 += __shfl_xor(m2, m, 64);
        }
        if (lane == 0) {
            coef[((size_t)b * G + g) * 2] = (float)(m1 / count);
            coef[((size_t)b * G + g) * 2 + 1] = (float)(m2 / count);
        }
    }
    if (hf == 0) {
        dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)d1;
        dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)d2;
    }
}

// The streaming kernels of the backward pass walk a channel plane in QUADS of four consecutive pixels -- 16-byte accesses at 4-byte
// alignment (the quarter-resolution planes of 125 x 125 pixels start at every 4-byte phase); the last P & 3 pixels go one each to
// the first threads of the plane's first block.  Rounds 1-4 moved one float per thread and access: 4-byte streams reach 3.4-4.1 TB/s
// on this chip where 16-byte ones reach 5.2.  URNN_PLANE_WALK(P, BODY): BODY(n_tag, p) handles N = 4 or 1 pixels at p.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <int N>
__device__ __forceinline__ void ldn(const float *p, float (&v)[N])
{
    if constexpr (N == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4u *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = p[0];
    }
}
template <int N>
__device__ __forceinline__ void stn(float *p, const float (&v)[N])
{
    if constexpr (N == 4) *reinterpret_cast<f32x4u *>(p) = f32x4{v[0], v[1], v[2], v[3]};
    else p[0] = v[0];
}
#define URNN_PLANE_WALK(P_, BODY)                                                                                       \
    do {                                                                                                                \
        const int Pq_ = (P_) & ~3;                                                                                      \
        for (int p_ = (blockIdx.x * 256 + threadIdx.x) * 4; p_ < Pq_; p_ += gridDim.x * 1024) BODY(std::integral_constant<int, 4>{}, p_); \
        if (blockIdx.x == 0 && (int)threadIdx.x < (P_) - Pq_) BODY(std::integral_constant<int, 1>{}, Pq_ + (int)threadIdx.x);               \
    } while (0)

// dv = rstd * (gamma * dy - m1 - xhat * m2), in place over dy.  grid (chunks, B*C)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(float *dy, const float *__restrict__ v, const float *__restrict__ stat,
                                                           const float *__restrict__ coef, const float *__restrict__ gamma, int C, int P)
{
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    const int G = C / 32, g = c / 32;
    const float mu = stat[((size_t)b * G + g) * 2], rs = stat[((size_t)b * G + g) * 2 + 1];
    const float m1 = coef[((size_t)b * G + g) * 2], m2 = coef[((size_t)b * G + g) * 2 + 1];
    const float gm = gamma[c];
    float *dp = dy + (size_t)bc * P;
    const float *vp = v + (size_t)bc * P;
    auto body = [&](auto n_tag, int p) {
        constexpr int N = decltype(n_tag)::value;
        float d[N], x[N];
        ldn<N>(dp + p, d);
        ldn<N>(vp + p, x);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float xh = (x[k] - mu) * rs;
            d[k] = rs * (gm * d[k] - m1 - xh * m2);
        }
        stn<N>(dp + p, d);
    };
    URNN_PLANE_WALK(P, body);
}

// Blend backward (h' = (1 - z) h + z n,  z = sigmoid(GN(gz)),  n = tanh(GN(c))):
//   dy2 = dout * z * (1 - n^2)          gradient w.r.t. the normalised candidate
//   dyz = dout * (n - h) * z * (1 - z)  gradient w.r.t. the normalised update gate  -> dy1[:, :F]
//   dh  = dout * (1 - z)
__device__ __forceinline__ void block_partials(float (&v)[4], int n, float *dst0, float *dst1)
{
    // n = 2: (v0, v1) -> dst0;  n = 4: also (v2, v3) -> dst1.  256-thread blocks, fixed order.
    __shared__ float sh[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < n; ++k) {
        const float s = wave_sum(v[k]);
        if (lane == 0) sh[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        dst0[0] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        dst0[1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
        if (n == 4) {
            dst1[0] = (sh[2][0] + sh[2][1]) + (sh[2][2] + sh[2][3]);
            dst1[1] = (sh[3][0] + sh[3][1]) + (sh[3][2] + sh[3][3]);
        }
    }
}

// Also emits, per (plane, blockIdx.x), the partial sums the two GroupNorm backward passes need (sum dy, sum dy * xhat):
// part2[(b*F + f)][gx][2] for the candidate, part1[(b*2F + f)][gx][2] for the update-gate half of the gates.
__global__ __launch_bounds__(256) void blend_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ dout2, const float *__restrict__ dout3,
                                                        const float *__restrict__ dout4, const float *__restrict__ g1, const float *__restrict__ c,
                                                        const float *__restrict__ h, const float *__restrict__ ss1, const float *__restrict__ ss2,
                                                        const float *__restrict__ st1, const float *__restrict__ st2, float *__restrict__ dy2,
                                                        float *__restrict__ dy1, float *__restrict__ dh, float *__restrict__ part1,
                                                        float *__restrict__ part2, int F, int P)
{
    const int bc = blockIdx.y, b = bc / F, f = bc - b * F;
    const float s1 = ss1[((size_t)b * 2 * F + f) * 2], t1 = ss1[((size_t)b * 2 * F + f) * 2 + 1];
    const float s2 = ss2[((size_t)b * F + f) * 2], t2 = ss2[((size_t)b * F + f) * 2 + 1];
    const float mu1 = st1[((size_t)b * (2 * F / 32) + f / 32) * 2], rs1 = st1[((size_t)b * (2 * F / 32) + f / 32) * 2 + 1];
    const float mu2 = st2[((size_t)b * (F / 32) + f / 32) * 2], rs2 = st2[((size_t)b * (F / 32) + f / 32) * 2 + 1];
    const float *gz = g1 + ((size_t)b * 2 * F + f) * P, *cc = c + (size_t)bc * P, *hh = h + (size_t)bc * P, *dd = dout + (size_t)bc * P;
    // dL/dh' may arrive as up to four terms (the layer above, the skip connection's consumer, and the next timestep's two): summed
    // here, left to right, instead of by passes of their own
    const float *dd2 = dout2 ? dout2 + (size_t)bc * P : nullptr;
    const float *dd3 = dout3 ? dout3 + (size_t)bc * P : nullptr;
    const float *dd4 = dout4 ? dout4 + (size_t)bc * P : nullptr;
    float *o2 = dy2 + (size_t)bc * P, *o1 = dy1 + ((size_t)b * 2 * F + f) * P, *oh = dh + (size_t)bc * P;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};       // sum dyz, sum dyz * xhat1, sum dy2, sum dy2 * xhat2
    auto body = [&](auto n_tag, int p) {
        constexpr int N = decltype(n_tag)::value;
        float gv[N], cv[N], d[N], hv[N], t[N], a1[N], a2[N], oh_[N];
        ldn<N>(gz + p, gv);
        ldn<N>(cc + p, cv);
        ldn<N>(dd + p, d);
        ldn<N>(hh + p, hv);
        if (dd2) {
            ldn<N>(dd2 + p, t);
#pragma unroll
            for (int k = 0; k < N; ++k) d[k] += t[k];
        }
        if (dd3) {
            ldn<N>(dd3 + p, t);
#pragma unroll
            for (int k = 0; k < N; ++k) d[k] += t[k];
        }
        if (dd4) {
            ldn<N>(dd4 + p, t);
#pragma unroll
            for (int k = 0; k < N; ++k) d[k] += t[k];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float z = 1.0f / (1.0f + expf(-(gv[k] * s1 + t1)));
            const float n = tanhf(cv[k] * s2 + t2);
            a2[k] = d[k] * z * (1.0f - n * n);
            a1[k] = d[k] * (n - hv[k]) * z * (1.0f - z);
            oh_[k] = d[k] * (1.0f - z);
            acc[0] += a1[k];
            acc[1] += a1[k] * ((gv[k] - mu1) * rs1);
            acc[2] += a2[k];
            acc[3] += a2[k] * ((cv[k] - mu2) * rs2);
        }
        stn<N>(o2 + p, a2);
        stn<N>(o1 + p, a1);
        stn<N>(oh + p, oh_);
    };
    URNN_PLANE_WALK(P, body);
    block_partials(acc, 4, part1 + (((size_t)b * 2 * F + f) * gridDim.x + blockIdx.x) * 2, part2 + ((size_t)bc * gridDim.x + blockIdx.x) * 2);
}

// rh = sigmoid(GN(gr)) * h   (the candidate GEMM's gated rows, materialised for the weight gradient)
__global__ __launch_bounds__(256) void reset_gate_kernel(const float *__restrict__ g1, const float *__restrict__ h, const float *__restrict__ ss1,
                                                         float *__restrict__ rh, int F, int P)
{
    const int bc = blockIdx.y, b = bc / F, f = bc - b * F;
    const float s = ss1[((size_t)b * 2 * F + F + f) * 2], t = ss1[((size_t)b * 2 * F + F + f) * 2 + 1];
    const float *gr = g1 + ((size_t)b * 2 * F + F + f) * P, *hh = h + (size_t)bc * P;
    float *o = rh + (size_t)bc * P;
    auto body = [&](auto n_tag, int p) {
        constexpr int N = decltype(n_tag)::value;
        float gv[N], hv[N];
        ldn<N>(gr + p, gv);
        ldn<N>(hh + p, hv);
#pragma unroll
        for (int k = 0; k < N; ++k) hv[k] = hv[k] / (1.0f + expf(-(gv[k] * s + t)));
        stn<N>(o + p, hv);
    };
    URNN_PLANE_WALK(P, body);
}

// From d(r*h) (rows K-F.. of dA2, batch stride K*P): dyr = drh * h * r (1 - r) -> dy1[:, F:];  dh += drh * r
__global__ __launch_bounds__(256) void reset_gate_bwd_kernel(const float *__restrict__ drh, long drh_bs, const float *__restrict__ g1,
                                                             const float *__restrict__ h, const float *__restrict__ ss1,
                                                             const float *__restrict__ st1, float *__restrict__ dy1, float *__restrict__ dh,
                                                             float *__restrict__ part1, int F, int P)
{
    const int bc = blockIdx.y, b = bc / F, f = bc - b * F;
    const float s = ss1[((size_t)b * 2 * F + F + f) * 2], t = ss1[((size_t)b * 2 * F + F + f) * 2 + 1];
    const int grp = F / 32 + f / 32;
    const float mu = st1[((size_t)b * (2 * F / 32) + grp) * 2], rs = st1[((size_t)b * (2 * F / 32) + grp) * 2 + 1];
    const float *gr = g1 + ((size_t)b * 2 * F + F + f) * P, *hh = h + (size_t)bc * P;
    const float *dd = drh + (size_t)b * drh_bs + (size_t)f * P;
    float *o1 = dy1 + ((size_t)b * 2 * F + F + f) * P, *oh = dh + (size_t)bc * P;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto body = [&](auto n_tag, int p) {
        constexpr int N = decltype(n_tag)::value;
        float gv[N], d[N], hv[N], a[N], dhv[N];
        ldn<N>(gr + p, gv);
        ldn<N>(dd + p, d);
        ldn<N>(hh + p, hv);
        ldn<N>(oh + p, dhv);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float r = 1.0f / (1.0f + expf(-(gv[k] * s + t)));
            a[k] = d[k] * hv[k] * r * (1.0f - r);
            dhv[k] += d[k] * r;
            acc[0] += a[k];
            acc[1] += a[k] * ((gv[k] - mu) * rs);
        }
        stn<N>(o1 + p, a);
        stn<N>(oh + p, dhv);
    };
    URNN_PLANE_WALK(P, body);
    block_partials(acc, 2, part1 + (((size_t)b * 2 * F + F + f) * gridDim.x + blockIdx.x) * 2, nullptr);
}

// out[b][c][p] (+)= a[b][c][p] (+ a2[b][c][p]) with independent batch strides (channel slices of (B,K,P) buffers)
__global__ __launch_bounds__(256) void add_slices_kernel(float *out, long out_bs, const float *__restrict__ a, long a_bs,
                                                         const float *__restrict__ a2, long a2_bs, int C, int P, int accumulate)
{
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    float *o = out + (size_t)b * out_bs + (size_t)c * P;
    const float *pa = a + (size_t)b * a_bs + (size_t)c * P;
    const float *pb = a2 ? a2 + (size_t)b * a2_bs + (size_t)c * P : nullptr;
    auto body = [&](auto n_tag, int p) {
        constexpr int N = decltype(n_tag)::value;
        float v[N], t[N];
        ldn<N>(pa + p, v);
        if (pb) {
            ldn<N>(pb + p, t);
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = v[k] + t[k];
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = v[k] + 0.f;
        }
        if (accumulate) {
            ldn<N>(o + p, t);
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = t[k] + v[k];
        }
        stn<N>(o + p, v);
    };
    URNN_PLANE_WALK(P, body);
}

// Zero-fill as a kernel.  hipMemsetAsync captured into a hipGraph did not execute on replay (ROCm 7.0: the classification
// branch's gradient blocks kept whatever an earlier tensor of the graph's memory pool had left there -> gradient norm 6e18 in
// every captured training window, fine in eager mode); a kernel node has no such problem.
__global__ __launch_bounds__(256) void zero_kernel(float *__restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}

hipError_t urnn_train_zero(float *p, size_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const size_t b = (n + 1023) / 1024;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)(b > 2048 ? 2048 : b)), dim3(256), 0, st, p, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient dW[n][k] = sum_{b,p} dY[b][n][p] * X[b][k][p]  (+ row sums of dY for the bias gradient).
// The contraction runs over PIXELS, which are contiguous in both operands' rows, and both operands are activations.  Block = 4
// waves on a 128 n x 128 k output tile and one pixel chunk.  Per 32-pixel stage every thread splits the 4 x 4 pixels of dY
// and of X it fetched (16-byte global loads, one stage ahead, in registers) into three exact bf16 pieces -- ONCE per element,
// not once per wave that multiplies it -- and parks them in LDS as [operand][piece][128 rows][32 px bf16 + 16 B pad]: a lane's
// MFMA operand (8 consecutive pixels of its row) is then one conflict-free ds_read_b128 per piece (row pitch 80 B: 16
// consecutive rows tile the 64 banks).  The compute phase is LDS reads and six v_mfma_f32_32x32x16_bf16 per 32 x 32 tile and
// 16 pixels (small products first: the arithmetic of the forward GEMMs, fp32-class error at 2.67x the fp32 matrix rate); the
// bf16 training variant (NP = 1) rounds the operands and issues one.  61 KB of LDS: two blocks per CU hide each other's barriers.
// X rows come from up to three tensors (x | e | h-or-rh), k < K.  partial[chunk][N][K].
// History: fp32 MFMA out of an fp32 LDS tile: 62 us per launch on average at 500 x 500; the same tile with the split done on the
// fragments of every wave: 50 us (each element split twice, VALU and LDS latency in front of every MFMA group).
// ------------------------------------------------------------------------------------------------------------------
struct WgradParams {
    const float *dy;          // (B,N,P)
    const float *seg[3];      // X segments, (B,segC[i],P)
    int segC[3];              // real channel counts (0 = unused)
    int segK0[3];             // first k of each segment
    int N, K, P, B;
    int chunkPix;             // pixels per chunk (multiple of 64), chunks cover B*ceil(P/chunkPix)
    int chunksPerSample;
    int tilesK, tilesN;       // 128 x 128 output tiles
    int xcdMap;               // 0: tiles of a chunk on consecutive workgroups = different XCDs (development knob URNN_TUNE_WGRAD_MAP=0)
    float *partial;           // [chunks][N][K]
    float *rowpart;           // [chunks][N] sums of dY (written by the k-column 0 blocks), may be nullptr
};

#ifndef WG_ABL
#define WG_ABL 0
#endif
constexpr int WG_T = 128, WG_BK = 32, WG_PITCH = 80, WG_PIECE = WG_T * WG_PITCH;
constexpr int WG_RED_BYTES = 2 * 2 * 2 * 16 * 64 * 4;      // accumulator hand-over of two waves

// one 16-pixel group of a wave: AN x AK MFMA tiles
template <int AN, int AK, int NP>
__device__ __forceinline__ void wgrad_group(const char *__restrict__ pa, const char *__restrict__ pb, f32x16 (&acc)[2][2])
{
    bf16x8 fa[AN][NP];
#pragma unroll
    for (int a = 0; a < AN; ++a)
#pragma unroll
        for (int q = 0; q < NP; ++q) fa[a][q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(pa + q * WG_PIECE + a * 32 * WG_PITCH));
#pragma unroll
    for (int c = 0; c < AK; ++c) {                         // one X fragment at a time: its pieces live only across its own MFMAs
        bf16x8 fb[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) fb[q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(pb + q * WG_PIECE + c * 32 * WG_PITCH));
        auto mm = [&](int qa, int qb) {
#pragma unroll
            for (int a = 0; a < AN; ++a) {
#if (WG_ABL & 1)
                acc[a][c][0] += __builtin_bit_cast(f32x4, fa[a][qa]).x * __builtin_bit_cast(f32x4, fb[qb]).x;
#else
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][qa], fb[qb], acc[a][c], 0, 0, 0);
#endif
            }
        };
        if constexpr (NP == 3) { mm(1, 1); mm(2, 0); mm(0, 2); mm(1, 0); mm(0, 1); mm(0, 0); }
        else mm(0, 0);
        __builtin_amdgcn_sched_barrier(0);                 // keep the next fragment's pieces from being loaded early (registers)
    }
}

template <int NP>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams prm)
{
    extern __shared__ __attribute__((aligned(16))) char wg_smem[];
    char *tA = wg_smem, *tB = wg_smem + NP * WG_PIECE;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, half = lane >> 5;
    // 1-D grid, XCD-aware: workgroups go round-robin over the 8 XCDs, so the KX x NY output tiles of one pixel chunk are given
    // consecutive slots of ONE XCD -- they stream the same dY / X rows at about the same time and all but the first read of a
    // row can hit that XCD's L2.
    const int per = prm.tilesK * prm.tilesN;
    const int slot = prm.xcdMap ? blockIdx.x >> 3 : blockIdx.x, tsel = slot % per;
    const int chunk = prm.xcdMap ? (slot / per) * 8 + (blockIdx.x & 7) : slot / per;
    if (chunk >= prm.B * prm.chunksPerSample) return;
    const int n0 = (tsel / prm.tilesK) * WG_T, k0 = (tsel % prm.tilesK) * WG_T;
    const int b = chunk / prm.chunksPerSample;
    const int p_lo = (chunk - b * prm.chunksPerSample) * prm.chunkPix;
    const int p_hi = min(prm.P, p_lo + prm.chunkPix);
    const bool vec = (prm.P & 3) == 0;                     // rows 16-byte aligned

    // Only the 32 x 32 tiles that hold real outputs are multiplied (the layers run from 16 x 16 to 384 x 96, 192 x 288: a
    // fixed 128 x 128 grid of MFMAs would mostly multiply padding).  A full block gives each wave a 2 x 2 quadrant on both
    // 16-pixel groups of a stage; with at most two valid tile rows (or columns) the waves pair up on the tiles and take one
    // group each, and add their accumulators through LDS at the end (fixed order).
    const int rowsV = min(4, (prm.N - n0 + 31) >> 5), colsV = min(4, (prm.K - k0 + 31) >> 5);
    int row0, col0, g0, gn, akmax = 2;
    if (rowsV > 2 && colsV > 2) { row0 = 2 * (wave >> 1); col0 = 2 * (wave & 1); g0 = 0; gn = 2; }
    else if (rowsV <= 2 && colsV <= 2) { row0 = 0; col0 = wave & 1; akmax = 1; g0 = wave >> 1; gn = 1; }
    else if (rowsV <= 2) { row0 = 0; col0 = 2 * (wave & 1); g0 = wave >> 1; gn = 1; }
    else { row0 = 2 * (wave & 1); col0 = 0; g0 = wave >> 1; gn = 1; }
    const int owners = gn == 2 ? 4 : 2;                    // waves 0 .. owners-1 own a distinct tile set each
    const int an = max(0, min(2, rowsV - row0)), ak = max(0, min(akmax, colsV - col0));

    // this thread stages rows r0 + 32*i (i < 4) of both operands, 4 consecutive pixels at column c4.  dY rows: one uniform base
    // and a 32-bit offset per row; X rows come from up to three tensors: a pointer each.  Rows that are not staged (beyond N /
    // K, or a missing segment) point at a valid dummy row so that the main loop's loads need no branch.
    const int r0 = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const float *dyb = prm.dy + (size_t)b * prm.N * prm.P;
    unsigned okA = 0, okB = 0, zeroB = 0;                  // bit i: row r0 + 32*i is staged / staged as zeros (missing segment inside K)
    int offA[4];
    const float *ptrB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + r0 + 32 * i;
        okA |= (n < prm.N ? 1u : 0u) << i;
        offA[i] = min(n, prm.N - 1) * prm.P + c4;
        const int k = k0 + r0 + 32 * i;
        ptrB[i] = dyb + c4;
        if (k < prm.K) {
            const int sg = k >= prm.segK0[2] ? 2 : (k >= prm.segK0[1] ? 1 : 0);
            const int ch = k - prm.segK0[sg];
            if (ch < prm.segC[sg]) {
                ptrB[i] = prm.seg[sg] + ((size_t)b * prm.segC[sg] + ch) * prm.P + c4;
                okB |= 1u << i;
            } else {
                zeroB |= 1u << i;
            }
        }
    }
    // rows beyond N / K are never staged -- an output only depends on ITS row of dY and ITS row of X, and the outputs beyond
    // N / K are not stored
    f32x4 ra[2][4], rb[2][4];                              // two 32-pixel stages in flight (bytes in flight are what bounds the stream)
    auto fetch = [&](const float *row, bool ok, int p) -> f32x4 {           // row already includes c4
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const int q = p + c4;
            if (vec && q + 3 < p_hi) v = *reinterpret_cast<const f32x4 *>(row + p);
            else {
                if (q < p_hi) v.x = row[p];
                if (q + 1 < p_hi) v.y = row[p + 1];
                if (q + 2 < p_hi) v.z = row[p + 2];
                if (q + 3 < p_hi) v.w = row[p + 3];
            }
        }
        return v;
    };
    auto load_stage = [&](int p0, f32x4 (&a4)[4], f32x4 (&b4)[4]) {      // any stage: bounds-checked, zero-filled (tails, unaligned planes)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a4[i] = fetch(dyb + offA[i], (okA >> i) & 1, p0);
            b4[i] = fetch(ptrB[i], (okB >> i) & 1, p0);
        }
    };
    // Full, aligned stages: unconditional 16-byte loads.  No control flow between issue and use, so the compiler keeps COUNTED
    // vmcnt waits and the stage after next really stays in flight -- with the bounds-checked fetch inside the loop it fell
    // back to vmcnt(0) at every park and the second stage in flight made the kernel slower (dec1 gates 116 -> 166 us).
    // (rows of an odd plane -- 125 x 125 at quarter resolution -- are only 4-byte aligned: global_load_dwordx4 takes that)
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    auto load_fast = [&](int p0, f32x4 (&a4)[4], f32x4 (&b4)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a4[i] = *reinterpret_cast<const f32x4u *>(dyb + offA[i] + p0);
            b4[i] = *reinterpret_cast<const f32x4u *>(ptrB[i] + p0);
        }
    };
    // four pixels -> two dwords of each piece, 8 bytes per piece at (row, c4)
    auto park = [&](char *base, int row, const f32x4 &v) {
        unsigned q0[NP], q1[NP];
        if constexpr (NP == 3) {
#if (WG_ABL & 2)
            q0[0] = __float_as_uint(v.x); q0[1] = __float_as_uint(v.y); q0[2] = __float_as_uint(v.z);
            q1[0] = __float_as_uint(v.w); q1[1] = __float_as_uint(v.x); q1[2] = __float_as_uint(v.y);
#else
            split_pair(v.x, v.y, q0[0], q0[1], q0[2]);
            split_pair(v.z, v.w, q1[0], q1[1], q1[2]);
#endif
        } else {
            q0[0] = round_pair(v.x, v.y);
            q1[0] = round_pair(v.z, v.w);
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2 *>(base + q * WG_PIECE + row * WG_PITCH + c4 * 2) = make_uint2(q0[q], q1[q]);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    const bool sums = prm.rowpart && k0 == 0;              // bias gradient: row sums of dY, taken from the staged registers
    float rsum[4] = {0.f, 0.f, 0.f, 0.f};
    const char *pa = tA + (row0 * 32 + j) * WG_PITCH + half * 16, *pb = tB + (col0 * 32 + j) * WG_PITCH + half * 16;

    auto park_stage = [&](f32x4 (&a4)[4], f32x4 (&b4)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if ((okA >> i) & 1) park(tA, r0 + 32 * i, a4[i]);
            if ((okB >> i) & 1) park(tB, r0 + 32 * i, b4[i]);
            else if ((zeroB >> i) & 1) park(tB, r0 + 32 * i, f32x4{0.f, 0.f, 0.f, 0.f});
            if (sums) rsum[i] += (a4[i].x + a4[i].y) + (a4[i].z + a4[i].w);
        }
    };
    auto multiply = [&]() {
        if (an > 0 && ak > 0) {
            const char *qa = pa + g0 * 32, *qb = pb + g0 * 32;
            if (an == 2 && ak == 2) {
                wgrad_group<2, 2, NP>(qa, qb, acc);
                if (gn == 2) wgrad_group<2, 2, NP>(qa + 32, qb + 32, acc);
            } else if (an == 2) {
                wgrad_group<2, 1, NP>(qa, qb, acc);
                if (gn == 2) wgrad_group<2, 1, NP>(qa + 32, qb + 32, acc);
            } else if (ak == 2) {
                wgrad_group<1, 2, NP>(qa, qb, acc);
                if (gn == 2) wgrad_group<1, 2, NP>(qa + 32, qb + 32, acc);
            } else {
                wgrad_group<1, 1, NP>(qa, qb, acc);
                if (gn == 2) wgrad_group<1, 1, NP>(qa + 32, qb + 32, acc);
            }
        }
    };
    // main part: pairs of full 32-pixel stages, two stages in flight; the last pair re-fetches itself instead of branching
    const int npair = (p_hi - p_lo) / (2 * WG_BK);
    if (npair > 0) {
        const int p_last = p_lo + (npair - 1) * 2 * WG_BK;
        load_fast(p_lo, ra[0], rb[0]);
        load_fast(p_lo + WG_BK, ra[1], rb[1]);
        for (int p0 = p_lo; p0 <= p_last; p0 += 2 * WG_BK) {
            const int pn = min(p0 + 2 * WG_BK, p_last);
            park_stage(ra[0], rb[0]);
            __syncthreads();
#if !(WG_ABL & 4)
            load_fast(pn, ra[0], rb[0]);
#endif
            multiply();
            __syncthreads();
            park_stage(ra[1], rb[1]);
            __syncthreads();
#if !(WG_ABL & 4)
            load_fast(pn + WG_BK, ra[1], rb[1]);
#endif
            multiply();
            __syncthreads();
        }
    }
    // the rest (< 64 pixels): bounds-checked stages, nothing in flight
    for (int p0 = p_lo + npair * 2 * WG_BK; p0 < p_hi; p0 += WG_BK) {
        load_stage(p0, ra[0], rb[0]);
        park_stage(ra[0], rb[0]);
        __syncthreads();
        multiply();
        __syncthreads();
    }
    if (owners < 4) {                                      // waves 2, 3 hand their accumulators to waves 0, 1
        float *red = reinterpret_cast<float *>(wg_smem);   // [wave - 2][2][2][16][64]
        if (wave >= owners) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (a < an && c < ak)
#pragma unroll
                        for (int r = 0; r < 16; ++r) red[((((wave - owners) * 2 + a) * 2 + c) * 16 + r) * 64 + lane] = acc[a][c][r];
        }
        __syncthreads();
        if (wave < owners)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (a < an && c < ak)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][c][r] += red[(((wave * 2 + a) * 2 + c) * 16 + r) * 64 + lane];
    }
    float *out = prm.partial + (size_t)chunk * prm.N * prm.K;
    if (wave < owners) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (a < an && c < ak)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = n0 + (row0 + a) * 32 + mfma_row(r, half), k = k0 + (col0 + c) * 32 + j;
                        if (n < prm.N && k < prm.K) out[(size_t)n * prm.K + k] = acc[a][c][r];
                    }
    }
    if (sums) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = rsum[i];                             // the row's 8 threads are 8 consecutive lanes
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            const int n = n0 + r0 + 32 * i;
            if ((threadIdx.x & 7) == 0 && n < prm.N) prm.rowpart[(size_t)chunk * prm.N + n] = v;
        }
    }
}

// dW[i] (+)= sum over chunks (double, fixed order); the same for the bias row sums
__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const float *__restrict__ partial, const float *__restrict__ rowpart, int nchunk,
                                                             long cntW, long cntB, float *__restrict__ dW, float *__restrict__ db, int accumulate)
{
    // outputs 0 .. cntW-1: weight gradient, cntW .. cntW+cntB-1: bias gradient (row sums).  A block finishes 16 outputs:
    // thread (q, io) sums chunks q, q+16, ... of output io in double (independent loads, 16 lanes on consecutive outputs),
    // then the 16 sub-sums of an output are added in a fixed order
    __shared__ double sub[16][17];
    const int io = threadIdx.x & 15, q = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + io;
    const bool isW = i < cntW, live = i < cntW + cntB;
    const float *src = isW ? partial + i : rowpart + (i - cntW);
    const long stride = isW ? cntW : cntB;
    double s = 0.0;
    if (live) {
        // 32 independent loads in flight per thread (512 chunks = one batch), added in chunk order: the sum is a chain of
        // memory round trips otherwise (8 us per launch, 22 launches per training timestep)
        for (int c0 = q; c0 < nchunk; c0 += 16 * 32) {
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {             // unconditional loads (clamped index): a predicated load is a branch and a wait each
                const int c = c0 + 16 * u;
                v[u] = src[(size_t)(c < nchunk ? c : nchunk - 1) * stride];
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) s += (c0 + 16 * u < nchunk) ? (double)v[u] : 0.0;
        }
    }
    sub[q][io] = s;
    __syncthreads();
    if (threadIdx.x < 16 && live) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sub[k][io];
        float *o = isW ? dW + i : db + (i - cntW);
        *o = (accumulate ? *o : 0.f) + (float)t;
    }
}

// "Weights" (Cout x Cin, row-major, the layout pack_conv takes) of the three input-gradient GEMMs of a cell, gathered from
// W1 (2F x K) and W2 (F x K), K ordered x | e | h, kh = K - F:
//   mode 0: d(r*h) = W2[:, h]^T . dc              out[j][n] = W2[n][kh + j]                           (F x F)
//   mode 1: d[x;e] = W1[:, xe]^T . dg + W2[:, xe]^T . dc   out[r][n] = n < 2F ? W1[n][rlo + r] : W2[n - 2F][rlo + r]   (nrows x 3F)
//   mode 2: dh    += W1[:, h]^T . dg              out[j][n] = W1[n][kh + j]                           (F x 2F)
__global__ void cell_bwd_weights_kernel(const float *__restrict__ W1, const float *__restrict__ W2, float *__restrict__ out, int F, int K,
                                        int rlo, int nrows, int mode)
{
    const int cin = mode == 0 ? F : (mode == 1 ? 3 * F : 2 * F);
    const int rows = mode == 1 ? nrows : F;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cin) return;
    const int r = i / cin, n = i - r * cin;
    const int col = mode == 1 ? rlo + r : K - F + r;
    float v;
    if (mode == 0) v = W2[(size_t)n * K + col];
    else if (mode == 2) v = W1[(size_t)n * K + col];
    else v = n < 2 * F ? W1[(size_t)n * K + col] : W2[(size_t)(n - 2 * F) * K + col];
    out[i] = v;
}

hipError_t urnn_train_cell_bwd_weights(const float *W1, const float *W2, float *out, int F, int K, int rlo, int nrows, int mode, hipStream_t st)
{
    const int cin = mode == 0 ? F : (mode == 1 ? 3 * F : 2 * F);
    const int total = (mode == 1 ? nrows : F) * cin;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(cell_bwd_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, st, W1, W2, out, F, K, rlo, nrows, mode);
    return hipGetLastError();
}

// weight (N,K) row-major -> transposed (K,N) row-major (the "weight" of the dX GEMM)
__global__ void transpose_kernel(const float *__restrict__ w, float *__restrict__ wt, int N, int K)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * K) {
        const int n = i / K, k = i - n * K;
        wt[(size_t)k * N + n] = w[i];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------------------------
static dim3 plane_grid(int P, int planes)
{
    int gx = (P + 1023) / 1024;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    return dim3(gx, planes);
}

hipError_t urnn_train_chan_sums(const float *a, long a_bs, const float *v, long v_bs, const float *stat, int B, int C, int P,
                                float *partial, double *sums, hipStream_t st)
{
    const int nchunk = urnn_train_nchunk(P);
    hipLaunchKernelGGL(chan_partial_kernel, dim3(nchunk, B * C), dim3(256), 0, st, a, a_bs, v, v_bs, stat, C, P, nchunk, partial);
    hipLaunchKernelGGL(chan_finalize_kernel, dim3(B * C), dim3(64), 0, st, partial, nchunk, sums);
    return hipGetLastError();
}

int urnn_train_nchunk(int P)
{
    const int n = (P + 4095) / 4096;
    return n < 1 ? 1 : (n > 64 ? 64 : n);
}

// have_partials: the producers of dy (blend / reset-gate backward) already wrote partial[(b*C + c)][plane_grid x][2]
hipError_t urnn_train_gn_backward(float *dy, const float *v, const float *stat, const float *gamma, int B, int C, int P, float *partial,
                                  double *sums, float *coef, float *dgamma, float *dbeta, int accumulate, int have_partials, hipStream_t st)
{
    if (have_partials) {
        hipLaunchKernelGGL(gn_bwd_sums_coef_kernel, dim3(C / 32), dim3(64), 0, st, partial, (int)plane_grid(P, 1).x, gamma, B, C,
                           32.0 * (double)P, coef, dgamma, dbeta, accumulate);
    } else {
        hipError_t e = urnn_train_chan_sums(dy, (long)C * P, v, (long)C * P, stat, B, C, P, partial, sums, st);
        if (e != hipSuccess) return e;
        const int n = B * (C / 32) > C ? B * (C / 32) : C;
        hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3((n + 127) / 128), dim3(128), 0, st, sums, gamma, B, C, 32.0 * (double)P, coef, dgamma,
                           dbeta, accumulate);
    }
    hipLaunchKernelGGL(gn_bwd_apply_kernel, plane_grid(P, B * C), dim3(256), 0, st, dy, v, stat, coef, gamma, C, P);
    return hipGetLastError();
}

hipError_t urnn_train_blend_bwd(const float *dout, const float *dout2, const float *dout3, const float *dout4, const float *g1, const float *c,
                                const float *h, const float *ss1, const float *ss2,
                                const float *st1, const float *st2, float *dy2, float *dy1, float *dh, float *part1, float *part2, int B,
                                int F, int P, hipStream_t st)
{
    hipLaunchKernelGGL(blend_bwd_kernel, plane_grid(P, B * F), dim3(256), 0, st, dout, dout2, dout3, dout4, g1, c, h, ss1, ss2, st1, st2, dy2, dy1, dh,
                       part1, part2, F, P);
    return hipGetLastError();
}

hipError_t urnn_train_reset_gate(const float *g1, const float *h, const float *ss1, float *rh, int B, int F, int P, hipStream_t st)
{
    hipLaunchKernelGGL(reset_gate_kernel, plane_grid(P, B * F), dim3(256), 0, st, g1, h, ss1, rh, F, P);
    return hipGetLastError();
}

hipError_t urnn_train_reset_gate_bwd(const float *drh, long drh_bs, const float *g1, const float *h, const float *ss1, const float *st1,
                                     float *dy1, float *dh, float *part1, int B, int F, int P, hipStream_t st)
{
    hipLaunchKernelGGL(reset_gate_bwd_kernel, plane_grid(P, B * F), dim3(256), 0, st, drh, drh_bs, g1, h, ss1, st1, dy1, dh, part1, F, P);
    return hipGetLastError();
}

hipError_t urnn_train_add_slices(float *out, long out_bs, const float *a, long a_bs, const float *a2, long a2_bs, int B, int C, int P,
                                 int accumulate, hipStream_t st)
{
    if (C < 1) return hipSuccess;
    hipLaunchKernelGGL(add_slices_kernel, plane_grid(P, B * C), dim3(256), 0, st, out, out_bs, a, a_bs, a2, a2_bs, C, P, accumulate);
    return hipGetLastError();
}

int urnn_train_wgrad_chunks(int B, int N, int K, int P);

size_t urnn_train_wgrad_partial_floats(int B, int N, int K, int P)
{
    return (size_t)B * urnn_train_wgrad_chunks(B, N, K, P) * ((size_t)N * K + N);
}

// pixel chunks per sample of the weight-gradient GEMM: ONE round of blocks (two per CU: 512) whatever the plane and the N x K
// tile count -- every block resident from the start keeps the most bytes in flight and leaves no half-empty second round --
// at least one 64-pixel stage pair per chunk, at most 512 chunks (their partial tiles are summed afterwards)
int urnn_train_wgrad_chunks(int B, int N, int K, int P)
{
    static const int target = (int)urnn_tune("URNN_TUNE_WGRAD_BLOCKS", 512);
    const int tiles = ((N + WG_T - 1) / WG_T) * ((K + WG_T - 1) / WG_T) * B;
    int n = target / tiles;
    const int most = (P + 63) / 64;
    n = n > most ? most : n;
    n = n > 512 ? 512 : n;
    return n < 1 ? 1 : n;
}

hipError_t urnn_train_wgrad(const float *dy, const float *const seg[3], const int segC[3], int B, int N, int K, int P, float *partial,
                            float *dW, float *db, int accumulate, hipStream_t st)
{
    WgradParams w = {};
    w.dy = dy;
    int k0 = 0;
    for (int i = 0; i < 3; ++i) {
        w.seg[i] = seg[i];
        w.segC[i] = seg[i] ? segC[i] : 0;
        w.segK0[i] = k0;
        k0 += segC[i];          // a missing segment (x == nullptr) still owns its weight columns: they get zero gradient
    }
    w.N = N; w.K = K; w.P = P; w.B = B;
    w.chunksPerSample = urnn_train_wgrad_chunks(B, N, K, P);
    w.chunkPix = ((P + w.chunksPerSample - 1) / w.chunksPerSample + 63) / 64 * 64;
    const int chunks = B * w.chunksPerSample;
    w.partial = partial;
    w.rowpart = db ? partial + (size_t)chunks * N * K : nullptr;
    // arithmetic follows the forward GEMMs: bf16x6 split (fp32-class) by default, rounded bf16 in the bf16 training variant
    const int np = urnn_get_matrix_mode() == URNN_MATRIX_BF16 ? 1 : 3;
    size_t lds = (size_t)2 * np * WG_PIECE;
    lds = lds < WG_RED_BYTES ? WG_RED_BYTES : lds;
    void (*kern)(const WgradParams) = np == 3 ? wgrad_kernel<3> : wgrad_kernel<1>;
    static bool big[2] = {false, false};
    if (!big[np == 3]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        big[np == 3] = true;
    }
    w.tilesK = (K + WG_T - 1) / WG_T;
    w.tilesN = (N + WG_T - 1) / WG_T;
    static const int xcd_map = (int)urnn_tune("URNN_TUNE_WGRAD_MAP", 1);
    w.xcdMap = xcd_map;
    hipLaunchKernelGGL(kern, dim3((unsigned)((chunks + 7) / 8 * 8 * w.tilesK * w.tilesN)), dim3(256), lds, st, w);
    const long cntW = (long)N * K, cntB = db ? N : 0;
    hipLaunchKernelGGL(wgrad_finalize_kernel, dim3((unsigned)((cntW + cntB + 15) / 16)), dim3(256), 0, st, partial, w.rowpart, chunks,
                       cntW, cntB, dW, db, accumulate);
    return hipGetLastError();
}

hipError_t urnn_train_transpose(const float *w, float *wt, int N, int K, hipStream_t st)
{
    hipLaunchKernelGGL(transpose_kernel, dim3((N * K + 255) / 256), dim3(256), 0, st, w, wt, N, K);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Stage conv / deconv backward helpers
// ------------------------------------------------------------------------------------------------------------------
// u holds the pre-activation W.x + b (B,C,P); du = dy_at(p) * lrelu'(u), written over u.  pool: dy is (B,C,P2) and every
// input pixel receives a quarter of its 2x2 cell's gradient (AvgPool2d(2,2) backward).
__global__ __launch_bounds__(256) void lrelu_pool_bwd_kernel(float *u, const float *__restrict__ dy, int P, int W, int P2, int W2, int pool,
                                                             float slope)
{
    const int bc = blockIdx.y;
    float *up = u + (size_t)bc * P;
    const float *dp = dy + (size_t)bc * (pool ? P2 : P);
    auto body = [&](auto n_tag, int p) {
        constexpr int N = decltype(n_tag)::value;
        float uv[N], d[N];
        ldn<N>(up + p, uv);
        if (pool) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int y = (p + k) / W, x = (p + k) - y * W;
                const int y2 = y >> 1, x2 = x >> 1;
                d[k] = (y2 * W2 + x2 < P2 && x2 < W2) ? 0.25f * dp[y2 * W2 + x2] : 0.f;   // odd last row / column: dropped by the floor pooling
            }
        } else {
            ldn<N>(dp + p, d);
        }
#pragma unroll
        for (int k = 0; k < N; ++k) uv[k] = uv[k] >= 0.f ? d[k] : d[k] * slope;
        stn<N>(up + p, uv);
    };
    URNN_PLANE_WALK(P, body);
}

// Deconv: D4[b][(a*2+bb)*Cout + co][i*W + j] = dy[b][co][2i+a][2j+bb] * lrelu'(y[...])   (y = forward output, sign-preserving)
__global__ __launch_bounds__(256) void deconv_unshuffle_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, float *__restrict__ d4,
                                                                   int Cout, int P, int W, float slope)
{
    const int bn = blockIdx.y;                 // b * 4*Cout + n
    const int b = bn / (4 * Cout), n = bn - b * 4 * Cout;
    const int ab = n / Cout, co = n - ab * Cout, a = ab >> 1, bb = ab & 1;
    const float *dp = dy + ((size_t)b * Cout + co) * 4 * P, *yp = y + ((size_t)b * Cout + co) * 4 * P;
    float *o = d4 + (size_t)bn * P;
    const int W2 = 2 * W;
    if (W % 4 == 0) {
        // four consecutive input pixels lie in one row: their outputs are every second float of eight consecutive ones
        for (int p = (blockIdx.x * 256 + threadIdx.x) * 4; p < P; p += gridDim.x * 1024) {
            const int i = p / W, j = p - i * W;
            const size_t q = (size_t)(2 * i + a) * W2 + 2 * j;
            float y8[8], d8[8], ov[4];
            ldn<4>(yp + q, *reinterpret_cast<float(*)[4]>(y8));
            ldn<4>(yp + q + 4, *reinterpret_cast<float(*)[4]>(y8 + 4));
            ldn<4>(dp + q, *reinterpret_cast<float(*)[4]>(d8));
            ldn<4>(dp + q + 4, *reinterpret_cast<float(*)[4]>(d8 + 4));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float yv = bb ? y8[2 * k + 1] : y8[2 * k], dv = bb ? d8[2 * k + 1] : d8[2 * k];
                ov[k] = yv >= 0.f ? dv : dv * slope;
            }
            stn<4>(o + p, ov);
        }
        return;
    }
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        const int i = p / W, j = p - i * W;
        const size_t q = (size_t)(2 * i + a) * W2 + 2 * j + bb;
        o[p] = yp[q] >= 0.f ? dp[q] : dp[q] * slope;
    }
}

// w (Cin,Cout,2,2) -> rows[(ab*Cout + co)][ci]
__global__ void deconv_weight_rows_kernel(const float *__restrict__ w, float *__restrict__ rows, int Cin, int Cout)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * Cout * Cin) return;
    const int n = i / Cin, ci = i - n * Cin;
    const int ab = n / Cout, co = n - ab * Cout;
    rows[i] = w[((size_t)ci * Cout + co) * 4 + ab];
}

// drows[(ab*Cout + co)][ci], drow_sum[(ab*Cout + co)] -> dw (Cin,Cout,2,2) (+)=, db[co] (+)= sum_ab
__global__ void deconv_rows_weight_kernel(const float *__restrict__ drows, const float *__restrict__ dsum, float *__restrict__ dw,
                                          float *__restrict__ db, int Cin, int Cout, int accumulate)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * Cout * Cin) {
        const int n = i / Cin, ci = i - n * Cin;
        const int ab = n / Cout, co = n - ab * Cout;
        float *o = dw + ((size_t)ci * Cout + co) * 4 + ab;
        *o = (accumulate ? *o : 0.f) + drows[i];
    }
    if (i < Cout) {
        const float s = (dsum[i] + dsum[Cout + i]) + (dsum[2 * Cout + i] + dsum[3 * Cout + i]);
        db[i] = (accumulate ? db[i] : 0.f) + s;
    }
}

hipError_t urnn_train_lrelu_pool_bwd(float *u, const float *dy, int B, int C, int H, int W, int pool, float slope, hipStream_t st)
{
    const int P = H * W, W2 = W / 2, P2 = (H / 2) * W2;
    hipLaunchKernelGGL(lrelu_pool_bwd_kernel, plane_grid(P, B * C), dim3(256), 0, st, u, dy, P, W, P2, W2, pool, slope);
    return hipGetLastError();
}

hipError_t urnn_train_deconv_unshuffle(const float *dy, const float *y, float *d4, int B, int Cout, int H, int W, float slope, hipStream_t st)
{
    const int P = H * W;
    hipLaunchKernelGGL(deconv_unshuffle_bwd_kernel, plane_grid(P, B * 4 * Cout), dim3(256), 0, st, dy, y, d4, Cout, P, W, slope);
    return hipGetLastError();
}

hipError_t urnn_train_deconv_weight_rows(const float *w, float *rows, int Cin, int Cout, hipStream_t st)
{
    hipLaunchKernelGGL(deconv_weight_rows_kernel, dim3((4 * Cout * Cin + 255) / 256), dim3(256), 0, st, w, rows, Cin, Cout);
    return hipGetLastError();
}

hipError_t urnn_train_deconv_rows_weight(const float *drows, const float *dsum, float *dw, float *db, int Cin, int Cout, int accumulate,
                                         hipStream_t st)
{
    hipLaunchKernelGGL(deconv_rows_weight_kernel, dim3((4 * Cout * Cin + 255) / 256), dim3(256), 0, st, drows, dsum, dw, db, Cin, Cout, accumulate);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Head backward (flood_head.py:131-202, regression branch: the classification branch only enters through the wet/dry mask,
// a comparison, and receives no gradient).  C = 16 channels, one thread per pixel.
//   stem:  u0 = Ws.f      t  = SiLU(LN0(u0))        block index 0
//   reg0:  v1 = Wq1.t     q1 = SiLU(LN3(v1))        block index 3
//   reg1:  v2 = Wq2.q1    q2 = SiLU(LN4(v2))        block index 4
//   pred:  raw = w_r.q2 + b_r,  reg = lrelu(raw),  out = reg * [cls >= thr]
// LayerNorm([16,H,W]): statistics over all 16*P values of a sample, element-wise affine (16,P).
// ------------------------------------------------------------------------------------------------------------------
#define HC 16

__device__ __forceinline__ void hb_matvec(const float *__restrict__ w, const float (&x)[HC], float (&u)[HC])
{
#pragma unroll
    for (int n = 0; n < HC; ++n) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HC; ++c) s = fmaf(w[n * HC + c], x[c], s);
        u[n] = s;
    }
}

__device__ __forceinline__ void hb_ln_silu(const float (&u)[HC], const float *__restrict__ g, const float *__restrict__ bt, int P, int p,
                                           float mean, float rstd, float (&s)[HC])
{
#pragma unroll
    for (int c = 0; c < HC; ++c) {
        const float y = (u[c] - mean) * rstd * g[(size_t)c * P + p] + bt[(size_t)c * P + p];
        s[c] = y / (1.0f + expf(-y));
    }
}

// recompute and store the regression branch: pre-norm u0, v1, v2 and layer inputs t, q1, q2 (each (B,16,P)); stats [5][B][2]
__global__ __launch_bounds__(256) void head_train_save_kernel(const float *__restrict__ feat, const float *__restrict__ conv_w,
                                                              const float *__restrict__ ln_w, const float *__restrict__ ln_b,
                                                              const float *__restrict__ stats, int B, int P, float *__restrict__ save)
{
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= P) return;
    const size_t CP = (size_t)HC * P, plane = (size_t)B * CP;
    float f[HC], u[HC], s[HC];
#pragma unroll
    for (int c = 0; c < HC; ++c) f[c] = feat[b * CP + (size_t)c * P + p];
    auto put = [&](int slot, const float (&v)[HC]) {
#pragma unroll
        for (int c = 0; c < HC; ++c) save[slot * plane + b * CP + (size_t)c * P + p] = v[c];
    };
    const int blk[3] = {0, 3, 4};
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        hb_matvec(conv_w + blk[l] * HC * HC, f, u);
        put(l, u);                                                            // slots 0..2: u0, v1, v2
        hb_ln_silu(u, ln_w + blk[l] * CP, ln_b + blk[l] * CP, P, p, stats[(blk[l] * B + b) * 2], stats[(blk[l] * B + b) * 2 + 1], s);
        put(3 + l, s);                                                        // slots 3..5: t, q1, q2
#pragma unroll
        for (int c = 0; c < HC; ++c) f[c] = s[c];
    }
}

// prediction layer: draw = dout * mask * lrelu'(reg);  ds[c] = w_r[c] * draw.  grid (chunks, B), pixels in quads (URNN_PLANE_WALK)
__global__ __launch_bounds__(256) void head_pred_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ cls,
                                                            const float *__restrict__ reg, const float *__restrict__ reg_w, float thr,
                                                            float slope, int P, float *__restrict__ draw, float *__restrict__ ds)
{
    const int b = blockIdx.y;
    const size_t base = (size_t)b * P;
    float w[HC];
#pragma unroll
    for (int c = 0; c < HC; ++c) w[c] = reg_w[c];
    auto body = [&](auto n_tag, int p) {
        constexpr int N = decltype(n_tag)::value;
        float cv[N], dv[N], rv[N], d[N], o[N];
        ldn<N>(cls + base + p, cv);
        ldn<N>(dout + base + p, dv);
        ldn<N>(reg + base + p, rv);
#pragma unroll
        for (int k = 0; k < N; ++k) d[k] = (cv[k] >= thr ? dv[k] : 0.f) * (rv[k] >= 0.f ? 1.f : slope);
        stn<N>(draw + base + p, d);
#pragma unroll
        for (int c = 0; c < HC; ++c) {
#pragma unroll
            for (int k = 0; k < N; ++k) o[k] = w[c] * d[k];
            stn<N>(ds + ((size_t)b * HC + c) * P + p, o);
        }
    };
    URNN_PLANE_WALK(P, body);
}

// LayerNorm + SiLU backward, first half: dy = ds * SiLU'(y); dgamma/dbeta (element-wise, summed over samples);
// dxhat = dy * gamma -> ds (in place); per-sample partial sums of dxhat and dxhat * xhat.  grid (chunks of 1024 pixels, 16 channels):
// a thread owns one quad of four pixels of one channel plane (16-byte accesses) and loops over the samples;
// partial[b][channel * chunks + chunk][2].
__global__ __launch_bounds__(256) void head_ln_bwd_a_kernel(float *ds, const float *__restrict__ u, const float *__restrict__ g,
                                                            const float *__restrict__ bt, const float *__restrict__ stats, int B, int P,
                                                            float *dg, float *dbt, int accumulate, float *__restrict__ partial)
{
    __shared__ float sh[2][4];
    const int c = blockIdx.y;
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int n = P - p >= 4 ? 4 : (P - p > 0 ? P - p : 0);          // pixels of this thread's quad inside the plane
    const size_t CP = (size_t)HC * P, row = (size_t)c * P + p;
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f}, gm[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (n == 4) {
        ldn<4>(g + row, gm);
        ldn<4>(bt + row, bv);
    } else {
        for (int k = 0; k < n; ++k) {
            gm[k] = g[row + k];
            bv[k] = bt[row + k];
        }
    }
    for (int b = 0; b < B; ++b) {
        const float mean = stats[b * 2], rstd = stats[b * 2 + 1];
        float s1 = 0.f, s2 = 0.f;
        float uv[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
        const size_t i = b * CP + row;
        if (n == 4) {
            ldn<4>(u + i, uv);
            ldn<4>(ds + i, dv);
        } else {
            for (int k = 0; k < n; ++k) {
                uv[k] = u[i + k];
                dv[k] = ds[i + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (uv[k] - mean) * rstd;
            const float y = xh * gm[k] + bv[k];
            const float sg = 1.0f / (1.0f + expf(-y));
            const float dy = k < n ? dv[k] * sg * (1.0f + y * (1.0f - sg)) : 0.f;
            ag[k] += dy * xh;
            ab[k] += dy;
            const float dxh = dy * gm[k];
            dv[k] = dxh;
            s1 += dxh;
            s2 += dxh * xh;
        }
        if (n == 4) stn<4>(ds + i, dv);
        else
            for (int k = 0; k < n; ++k) ds[i + k] = dv[k];
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) { sh[0][wave] = s1; sh[1][wave] = s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float *pp = partial + ((size_t)b * (gridDim.x * gridDim.y) + (size_t)c * gridDim.x + blockIdx.x) * 2;
            pp[0] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
            pp[1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
        }
    }
    float og[4], ob[4];
    if (n == 4) {
        if (accumulate) {
            ldn<4>(dg + row, og);
            ldn<4>(dbt + row, ob);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            og[k] = (accumulate ? og[k] : 0.f) + ag[k];
            ob[k] = (accumulate ? ob[k] : 0.f) + ab[k];
        }
        stn<4>(dg + row, og);
        stn<4>(dbt + row, ob);
    } else {
        for (int k = 0; k < n; ++k) {
            dg[row + k] = (accumulate ? dg[row + k] : 0.f) + ag[k];
            dbt[row + k] = (accumulate ? dbt[row + k] : 0.f) + ab[k];
        }
    }
}

// per sample: m1 = sum(dxhat) / (16 P), m2 = sum(dxhat * xhat) / (16 P); one block of 256 threads per sample: thread t adds partials
// t, t + 256, ... in double (eight loads in flight), xor butterfly per wave, waves 0..3 in order -- a fixed order
__global__ __launch_bounds__(256) void head_ln_coef_kernel(const float *__restrict__ partial, int nblk, double count, float *__restrict__ coef)
{
    __shared__ double red[2][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *pp = partial + (size_t)b * nblk * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int t0 = 0; t0 < nblk; t0 += 256 * 8) {
        f32x2 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int t = t0 + q * 256 + tid;
            v[q] = t < nblk ? *reinterpret_cast<const f32x2 *>(pp + 2 * t) : f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            s1 += (double)v[q].x;
            s2 += (double)v[q].y;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s1 += __shfl_xor(s1, m, 64);
        s2 += __shfl_xor(s2, m, 64);
    }
    if ((tid & 63) == 0) {
        red[0][tid >> 6] = s1;
        red[1][tid >> 6] = s2;
    }
    __syncthreads();
    if (tid == 0) {
        coef[b * 2] = (float)(((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / count);
        coef[b * 2 + 1] = (float)(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / count);
    }
}

// second half: du = rstd * (dxhat - m1 - xhat * m2), in place
__global__ __launch_bounds__(256) void head_ln_bwd_b_kernel(float *ds, const float *__restrict__ u, const float *__restrict__ stats,
                                                            const float *__restrict__ coef, int P)
{
    const int bc = blockIdx.y, b = bc / HC;                  // grid (chunks, B * 16): one channel plane per blockIdx.y, pixels in quads
    const float mean = stats[b * 2], rstd = stats[b * 2 + 1], m1 = coef[b * 2], m2 = coef[b * 2 + 1];
    const float *up = u + (size_t)bc * P;
    float *dp = ds + (size_t)bc * P;
    auto body = [&](auto n_tag, int p) {
        constexpr int N = decltype(n_tag)::value;
        float uv[N], dv[N];
        ldn<N>(up + p, uv);
        ldn<N>(dp + p, dv);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float xh = (uv[k] - mean) * rstd;
            dv[k] = rstd * (dv[k] - m1 - xh * m2);
        }
        stn<N>(dp + p, dv);
    };
    URNN_PLANE_WALK(P, body);
}

hipError_t urnn_train_head_save(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *stats, int B,
                                int P, float *save, hipStream_t st)
{
    hipLaunchKernelGGL(head_train_save_kernel, dim3((P + 255) / 256, B), dim3(256), 0, st, feat, conv_w, ln_w, ln_b, stats, B, P, save);
    return hipGetLastError();
}

hipError_t urnn_train_head_pred_bwd(const float *dout, const float *cls, const float *reg, const float *reg_w, float thr, float slope,
                                    int B, int P, float *draw, float *ds, hipStream_t st)
{
    hipLaunchKernelGGL(head_pred_bwd_kernel, plane_grid(P, B), dim3(256), 0, st, dout, cls, reg, reg_w, thr, slope, P, draw, ds);
    return hipGetLastError();
}

// blocks per channel plane of head_ln_bwd_a_kernel (1024 pixels each); its partial buffer holds B x 16 x chunks pairs
int urnn_train_head_ln_chunks(int P) { return (P + 1023) / 1024; }

hipError_t urnn_train_head_ln_bwd(float *ds, const float *u, const float *g, const float *bt, const float *stats, int B, int P, float *dg,
                                  float *dbt, int accumulate, float *partial, float *coef, hipStream_t st)
{
    const int nchunk = urnn_train_head_ln_chunks(P);
    hipLaunchKernelGGL(head_ln_bwd_a_kernel, dim3(nchunk, HC), dim3(256), 0, st, ds, u, g, bt, stats, B, P, dg, dbt, accumulate, partial);
    hipLaunchKernelGGL(head_ln_coef_kernel, dim3(B), dim3(256), 0, st, partial, nchunk * HC, (double)HC * (double)P, coef);
    hipLaunchKernelGGL(head_ln_bwd_b_kernel, plane_grid(P, B * HC), dim3(256), 0, st, ds, u, stats, coef, P);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Training loss  FocalBCE_and_WMSE (losses.py:44-249) as the SWP loop applies it (main.py:489-539): reg = network output,
// cls = [reg >= cls_thred] (non-differentiable), wet = target > 0:
//   loss_reg = 20 * mean((reg - t)^2 | wet) + mean((reg - t)^2 | dry);   loss_cls = mean focal BCE of the 0/1 "probabilities";
//   loss = loss_reg + 0.1 * loss_cls;   d loss / d reg = 40 (reg - t) / n_wet on wet cells, 2 (reg - t) / n_dry on dry ones.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_partial_kernel(const float *__restrict__ reg, const float *__restrict__ tgt, float thr, long n,
                                                           float *__restrict__ partial)
{
    __shared__ float sh[5][4];
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // n_wet, sse_wet, sse_dry, n(p=0,y=1), n(p=1,y=0)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float r = reg[i], t = tgt[i], e = r - t;
        const bool wet = t > 0.f, pos = r >= thr;
        if (wet) { v[0] += 1.f; v[1] += e * e; } else v[2] += e * e;
        if (wet && !pos) v[3] += 1.f;
        if (!wet && pos) v[4] += 1.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float s = wave_sum(v[k]);
        if (lane == 0) sh[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < 5) partial[(size_t)blockIdx.x * 5 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// one wave: comps[5] = loss, loss_reg, wet MSE, dry MSE, loss_cls; scales[2] = 40 / n_wet, 2 / n_dry
__global__ __launch_bounds__(64) void loss_finalize_kernel(const float *__restrict__ partial, int nblk, long n, float *__restrict__ comps,
                                                           float *__restrict__ scales)
{
    const int lane = threadIdx.x;
    double s[5] = {0, 0, 0, 0, 0};
    for (int t = lane; t < nblk; t += 64)
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] += (double)partial[(size_t)t * 5 + k];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s[k] += __shfl_xor(s[k], m, 64);
    if (lane == 0) {
        const double n_wet = s[0], n_dry = (double)n - s[0];
        const double flood = s[1] / n_wet, dry = s[2] / n_dry;        // empty class -> nan, as torch's mse_loss of an empty tensor
        const double L = -log(1e-9);
        const double cls = (0.25 * L * s[3] + 0.75 * L * s[4]) / (double)n;
        const double lreg = 20.0 * flood + dry;
        comps[0] = (float)(lreg + 0.1 * cls);
        comps[1] = (float)lreg;
        comps[2] = (float)flood;
        comps[3] = (float)dry;
        comps[4] = (float)cls;
        scales[0] = (float)(40.0 / n_wet);
        scales[1] = (float)(2.0 / n_dry);
    }
}

__global__ __launch_bounds__(256) void loss_grad_kernel(const float *__restrict__ reg, const float *__restrict__ tgt, const float *__restrict__ scales,
                                                        long n, float *__restrict__ dreg)
{
    const float sw = scales[0], sd = scales[1];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float t = tgt[i];
        dreg[i] = (t > 0.f ? sw : sd) * (reg[i] - t);
    }
}

int urnn_train_loss_nblk(long n)
{
    const long b = (n + 4095) / 4096;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

hipError_t urnn_train_loss(const float *reg, const float *tgt, float thr, long n, float *partial, float *scales, float *comps, float *dreg,
                           hipStream_t st)
{
    const int nblk = urnn_train_loss_nblk(n);
    hipLaunchKernelGGL(loss_partial_kernel, dim3(nblk), dim3(256), 0, st, reg, tgt, thr, n, partial);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, partial, nblk, n, comps, scales);
    if (dreg) hipLaunchKernelGGL(loss_grad_kernel, dim3(nblk), dim3(256), 0, st, reg, tgt, scales, n, dreg);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Optimizer step on the flat parameter buffer: global gradient-norm clipping (torch.nn.utils.clip_grad_norm_, main.py:760-761)
// and Adam (torch.optim.Adam defaults, main.py:306).  Deterministic (fixed-order double sum of the per-block partials).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float *__restrict__ g, long n, float *__restrict__ partial)
{
    __shared__ float sh[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += g[i] * g[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// out[0] = clip coefficient min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0), out[1] = the norm
__global__ __launch_bounds__(64) void clip_coef_kernel(const float *__restrict__ partial, int nblk, float max_norm, float *__restrict__ out)
{
    double s = 0.0;
    for (int t = threadIdx.x; t < nblk; t += 64) s += (double)partial[t];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (threadIdx.x == 0) {
        const double norm = sqrt(s);
        double c = max_norm > 0.f ? (double)max_norm / (norm + 1e-6) : 1.0;
        out[0] = (float)(c < 1.0 ? c : 1.0);
        out[1] = (float)norm;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                                   long n, float lr, float b1, float b2, float eps, int step_host, const int *__restrict__ step_dev,
                                                   const float *__restrict__ coef)
{
    // step_dev (device counter, for hipGraph replay) overrides step_host; every thread derives the same bias corrections
    const int step = step_dev ? *step_dev : step_host;
    const float bc1 = 1.f - powf(b1, (float)step), bc2_sqrt = sqrtf(1.f - powf(b2, (float)step));
    const float c = coef ? coef[0] : 1.f;
    const float stepsz = lr / bc1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * c;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= stepsz * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

hipError_t urnn_train_clip_coef(const float *g, long n, float max_norm, float *partial, float *out, hipStream_t st)
{
    const int nblk = urnn_train_loss_nblk(n);
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nblk), dim3(256), 0, st, g, n, partial);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, st, partial, nblk, max_norm, out);
    return hipGetLastError();
}

hipError_t urnn_train_adam(float *p, const float *g, float *m, float *v, long n, float lr, float b1, float b2, float eps, int step,
                           const int *step_dev, const float *coef, hipStream_t st)
{
    const int nblk = urnn_train_loss_nblk(n);
    hipLaunchKernelGGL(adam_kernel, dim3(nblk), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, step, step_dev, coef);
    return hipGetLastError();
}
