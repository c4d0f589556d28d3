// Tail of the three-space adversarial model of drl_classifier_en.py (config 4) and its backward:
//   pooler -> six latent heads (:227-232) -> three reparameterised samples (:238-240, :417-423) -> content
//   discriminator on the emotion and the cause sample (softmax over the vocabulary, BCE + entropy, :246-254),
//   content classifier (:256), four one-logit discriminators with entropies (:258-278), emotion / cause /
//   pair classifiers (:268, :280, :283), three annealed KL terms (:285-303), decoder reconstruction (:306-307),
//   weighted sum (:325-332).  Everything is fp32.
// All losses are roots of the graph, so every gradient is produced here for a unit upstream gradient and the six
// backward calls of the training step (:919-939) only rescale / accumulate them.
// The vocabulary-wide heads (K = 24, 24, 384, 432 inputs -> V outputs) go through an fp32 tiled GEMM: logits [B, V]
// are written once, a row kernel turns them into d(loss)/d(logits) in place, and two more GEMMs give the weight and
// input gradients.  HBM-bound on the [V, K] weight images (41 MB for the decoder at V = 23 771).
#include "carel_hip_internal.h"
#include "rowvec_device.h"

namespace carel {

// ------------------------------------------------------------------------------------------
// fp32 GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, an fmaf chain per output).
// 64 x 64 tile per workgroup, four waves of one 32 x 32 accumulator each, 32-deep steps staged through LDS
// (k-major images, so lane l reads A[k = 2s + (l >> 5)][m = l & 31] as one conflict-free b32), the next step's
// global loads in flight while the current step's 16 MFMAs run (64-deep steps measured 8 % slower).  All edges are guarded (any M, N, K).
//   TA = false: A is [M, K] row-major;  true: A is [K, M].   TB = false: B is [N, K];  true: B is [K, N].
// blockIdx.z splits the reduction into chunks of `kchunk`; split z writes to C + z * c_split_stride.
// ------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SG_BK = 32, SG_LD = 65, SG_NR = SG_BK * 64 / 256;     // elements per thread and operand per step

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void sgemm64_kernel(const float* __restrict__ A, long lda, const float* __restrict__ Bm, long ldb,
                                                      float* __restrict__ C, long ldc, int M, int N, int K, const float* __restrict__ bias,
                                                      int accumulate, int kchunk, long c_split_stride) {
  __shared__ float As[SG_BK][SG_LD];
  __shared__ float Bs[SG_BK][SG_LD];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int kb = blockIdx.z * kchunk, ke = min(K, kb + kchunk);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float ra[SG_NR], rb[SG_NR];
  // loop-invariant element offsets; rows past M / N are CLAMPED, not skipped (their products only reach outputs that are never
  // stored), so the steady-state loads carry no per-element guard -- only the last, partial K step is guarded (zeros matter there).
  // (With per-element guards the 16 address computations + exec-mask branches took ~0.6 us of issue time per step, in program order
  // AHEAD of the MFMA chain: the load phase added to the MFMA phase instead of hiding under it.)
  long offa[SG_NR], offb[SG_NR];
#pragma unroll
  for (int i = 0; i < SG_NR; ++i) {
    const int e = t + i * 256;
    if (TA) offa[i] = (long)(e >> 6) * lda + min(m0 + (e & 63), M - 1); else offa[i] = (long)min(m0 + e / SG_BK, M - 1) * lda + e % SG_BK;
    if (TB) offb[i] = (long)(e >> 6) * ldb + min(n0 + (e & 63), N - 1); else offb[i] = (long)min(n0 + e / SG_BK, N - 1) * ldb + e % SG_BK;
  }
  auto gload = [&](int k0) {
    const float* Ak = A + (TA ? (long)k0 * lda : (long)k0);
    const float* Bk = Bm + (TB ? (long)k0 * ldb : (long)k0);
    if (k0 + SG_BK <= ke) {
#pragma unroll
      for (int i = 0; i < SG_NR; ++i) { ra[i] = Ak[offa[i]]; rb[i] = Bk[offb[i]]; }
    } else {
#pragma unroll
      for (int i = 0; i < SG_NR; ++i) {
        const int e = t + i * 256;
        const int ka = TA ? (e >> 6) : (e % SG_BK), kb2 = TB ? (e >> 6) : (e % SG_BK);
        ra[i] = (k0 + ka < ke) ? Ak[offa[i]] : 0.f;
        rb[i] = (k0 + kb2 < ke) ? Bk[offb[i]] : 0.f;
      }
    }
  };
  gload(kb);
  for (int k0 = kb; k0 < ke; k0 += SG_BK) {
#pragma unroll
    for (int i = 0; i < SG_NR; ++i) {
      const int e = t + i * 256;
      if (TA) As[e >> 6][e & 63] = ra[i]; else As[e % SG_BK][e / SG_BK] = ra[i];
      if (TB) Bs[e >> 6][e & 63] = rb[i]; else Bs[e % SG_BK][e / SG_BK] = rb[i];
    }
    __syncthreads();
    if (k0 + SG_BK < ke) gload(k0 + SG_BK);
    float av[SG_BK / 2], bv2[SG_BK / 2];          // all operand reads first, then the MFMAs back to back
#pragma unroll
    for (int s = 0; s < SG_BK / 2; ++s) {
      av[s] = As[2 * s + (l >> 5)][wm * 32 + (l & 31)];
      bv2[s] = Bs[2 * s + (l >> 5)][wn * 32 + (l & 31)];
    }
#pragma unroll
    for (int s = 0; s < SG_BK / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv2[s], acc, 0, 0, 0);
    __syncthreads();
  }
  float* Cz = C + (long)blockIdx.z * c_split_stride;
  const int gn = n0 + wn * 32 + (l & 31);
  if (gn >= N) return;
  const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    if (gm >= M) continue;
    float v = acc[r] + bv;
    if (accumulate) v += Cz[(long)gm * ldc + gn];
    Cz[(long)gm * ldc + gn] = v;
  }
}

// ------------------------------------------------------------------------------------------
// One workgroup per sample: logits row -> softmax -> BCE against the smoothed bag of words (+ entropy);
// the row is overwritten with d(loss)/d(logits) * s_bce, and (ENT) d(entropy)/d(logits) * s_ent goes to L2.
// rowstat[b] = {sum_j bce, sum_j p log(p + eps)}.
// ------------------------------------------------------------------------------------------
struct BowRowArgs {
  float* L; float* L2; const float* bow; int B, V; float ls, eps, s_bce, s_ent; float* rowstat;
};
// Three passes over a row, each spread over BR_CHUNK-wide chunks (grid = chunks x B) so that all CUs take part (the work
// is transcendental-bound: exp / log / log1p per element):
//   PASS 1: chunk maximum and sum of exp                                      -> part[b][c][0..1]
//   PASS 2: log-sum-exp from part; chunk sums of BCE, p*dBCE/dp (+ entropy, p*dEnt/dp) -> part2[b][c][0..3]
//   PASS 3: row sums from part2; writes the gradients; chunk 0 writes rowstat
// hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp) for the per-element work of the row passes; log(1 - p)
// by its series for small p (most entries have p ~ 1/V), where 1 - p would round away the information
__device__ __forceinline__ float br_log1m(float p) {
  if (p < 0.0625f) return -p * (1.f + p * (0.5f + p * (0.33333334f + p * (0.25f + p * (0.2f + p * (0.16666667f + p * 0.14285715f))))));
  return __logf(1.f - p);
}
constexpr int BR_CHUNK = 2048, BR_THREADS = 256, BR_NPT = BR_CHUNK / BR_THREADS;
template <int ENT, int PASS>
__global__ __launch_bounds__(BR_THREADS) void bow_row_kernel(BowRowArgs a, float* __restrict__ part, float* __restrict__ part2, int chunks) {
  __shared__ float red[16];
  const int c = blockIdx.x, b = blockIdx.y, t = threadIdx.x, V = a.V;
  float* L = a.L + (long)b * V;
  const float* bw = a.bow + (long)b * V;
  const int j0 = c * BR_CHUNK;
  float x[BR_NPT];
#pragma unroll
  for (int i = 0; i < BR_NPT; ++i) { const int j = j0 + t + i * BR_THREADS; x[i] = j < V ? L[j] : -INFINITY; }
  if (PASS == 1) {
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < BR_NPT; ++i) m = fmaxf(m, x[i]);
    m = block_max(m, red);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < BR_NPT; ++i) s += expf(x[i] - m);
    s = block_sum(s, red);
    if (t == 0) { part[((long)b * chunks + c) * 2] = m; part[((long)b * chunks + c) * 2 + 1] = s; }
    return;
  }
  float m = -INFINITY;            // every thread combines the (few) chunk partials itself, in fixed order
  for (int q = 0; q < chunks; ++q) m = fmaxf(m, part[((long)b * chunks + q) * 2]);
  float s = 0.f;
  for (int q = 0; q < chunks; ++q) s += part[((long)b * chunks + q) * 2 + 1] * expf(part[((long)b * chunks + q) * 2] - m);
  const float lse = m + logf(s);
  const float t0 = a.ls / (float)V, t1 = 1.f - a.ls;
  float tg[BR_NPT];
#pragma unroll
  for (int i = 0; i < BR_NPT; ++i) { const int j = j0 + t + i * BR_THREADS; tg[i] = j < V ? bw[j] * t1 + t0 : 0.f; }
  if (PASS == 2) {
    float le = 0.f, db = 0.f, en = 0.f, de = 0.f;
#pragma unroll
    for (int i = 0; i < BR_NPT; ++i) {
      if (j0 + t + i * BR_THREADS < V) {
        const float lp = x[i] - lse, p = __expf(lp);
        le += -(tg[i] * fmaxf(lp, -100.f) + (1.f - tg[i]) * fmaxf(br_log1m(p), -100.f));
        db += p * ((p - tg[i]) * __frcp_rn(fmaxf((1.f - p) * p, 1e-12f)));
        if (ENT) { const float lg = __logf(p + a.eps); en += p * lg; de += p * (lg + p * __frcp_rn(p + a.eps)); }
      }
    }
    le = block_sum(le, red); db = block_sum(db, red);
    if (ENT) { en = block_sum(en, red); de = block_sum(de, red); }
    if (t == 0) { float* o = part2 + ((long)b * chunks + c) * 4; o[0] = le; o[1] = db; o[2] = en; o[3] = de; }
    return;
  }
  float le = 0.f, db = 0.f, en = 0.f, de = 0.f;
  for (int q = 0; q < chunks; ++q) {
    const float* o = part2 + ((long)b * chunks + q) * 4;
    le += o[0]; db += o[1]; en += o[2]; de += o[3];
  }
#pragma unroll
  for (int i = 0; i < BR_NPT; ++i) {
    const int j = j0 + t + i * BR_THREADS;
    if (j < V) {
      const float p = __expf(x[i] - lse);
      const float gp = (p - tg[i]) * __frcp_rn(fmaxf((1.f - p) * p, 1e-12f));
      if (ENT) { const float lg = __logf(p + a.eps); a.L2[(long)b * V + j] = p * ((lg + p * __frcp_rn(p + a.eps)) - de) * a.s_ent; }
      L[j] = p * (gp - db) * a.s_bce;
    }
  }
  if (c == 0 && t == 0) { a.rowstat[b * 2] = le; a.rowstat[b * 2 + 1] = en; }
}

// out[j] (+)= sum_b X[b][j]
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float* __restrict__ X, int B, int V, float* __restrict__ out, int accumulate) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= V) return;
  float s = 0.f;
  int b = 0;
  for (; b + 8 <= B; b += 8) {        // eight independent loads in flight; summation order stays b = 0, 1, 2, ...
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = X[(long)(b + u) * V + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; b < B; ++b) s += X[(long)b * V + j];
  out[j] = accumulate ? out[j] + s : s;
}

// ------------------------------------------------------------------------------------------
// z = [emotion | cause | content] samples and the ten dropped-out copies the heads read
// ------------------------------------------------------------------------------------------
constexpr int EN_NSEG = 10;
struct EnSegs { int src[EN_NSEG]; int K[EN_NSEG]; long dst[EN_NSEG]; Dropout d[EN_NSEG]; };

__device__ __forceinline__ void en_lat_index(int k, int D, int Cd, int& mu, int& lv) {
  if (k < D) { mu = 2 * Cd + k; lv = 2 * Cd + D + k; }
  else if (k < 2 * D) { mu = 2 * Cd + 2 * D + (k - D); lv = 2 * Cd + 3 * D + (k - D); }
  else { mu = k - 2 * D; lv = Cd + (k - 2 * D); }
}

// One workgroup per sample: z, the dropped-out copies, the sample's three KL sums (klrow[b][3], unweighted) and the seven
// one-logit heads' pre-activations lg[h][b] (head h reads segment hseg[h]).
struct EnHeadIn { int hseg[7]; const float* w[7]; const float* b[7]; };
__global__ __launch_bounds__(512) void en_sample_kernel(const float* __restrict__ lat, const float* __restrict__ eps, int B, int D, int Cd,
                                                        float* __restrict__ z, float* __restrict__ xd, EnSegs sg, EnHeadIn hi,
                                                        float* __restrict__ lg, float* __restrict__ klrow) {
  extern __shared__ float zrow[];           // [2D + Cd] + 16
  const int ZW = 2 * D + Cd, LW = 2 * Cd + 4 * D;
  float* red = zrow + ZW;
  const int b = blockIdx.x, t = threadIdx.x, nthr = blockDim.x;
  float kl[3] = {0.f, 0.f, 0.f};
  for (int k = t; k < ZW; k += nthr) {
    int mu, lv;
    en_lat_index(k, D, Cd, mu, lv);
    const float m = lat[(long)b * LW + mu], l = lat[(long)b * LW + lv], ex = expf(l);
    const float zv = m + eps[k] * ex;
    z[(long)b * ZW + k] = zv;
    zrow[k] = zv;
    const float v = -0.5f * (1.f + l - ex - m * m);          // :615-624
    if (k < D) kl[0] += v; else if (k < 2 * D) kl[1] += v; else kl[2] += v;
#pragma unroll
    for (int s = 0; s < EN_NSEG; ++s) {
      const int kk = k - sg.src[s];
      if (kk >= 0 && kk < sg.K[s]) xd[sg.dst[s] + (long)b * sg.K[s] + kk] = zv * dropout_mult(sg.d[s], (uint32_t)(b * sg.K[s] + kk));
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) { kl[i] = block_sum(kl[i], red); if (t == 0) klrow[b * 3 + i] = kl[i]; }      // also publishes zrow
#pragma unroll
  for (int h = 0; h < 7; ++h) {
    const int sgi = hi.hseg[h], K = sg.K[sgi], src = sg.src[sgi];
    float acc = 0.f;
    for (int kk = t; kk < K; kk += nthr) acc = fmaf(zrow[src + kk] * dropout_mult(sg.d[sgi], (uint32_t)(b * K + kk)), hi.w[h][kk], acc);
    acc = block_sum(acc, red);
    if (t == 0) lg[h * B + b] = acc + hi.b[h][0];
  }
}

// ------------------------------------------------------------------------------------------
// The seven one-logit heads, the KL terms and the totals: one workgroup.
//   h0 emotion_disc(content)  h1 cause_disc(content)  h2 ec_disc(cause)  h3 ce_disc(emotion)
//   h4 emotion_classifier(emotion)  h5 cause_classifier(cause)  h6 pair_classifier([emotion, cause])
// ------------------------------------------------------------------------------------------
struct EnHeads {
  int B, D, Cd, V;
  const float* xd; long xoff[7]; int K[7];
  const float* w[7]; const float* b[7];
  const float* emo; const float* cau; const float* pair;
  const float* label_sum_override; float n_override;      // data parallel: pos_weight of the global batch
  const float* lg_in;                // [7][B] pre-activations (en_sample_kernel)
  const float* klrow;                // [B][3]
  float w_con_adv, w_ec_adv, w_ecce_adv, w_ec_mul, w_con_mul, w_pair, kl_w_ec, kl_w_con, ls, eps;
  Dropout d_emul, d_caumul, d_pair;
  const float* rowstat;              // [4][B][2]: content_disc(emotion), content_disc(cause), content_classifier, decoder
  float* terms;
  float* gw[7]; float* gb[7];
  float* gew[4]; float* geb[4];
  float* dz_heads;                   // [B][2D]
};

__global__ __launch_bounds__(1024) void en_heads_kernel(EnHeads a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int B = a.B, D = a.D;
  float* red = sm;                   // 16
  float* lg = sm + 16;               // [7][B]
  float* dl = lg + 7 * B;            // [7][B]
  float* de = dl + 7 * B;            // [4][B]
  const int t = threadIdx.x, nthr = blockDim.x;
  for (int e = t; e < 7 * B; e += nthr) lg[e] = a.lg_in[e];
  float ys = 0.f;
  for (int b = t; b < B; b += nthr) ys += a.pair[b];
  ys = block_sum(ys, red);           // also publishes lg
  if (a.label_sum_override) ys = a.label_sum_override[0];
  const float pw = ((a.label_sum_override ? a.n_override : (float)B) - ys) / ys;
  const float invB = 1.0f / (float)B;
  float loss[7], ent[4];
#pragma unroll
  for (int h = 0; h < 7; ++h) loss[h] = 0.f;
#pragma unroll
  for (int h = 0; h < 4; ++h) ent[h] = 0.f;
  const float went[4] = {a.w_ec_adv, a.w_ec_adv, a.w_ecce_adv, a.w_ecce_adv};
  for (int b = t; b < B; b += nthr) {
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      const float y = (h == 0 || h == 2 || h == 4) ? a.emo[b] : a.cau[b];
      const float tg = y * (1.f - a.ls) + a.ls;                 // label_smoothing / ec_num_class with one class
      const float p = 1.0f / (1.0f + expf(-lg[h * B + b]));
      loss[h] += -(tg * fmaxf(logf(p), -100.f) + (1.f - tg) * fmaxf(logf(1.f - p), -100.f));
      const float gp = (p - tg) / fmaxf((1.f - p) * p, 1e-12f);
      const float dp = p * (1.f - p);
      dl[h * B + b] = gp * dp * invB * (h >= 4 ? a.w_ec_mul : 1.f);
      if (h < 4) {
        const float l = logf(p + a.eps);
        ent[h] += p * l;
        de[h * B + b] = (l + p / (p + a.eps)) * dp * invB * went[h];
      }
    }
    const float x = lg[6 * B + b];
    const float tg = a.pair[b] * (1.f - a.ls) + a.ls;
    const float lw = (pw - 1.f) * tg + 1.f;
    loss[6] += (1.f - tg) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f));
    const float sg = 1.0f / (1.0f + expf(-x));
    dl[6 * B + b] = ((1.f - tg) - lw * (1.f - sg)) * invB * a.w_pair;
  }
#pragma unroll
  for (int h = 0; h < 7; ++h) loss[h] = block_sum(loss[h], red) * invB;
#pragma unroll
  for (int h = 0; h < 4; ++h) ent[h] = block_sum(ent[h], red) * invB;
  float kl[3] = {0.f, 0.f, 0.f};          // per-sample sums from en_sample_kernel (:615-624)
  for (int b = t; b < B; b += nthr) { kl[0] += a.klrow[b * 3]; kl[1] += a.klrow[b * 3 + 1]; kl[2] += a.klrow[b * 3 + 2]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) kl[i] = block_sum(kl[i], red) * invB;
  float rs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // cd_e loss, cd_e ent, cd_c loss, cd_c ent, cmul loss, dec loss
  for (int b = t; b < B; b += nthr) {
    rs[0] += a.rowstat[(0 * B + b) * 2]; rs[1] += a.rowstat[(0 * B + b) * 2 + 1];
    rs[2] += a.rowstat[(1 * B + b) * 2]; rs[3] += a.rowstat[(1 * B + b) * 2 + 1];
    rs[4] += a.rowstat[(2 * B + b) * 2]; rs[5] += a.rowstat[(3 * B + b) * 2];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) rs[i] = block_sum(rs[i], red);
  if (t == 0) {
    const float inv_bv = invB / (float)a.V;
    float* T = a.terms;
    const float cd_e = rs[0] * inv_bv, cd_c = rs[2] * inv_bv, cent_e = rs[1] * invB, cent_c = rs[3] * invB;
    const float con_mul = rs[4] * inv_bv, rec = rs[5] * inv_bv;
    const float kl_e = kl[0] * a.kl_w_ec, kl_c = kl[1] * a.kl_w_ec, kl_con = kl[2] * a.kl_w_con;
    T[0] = cd_e; T[1] = cd_c; T[2] = loss[0]; T[3] = loss[2]; T[4] = loss[1]; T[5] = loss[3];
    T[7] = cent_e; T[8] = cent_c; T[9] = ent[0]; T[10] = ent[1]; T[11] = ent[2]; T[12] = ent[3];
    T[13] = loss[4]; T[14] = loss[5]; T[15] = con_mul; T[16] = loss[6]; T[17] = kl_e; T[18] = kl_c; T[19] = kl_con; T[20] = rec;
    T[6] = a.w_con_adv * (cent_e + cent_c) + a.w_ec_adv * (ent[0] + ent[1]) + a.w_ecce_adv * (ent[2] + ent[3]) +
           a.w_ec_mul * (loss[4] + loss[5]) + a.w_con_mul * con_mul + a.w_pair * loss[6] + kl_e + kl_c + kl_con + rec;
  }
  __syncthreads();
  // weight gradients: one thread per (head, input column); bias gradients: one thread per head
  int tot = 0;
#pragma unroll
  for (int h = 0; h < 7; ++h) tot += a.K[h];
  for (int idx = t; idx < tot + 7; idx += nthr) {
    if (idx < tot) {
      int h = 0, k = idx;
      while (k >= a.K[h]) { k -= a.K[h]; ++h; }
      const float* x = a.xd + a.xoff[h] + k;
      const int Kh = a.K[h];
      float s = 0.f, se = 0.f;
      const float* deh = de + (h < 4 ? h : 0) * B;
      int b = 0;
      for (; b + 8 <= B; b += 8) {          // eight loads in flight; accumulation order stays b = 0, 1, 2, ...
        float xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = x[(long)(b + u) * Kh];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s = fmaf(dl[h * B + b + u], xv[u], s); se = fmaf(deh[b + u], xv[u], se); }
      }
      for (; b < B; ++b) {
        const float xv = x[(long)b * Kh];
        s = fmaf(dl[h * B + b], xv, s);
        se = fmaf(deh[b], xv, se);
      }
      a.gw[h][k] = s;
      if (h < 4) a.gew[h][k] = se;
    } else {
      const int h = idx - tot;
      float s = 0.f, se = 0.f;
      for (int b = 0; b < B; ++b) { s += dl[h * B + b]; if (h < 4) se += de[h * B + b]; }
      a.gb[h][0] = s;
      if (h < 4) a.geb[h][0] = se;
    }
  }
  // d vae / d [z_e, z_c] through the emotion / cause / pair classifiers
  for (int e = t; e < B * 2 * D; e += nthr) {
    const int b = e / (2 * D), k = e - b * 2 * D;
    float v = dl[6 * B + b] * a.w[6][k] * dropout_mult(a.d_pair, (uint32_t)(b * 2 * D + k));
    if (k < D) v += dl[4 * B + b] * a.w[4][k] * dropout_mult(a.d_emul, (uint32_t)(b * D + k));
    else v += dl[5 * B + b] * a.w[5][k - D] * dropout_mult(a.d_caumul, (uint32_t)(b * D + (k - D)));
    a.dz_heads[e] = v;
  }
}

// d vae / d lat: classifier + decoder gradients through the samples, plus the direct KL part
__global__ __launch_bounds__(256) void en_dlat_kernel(const float* __restrict__ dz_dec, const float* __restrict__ dz_heads,
                                                      const float* __restrict__ dxd_cmul, Dropout d_cmul, const float* __restrict__ lat,
                                                      const float* __restrict__ eps, int B, int D, int Cd, float kl_w_ec, float kl_w_con,
                                                      float* __restrict__ dlat) {
  const int ZW = 2 * D + Cd, LW = 2 * Cd + 4 * D;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * ZW) return;
  const int b = e / ZW, k = e - b * ZW;
  float g = dz_dec[e];
  if (k < 2 * D) g += dz_heads[b * 2 * D + k];
  else g += dxd_cmul[(long)b * Cd + (k - 2 * D)] * dropout_mult(d_cmul, (uint32_t)(b * Cd + (k - 2 * D)));
  int mu, lv;
  en_lat_index(k, D, Cd, mu, lv);
  const float klw = (k < 2 * D ? kl_w_ec : kl_w_con) / (float)B;
  const float m = lat[(long)b * LW + mu], ex = expf(lat[(long)b * LW + lv]);
  dlat[(long)b * LW + mu] = g + klw * m;
  dlat[(long)b * LW + lv] = g * eps[k] * ex + klw * 0.5f * (ex - 1.f);
}

__global__ __launch_bounds__(256) void en_pair_logits_kernel(const float* __restrict__ lat, int stride, int emo_off, int cau_off,
                                                             const float* __restrict__ eps_e, const float* __restrict__ eps_c,
                                                             const float* __restrict__ w, const float* __restrict__ bias, int B, int D,
                                                             float* __restrict__ out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float* row = lat + (long)b * stride;
  float s = bias[0];
  for (int k = 0; k < D; ++k) {
    s = fmaf(w[k], row[emo_off + k] + eps_e[k] * expf(row[emo_off + D + k]), s);
    s = fmaf(w[D + k], row[cau_off + k] + eps_c[k] * expf(row[cau_off + D + k]), s);
  }
  out[b] = s;
}

__global__ void axpy_kernel(float* __restrict__ dst, const float* __restrict__ src, long n, const float* __restrict__ scale, int accumulate) {
  const float s = scale ? scale[0] : 1.f;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = (accumulate ? dst[i] : 0.f) + s * src[i];
}

}  // namespace carel

using namespace carel;

static size_t en_align(size_t x) { return (x + 63) & ~(size_t)63; }
constexpr int EN_MAX_SPLITS = 64;

struct EnWork {
  float* xd; long xoff[EN_NSEG]; float* L1; float* L2; float* rowstat; float* dxd_cmul; float* dz_dec; float* dz_heads; float* parts;
  float* dlat; float* dpooled; float* dpre; float* dcls; float* dgpart; float* rowpart; float* rowpart2; float* lg; float* klrow;
  size_t total;
};
// inputs of the heads, in the order of the ten nn.Dropout calls (sites 110..119)
static const int kSegSource[EN_NSEG] = {0, 1, 2, 2, 1, 0, 2, 0, 1, 3};   // 0 emotion, 1 cause, 2 content, 3 [emotion, cause]
static void en_seg(int s, int D, int Cd, int& src, int& K) {
  switch (kSegSource[s]) {
    case 0: src = 0; K = D; break;
    case 1: src = D; K = D; break;
    case 2: src = 2 * D; K = Cd; break;
    default: src = 0; K = 2 * D; break;
  }
}
static int en_splits(int V) { int s = (V + 255) / 256; return s > EN_MAX_SPLITS ? EN_MAX_SPLITS : (s < 1 ? 1 : s); }

static EnWork en_carve(float* base, int B, int D, int Cd, int V) {
  EnWork w; size_t o = 0;
  auto take = [&](size_t n) { float* p = base ? base + o : nullptr; o += en_align(n); return p; };
  size_t xd_total = 0;
  for (int s = 0; s < EN_NSEG; ++s) { int src, K; en_seg(s, D, Cd, src, K); w.xoff[s] = (long)xd_total; xd_total += en_align((size_t)B * K); }
  const int ZW = 2 * D + Cd, LW = 2 * Cd + 4 * D;
  w.xd = take(xd_total);
  w.L1 = take((size_t)B * V); w.L2 = take((size_t)B * V);
  w.rowstat = take((size_t)4 * B * 2); w.dxd_cmul = take((size_t)B * Cd); w.dz_dec = take((size_t)B * ZW); w.dz_heads = take((size_t)B * 2 * D);
  w.parts = take((size_t)en_splits(V) * B * ZW);
  w.dlat = take((size_t)B * LW); w.dpooled = take((size_t)B * TH); w.dpre = take((size_t)B * TH); w.dcls = take((size_t)B * TH);
  const int hc = (2 * Cd + DG_CHUNK - 1) / DG_CHUNK + (4 * D + DG_CHUNK - 1) / DG_CHUNK, pc = (TH + DG_CHUNK - 1) / DG_CHUNK;
  w.dgpart = take((size_t)(hc > pc ? hc : pc) * B * TH);
  const size_t chunks = (size_t)(V + BR_CHUNK - 1) / BR_CHUNK;
  w.rowpart = take((size_t)B * chunks * 2); w.rowpart2 = take((size_t)B * chunks * 4); w.lg = take((size_t)7 * B); w.klrow = take((size_t)3 * B);
  w.total = o;
  return w;
}

extern "C" int64_t carel_en_tail_workspace_floats(int32_t batch, int32_t ec_dim, int32_t con_dim, int32_t bow_dim) {
  return (int64_t)en_carve(nullptr, batch, ec_dim, con_dim, bow_dim).total;
}

static int en_check(const carel_en_tail_args* a, const char* who) {
  if (!a) return set_error(CAREL_ERR_ARG, "%s: null args", who);
  if (a->hidden != TH) return set_error(CAREL_ERR_SHAPE, "%s: hidden must be %d", who, TH);
  if (a->batch < 1 || a->batch > 1024 || a->seq_len < 1) return set_error(CAREL_ERR_SHAPE, "%s: batch must be 1..1024", who);
  if (a->ec_dim < 1 || a->ec_dim > 64 || a->con_dim < 1 || a->con_dim > 1024)
    return set_error(CAREL_ERR_SHAPE, "%s: ec_dim must be <= 64 and con_dim <= 1024", who);
  if (!a->x_last_f32 || !a->pooler_w || !a->pooler_b || !a->pooled || !a->lat) return set_error(CAREL_ERR_ARG, "%s: null tensor", who);
  for (int i = 0; i < 6; ++i) if (!a->head_w[i] || !a->head_b[i]) return set_error(CAREL_ERR_ARG, "%s: null latent head", who);
  return CAREL_OK;
}

static int sgemm(const float* A, long lda, bool ta, const float* Bm, long ldb, bool tb, float* C, long ldc, int M, int N, int K,
                 const float* bias, int accumulate, int splits, long c_split_stride, hipStream_t stream) {
  if (M < 1 || N < 1 || K < 1 || splits < 1) return set_error(CAREL_ERR_SHAPE, "carel_sgemm_f32: bad shape");
  int kchunk = (K + splits - 1) / splits;
  kchunk = (kchunk + SG_BK - 1) & ~(SG_BK - 1);
  const int nz = (K + kchunk - 1) / kchunk;
  if (nz != splits && splits > 1) {       // fewer non-empty chunks than requested: the caller sums `splits` slabs, so clear the tail
    (void)hipMemsetAsync(C + (long)nz * c_split_stride, 0, sizeof(float) * (size_t)(splits - nz) * (size_t)c_split_stride, stream);
  }
  dim3 grid((M + 63) / 64, (N + 63) / 64, nz);
  if (grid.y > 65535 || grid.z > 65535) return set_error(CAREL_ERR_SHAPE, "carel_sgemm_f32: N too large");
#define SG(TA, TB) hipLaunchKernelGGL((sgemm64_kernel<TA, TB>), grid, dim3(256), 0, stream, A, lda, Bm, ldb, C, ldc, M, N, K, bias, accumulate, kchunk, c_split_stride)
  if (!ta && !tb) SG(false, false); else if (!ta && tb) SG(false, true); else if (ta && !tb) SG(true, false); else SG(true, true);
#undef SG
  return check_launch("sgemm64_kernel");
}

extern "C" int carel_sgemm_f32(const void* A, int64_t lda, int32_t ta, const void* B, int64_t ldb, int32_t tb, void* C, int64_t ldc,
                               int32_t M, int32_t N, int32_t K, const void* bias, int32_t accumulate, int32_t splits, int64_t c_split_stride,
                               void* stream) {
  if (!A || !B || !C) return set_error(CAREL_ERR_ARG, "carel_sgemm_f32: null tensor");
  if (splits > 1 && (bias || accumulate)) return set_error(CAREL_ERR_ARG, "carel_sgemm_f32: split results take neither bias nor accumulate");
  return sgemm((const float*)A, (long)lda, ta != 0, (const float*)B, (long)ldb, tb != 0, (float*)C, (long)ldc, M, N, K, (const float*)bias,
               accumulate, splits < 1 ? 1 : splits, (long)c_split_stride, (hipStream_t)stream);
}

extern "C" int carel_axpy_f32(void* dst, const void* src, int64_t n, const void* scale_dev, int32_t accumulate, void* stream) {
  if (!dst || !src || n <= 0) return set_error(CAREL_ERR_ARG, "carel_axpy_f32: bad arguments");
  long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float*)dst, (const float*)src, (long)n,
                     (const float*)scale_dev, accumulate);
  return check_launch("axpy_kernel");
}

extern "C" int carel_en_pair_logits(const void* lat, int32_t lat_stride, int32_t emo_off, int32_t cau_off, const void* eps_e, const void* eps_c,
                                    const void* pair_w, const void* pair_b, int32_t batch, int32_t ec_dim, void* logits, void* stream) {
  if (!lat || !eps_e || !eps_c || !pair_w || !pair_b || !logits || batch < 1) return set_error(CAREL_ERR_ARG, "carel_en_pair_logits: bad arguments");
  hipLaunchKernelGGL(en_pair_logits_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)lat, lat_stride, emo_off,
                     cau_off, (const float*)eps_e, (const float*)eps_c, (const float*)pair_w, (const float*)pair_b, batch, ec_dim, (float*)logits);
  return check_launch("en_pair_logits_kernel");
}

extern "C" int carel_en_tail_latents(const carel_en_tail_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = en_check(a, "carel_en_tail_latents");
  if (rc) return rc;
  const int B = a->batch, D = a->ec_dim, Cd = a->con_dim, LW = 2 * Cd + 4 * D;
  auto groups = [&](int ncols) { int g = (4096 + ncols - 1) / ncols; const int mx = (B + 3) / 4; g = g > mx ? mx : g; return g < 1 ? 1 : g; };
  PtrSet4 pp; for (int i = 0; i < 4; ++i) { pp.w[i] = nullptr; pp.b[i] = nullptr; }
  pp.w[0] = (const float*)a->pooler_w; pp.b[0] = (const float*)a->pooler_b;
  hipLaunchKernelGGL(rowvec_linear_kernel<1>, dim3(TH / 4, groups(TH)), dim3(256), 0, stream, (const float*)a->x_last_f32, (long)a->seq_len * TH,
                     (const int*)a->cls_rows, B, TH, TH, pp, (float*)a->pooled, (long)TH);
  PtrSet4 cp; for (int i = 0; i < 4; ++i) { cp.w[i] = nullptr; cp.b[i] = nullptr; }
  cp.w[0] = (const float*)a->head_w[0]; cp.b[0] = (const float*)a->head_b[0]; cp.w[1] = (const float*)a->head_w[1]; cp.b[1] = (const float*)a->head_b[1];
  hipLaunchKernelGGL(rowvec_linear_kernel<0>, dim3((2 * Cd + 3) / 4, groups(2 * Cd)), dim3(256), 0, stream, (const float*)a->pooled, (long)TH,
                     (const int*)nullptr, B, 2 * Cd, Cd, cp, (float*)a->lat, (long)LW);
  PtrSet4 hp; for (int i = 0; i < 4; ++i) { hp.w[i] = (const float*)a->head_w[2 + i]; hp.b[i] = (const float*)a->head_b[2 + i]; }
  hipLaunchKernelGGL(rowvec_linear_kernel<0>, dim3((4 * D + 3) / 4, groups(4 * D)), dim3(256), 0, stream, (const float*)a->pooled, (long)TH,
                     (const int*)nullptr, B, 4 * D, D, hp, (float*)a->lat + 2 * Cd, (long)LW);
  return check_launch("en tail latents");
}

// one vocabulary-wide head: logits -> row kernel -> weight / bias / input gradients
struct BowHead {
  const float* x; int K; const float* w; const float* b;
  float s_bce, s_ent; bool ent;
  float* gw; float* gb; int acc_g;             // gradient of the BCE part
  float* gew; float* geb; int acc_ge;          // gradient of the entropy part (ent only)
  float* dx;                                   // [B, K] or null
  float* rowstat;
};
static int bow_head(const carel_en_tail_args* a, const EnWork& w, const BowHead& h, hipStream_t stream) {
  const int B = a->batch, V = a->bow_dim;
  int rc = sgemm(h.x, h.K, false, h.w, h.K, false, w.L1, V, B, V, h.K, h.b, 0, 1, 0, stream);
  if (rc) return rc;
  BowRowArgs r; r.L = w.L1; r.L2 = w.L2; r.bow = (const float*)a->bow; r.B = B; r.V = V; r.ls = a->label_smoothing; r.eps = a->epsilon;
  r.s_bce = h.s_bce; r.s_ent = h.s_ent; r.rowstat = h.rowstat;
  const int chunks = (V + BR_CHUNK - 1) / BR_CHUNK;
  const dim3 rg(chunks, B);
#define BOW_ROW(PASS)                                                                                                         \
  do {                                                                                                                        \
    if (h.ent) hipLaunchKernelGGL((bow_row_kernel<1, PASS>), rg, dim3(BR_THREADS), 0, stream, r, w.rowpart, w.rowpart2, chunks);  \
    else hipLaunchKernelGGL((bow_row_kernel<0, PASS>), rg, dim3(BR_THREADS), 0, stream, r, w.rowpart, w.rowpart2, chunks);        \
  } while (0)
  BOW_ROW(1); BOW_ROW(2); BOW_ROW(3);
#undef BOW_ROW
  if ((rc = check_launch("bow_row_kernel"))) return rc;
  // dW[j][k] = sum_b dL[b][j] x[b][k]
  if ((rc = sgemm(w.L1, V, true, h.x, h.K, true, h.gw, h.K, V, h.K, B, nullptr, h.acc_g, 1, 0, stream))) return rc;
  hipLaunchKernelGGL(colsum_rows_kernel, dim3((V + 255) / 256), dim3(256), 0, stream, (const float*)w.L1, B, V, h.gb, h.acc_g);
  if (h.ent) {
    if ((rc = sgemm(w.L2, V, true, h.x, h.K, true, h.gew, h.K, V, h.K, B, nullptr, h.acc_ge, 1, 0, stream))) return rc;
    hipLaunchKernelGGL(colsum_rows_kernel, dim3((V + 255) / 256), dim3(256), 0, stream, (const float*)w.L2, B, V, h.geb, h.acc_ge);
  }
  if (h.dx) {      // dx[b][k] = sum_j dL[b][j] W[j][k], reduction over the vocabulary split across workgroups
    const int sp = en_splits(V);
    const long slab = (long)B * h.K;
    if ((rc = sgemm(w.L1, V, false, h.w, h.K, true, w.parts, h.K, B, h.K, V, nullptr, 0, sp, slab, stream))) return rc;
    const long n4 = (slab + 3) / 4;
    if (slab % 4) return set_error(CAREL_ERR_SHAPE, "carel_en_tail_losses: batch * width must be a multiple of 4");
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (const float*)w.parts, h.dx, slab, sp);
  }
  return check_launch("bow head");
}

extern "C" int carel_en_tail_losses(const carel_en_tail_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = en_check(a, "carel_en_tail_losses");
  if (rc) return rc;
  const int B = a->batch, D = a->ec_dim, Cd = a->con_dim, V = a->bow_dim, ZW = 2 * D + Cd;
  if (V < 1) return set_error(CAREL_ERR_SHAPE, "carel_en_tail_losses: bow_dim must be positive");
  if ((D & 1) || (Cd & 3)) return set_error(CAREL_ERR_SHAPE, "carel_en_tail_losses: ec_dim must be even and con_dim a multiple of 4");
  if (!a->cdisc_w || !a->cdisc_b || !a->ccls_w || !a->ccls_b || !a->emo_w || !a->emo_b || !a->cau_w || !a->cau_b || !a->pair_w || !a->pair_b ||
      !a->dec_w || !a->dec_b || !a->emo_labels || !a->cau_labels || !a->pair_labels || !a->bow || !a->eps || !a->z || !a->terms || !a->work ||
      !a->d_ccls_w || !a->d_ccls_b || !a->d_emo_w || !a->d_emo_b || !a->d_cau_w || !a->d_cau_b || !a->d_pair_w || !a->d_pair_b || !a->d_dec_w ||
      !a->d_dec_b)
    return set_error(CAREL_ERR_ARG, "carel_en_tail_losses: null tensor");
  for (int i = 0; i < 3; ++i) if (!a->g_cdisc_w[i] || !a->g_cdisc_b[i]) return set_error(CAREL_ERR_ARG, "carel_en_tail_losses: null content_disc gradient");
  for (int i = 0; i < 4; ++i)
    if (!a->sdisc_w[i] || !a->sdisc_b[i] || !a->g_sdisc_w[i] || !a->g_sdisc_b[i] || !a->g_sdisc_ent_w[i] || !a->g_sdisc_ent_b[i])
      return set_error(CAREL_ERR_ARG, "carel_en_tail_losses: null discriminator tensor");
  EnWork w = en_carve((float*)a->work, B, D, Cd, V);
  EnSegs sg;
  for (int s = 0; s < EN_NSEG; ++s) {
    en_seg(s, D, Cd, sg.src[s], sg.K[s]);
    sg.dst[s] = w.xoff[s];
    sg.d[s] = make_dropout(a->drop_seed, 110u + (uint32_t)s, a->drop_p, a->drop_row_offset * (uint32_t)sg.K[s]);
  }
  const int seg_of_head[7] = {3, 6, 4, 7, 5, 8, 9};
  const void* hw[7] = {a->sdisc_w[0], a->sdisc_w[1], a->sdisc_w[2], a->sdisc_w[3], a->emo_w, a->cau_w, a->pair_w};
  const void* hb[7] = {a->sdisc_b[0], a->sdisc_b[1], a->sdisc_b[2], a->sdisc_b[3], a->emo_b, a->cau_b, a->pair_b};
  EnHeadIn hin;
  for (int i = 0; i < 7; ++i) { hin.hseg[i] = seg_of_head[i]; hin.w[i] = (const float*)hw[i]; hin.b[i] = (const float*)hb[i]; }
  hipLaunchKernelGGL(en_sample_kernel, dim3(B), dim3(512), sizeof(float) * (size_t)(ZW + 16), stream, (const float*)a->lat, (const float*)a->eps,
                     B, D, Cd, (float*)a->z, w.xd, sg, hin, w.lg, w.klrow);
  if ((rc = check_launch("en_sample_kernel"))) return rc;
  const float inv_b = 1.0f / (float)B, inv_bv = inv_b / (float)V;
  BowHead h;
  // content discriminator on the emotion sample, then on the cause sample (shared weights; the entropy image accumulates)
  for (int i = 0; i < 2; ++i) {
    h.x = w.xd + w.xoff[i]; h.K = D; h.w = (const float*)a->cdisc_w; h.b = (const float*)a->cdisc_b;
    h.s_bce = inv_bv; h.s_ent = inv_b * a->w_con_adv; h.ent = true;
    h.gw = (float*)a->g_cdisc_w[i]; h.gb = (float*)a->g_cdisc_b[i]; h.acc_g = 0;
    h.gew = (float*)a->g_cdisc_w[2]; h.geb = (float*)a->g_cdisc_b[2]; h.acc_ge = i;
    h.dx = nullptr; h.rowstat = w.rowstat + (size_t)i * B * 2;
    if ((rc = bow_head(a, w, h, stream))) return rc;
  }
  // content classifier on the content sample
  h.x = w.xd + w.xoff[2]; h.K = Cd; h.w = (const float*)a->ccls_w; h.b = (const float*)a->ccls_b;
  h.s_bce = inv_bv * a->w_con_mul; h.s_ent = 0.f; h.ent = false; h.gw = (float*)a->d_ccls_w; h.gb = (float*)a->d_ccls_b; h.acc_g = 0;
  h.gew = nullptr; h.geb = nullptr; h.acc_ge = 0; h.dx = w.dxd_cmul; h.rowstat = w.rowstat + (size_t)2 * B * 2;
  if ((rc = bow_head(a, w, h, stream))) return rc;
  // decoder on [emotion, cause, content] (no dropout)
  h.x = (const float*)a->z; h.K = ZW; h.w = (const float*)a->dec_w; h.b = (const float*)a->dec_b;
  h.s_bce = inv_bv; h.gw = (float*)a->d_dec_w; h.gb = (float*)a->d_dec_b; h.dx = w.dz_dec; h.rowstat = w.rowstat + (size_t)3 * B * 2;
  if ((rc = bow_head(a, w, h, stream))) return rc;

  EnHeads e;
  e.B = B; e.D = D; e.Cd = Cd; e.V = V; e.xd = w.xd;
  void* gw[7] = {a->g_sdisc_w[0], a->g_sdisc_w[1], a->g_sdisc_w[2], a->g_sdisc_w[3], a->d_emo_w, a->d_cau_w, a->d_pair_w};
  void* gb[7] = {a->g_sdisc_b[0], a->g_sdisc_b[1], a->g_sdisc_b[2], a->g_sdisc_b[3], a->d_emo_b, a->d_cau_b, a->d_pair_b};
  for (int i = 0; i < 7; ++i) {
    e.xoff[i] = w.xoff[seg_of_head[i]]; e.K[i] = sg.K[seg_of_head[i]];
    e.w[i] = (const float*)hw[i]; e.b[i] = (const float*)hb[i]; e.gw[i] = (float*)gw[i]; e.gb[i] = (float*)gb[i];
  }
  for (int i = 0; i < 4; ++i) { e.gew[i] = (float*)a->g_sdisc_ent_w[i]; e.geb[i] = (float*)a->g_sdisc_ent_b[i]; }
  e.emo = (const float*)a->emo_labels; e.cau = (const float*)a->cau_labels; e.pair = (const float*)a->pair_labels; e.lg_in = w.lg; e.klrow = w.klrow;
  e.label_sum_override = (const float*)a->global_label_sum; e.n_override = (float)a->global_n;
  e.w_con_adv = a->w_con_adv; e.w_ec_adv = a->w_ec_adv; e.w_ecce_adv = a->w_ecce_adv; e.w_ec_mul = a->w_ec_mul; e.w_con_mul = a->w_con_mul;
  e.w_pair = a->w_pair; e.kl_w_ec = a->kl_w_ec; e.kl_w_con = a->kl_w_con; e.ls = a->label_smoothing; e.eps = a->epsilon;
  e.d_emul = sg.d[5]; e.d_caumul = sg.d[8]; e.d_pair = sg.d[9];
  e.rowstat = w.rowstat; e.terms = (float*)a->terms; e.dz_heads = w.dz_heads;
  const size_t lds = sizeof(float) * (16 + (size_t)18 * B);
  if (lds > 64 * 1024) {
    hipError_t er = hipFuncSetAttribute((const void*)en_heads_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (er != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_en_tail_losses: hipFuncSetAttribute: %s", hipGetErrorString(er));
  }
  hipLaunchKernelGGL(en_heads_kernel, dim3(1), dim3(1024), lds, stream, e);
  if ((rc = check_launch("en_heads_kernel"))) return rc;
  hipLaunchKernelGGL(en_dlat_kernel, dim3((B * ZW + 255) / 256), dim3(256), 0, stream, (const float*)w.dz_dec, (const float*)w.dz_heads,
                     (const float*)w.dxd_cmul, sg.d[2], (const float*)a->lat, (const float*)a->eps, B, D, Cd, a->kl_w_ec, a->kl_w_con, w.dlat);
  return check_launch("en_dlat_kernel");
}

extern "C" int carel_en_tail_backward(const carel_en_tail_args* a, const void* grad_out_dev, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = en_check(a, "carel_en_tail_backward");
  if (rc) return rc;
  if (!a->work || !a->dx_last_f32 || !a->d_pooler_w || !a->d_pooler_b) return set_error(CAREL_ERR_ARG, "carel_en_tail_backward: null tensor");
  const int B = a->batch, D = a->ec_dim, Cd = a->con_dim, LW = 2 * Cd + 4 * D;
  EnWork w = en_carve((float*)a->work, B, D, Cd, a->bow_dim);
  if (grad_out_dev)
    hipLaunchKernelGGL(scale_inplace_kernel, dim3((B * LW + 255) / 256), dim3(256), 0, stream, w.dlat, (long)B * LW, (const float*)grad_out_dev);
  const int hc1 = (2 * Cd + DG_CHUNK - 1) / DG_CHUNK, hc2 = (4 * D + DG_CHUNK - 1) / DG_CHUNK, pc = (TH + DG_CHUNK - 1) / DG_CHUNK;
  const long bt = (long)B * TH;
  PtrSet4 cp; for (int i = 0; i < 4; ++i) { cp.w[i] = nullptr; cp.b[i] = nullptr; }
  cp.w[0] = (const float*)a->head_w[0]; cp.w[1] = (const float*)a->head_w[1];
  hipLaunchKernelGGL(rowvec_dgrad_kernel<0>, dim3((B + 3) / 4, hc1), dim3(256), 0, stream, (const float*)w.dlat, (long)LW, B, 2 * Cd, Cd, cp,
                     (const float*)nullptr, (float*)nullptr, w.dgpart);
  PtrSet4 hp; for (int i = 0; i < 4; ++i) { hp.w[i] = (const float*)a->head_w[2 + i]; hp.b[i] = nullptr; }
  hipLaunchKernelGGL(rowvec_dgrad_kernel<0>, dim3((B + 3) / 4, hc2), dim3(256), 0, stream, (const float*)w.dlat + 2 * Cd, (long)LW, B, 4 * D, D, hp,
                     (const float*)nullptr, (float*)nullptr, w.dgpart + (size_t)hc1 * bt);
  hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((bt / 4 + 255) / 256)), dim3(256), 0, stream, (const float*)w.dgpart, w.dpooled, bt, hc1 + hc2);
  PtrSet4 pp; OutSet4 po;
  for (int i = 0; i < 4; ++i) { pp.w[i] = nullptr; pp.b[i] = nullptr; po.w[i] = nullptr; po.b[i] = nullptr; }
  pp.w[0] = (const float*)a->pooler_w; po.w[0] = (float*)a->d_pooler_w; po.b[0] = (float*)a->d_pooler_b;
  hipLaunchKernelGGL(rowvec_dgrad_kernel<1>, dim3((B + 3) / 4, pc), dim3(256), 0, stream, (const float*)w.dpooled, (long)TH, B, TH, TH, pp,
                     (const float*)a->pooled, w.dpre, w.dgpart);
  hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((bt / 4 + 255) / 256)), dim3(256), 0, stream, (const float*)w.dgpart, w.dcls, bt, pc);
  hipLaunchKernelGGL(rowvec_wgrad_kernel, dim3(TH / 4), dim3(256), 0, stream, (const float*)w.dpre, (long)TH, (const float*)a->x_last_f32,
                     (long)a->seq_len * TH, (const int*)a->cls_rows, B, TH, TH, po);
  const size_t dx_rows = a->n_rows > 0 ? (size_t)a->n_rows : (size_t)B * a->seq_len;
  (void)hipMemsetAsync(a->dx_last_f32, 0, dx_rows * TH * sizeof(float), stream);
  hipLaunchKernelGGL(scatter_cls_kernel, dim3(B), dim3(256), 0, stream, (const float*)w.dcls, B, a->seq_len, (const int*)a->cls_rows,
                     (float*)a->dx_last_f32);
  return check_launch("en tail backward");
}
