// VAE tail of DrlClassifier.forward (drl_classifier_ec_mmd_final_mul.py:202-261) and its backward:
//   pooler (HF BertPooler: tanh(W_p x[:,0] + b_p)) -> 4 latent heads (:312-336) -> reparameterised sample
//   (:345-351, one eps vector shared by the batch, std = exp(log_var)) -> emotion CE (:461-476), cause BCE
//   (:478-492), pair BCE-with-logits + pos_weight (:494-513, replaced by 0 when infinite), RBF-MMD
//   (:231-233), annealed KL (:515-534), decoder softmax + BCE reconstruction (:253-254, :381-387),
//   weighted sum (:256-261).
// Everything is fp32.  The scalar loss and the gradients w.r.t. every tail tensor are produced together
// (the loss is the root of the graph, so backward only rescales them by grad_output).
#include "carel_hip_internal.h"
#include "mmd_device.h"
#include "hsic_device.h"
#include "rowvec_device.h"

namespace carel {

// ------------------------------------------------------------------------------------------
// Tail core: one workgroup.  lat = [mu_e | lv_e | mu_c | lv_c] (B x 4D).
// ------------------------------------------------------------------------------------------
struct TailCore {
  int B, D, EC;                 // batch, ec_dim (<= 32), emotion classes (<= 8)
  const float* lat;             // [B, 4D]
  const float* eps_e; const float* eps_c;
  const float* emo_w; const float* emo_b;      // [EC, D], [EC]
  const float* cau_w; const float* cau_b;      // [1, D], [1]
  const float* pair_w; const float* pair_b;    // [1, 2D], [1]
  const long* emo_labels; const float* cau_labels; const float* pair_labels;
  float w_mmd, w_emo, w_cau, w_pair, kl_w, ls;
  int dis_mode;                 // 0: -w_mmd * RBF-MMD (ref :231-233); 1: +w_mmd * HSIC (drl_classifier_ec_hsic.py:214); 2: none
  int emo_bce;                  // 1: one-logit sigmoid + BCE emotion head (ec_hsic/ec_vi scripts) instead of the softmax CE
  Dropout d_emo, d_cau, d_pair;                // element index b*D + k (pair: b*2D + k)
  float alpha, mmd_eps;
  const float* label_sum_override; float n_override;   // global-batch pos_weight (DP); null/0 = local
  int label_sum_ranks;          // > 1: the global label sum is the sum of this many floats, rank_stride apart
  const float* z_global; int n_global, row_offset; float mmd_grad_scale;  // global-batch MMD (DP)
  int rank_stride;              // floats between consecutive ranks' blocks of B rows in z_global (B*2D when dense)
  const float* mmd_part; int mmd_nblk; const float* dz_mmd;   // global-batch MMD computed by mmd_global_kernel (else null)
  // outputs
  float* z;                     // [B, 2D]; null = do not write it (the caller's own sample_z_kernel did: overlapped tail)
  float* terms;                 // [16]: 0 partial loss (everything but rec), 1 mmd, 2 emo, 3 cau, 4 pair, 5 kl_e, 6 kl_c
  float* dz;                    // [B, 2D]  d(loss)/dz (without reconstruction)
  float* dlat_direct;           // [B, 4D]  KL part of d(loss)/d lat
  float* d_emo_w; float* d_emo_b; float* d_cau_w; float* d_cau_b; float* d_pair_w; float* d_pair_b;
  float* pair_dead;             // [1]: 1.0 if the pair loss was replaced by 0 (ref :510-511)
  long long* prof;              // diagnostics (carel_tail_profile): phase time stamps, or null
};
#define TAIL_STAMP(i) do { if (a.prof && threadIdx.x == 0) a.prof[i] = (long long)wall_clock64(); } while (0)

// ------------------------------------------------------------------------------------------
// Global-batch RBF-MMD for data parallel runs (z_global = every rank's sampled latents): the statistic needs all
// (2n)^2 pairs -- 1 M at 8 x 64 samples -- which one workgroup would chew on for ~0.4 ms.  Here each workgroup takes 32
// rows (8 lanes per row split the partners, streamed through LDS in chunks of 256 rows) and writes its partial sums of
// the three blocks; rows of THIS rank's samples also get d(gscale * mmd)/dz.  tail_core_kernel adds the partials in
// block order.  Feature width padded to CD with zeros (CD = 24: the reference's ec_dim; 32: anything else <= 32).
// ------------------------------------------------------------------------------------------
struct MmdGlobalArgs {
  const float* zg; int n, B, D, rank_stride, row_offset;     // n samples per side in total, B per rank
  float alpha, eps, gscale;
  float* part;      // [gridDim.x][4]: s11, s22, s12
  float* dz;        // [B][2D] (local rows)
};
template <int CD>
__global__ __launch_bounds__(256) void mmd_global_kernel(MmdGlobalArgs a) {
  constexpr int CH = 256, ZS = CD + 1;
  __shared__ float zc[CH * ZS];
  __shared__ float nc[CH];
  __shared__ float red[64];
  const int t = threadIdx.x, grp = t & 7;
  const int n = a.n, n2 = 2 * n, D = a.D, D2 = 2 * D;
  auto zrow = [&](int r) -> const float* {               // row r of Z = [all emotion samples ; all cause samples]
    const int side = r >= n, s = r - side * n;
    return a.zg + (long)(s / a.B) * a.rank_stride + (long)(s % a.B) * D2 + side * D;
  };
  const int i = blockIdx.x * 32 + (t >> 3);
  const bool live = i < n2;
  float zi[CD], g[CD];
  float ni = 0.f;
  {
    const float* zp = zrow(live ? i : 0);
#pragma unroll
    for (int k = 0; k < CD; ++k) { zi[k] = (k < D) ? zp[k] : 0.f; g[k] = 0.f; ni = fmaf(zi[k], zi[k], ni); }
  }
  const bool i1 = i < n;
  const int si = i - (i1 ? 0 : n);
  const bool local = live && si >= a.row_offset && si < a.row_offset + a.B;
  const float a00 = (float)(1.0 / ((double)n * (n - 1))), a01 = (float)(-1.0 / ((double)n * n));
  float s11 = 0.f, s22 = 0.f, s12 = 0.f;
  for (int c0 = 0; c0 < n2; c0 += CH) {
    const int cn = min(CH, n2 - c0);
    __syncthreads();
    for (int e = t; e < cn * CD; e += 256) {
      const int r = e / CD, k = e - r * CD;
      zc[r * ZS + k] = (k < D) ? zrow(c0 + r)[k] : 0.f;
    }
    __syncthreads();
    for (int r = t; r < cn; r += 256) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < CD; ++k) s = fmaf(zc[r * ZS + k], zc[r * ZS + k], s);
      nc[r] = s;
    }
    __syncthreads();
    if (live) {
      for (int jj = grp; jj < cn; jj += 8) {
        const int j = c0 + jj;
        if (j == i) continue;
        float zj[CD], dot = 0.f;
#pragma unroll
        for (int k = 0; k < CD; ++k) { zj[k] = zc[jj * ZS + k]; dot = fmaf(zi[k], zj[k], dot); }
        const float d2 = ni + nc[jj] - 2.0f * dot;
        const float e = expf(-a.alpha * (a.eps + fabsf(d2)));
        const bool j1 = j < n;
        if (i1 && j1) s11 += e; else if (!i1 && !j1) s22 += e; else if (i1 && !j1) s12 += e;
        if (local) {
          const float dcs = (d2 > 0.f) ? a.alpha * e : ((d2 < 0.f) ? -a.alpha * e : 0.f);
          const float coef = -4.0f * ((i1 == j1) ? a00 : a01) * dcs * a.gscale;
#pragma unroll
          for (int k = 0; k < CD; ++k) g[k] = fmaf(coef, zi[k] - zj[k], g[k]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < CD; ++k) {
    float v = g[k];
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    if (local && grp == 0 && k < D) a.dz[(long)(si - a.row_offset) * D2 + (i1 ? 0 : D) + k] = v;
  }
  __syncthreads();
  s11 = block_sum(s11, red);
  s22 = block_sum(s22, red + 16);
  s12 = block_sum(s12, red + 32);
  if (t == 0) { a.part[blockIdx.x * 4] = s11; a.part[blockIdx.x * 4 + 1] = s22; a.part[blockIdx.x * 4 + 2] = s12; }
}

static long long* g_tail_prof = nullptr;
// tuning hook (carel_gemm_set_variant(210 / 211)): the single-workgroup loss kernel (~50 us on ONE CU) on the library's low-priority side stream
// while the reconstruction decoder's passes (~100 us, 186 workgroups) run on the caller's stream, joined before the two results meet
CAREL_TUNABLE(int, g_tail_overlap, 1);
#ifdef CAREL_EXPERIMENTS
void tail_overlap_enable(int on) { g_tail_overlap = on ? 1 : 0; }
#endif
struct TailEvents { hipEvent_t fork, join; bool ok; };
static TailEvents* tail_events() {               // two events per device, created on first use and kept for the life of the process
  static TailEvents per_dev[16];
  static bool made[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!made[dev]) {
    TailEvents& t = per_dev[dev];
    t.ok = hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&t.join, hipEventDisableTiming) == hipSuccess;
    made[dev] = true;
  }
  return per_dev[dev].ok ? &per_dev[dev] : nullptr;
}

__global__ __launch_bounds__(1024) void tail_core_kernel(TailCore a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;                 // 64
  float* sc = red + 64;                          // 64 scalars
  const int B = a.B, D = a.D, D2 = 2 * a.D, D4 = 4 * a.D;
  float* zl = sc + 64;                           // [B][2D] sampled latents
  float* dzl = zl + B * D2;                      // [B][2D] gradient accumulator
  float* elog = dzl + B * D2;                    // [B][8] emotion dlogits
  float* gx = elog + B * 8;                      // [B][2]  cause / pair dlogit
  float* mlt = gx + B * 2;                       // [B][4D] dropout multipliers: emotion [D] | cause [D] | pair [2D]
  float* lg = mlt + B * D4;                      // [B][16] head logits: emotion classes, then cause, then pair
  float* hw = lg + B * 16;                       // head weights: emo_w [EC*D], emo_b [EC], cau_w [D], cau_b, pair_w [2D], pair_b
  const int n_emo = a.EC * D;
  const int o_eb = n_emo, o_cw = o_eb + a.EC, o_cb = o_cw + D, o_pw = o_cb + 1, o_pb = o_pw + D2, n_hw = o_pb + 1;
  const int nm = a.z_global ? a.n_global : B;    // samples per side in the MMD
  float* nrm = hw + ((n_hw + 3) & ~3);           // [2 nm]
  float* Z = nrm + ((2 * nm + 3) & ~3);          // MMD samples [2 nm][D|1]
  const int zs = D | 1;
  const int t = threadIdx.x;
  TAIL_STAMP(0);

  for (int e = t; e < B * D2; e += blockDim.x) {
    const int b = e / D2, k = e - b * D2;
    const float* row = a.lat + (long)b * 4 * D;
    const float z = (k < D) ? row[k] + a.eps_e[k] * expf(row[D + k]) : row[2 * D + (k - D)] + a.eps_c[k - D] * expf(row[3 * D + (k - D)]);
    zl[e] = z; dzl[e] = 0.f;
    if (a.z) a.z[e] = z;
  }
  for (int e = t; e < B * D4; e += blockDim.x) {
    const int b = e / D4, k = e - b * D4;
    mlt[e] = k < D ? dropout_mult(a.d_emo, b * D + k) : (k < D2 ? dropout_mult(a.d_cau, b * D + (k - D)) : dropout_mult(a.d_pair, b * D2 + (k - D2)));
  }
  for (int e = t; e < n_hw; e += blockDim.x)
    hw[e] = e < o_eb ? a.emo_w[e] : e < o_cw ? a.emo_b[e - o_eb] : e < o_cb ? a.cau_w[e - o_cw] : e < o_pw ? a.cau_b[0]
          : e < o_pb ? a.pair_w[e - o_pw] : a.pair_b[0];
  __syncthreads();
  // ---- MMD samples: rows [0,nm) emotion, [nm,2nm) cause   (not needed when mmd_global_kernel did the statistic)
  for (int e = t; e < (a.mmd_part ? 0 : 2 * nm * D); e += blockDim.x) {
    const int i = e / D, k = e - i * D;
    const int s = i < nm ? i : i - nm, off = i < nm ? 0 : D;
    Z[i * zs + k] = a.z_global ? a.z_global[(long)(s / B) * a.rank_stride + (long)(s % B) * D2 + off + k] : zl[s * D2 + off + k];
  }
  __syncthreads();
  TAIL_STAMP(1);
  float mmd = 0.f;                     // value of the disentanglement statistic (terms[1])
  float dis_term = 0.f;                // its contribution to the loss
  if (a.dis_mode == 0 && a.mmd_part) {
    // partial block sums in fixed order (double accumulation as in mmd_forward_block), gradient rows already computed
    if (t == 0) {
      double s11 = 0.0, s22 = 0.0, s12 = 0.0;
      for (int q = 0; q < a.mmd_nblk; ++q) { s11 += a.mmd_part[q * 4]; s22 += a.mmd_part[q * 4 + 1]; s12 += a.mmd_part[q * 4 + 2]; }
      const double b00 = 1.0 / ((double)nm * (nm - 1)), b01 = -1.0 / ((double)nm * nm);
      sc[0] = (float)(2.0 * b01 * s12 + b00 * s11 + b00 * s22);
    }
    __syncthreads();
    mmd = sc[0];
    dis_term = -a.w_mmd * mmd;
    for (int e = t; e < B * D2; e += blockDim.x) dzl[e] += a.dz_mmd[e];
    TAIL_STAMP(2);
  } else if (a.dis_mode == 0) {
    MmdCfg mc; mc.n1 = nm; mc.n2 = nm; mc.d = D; mc.zs = zs; mc.n_alphas = 1; mc.alphas[0] = a.alpha; mc.eps = a.mmd_eps;
    if (!a.z_global && D == 24) {      // the configuration of every reference script: one fused forward + backward sweep
      mmd = mmd_forward_backward_block<24>(mc, Z, nrm, red, -a.w_mmd * a.mmd_grad_scale, dzl, D2);
      dis_term = -a.w_mmd * mmd;
      TAIL_STAMP(2);
    } else {
      mmd = mmd_forward_block(mc, Z, nrm, red, nullptr);
      dis_term = -a.w_mmd * mmd;
      TAIL_STAMP(2);
      // d(-w_mmd * mmd)/dz for the local rows: 8 threads per row
      const int rows = 2 * B;
      const int grp = t & 7;
      for (int base = 0; base < rows; base += blockDim.x / 8) {
        const int rloc = base + (t >> 3);
        float g[32];
        const bool live = rloc < rows;
        if (live) {
          const int side = rloc >= B, bl = rloc - side * B;
          const int i = side * nm + a.row_offset * (a.z_global ? 1 : 0) + bl;
          mmd_backward_row(mc, Z, nrm, i, -a.w_mmd * a.mmd_grad_scale, g, grp, 8);
        } else {
          for (int k = 0; k < D; ++k) g[k] = 0.f;
        }
        for (int k = 0; k < D; ++k) {
          float v = g[k];
          v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
          if (live && grp == 0) { const int side = rloc >= B, bl = rloc - side * B; dzl[bl * D2 + side * D + k] += v; }
        }
      }
    }
  } else if (a.dis_mode == 1) {        // HSIC between the local emotion (rows 0..B) and cause (rows B..2B) samples
    HsicCfg hc; hc.m = B; hc.d = D; hc.xs = zs; hc.inv_sx = 1.0f; hc.inv_sy = 1.0f;
    float* rk = Z + 2 * nm * zs;       // [B] row sums of K, then [B] of L (extra LDS reserved by the host)
    float* rl = rk + B;
    mmd = hsic_forward_block(hc, Z, Z + B * zs, nrm, nrm + B, rk, rl, red);
    dis_term = a.w_mmd * mmd;
    const float tk = red[60], tl = red[61];
    for (int i = t; i < B; i += blockDim.x) {
      float gx[32], gy[32];
      hsic_backward_row(hc, Z, Z + B * zs, nrm, nrm + B, rk, rl, tk, tl, i, a.w_mmd, gx, gy);
      for (int k = 0; k < D; ++k) { dzl[i * D2 + k] += gx[k]; dzl[i * D2 + D + k] += gy[k]; }
    }
  }
  __syncthreads();
  TAIL_STAMP(3);
  // ---- classifier heads: 16 threads per sample, one per logit (emotion classes, cause, pair)
  for (int e = t; e < B * 16; e += blockDim.x) {
    const int b = e >> 4, h = e & 15;
    if (h < a.EC) {
      float s = hw[o_eb + h];
      for (int k = 0; k < D; ++k) s = fmaf(hw[h * D + k], zl[b * D2 + k] * mlt[b * D4 + k], s);
      lg[e] = s;
    } else if (h == a.EC) {
      float s = hw[o_cb];
      for (int k = 0; k < D; ++k) s = fmaf(hw[o_cw + k], zl[b * D2 + D + k] * mlt[b * D4 + D + k], s);
      lg[e] = s;
    } else if (h == a.EC + 1) {
      float s = hw[o_pb];
      for (int k = 0; k < D2; ++k) s = fmaf(hw[o_pw + k], zl[b * D2 + k] * mlt[b * D4 + D2 + k], s);
      lg[e] = s;
    }
  }
  float l_emo = 0.f, l_cau = 0.f, l_pair = 0.f, ysum = 0.f;
  if (t < B) ysum = a.pair_labels[t];
  ysum = block_sum(t < B ? ysum : 0.f, red);        // (its barriers also publish lg)
  const float ntot = a.label_sum_override ? a.n_override : (float)B;
  float ytot = ysum;
  if (a.label_sum_override) {
    ytot = a.label_sum_override[0];
    for (int r = 1; r < a.label_sum_ranks; ++r) ytot += a.label_sum_override[(long)r * a.rank_stride];
  }
  const float pw = (ntot - ytot) / ytot;        // inf when there is no positive in the batch
  float xp = 0.f, tp = 0.f;
  for (int b = t; b < B; b += blockDim.x) {
    if (!a.emo_bce) {
      // emotion: CE(W (z_e * m) + b, label)
      float mx = -INFINITY;
      for (int c = 0; c < a.EC; ++c) mx = fmaxf(mx, lg[b * 16 + c]);
      float se = 0.f;
      for (int c = 0; c < a.EC; ++c) se += expf(lg[b * 16 + c] - mx);
      const float lse = mx + logf(se);
      long lab = a.emo_labels[b]; lab = lab < 0 ? 0 : (lab >= a.EC ? a.EC - 1 : lab);
      l_emo += lse - lg[b * 16 + lab];
      for (int c = 0; c < a.EC; ++c) elog[b * 8 + c] = (expf(lg[b * 16 + c] - lse) - (c == lab ? 1.f : 0.f)) * (a.w_emo / B);
    } else {
      // emotion, 1-logit variant: BCE(sigmoid(w (z_e * m) + b), smoothed label)   (drl_classifier_ec_hsic.py:455-470)
      const float pe = 1.0f / (1.0f + expf(-lg[b * 16]));
      const float te = (float)a.emo_labels[b] * (1.f - a.ls) + a.ls;
      l_emo += -(te * fmaxf(logf(pe), -100.f) + (1.f - te) * fmaxf(logf(1.f - pe), -100.f));
      const float gpe = (pe - te) / fmaxf((1.f - pe) * pe, 1e-12f);
      elog[b * 8] = gpe * pe * (1.f - pe) * (a.w_emo / B);
    }
    // cause: BCE(sigmoid(w (z_c * m) + b), smoothed)
    const float pc = 1.0f / (1.0f + expf(-lg[b * 16 + a.EC]));
    const float tc = a.cau_labels[b] * (1.f - a.ls) + a.ls;
    l_cau += -(tc * fmaxf(logf(pc), -100.f) + (1.f - tc) * fmaxf(logf(1.f - pc), -100.f));
    const float gp = (pc - tc) / fmaxf((1.f - pc) * pc, 1e-12f);
    gx[b * 2] = gp * pc * (1.f - pc) * (a.w_cau / B);
    // pair: BCEWithLogits(w (z * m) + b, smoothed, pos_weight)
    xp = lg[b * 16 + a.EC + 1];
    tp = a.pair_labels[b] * (1.f - a.ls) + a.ls;
    const float lw = (pw - 1.f) * tp + 1.f;
    l_pair += (1.f - tp) * xp + lw * (log1pf(expf(-fabsf(xp))) + fmaxf(-xp, 0.f));
  }
  l_emo = block_sum(l_emo, red) / B;
  l_cau = block_sum(l_cau, red) / B;
  l_pair = block_sum(l_pair, red) / B;
  const bool dead = isinf(l_pair);
  if (dead) l_pair = 0.f;
  for (int b = t; b < B; b += blockDim.x) {
    const float xb = lg[b * 16 + a.EC + 1], tb = a.pair_labels[b] * (1.f - a.ls) + a.ls;
    const float lw = (pw - 1.f) * tb + 1.f;
    const float sg = 1.0f / (1.0f + expf(-xb));
    gx[b * 2 + 1] = dead ? 0.f : ((1.f - tb) - lw * (1.f - sg)) * (a.w_pair / B);
  }
  __syncthreads();
  TAIL_STAMP(4);
  // ---- KL (:525-534) and its direct gradient on lat
  float kle = 0.f, klc = 0.f;
  for (int e = t; e < B * D; e += blockDim.x) {
    const int b = e / D, k = e - b * D;
    const float* row = a.lat + (long)b * 4 * D;
    const float mue = row[k], lve = row[D + k], muc = row[2 * D + k], lvc = row[3 * D + k];
    kle += -0.5f * (1.f + lve - expf(lve) - mue * mue);
    klc += -0.5f * (1.f + lvc - expf(lvc) - muc * muc);
    float* dl = a.dlat_direct + (long)b * 4 * D;
    dl[k] = a.kl_w * mue / B; dl[D + k] = a.kl_w * (-0.5f) * (1.f - expf(lve)) / B;
    dl[2 * D + k] = a.kl_w * muc / B; dl[3 * D + k] = a.kl_w * (-0.5f) * (1.f - expf(lvc)) / B;
  }
  kle = block_sum(kle, red) / B * a.kl_w;
  klc = block_sum(klc, red) / B * a.kl_w;
  TAIL_STAMP(5);
  // ---- dz from the three heads, and the head parameter gradients
  for (int e = t; e < B * D2; e += blockDim.x) {
    const int b = e / D2, k = e - b * D2;
    float g = hw[o_pw + k] * gx[b * 2 + 1] * mlt[b * D4 + D2 + k];
    if (k < D) {
      float s = 0.f;
      for (int c = 0; c < a.EC; ++c) s = fmaf(hw[c * D + k], elog[b * 8 + c], s);
      g += s * mlt[b * D4 + k];
    } else {
      g += hw[o_cw + (k - D)] * gx[b * 2] * mlt[b * D4 + k];
    }
    dzl[e] += g;
  }
  TAIL_STAMP(6);
  // parameter gradients: thread per parameter element, loop over the batch (fixed order)
  for (int e = t; e < n_hw; e += blockDim.x) {
    float s = 0.f;
    if (e < o_eb) {
      const int c = e / D, k = e - c * D;
      for (int b = 0; b < B; ++b) s = fmaf(elog[b * 8 + c], zl[b * D2 + k] * mlt[b * D4 + k], s);
      a.d_emo_w[e] = s;
    } else if (e < o_cw) {
      const int c = e - o_eb;
      for (int b = 0; b < B; ++b) s += elog[b * 8 + c];
      a.d_emo_b[c] = s;
    } else if (e < o_cb) {
      const int k = e - o_cw;
      for (int b = 0; b < B; ++b) s = fmaf(gx[b * 2], zl[b * D2 + D + k] * mlt[b * D4 + D + k], s);
      a.d_cau_w[k] = s;
    } else if (e == o_cb) {
      for (int b = 0; b < B; ++b) s += gx[b * 2];
      a.d_cau_b[0] = s;
    } else if (e < o_pb) {
      const int k = e - o_pw;
      for (int b = 0; b < B; ++b) s = fmaf(gx[b * 2 + 1], zl[b * D2 + k] * mlt[b * D4 + D2 + k], s);
      a.d_pair_w[k] = s;
    } else {
      for (int b = 0; b < B; ++b) s += gx[b * 2 + 1];
      a.d_pair_b[0] = s;
    }
  }
  __syncthreads();
  TAIL_STAMP(7);
  for (int e = t; e < B * D2; e += blockDim.x) a.dz[e] = dzl[e];
  if (t == 0) {
    a.terms[1] = mmd; a.terms[2] = l_emo; a.terms[3] = l_cau; a.terms[4] = l_pair; a.terms[5] = kle; a.terms[6] = klc;
    a.terms[0] = dis_term + a.w_emo * l_emo + a.w_cau * l_cau + a.w_pair * l_pair + kle + klc;
    a.pair_dead[0] = dead ? 1.f : 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// Decoder: logits[b][j] = z[b] . Wd[j] + bd[j] ; p = softmax_j ; rec = mean_{b,j} BCE(p, 0.9 bow + 0.1/V)
// Work is split over chunks of DEC_J vocabulary entries (one thread per entry holding its 2D weights in
// registers, the batch's latents in LDS).  Three passes (row max/sum, loss + softmax-backward dot,
// gradients); partial results are combined in fixed order so the result is run-to-run reproducible.
// ------------------------------------------------------------------------------------------
constexpr int DEC_J = 128;          // vocabulary entries per workgroup
constexpr int DEC_MAXD2 = 64;
constexpr int DEC_MAXG = 4;         // sample groups per workgroup (threads = DEC_J * groups)

struct DecArgs {
  int B, D2, V;
  const float* z;          // [B, D2]
  const float* w; const float* b;          // [V, D2], [V]
  const float* bow;        // [B, V]
  float ls;
  float* part;             // pass1: [chunks][B][2] (max, sumexp) ; pass2: [chunks][B][2] (loss, dot)
  float* rowstat;          // [B][4]: M, log L, dot, loss-sum
  float* dz_part;          // [chunks][B][D2]
  float* dw; float* db;    // [V, D2], [V]
  float gscale;            // 1/(B V)
};

template <int CD2>
__device__ __forceinline__ float dec_logit(const float* wr, const float* zrow, int D2rt, float bias) {
  const int D2 = CD2 ? CD2 : D2rt;
  float s = bias;
  if (CD2 && (CD2 & 3) == 0) {
#pragma unroll
    for (int k = 0; k < D2; k += 4) {
      const float4 zv = *(const float4*)(zrow + k);          // LDS broadcast read, 16 B
      s = fmaf(wr[k], zv.x, s); s = fmaf(wr[k + 1], zv.y, s); s = fmaf(wr[k + 2], zv.z, s); s = fmaf(wr[k + 3], zv.w, s);
    }
  } else {
#pragma unroll
    for (int k = 0; k < D2; ++k) s = fmaf(wr[k], zrow[k], s);
  }
  return s;
}

// CD2 > 0: 2*ec_dim known at compile time (weights stay in registers); CD2 == 0: generic run-time width.
// Thread t: entry tj = t % DEC_J of the chunk, sample group g = t / DEC_J (G = blockDim.x / DEC_J groups, each a
// contiguous range of ceil(B/G) samples): 4x the waves of a thread-per-entry layout, bow values prefetched four samples
// ahead.  Per-group results are combined in fixed order, so the outputs are run-to-run reproducible.
template <int PASS, int CD2>
__global__ __launch_bounds__(DEC_J * DEC_MAXG) void decoder_kernel(DecArgs a) {
  const int D2 = CD2 ? CD2 : a.D2;
  const int ws = D2 + 1;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* zl = (float*)smem_raw;                       // [B][D2]
  float* red = zl + a.B * D2;                         // 16
  float* wl = red + 16;                               // [DEC_J][D2+1] weights of the chunk
  float* tile = wl + DEC_J * ws;                      // [B][DEC_J] per-(sample, entry) values (pass 3: dlogits)
  float* tile2 = tile + a.B * DEC_J;                  // pass 2: second tile; pass 3: [G-1][DEC_J][D2+1] dW partials + [G-1][DEC_J] db
  const int t = threadIdx.x, nthr = blockDim.x, G = nthr / DEC_J;
  const int tj = t & (DEC_J - 1), g = t / DEC_J;
  const int j0 = blockIdx.x * DEC_J, j = j0 + tj;
  const bool live = j < a.V;
  const int Bg = (a.B + G - 1) / G, b0 = g * Bg, b1 = min(a.B, b0 + Bg);
  for (int e = t; e < a.B * D2; e += nthr) zl[e] = a.z[e];
  {   // coalesced load of the chunk's weights, then each thread takes its row
    const int nrow = min(DEC_J, a.V - j0);
    for (int e = t; e < nrow * D2; e += nthr) { const int r = e / D2, k = e - r * D2; wl[r * ws + k] = a.w[(long)j0 * D2 + e]; }
  }
  __syncthreads();
  float wr[CD2 ? CD2 : DEC_MAXD2];
  _Pragma("unroll") for (int k = 0; k < D2; ++k) wr[k] = live ? wl[tj * ws + k] : 0.f;
  const float bias = live ? a.b[j] : 0.f;
  const int lane = t & 63, wv = t >> 6, nwv = nthr >> 6;
  if (PASS == 1) {
    for (int b = b0; b < b1; ++b) tile[b * DEC_J + tj] = live ? dec_logit<CD2>(wr, zl + b * D2, D2, bias) : -INFINITY;
    __syncthreads();
    for (int b = wv; b < a.B; b += nwv) {
      const float x0 = tile[b * DEC_J + lane], x1 = tile[b * DEC_J + 64 + lane];
      const float m = wave_max(fmaxf(x0, x1));
      const float s2 = wave_sum(expf(x0 - m) + expf(x1 - m));            // exp(-inf - m) = 0 for dead entries
      if (lane == 0) { a.part[((long)blockIdx.x * a.B + b) * 2] = m; a.part[((long)blockIdx.x * a.B + b) * 2 + 1] = s2; }
    }
  } else if (PASS == 2) {
    for (int bb = b0; bb < b1; bb += 4) {
      float bw[4];
      _Pragma("unroll") for (int u = 0; u < 4; ++u) bw[u] = (live && bb + u < b1) ? a.bow[(long)(bb + u) * a.V + j] : 0.f;
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {
        const int b = bb + u;
        if (b < b1) {
          float le = 0.f, dt = 0.f;
          if (live) {
            const float lp = dec_logit<CD2>(wr, zl + b * D2, D2, bias) - a.rowstat[b * 4] - a.rowstat[b * 4 + 1];
            const float p = expf(lp);
            const float tg = bw[u] * (1.f - a.ls) + a.ls / a.V;
            le = -(tg * fmaxf(lp, -100.f) + (1.f - tg) * fmaxf(log1pf(-p), -100.f));
            dt = p * ((p - tg) / fmaxf((1.f - p) * p, 1e-12f));
          }
          tile[b * DEC_J + tj] = le; tile2[b * DEC_J + tj] = dt;
        }
      }
    }
    __syncthreads();
    for (int b = wv; b < a.B; b += nwv) {
      const float le = wave_sum(tile[b * DEC_J + lane] + tile[b * DEC_J + 64 + lane]);
      const float dt = wave_sum(tile2[b * DEC_J + lane] + tile2[b * DEC_J + 64 + lane]);
      if (lane == 0) { a.part[((long)blockIdx.x * a.B + b) * 2] = le; a.part[((long)blockIdx.x * a.B + b) * 2 + 1] = dt; }
    }
  } else {
    float dwr[CD2 ? CD2 : DEC_MAXD2];
    _Pragma("unroll") for (int k = 0; k < D2; ++k) dwr[k] = 0.f;
    float dbj = 0.f;
    for (int bb = b0; bb < b1; bb += 4) {
      float bw[4];
      _Pragma("unroll") for (int u = 0; u < 4; ++u) bw[u] = (live && bb + u < b1) ? a.bow[(long)(bb + u) * a.V + j] : 0.f;
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {
        const int b = bb + u;
        if (b < b1) {
          float gg = 0.f;
          if (live) {
            const float lp = dec_logit<CD2>(wr, zl + b * D2, D2, bias) - a.rowstat[b * 4] - a.rowstat[b * 4 + 1];
            const float p = expf(lp);
            const float tg = bw[u] * (1.f - a.ls) + a.ls / a.V;
            const float gp = (p - tg) / fmaxf((1.f - p) * p, 1e-12f);
            gg = p * (gp - a.rowstat[b * 4 + 2]) * a.gscale;
            _Pragma("unroll") for (int k = 0; k < D2; ++k) dwr[k] = fmaf(gg, zl[b * D2 + k], dwr[k]);
            dbj += gg;
          }
          tile[b * DEC_J + tj] = gg;
        }
      }
    }
    float* dwp = tile2;                               // [G-1][DEC_J][ws]
    float* dbp = dwp + (G - 1) * DEC_J * ws;          // [G-1][DEC_J]
    if (g > 0) {
      _Pragma("unroll") for (int k = 0; k < D2; ++k) dwp[((g - 1) * DEC_J + tj) * ws + k] = dwr[k];
      dbp[(g - 1) * DEC_J + tj] = dbj;
    }
    __syncthreads();
    if (g == 0 && live) {
      for (int q = 0; q < G - 1; ++q) {               // fixed order: group 0 + group 1 + ...
        _Pragma("unroll") for (int k = 0; k < D2; ++k) dwr[k] += dwp[(q * DEC_J + tj) * ws + k];
        dbj += dbp[q * DEC_J + tj];
      }
      _Pragma("unroll") for (int k = 0; k < D2; ++k) a.dw[(long)j * D2 + k] = dwr[k];
      a.db[j] = dbj;
    }
    // dz_part[chunk][b][k] = sum_{j in chunk} dlogit[b][j] * W[j][k]
    for (int e = t; e < a.B * D2; e += nthr) {
      const int b = e / D2, k = e - b * D2;
      const int nrow = min(DEC_J, a.V - j0);
      float s = 0.f;
      for (int jj = 0; jj < nrow; ++jj) s = fmaf(tile[b * DEC_J + jj], wl[jj * ws + k], s);
      a.dz_part[(long)blockIdx.x * a.B * D2 + e] = s;
    }
  }
}

// combine pass-1 partials -> rowstat[b] = {M, log L}; pass-2 partials -> rowstat[b][2] = dot, [3] = row loss.
// One wave per sample (lanes over chunks).
__global__ __launch_bounds__(256) void decoder_combine_kernel(const float* __restrict__ part, int chunks, int B, int pass,
                                                              float* __restrict__ rowstat) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  if (pass == 1) {
    float m = -INFINITY;
    for (int c = lane; c < chunks; c += 64) m = fmaxf(m, part[((long)c * B + b) * 2]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < chunks; c += 64) s += part[((long)c * B + b) * 2 + 1] * expf(part[((long)c * B + b) * 2] - m);
    s = wave_sum(s);
    if (lane == 0) { rowstat[b * 4] = m; rowstat[b * 4 + 1] = logf(s); }
  } else {
    float le = 0.f, dt = 0.f;
    for (int c = lane; c < chunks; c += 64) { le += part[((long)c * B + b) * 2]; dt += part[((long)c * B + b) * 2 + 1]; }
    le = wave_sum(le); dt = wave_sum(dt);
    if (lane == 0) { rowstat[b * 4 + 2] = dt; rowstat[b * 4 + 3] = le; }
  }
}
// rec = sum_b rowloss[b] / (B V); terms[7] = rec, terms[8] = terms[0] + rec
__global__ __launch_bounds__(256) void decoder_total_kernel(const float* __restrict__ rowstat, int B, float inv_bv, float* __restrict__ terms) {
  __shared__ float red[16];
  float tot = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) tot += rowstat[b * 4 + 3];
  tot = block_sum(tot, red);
  if (threadIdx.x == 0) { terms[7] = tot * inv_bv; terms[8] = terms[0] + tot * inv_bv; }
}

// out[c] = sum_p parts[p][c]: 16 columns x 16 part-lanes per block, fixed order
__global__ __launch_bounds__(256) void reduce_parts16_kernel(const float* __restrict__ parts, float* __restrict__ out, int n, int nparts) {
  __shared__ float lds[256];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < n) for (int p = rl; p < nparts; p += 16) s += parts[(long)p * n + c];
  lds[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && c < n) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += lds[r * 16 + cl];
    out[c] = t;
  }
}

// dlat = dlat_direct + [dz_e, dz_e*eps_e*exp(lv_e), dz_c, dz_c*eps_c*exp(lv_c)], dz = dz_core + sum_chunks dz_part
__global__ __launch_bounds__(256) void tail_dlat_kernel(const float* __restrict__ dz_core, const float* __restrict__ dz_part, int chunks,
                                                        const float* __restrict__ dlat_direct, const float* __restrict__ lat,
                                                        const float* __restrict__ eps_e, const float* __restrict__ eps_c, int B, int D,
                                                        float gout, float* __restrict__ dlat) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * 2 * D) return;
  const int b = e / (2 * D), k = e - b * 2 * D;
  float g = dz_core[e];
  for (int c = 0; c < chunks; ++c) g += dz_part[(long)c * B * 2 * D + e];
  const int side = k >= D, kk = k - side * D;
  const float* row = lat + (long)b * 4 * D;
  const float ep = side ? eps_c[kk] : eps_e[kk];
  const float lv = row[(2 * side + 1) * D + kk];
  dlat[(long)b * 4 * D + 2 * side * D + kk] = (dlat_direct[(long)b * 4 * D + 2 * side * D + kk] + g) * gout;
  dlat[(long)b * 4 * D + (2 * side + 1) * D + kk] = (dlat_direct[(long)b * 4 * D + (2 * side + 1) * D + kk] + g * ep * expf(lv)) * gout;
}

// z = [mu_e + eps_e * exp(lv_e) | mu_c + eps_c * exp(lv_c)]   (sample_prior :345-351; same expression as in tail_core_kernel)
__global__ __launch_bounds__(256) void sample_z_kernel(const float* __restrict__ lat, const float* __restrict__ eps_e,
                                                       const float* __restrict__ eps_c, int B, int D, float* __restrict__ z) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * 2 * D) return;
  const int b = e / (2 * D), k = e - b * 2 * D;
  const float* row = lat + (long)b * 4 * D;
  z[e] = (k < D) ? row[k] + eps_e[k] * expf(row[D + k]) : row[2 * D + (k - D)] + eps_c[k - D] * expf(row[3 * D + (k - D)]);
}

// eval-mode pair probabilities (get_pair_preds :265-282): sigmoid(w . [mu_e + eps_e e^lv_e, mu_c + eps_c e^lv_c] + b)
__global__ __launch_bounds__(256) void pair_prob_kernel(const float* __restrict__ lat, const float* __restrict__ eps_e,
                                                        const float* __restrict__ eps_c, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int B, int D, float* __restrict__ prob) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float* row = lat + (long)b * 4 * D;
  float s = bias[0];
  for (int k = 0; k < D; ++k) {
    s = fmaf(w[k], row[k] + eps_e[k] * expf(row[D + k]), s);
    s = fmaf(w[D + k], row[2 * D + k] + eps_c[k] * expf(row[3 * D + k]), s);
  }
  prob[b] = 1.0f / (1.0f + expf(-s));
}

}  // namespace carel

using namespace carel;

static size_t align_up(size_t x) { return (x + 63) & ~(size_t)63; }
constexpr int MMD_MAX_BLOCKS = 1024;          // mmd_global_kernel: 32 rows per block -> up to 16 384 samples per side

struct TailWork {     // carve-up of the caller's f32 workspace
  float* dz_core; float* dlat_direct; float* dlat; float* dpooled; float* dpre; float* part; float* rowstat; float* dz_part;
  float* dcls; float* dgpart; float* pair_dead; float* mmd_part; float* dz_mmd;
  size_t total;
};
static TailWork carve(float* base, int B, int D, int V) {
  TailWork w; size_t o = 0;
  const int chunks = (V + DEC_J - 1) / DEC_J;
  auto take = [&](size_t n) { float* p = base ? base + o : nullptr; o += align_up(n); return p; };
  w.dz_core = take((size_t)B * 2 * D); w.dlat_direct = take((size_t)B * 4 * D); w.dlat = take((size_t)B * 4 * D);
  w.dpooled = take((size_t)B * TH); w.dpre = take((size_t)B * TH); w.part = take((size_t)chunks * B * 2);
  w.rowstat = take((size_t)B * 4); w.dz_part = take((size_t)chunks * B * 2 * D); w.dcls = take((size_t)B * TH);
  w.dgpart = take((size_t)((TH + DG_CHUNK - 1) / DG_CHUNK) * B * TH);
  w.pair_dead = take(16);
  w.mmd_part = take(4 * MMD_MAX_BLOCKS); w.dz_mmd = take((size_t)B * 2 * D);
  w.total = o;
  return w;
}

extern "C" int64_t carel_tail_workspace_floats(int32_t batch, int32_t ec_dim, int32_t bow_dim) {
  return (int64_t)carve(nullptr, batch, ec_dim, bow_dim).total;
}

static int tail_check(const carel_tail_args* a, const char* who) {
  if (!a) return set_error(CAREL_ERR_ARG, "%s: null args", who);
  if (a->hidden != TH) return set_error(CAREL_ERR_SHAPE, "%s: hidden must be %d", who, TH);
  if (a->batch < 1 || a->seq_len < 1) return set_error(CAREL_ERR_SHAPE, "%s: bad batch/seq_len", who);
  if (a->ec_dim < 1 || a->ec_dim > 32 || a->e_classes < 1 || a->e_classes > 8)
    return set_error(CAREL_ERR_SHAPE, "%s: ec_dim must be <= 32 and e_num_class <= 8", who);
  if (!a->x_last_f32 || !a->pooler_w || !a->pooler_b || !a->pooled || !a->lat)
    return set_error(CAREL_ERR_ARG, "%s: null tensor", who);
  for (int i = 0; i < 4; ++i) if (!a->head_w[i] || !a->head_b[i]) return set_error(CAREL_ERR_ARG, "%s: null latent head", who);
  return CAREL_OK;
}

// pooler + latent heads:  pooled [B,768], lat [B,4D]
extern "C" int carel_tail_latents(const carel_tail_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = tail_check(a, "carel_tail_latents");
  if (rc) return rc;
  PtrSet4 pp; for (int i = 0; i < 4; ++i) { pp.w[i] = nullptr; pp.b[i] = nullptr; }
  pp.w[0] = (const float*)a->pooler_w; pp.b[0] = (const float*)a->pooler_b;
  auto sample_groups = [&](int ncols) { int g = (4096 + ncols - 1) / ncols;      /* ~4 k waves: each walks its samples four at a time, one round trip to memory per trip */ const int mx = (a->batch + 3) / 4; g = g > mx ? mx : g; return g < 1 ? 1 : g; };
  hipLaunchKernelGGL(rowvec_linear_kernel<1>, dim3(TH / 4, sample_groups(TH)), dim3(256), 0, stream, (const float*)a->x_last_f32,
                     (long)a->seq_len * TH, (const int*)a->cls_rows, a->batch, TH, TH, pp, (float*)a->pooled, (long)TH);
  PtrSet4 hp; for (int i = 0; i < 4; ++i) { hp.w[i] = (const float*)a->head_w[i]; hp.b[i] = (const float*)a->head_b[i]; }
  const int N = 4 * a->ec_dim;
  hipLaunchKernelGGL(rowvec_linear_kernel<0>, dim3((N + 3) / 4, sample_groups(N)), dim3(256), 0, stream, (const float*)a->pooled, (long)TH,
                     (const int*)nullptr, a->batch, N, a->ec_dim, hp, (float*)a->lat, (long)N);
  if (a->eps_e && a->eps_c && a->z)      // the sampled embeddings, for callers that exchange them before carel_tail_losses (data parallel)
    hipLaunchKernelGGL(sample_z_kernel, dim3((a->batch * 2 * a->ec_dim + 255) / 256), dim3(256), 0, stream, (const float*)a->lat,
                       (const float*)a->eps_e, (const float*)a->eps_c, a->batch, a->ec_dim, (float*)a->z);
  return check_launch("tail latents");
}

extern "C" int carel_pair_probs(const void* lat, const void* eps_e, const void* eps_c, const void* pair_w, const void* pair_b,
                                int32_t batch, int32_t ec_dim, void* prob, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!lat || !eps_e || !eps_c || !pair_w || !pair_b || !prob || batch < 1) return set_error(CAREL_ERR_ARG, "carel_pair_probs: bad arguments");
  hipLaunchKernelGGL(pair_prob_kernel, dim3((batch + 255) / 256), dim3(256), 0, stream, (const float*)lat, (const float*)eps_e,
                     (const float*)eps_c, (const float*)pair_w, (const float*)pair_b, batch, ec_dim, (float*)prob);
  return check_launch("pair_prob_kernel");
}

// losses + gradients of everything after the latents.  Needs carel_tail_latents() first.
extern "C" int carel_tail_losses(const carel_tail_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = tail_check(a, "carel_tail_losses");
  if (rc) return rc;
  const int B = a->batch, D = a->ec_dim, V = a->bow_dim;
  if (V < 1) return set_error(CAREL_ERR_SHAPE, "carel_tail_losses: bow_dim must be positive");
  if (2 * D > DEC_MAXD2) return set_error(CAREL_ERR_SHAPE, "carel_tail_losses: 2*ec_dim must be <= %d", DEC_MAXD2);
  if (B > 1024) return set_error(CAREL_ERR_SHAPE, "carel_tail_losses: batch must be <= 1024 per rank");
  if (B < 2 && !a->z_global) return set_error(CAREL_ERR_SHAPE, "carel_tail_losses: batch must be >= 2 (MMD divides by n(n-1))");
  if (!a->emo_w || !a->emo_b || !a->cau_w || !a->cau_b || !a->pair_w || !a->pair_b || !a->dec_w || !a->dec_b ||
      !a->emo_labels || !a->cau_labels || !a->pair_labels || !a->bow || !a->eps_e || !a->eps_c || !a->z || !a->terms ||
      !a->work || !a->d_emo_w || !a->d_emo_b || !a->d_cau_w || !a->d_cau_b || !a->d_pair_w || !a->d_pair_b || !a->d_dec_w ||
      !a->d_dec_b)
    return set_error(CAREL_ERR_ARG, "carel_tail_losses: null tensor");
  TailWork w = carve((float*)a->work, B, D, V);
  TailCore c;
  c.B = B; c.D = D; c.EC = a->e_classes; c.lat = (const float*)a->lat; c.eps_e = (const float*)a->eps_e; c.eps_c = (const float*)a->eps_c;
  c.emo_w = (const float*)a->emo_w; c.emo_b = (const float*)a->emo_b; c.cau_w = (const float*)a->cau_w; c.cau_b = (const float*)a->cau_b;
  c.pair_w = (const float*)a->pair_w; c.pair_b = (const float*)a->pair_b;
  c.emo_labels = (const long*)a->emo_labels; c.cau_labels = (const float*)a->cau_labels; c.pair_labels = (const float*)a->pair_labels;
  c.w_mmd = a->w_mmd; c.w_emo = a->w_emo; c.w_cau = a->w_cau; c.w_pair = a->w_pair; c.kl_w = a->kl_weight; c.ls = a->label_smoothing;
  c.dis_mode = a->dis_mode; c.emo_bce = a->emo_bce;
  if (c.dis_mode < 0 || c.dis_mode > 2) return set_error(CAREL_ERR_ARG, "carel_tail_losses: dis_mode must be 0 (MMD), 1 (HSIC) or 2 (none)");
  if (c.emo_bce && a->e_classes != 1) return set_error(CAREL_ERR_ARG, "carel_tail_losses: the BCE emotion head has exactly one logit");
  if (c.dis_mode == 1 && a->z_global) return set_error(CAREL_ERR_ARG, "carel_tail_losses: HSIC has no global-batch mode");
  c.d_emo = make_dropout(a->drop_seed, 100u, a->drop_p, a->drop_row_offset * (uint32_t)D);
  c.d_cau = make_dropout(a->drop_seed, 101u, a->drop_p, a->drop_row_offset * (uint32_t)D);
  c.d_pair = make_dropout(a->drop_seed, 102u, a->drop_p, a->drop_row_offset * (uint32_t)(2 * D));
  c.alpha = a->mmd_alpha; c.mmd_eps = a->mmd_eps;
  c.label_sum_override = (const float*)a->global_label_sum; c.n_override = (float)a->global_n;
  c.label_sum_ranks = a->global_label_ranks > 1 ? a->global_label_ranks : 1;
  if (c.label_sum_ranks > 1 && a->global_rank_stride <= 0)
    return set_error(CAREL_ERR_ARG, "carel_tail_losses: global_label_ranks > 1 needs global_rank_stride");
  c.z_global = (const float*)a->z_global; c.n_global = a->global_n; c.row_offset = a->global_row_offset;
  c.rank_stride = a->global_rank_stride > 0 ? a->global_rank_stride : B * 2 * D;
  if (c.z_global && a->global_rank_stride > 0 && (a->global_rank_stride < B * 2 * D || c.n_global % B))
    return set_error(CAREL_ERR_ARG, "carel_tail_losses: global_rank_stride needs global_n to be a multiple of batch and stride >= batch*2*ec_dim");
  c.mmd_grad_scale = a->mmd_grad_scale > 0.f ? a->mmd_grad_scale : 1.f;
  if (c.z_global && (c.n_global < 2 || c.row_offset < 0 || c.row_offset + B > c.n_global))
    return set_error(CAREL_ERR_ARG, "carel_tail_losses: inconsistent global batch description");
  c.z = (float*)a->z; c.terms = (float*)a->terms; c.dz = w.dz_core; c.dlat_direct = w.dlat_direct;
  c.d_emo_w = (float*)a->d_emo_w; c.d_emo_b = (float*)a->d_emo_b; c.d_cau_w = (float*)a->d_cau_w; c.d_cau_b = (float*)a->d_cau_b;
  c.d_pair_w = (float*)a->d_pair_w; c.d_pair_b = (float*)a->d_pair_b; c.pair_dead = w.pair_dead;
  const int nm = (c.z_global && c.dis_mode != 0) ? c.n_global : B;        // the global-batch MMD runs in its own kernel
  const int n_hw = a->e_classes * D + a->e_classes + D + 1 + 2 * D + 1;
  const size_t lds = sizeof(float) * (64 + 64 + (size_t)2 * B * 2 * D + (size_t)B * 8 + (size_t)B * 2 + (size_t)B * 4 * D + (size_t)B * 16 +
                                      ((n_hw + 3) & ~3) + ((2 * nm + 3) & ~3) + (size_t)2 * nm * (D | 1) + 2 * (size_t)B);
  if (lds > 160 * 1024) return set_error(CAREL_ERR_SHAPE, "carel_tail_losses: batch too large for the single-workgroup tail (%zu B LDS)", lds);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)tail_core_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_tail_losses: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  c.mmd_part = nullptr; c.mmd_nblk = 0; c.dz_mmd = nullptr;
  // (every shape check comes BEFORE the fork below -- ADVICE r03: an error return between fork and join would leave the side stream un-joined)
  if (c.z_global && c.dis_mode == 0 && (2 * c.n_global + 31) / 32 > MMD_MAX_BLOCKS)
    return set_error(CAREL_ERR_SHAPE, "carel_tail_losses: global batch too large for the MMD partial buffer");
  // The loss kernel and the decoder passes only share their input z and meet again in decoder_total_kernel (terms[0]) and tail_dlat_kernel:
  // fork the former onto the side stream (the profiling hook keeps the serial order)
  hipStream_t core_stream = stream;
  TailEvents* tev = nullptr;
  // after the fork every early return first joins the side stream back into the caller's stream (whatever was enqueued there completes
  // before the caller's next kernel); the fork / join events are one pair per device: carel_tail_losses with serial = 0 is NOT re-entrant
  // per device (two host threads, or two models interleaving calls on one device, must pass serial = 1)
  auto join_on_error = [&](int code) -> int {
    if (tev && core_stream != stream) {
      (void)hipEventRecord(tev->join, core_stream);
      (void)hipStreamWaitEvent(stream, tev->join, 0);
    }
    return code;
  };
  if (g_tail_overlap && !a->serial && !g_tail_prof) {
    hipStream_t side = (hipStream_t)carel_side_stream(0);     // the weight-gradient stream (idle here: the backward pass has not started); a THIRD stream in play made every later kernel slower
    tev = side ? tail_events() : nullptr;
    if (tev) {
      hipLaunchKernelGGL(sample_z_kernel, dim3((B * 2 * D + 255) / 256), dim3(256), 0, stream, c.lat, c.eps_e, c.eps_c, B, D, c.z);
      if (hipEventRecord(tev->fork, stream) != hipSuccess || hipStreamWaitEvent(side, tev->fork, 0) != hipSuccess)
        return set_error(CAREL_ERR_HIP, "carel_tail_losses: event fork failed");
      core_stream = side;
      c.z = nullptr;
    }
  }
  if (c.z_global && c.dis_mode == 0) {
    MmdGlobalArgs g;
    g.zg = c.z_global; g.n = c.n_global; g.B = B; g.D = D; g.rank_stride = c.rank_stride; g.row_offset = c.row_offset;
    g.alpha = c.alpha; g.eps = c.mmd_eps; g.gscale = -c.w_mmd * c.mmd_grad_scale; g.part = w.mmd_part; g.dz = w.dz_mmd;
    const int nblk = (2 * c.n_global + 31) / 32;
    if (D == 24) hipLaunchKernelGGL(mmd_global_kernel<24>, dim3(nblk), dim3(256), 0, core_stream, g);
    else hipLaunchKernelGGL(mmd_global_kernel<32>, dim3(nblk), dim3(256), 0, core_stream, g);
    if ((rc = check_launch("mmd_global_kernel"))) return join_on_error(rc);
    c.mmd_part = w.mmd_part; c.mmd_nblk = nblk; c.dz_mmd = w.dz_mmd;
  }
  c.prof = g_tail_prof;
  hipLaunchKernelGGL(tail_core_kernel, dim3(1), dim3(1024), lds, core_stream, c);
  if ((rc = check_launch("tail_core_kernel"))) return join_on_error(rc);
  if (tev && hipEventRecord(tev->join, core_stream) != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_tail_losses: event record failed");

  DecArgs d;
  d.B = B; d.D2 = 2 * D; d.V = V; d.z = (const float*)a->z; d.w = (const float*)a->dec_w; d.b = (const float*)a->dec_b;
  d.bow = (const float*)a->bow; d.ls = a->label_smoothing; d.part = w.part; d.rowstat = w.rowstat; d.dz_part = w.dz_part;
  d.dw = (float*)a->d_dec_w; d.db = (float*)a->d_dec_b; d.gscale = 1.0f / ((float)B * (float)V);
  const int chunks = (V + DEC_J - 1) / DEC_J;
  // LDS: z + red + weights + tile [B][DEC_J] (+ pass 2: second tile; pass 3: the other groups' dW/db partials)
  const size_t lds0 = sizeof(float) * ((size_t)B * 2 * D + 16 + (size_t)DEC_J * (2 * D + 1) + (size_t)B * DEC_J);
  int G = DEC_MAXG;
  auto lds3_of = [&](int g) { return lds0 + sizeof(float) * (size_t)(g - 1) * DEC_J * (2 * D + 2); };
  while (G > 1 && (lds3_of(G) > 160 * 1024 || (B + G - 1) / G < 4)) G >>= 1;
  const size_t lds1 = lds0, lds2 = lds0 + sizeof(float) * (size_t)B * DEC_J, lds3 = lds3_of(G);
  const size_t ldsmax = lds2 > lds3 ? lds2 : lds3;
  if (ldsmax > 160 * 1024) return set_error(CAREL_ERR_SHAPE, "carel_tail_losses: batch too large for the decoder kernel");
  if (ldsmax > 64 * 1024) {
    hipError_t e = hipSuccess;
    const void* fns[6] = {(const void*)decoder_kernel<1, 48>, (const void*)decoder_kernel<2, 48>, (const void*)decoder_kernel<3, 48>,
                          (const void*)decoder_kernel<1, 0>, (const void*)decoder_kernel<2, 0>, (const void*)decoder_kernel<3, 0>};
    for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsmax);
    if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_tail_losses: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
#define DEC_LAUNCH(PASS, LDS)                                                                              \
  do {                                                                                                  \
    if (2 * D == 48) hipLaunchKernelGGL((decoder_kernel<PASS, 48>), dim3(chunks), dim3(DEC_J * G), LDS, stream, d); \
    else hipLaunchKernelGGL((decoder_kernel<PASS, 0>), dim3(chunks), dim3(DEC_J * G), LDS, stream, d);      \
  } while (0)
  DEC_LAUNCH(1, lds1);
  hipLaunchKernelGGL(decoder_combine_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, (const float*)w.part, chunks, B, 1, w.rowstat);
  DEC_LAUNCH(2, lds2);
  hipLaunchKernelGGL(decoder_combine_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, (const float*)w.part, chunks, B, 2, w.rowstat);
  DEC_LAUNCH(3, lds3);
#undef DEC_LAUNCH
  if ((rc = check_launch("decoder kernels"))) return rc;
  // d(loss)/d lat, unscaled (grad_output is applied in carel_tail_backward)
  // sum the decoder's per-chunk dz partials into slot 0 (in place is safe: block c only reads column c of every part)
  hipLaunchKernelGGL(reduce_parts16_kernel, dim3((B * 2 * D + 15) / 16), dim3(256), 0, stream, (const float*)w.dz_part, w.dz_part, B * 2 * D, chunks);
  if (tev && hipStreamWaitEvent(stream, tev->join, 0) != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_tail_losses: event join failed");
  hipLaunchKernelGGL(decoder_total_kernel, dim3(1), dim3(256), 0, stream, (const float*)w.rowstat, B, d.gscale, (float*)a->terms);   // reads terms[0] of the loss kernel
  hipLaunchKernelGGL(tail_dlat_kernel, dim3((B * 2 * D + 255) / 256), dim3(256), 0, stream, (const float*)w.dz_core, (const float*)w.dz_part,
                     1, (const float*)w.dlat_direct, (const float*)a->lat, (const float*)a->eps_e, (const float*)a->eps_c, B, D, 1.0f, w.dlat);
  return check_launch("tail_dlat_kernel");
}

// Back-propagate d(loss)/d lat through the latent heads and the pooler into the encoder output.
// grad_out_dev (device f32 scalar, or NULL for 1.0) is the upstream gradient of the loss: it scales dlat
// here; the classifier / decoder gradients written by carel_tail_losses are scaled by the caller with
// carel_scale_f32 (they are linear in it).
extern "C" int carel_scale_f32(void* x, int64_t n, const void* scale_dev, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !scale_dev || n <= 0) return set_error(CAREL_ERR_ARG, "carel_scale_f32: bad arguments");
  long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scale_inplace_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (float*)x, (long)n, (const float*)scale_dev);
  return check_launch("scale_inplace_kernel");
}

// Bag-of-words targets shipped as (row, column, value) triples (3-30 entries per clause pair against V = 23 771 columns:
// ~10 KB instead of the 6 MB dense block `bow_reps` of ref :829) and expanded on the device.  `trip` = int32 [nnz] rows,
// int32 [nnz] columns, f32 [nnz] values back to back; entries of one batch are distinct (ECPEDataset builds one entry per
// vocabulary word of the pair), so plain stores suffice.
__global__ void bow_scatter_kernel(const int* __restrict__ rows, const int* __restrict__ cols, const float* __restrict__ vals, int nnz,
                                   float* __restrict__ out, int B, int V) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  const int r = rows[i], c = cols[i];
  if (r >= 0 && r < B && c >= 0 && c < V) out[(long)r * V + c] = vals[i];
}
extern "C" int carel_bow_expand(const void* trip, int32_t nnz, void* out, int32_t B, int32_t V, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!out || B < 1 || V < 1 || nnz < 0 || (nnz > 0 && !trip)) return set_error(CAREL_ERR_ARG, "carel_bow_expand: bad arguments");
  hipError_t e = hipMemsetAsync(out, 0, (size_t)B * V * 4, stream);
  if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_bow_expand: memset: %s", hipGetErrorString(e));
  if (nnz == 0) return CAREL_OK;
  const int* rows = (const int*)trip;
  hipLaunchKernelGGL(bow_scatter_kernel, dim3((nnz + 255) / 256), dim3(256), 0, stream, rows, rows + nnz, (const float*)(rows + 2 * (long)nnz), nnz,
                     (float*)out, B, V);
  return check_launch("bow_scatter_kernel");
}

extern "C" int64_t carel_tail_pair_dead_offset(int32_t batch, int32_t ec_dim, int32_t bow_dim) {
  TailWork w = carve((float*)nullptr + 1, batch, ec_dim, bow_dim);   // offsets relative to a fake base
  return (int64_t)(w.pair_dead - ((float*)nullptr + 1));
}

extern "C" int carel_tail_profile(void* dev_i64_x16) { g_tail_prof = (long long*)dev_i64_x16; return CAREL_OK; }

extern "C" int carel_tail_backward(const carel_tail_args* a, const void* grad_out_dev, void* stream_) {
  return carel_tail_backward_dz(a, grad_out_dev, nullptr, stream_);
}

// dlat += [dz_e, dz_e*eps_e*exp(lv_e), dz_c, dz_c*eps_c*exp(lv_c)] for an extra gradient dz on the sampled embeddings
__global__ __launch_bounds__(256) void tail_add_dz_kernel(const float* __restrict__ dz, const float* __restrict__ lat,
                                                          const float* __restrict__ eps_e, const float* __restrict__ eps_c, int B, int D,
                                                          float* __restrict__ dlat) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * 2 * D) return;
  const int b = e / (2 * D), k = e - b * 2 * D, side = k >= D, kk = k - side * D;
  const float g = dz[e];
  const float ep = side ? eps_c[kk] : eps_e[kk];
  const float lv = lat[(long)b * 4 * D + (2 * side + 1) * D + kk];
  dlat[(long)b * 4 * D + 2 * side * D + kk] += g;
  dlat[(long)b * 4 * D + (2 * side + 1) * D + kk] += g * ep * expf(lv);
}

extern "C" int carel_tail_backward_dz(const carel_tail_args* a, const void* grad_out_dev, const void* dz_extra, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = tail_check(a, "carel_tail_backward");
  if (rc) return rc;
  if (!a->work || !a->dx_last_f32 || !a->d_pooler_w || !a->d_pooler_b) return set_error(CAREL_ERR_ARG, "carel_tail_backward: null tensor");
  const int B = a->batch, D = a->ec_dim, N = 4 * D;
  TailWork w = carve((float*)a->work, B, D, a->bow_dim);
  if (grad_out_dev) {
    hipLaunchKernelGGL(scale_inplace_kernel, dim3((B * N + 255) / 256), dim3(256), 0, stream, w.dlat, (long)B * N, (const float*)grad_out_dev);
  }
  if (dz_extra) {
    if (!a->lat || !a->eps_e || !a->eps_c) return set_error(CAREL_ERR_ARG, "carel_tail_backward_dz: lat / eps needed");
    hipLaunchKernelGGL(tail_add_dz_kernel, dim3((B * 2 * D + 255) / 256), dim3(256), 0, stream, (const float*)dz_extra, (const float*)a->lat,
                       (const float*)a->eps_e, (const float*)a->eps_c, B, D, w.dlat);
  }
  PtrSet4 hp; OutSet4 ho;
  for (int i = 0; i < 4; ++i) { hp.w[i] = (const float*)a->head_w[i]; hp.b[i] = nullptr; ho.w[i] = (float*)a->d_head_w[i]; ho.b[i] = (float*)a->d_head_b[i]; }
  if (ho.w[0] && ho.w[1] && ho.w[2] && ho.w[3])
    hipLaunchKernelGGL(rowvec_wgrad_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, (const float*)w.dlat, (long)N, (const float*)a->pooled,
                       (long)TH, (const int*)nullptr, B, N, D, ho);
  const int hc = (N + DG_CHUNK - 1) / DG_CHUNK, pc = (TH + DG_CHUNK - 1) / DG_CHUNK;
  const long bt = (long)B * TH;
  hipLaunchKernelGGL(rowvec_dgrad_kernel<0>, dim3((B + 3) / 4, hc), dim3(256), 0, stream, (const float*)w.dlat, (long)N, B, N, D, hp,
                     (const float*)nullptr, (float*)nullptr, w.dgpart);
  hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((bt / 4 + 255) / 256)), dim3(256), 0, stream, (const float*)w.dgpart, w.dpooled, bt, hc);
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
  return check_launch("tail backward");
}
