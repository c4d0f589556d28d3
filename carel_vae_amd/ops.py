"""Thin torch-tensor wrappers over the C ABI (include/carel_hip.h).  PyTorch is used for device memory
and streams only; every numeric operation happens inside libcarel_hip.so.  No fallbacks."""
import ctypes as C
import math

import torch

from . import _lib as L

H = 768
NH, HD = 12, 64


def _chk_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.CarelError("carel_vae_amd kernels need CUDA/HIP tensors; got a %s tensor. "
                               "There is no CPU fallback in the product path." % t.device)


def gemm(A, B, form, epi, M, N, K, *, splits=1, out_bf16=None, out2_bf16=None, out_f32=None, bias=None,
         resid=None, aux=None, drop=(0, 0, 0, 0.0), lda=None, ldb=None, ldc=None, colsum_part=None, colsum_a=None, splitk_ws=None):
    a = L.GemmArgs()
    a.A, a.B = A.data_ptr(), B.data_ptr()
    a.lda = lda if lda is not None else A.stride(0)
    a.ldb = ldb if ldb is not None else B.stride(0)
    a.ldc = ldc if ldc is not None else N
    a.M, a.N, a.K = M, N, K
    a.form, a.epilogue, a.splits = form, epi, splits
    a.out_bf16 = None if out_bf16 is None else out_bf16.data_ptr()
    a.out2_bf16 = None if out2_bf16 is None else out2_bf16.data_ptr()
    a.out_f32 = None if out_f32 is None else out_f32.data_ptr()
    a.bias = None if bias is None else bias.data_ptr()
    a.resid_f32 = None if resid is None else resid.data_ptr()
    a.aux_bf16 = None if aux is None else aux.data_ptr()
    a.drop_seed, a.drop_site, a.drop_idx_offset, a.drop_p = drop
    a.colsum_part = None if colsum_part is None else colsum_part.data_ptr()
    a.colsum_a = None if colsum_a is None else colsum_a.data_ptr()
    a.splitk_ws = None if splitk_ws is None else splitk_ws.data_ptr()
    a.splitk_ws_bytes = 0 if splitk_ws is None else splitk_ws.numel() * splitk_ws.element_size()
    L.check(L.load().carel_gemm_bf16(C.byref(a), L.current_stream()), "carel_gemm_bf16")


def rbf_mmd(s1, s2, alphas, eps=1e-5, ret_matrix=False):
    """MMDStatistic.__call__ forward (ref :547-569).  Returns (mmd[1], kernels or None)."""
    _chk_cuda(s1, s2)
    a = _mmd_args(s1, s2, alphas, eps)
    out = torch.empty(1, device=s1.device, dtype=torch.float32)
    n = s1.shape[0] + s2.shape[0]
    kern = torch.empty((n, n), device=s1.device, dtype=torch.float32) if ret_matrix else None
    a.mmd_out = out.data_ptr()
    a.kernels_out = None if kern is None else kern.data_ptr()
    L.check(L.load().carel_rbf_mmd_fwd(C.byref(a), L.current_stream()), "carel_rbf_mmd_fwd")
    return out, kern


def rbf_mmd_backward(s1, s2, alphas, grad, eps=1e-5):
    a = _mmd_args(s1, s2, alphas, eps)
    g1 = torch.empty((s1.shape[0], s1.shape[1]), device=s1.device, dtype=torch.float32)
    g2 = torch.empty((s2.shape[0], s2.shape[1]), device=s1.device, dtype=torch.float32)
    grad = grad.reshape(1).to(torch.float32).contiguous()
    a.grad_mmd, a.g1, a.g2 = grad.data_ptr(), g1.data_ptr(), g2.data_ptr()
    L.check(L.load().carel_rbf_mmd_bwd(C.byref(a), L.current_stream()), "carel_rbf_mmd_bwd")
    return g1, g2


def _pdist_args(s1, s2, eps):
    if s1.dtype != torch.float32 or s2.dtype != torch.float32:
        raise L.CarelError("pdist: float32 samples required")
    if s1.stride(1) != 1 or s2.stride(1) != 1 or s1.shape[1] != s2.shape[1]:
        raise L.CarelError("pdist: samples must share the feature width and be contiguous along it")
    a = L.PdistArgs()
    a.s1, a.s2, a.ld1, a.ld2 = s1.data_ptr(), s2.data_ptr(), s1.stride(0), s2.stride(0)
    a.n1, a.n2, a.d, a.eps = s1.shape[0], s2.shape[0], s1.shape[1], eps
    return a


def pdist(s1, s2, eps=1e-5):
    """ref :580-589 (norm = 2): [n1, n2] distances."""
    _chk_cuda(s1, s2)
    a = _pdist_args(s1, s2, eps)
    out = torch.empty((s1.shape[0], s2.shape[0]), device=s1.device, dtype=torch.float32)
    a.dist_out = out.data_ptr()
    L.check(L.load().carel_pdist_fwd(C.byref(a), L.current_stream()), "carel_pdist_fwd")
    return out


def pdist_backward(s1, s2, grad, eps=1e-5):
    a = _pdist_args(s1, s2, eps)
    grad = grad.to(torch.float32).contiguous()
    g1 = torch.empty((s1.shape[0], s1.shape[1]), device=s1.device, dtype=torch.float32)
    g2 = torch.empty((s2.shape[0], s2.shape[1]), device=s1.device, dtype=torch.float32)
    a.grad_dist, a.g1, a.g2 = grad.data_ptr(), g1.data_ptr(), g2.data_ptr()
    L.check(L.load().carel_pdist_bwd(C.byref(a), L.current_stream()), "carel_pdist_bwd")
    return g1, g2


def _mmd_args(s1, s2, alphas, eps):
    if s1.dtype != torch.float32 or s2.dtype != torch.float32:
        raise L.CarelError("rbf_mmd: float32 samples required")
    if s1.stride(1) != 1 or s2.stride(1) != 1:
        raise L.CarelError("rbf_mmd: samples must be contiguous along the feature dimension")
    a = L.MmdArgs()
    a.s1, a.s2, a.ld1, a.ld2 = s1.data_ptr(), s2.data_ptr(), s1.stride(0), s2.stride(0)
    a.n1, a.n2, a.d, a.n_alphas, a.eps = s1.shape[0], s2.shape[0], s1.shape[1], len(alphas), eps
    for i, v in enumerate(alphas):
        a.alphas[i] = float(v)
    return a


def kl_anneal_weight(iteration, opt):
    """ref :515-523 (host double arithmetic) gated by :242/:248."""
    if iteration < opt.kl_ann_iterations:
        return (math.tanh((iteration - opt.kl_ann_iterations * 1.5) / (opt.kl_ann_iterations / 3)) + 1) * opt.ec_kl_lambda
    return 1.0


class TailBuffers:
    """Outputs + workspace of the tail calls for one (batch, ec_dim, bow_dim) shape."""

    def __init__(self, B, S, D, EC, V, device, rows=None):
        f = dict(device=device, dtype=torch.float32)
        self.B, self.S, self.D, self.EC, self.V = B, S, D, EC, V
        self.rows = rows if rows is not None else B * S        # rows of the encoder's last hidden state / its gradient
        self.pooled = torch.empty((B, H), **f)
        self.lat = torch.empty((B, 4 * D), **f)
        # z sits at the head of a buffer with 16 spare floats: a data-parallel caller all-gathers the whole thing (z + its
        # label sum in [n]) in ONE collective without packing anything (carel_vae_amd.dp.DataParallel.fill_global)
        self.zpack = torch.zeros(B * 2 * D + 16, **f)
        self.z = self.zpack[:B * 2 * D].view(B, 2 * D)
        self.terms = torch.zeros(16, **f)
        self.work = torch.empty(L.load().carel_tail_workspace_floats(B, D, V), **f)
        self.dx_last = torch.empty((self.rows, H), **f)


def tail_args(buf: TailBuffers, x_last, W, labels, eps_e, eps_c, opt, kl_weight, *, grads=None, drop=(0.0, 0, 0),
              global_label_sum=None, global_n=0, global_row_offset=0, z_global=None, mmd_grad_scale=1.0, global_rank_stride=0, global_label_ranks=0, cls_rows=None,
              n_rows=0):
    """W / grads: dicts keyed by the reference's state_dict names (tail part)."""
    a = L.TailArgs()
    a.batch, a.seq_len, a.hidden, a.ec_dim, a.e_classes, a.bow_dim = buf.B, buf.S, H, buf.D, buf.EC, buf.V
    a.x_last_f32 = x_last.data_ptr()
    a.pooler_w, a.pooler_b = W["encoder.pooler.dense.weight"].data_ptr(), W["encoder.pooler.dense.bias"].data_ptr()
    heads = ("emotion_mu", "emotion_log_var", "cause_mu", "cause_log_var")
    for i, n in enumerate(heads):
        a.head_w[i] = W[n + ".weight"].data_ptr()
        a.head_b[i] = W[n + ".bias"].data_ptr()
    a.emo_w, a.emo_b = W["emotion_classifier.weight"].data_ptr(), W["emotion_classifier.bias"].data_ptr()
    a.cau_w, a.cau_b = W["cause_classifier.weight"].data_ptr(), W["cause_classifier.bias"].data_ptr()
    a.pair_w, a.pair_b = W["pair_classifier.weight"].data_ptr(), W["pair_classifier.bias"].data_ptr()
    a.dec_w, a.dec_b = W["decoder.weight"].data_ptr(), W["decoder.bias"].data_ptr()
    if labels is not None:
        a.emo_labels, a.cau_labels = labels["emo"].data_ptr(), labels["cau"].data_ptr()
        a.pair_labels, a.bow = labels["pair"].data_ptr(), labels["bow"].data_ptr()
    a.eps_e = None if eps_e is None else eps_e.data_ptr()
    a.eps_c = None if eps_c is None else eps_c.data_ptr()
    a.w_mmd, a.w_emo, a.w_cau, a.w_pair = (opt.mmd_loss_weight, opt.emo_mul_loss_weight, opt.cau_mul_loss_weight,
                                            opt.pair_mul_loss_weight)
    a.kl_weight, a.label_smoothing = kl_weight, opt.label_smoothing
    a.drop_p, a.drop_seed, a.drop_row_offset = drop
    a.mmd_alpha, a.mmd_eps = 0.1, 1e-5
    mode = getattr(opt, "disentangle", "mmd")
    if mode not in ("mmd", "hsic", "none", "vi"):
        raise L.CarelError("opt.disentangle must be 'mmd', 'hsic', 'vi' or 'none'")
    a.dis_mode = {"mmd": 0, "hsic": 1, "none": 2, "vi": 2}[mode]      # vi: the CLUB term is added outside the tail (vi_upper)
    if mode == "hsic":
        a.w_mmd = getattr(opt, "hsic_loss_weight", 1.0)      # the HSIC script adds the statistic unweighted
    a.emo_bce = int(getattr(opt, "emotion_head", "ce") == "bce")
    a.global_label_sum = None if global_label_sum is None else global_label_sum.data_ptr()
    a.global_n, a.global_row_offset = global_n, global_row_offset
    a.z_global = None if z_global is None else z_global.data_ptr()
    a.mmd_grad_scale = mmd_grad_scale
    a.global_rank_stride = global_rank_stride
    a.global_label_ranks = global_label_ranks
    a.pooled, a.lat, a.z, a.terms, a.work = (buf.pooled.data_ptr(), buf.lat.data_ptr(), buf.z.data_ptr(),
                                             buf.terms.data_ptr(), buf.work.data_ptr())
    a.dx_last_f32 = buf.dx_last.data_ptr()
    a.cls_rows = None if cls_rows is None else cls_rows.data_ptr()
    a.n_rows = n_rows if n_rows else buf.rows
    if grads is not None:
        a.d_emo_w, a.d_emo_b = grads["emotion_classifier.weight"].data_ptr(), grads["emotion_classifier.bias"].data_ptr()
        a.d_cau_w, a.d_cau_b = grads["cause_classifier.weight"].data_ptr(), grads["cause_classifier.bias"].data_ptr()
        a.d_pair_w, a.d_pair_b = grads["pair_classifier.weight"].data_ptr(), grads["pair_classifier.bias"].data_ptr()
        a.d_dec_w, a.d_dec_b = grads["decoder.weight"].data_ptr(), grads["decoder.bias"].data_ptr()
        for i, n in enumerate(heads):
            a.d_head_w[i] = grads[n + ".weight"].data_ptr() if (n + ".weight") in grads else None
            a.d_head_b[i] = grads[n + ".bias"].data_ptr() if (n + ".bias") in grads else None
        a.d_pooler_w = grads["encoder.pooler.dense.weight"].data_ptr()
        a.d_pooler_b = grads["encoder.pooler.dense.bias"].data_ptr()
    return a


def tail_latents(a):
    L.check(L.load().carel_tail_latents(C.byref(a), L.current_stream()), "carel_tail_latents")


def tail_losses(a):
    L.check(L.load().carel_tail_losses(C.byref(a), L.current_stream()), "carel_tail_losses")


def tail_backward(a, grad_out=None, dz_extra=None):
    """grad_out: device f32 scalar tensor (upstream gradient of the loss) or None for 1.
    dz_extra: optional f32 [B, 2*ec_dim] additional gradient on the sampled embeddings (not scaled by grad_out)."""
    if dz_extra is not None:
        _chk_cuda(dz_extra)
        if dz_extra.dtype != torch.float32 or not dz_extra.is_contiguous() or dz_extra.numel() != a.batch * 2 * a.ec_dim:
            raise L.CarelError("dz_extra must be a contiguous f32 [batch, 2*ec_dim] tensor")
    L.check(L.load().carel_tail_backward_dz(C.byref(a), None if grad_out is None else grad_out.data_ptr(),
                                            None if dz_extra is None else dz_extra.data_ptr(), L.current_stream()),
            "carel_tail_backward_dz")


def _vi_args(z, net):
    _chk_cuda(z, *net)
    if z.dtype != torch.float32 or not z.is_contiguous() or z.dim() != 2 or z.shape[1] % 2:
        raise L.CarelError("z must be a contiguous f32 [batch, 2*ec_dim] tensor")
    D = z.shape[1] // 2
    shapes = [(D, D), (D,), (D, D), (D,)] * 2
    if len(net) != 8 or any(tuple(w.shape) != sh or w.dtype != torch.float32 or not w.is_contiguous() for w, sh in zip(net, shapes)):
        raise L.CarelError("the approximation network is 8 contiguous f32 tensors: (w1 [D,D], b1 [D], w2 [D,D], b2 [D]) x (mu, log_var)")
    a = L.ViArgs()
    a.z, a.batch, a.ec_dim = z.data_ptr(), z.shape[0], D
    for i, w in enumerate(net):
        a.net[i] = w.data_ptr()
    return a


def vi_aprx(z, net):
    """-> (loss [1], [8 gradients of the approximation network]); drl_classifier_ec_vi.py:422-427."""
    a = _vi_args(z, net)
    loss = torch.empty(1, device=z.device, dtype=torch.float32)
    grads = [torch.empty_like(w) for w in net]
    a.loss_out = loss.data_ptr()
    for i, g in enumerate(grads):
        a.d_net[i] = g.data_ptr()
    L.check(L.load().carel_vi_aprx(C.byref(a), L.current_stream()), "carel_vi_aprx")
    return loss, grads


def vi_upper(z, net, perm):
    """-> (CLUB upper bound [1], d bound / d z [B, 2D]); drl_classifier_ec_vi.py:429-440; perm: int32 [B] device tensor."""
    a = _vi_args(z, net)
    _chk_cuda(perm)
    if perm.dtype != torch.int32 or perm.numel() != z.shape[0]:
        raise L.CarelError("perm must be an int32 [batch] tensor")
    loss = torch.empty(1, device=z.device, dtype=torch.float32)
    dz = torch.empty_like(z)
    a.perm, a.loss_out, a.dz = perm.data_ptr(), loss.data_ptr(), dz.data_ptr()
    L.check(L.load().carel_vi_upper(C.byref(a), L.current_stream()), "carel_vi_upper")
    return loss, dz


def bow_expand(trip, nnz, out):
    """trip: device int32 [3 * nnz] (rows, cols, f32 value bits); out: device f32 [B, V] (overwritten)."""
    L.check(L.load().carel_bow_expand(trip.data_ptr() if nnz else None, int(nnz), out.data_ptr(), out.shape[0], out.shape[1], L.current_stream()),
            "carel_bow_expand")
    return out


def scale_(x, scale_dev):
    L.check(L.load().carel_scale_f32(x.data_ptr(), x.numel(), scale_dev.data_ptr(), L.current_stream()), "carel_scale_f32")


def pair_probs(lat, eps_e, eps_c, pair_w, pair_b, D):
    B = lat.shape[0]
    prob = torch.empty(B, device=lat.device, dtype=torch.float32)
    L.check(L.load().carel_pair_probs(lat.data_ptr(), eps_e.data_ptr(), eps_c.data_ptr(), pair_w.data_ptr(),
                                      pair_b.data_ptr(), B, D, prob.data_ptr(), L.current_stream()), "carel_pair_probs")
    return prob


def _hsic_args(x, y, s_x, s_y):
    _chk_cuda(x, y)
    if x.shape != y.shape or x.dtype != torch.float32 or y.dtype != torch.float32 or x.stride(1) != 1 or y.stride(1) != 1:
        raise L.CarelError("hsic: two float32 [m, d] samples of equal shape, contiguous rows")
    a = L.HsicArgs()
    a.x, a.y, a.ldx, a.ldy, a.m, a.d, a.s_x, a.s_y = x.data_ptr(), y.data_ptr(), x.stride(0), y.stride(0), x.shape[0], x.shape[1], s_x, s_y
    return a


def hsic(x, y, s_x=1.0, s_y=1.0):
    a = _hsic_args(x, y, s_x, s_y)
    out = torch.empty(1, device=x.device, dtype=torch.float32)
    a.hsic_out = out.data_ptr()
    L.check(L.load().carel_hsic_fwd(C.byref(a), L.current_stream()), "carel_hsic_fwd")
    return out


def hsic_backward(x, y, grad, s_x=1.0, s_y=1.0):
    a = _hsic_args(x, y, s_x, s_y)
    gx, gy = torch.empty(x.shape, device=x.device, dtype=torch.float32), torch.empty(y.shape, device=x.device, dtype=torch.float32)
    grad = grad.reshape(1).to(torch.float32).contiguous()
    a.grad_hsic, a.gx, a.gy = grad.data_ptr(), gx.data_ptr(), gy.data_ptr()
    L.check(L.load().carel_hsic_bwd(C.byref(a), L.current_stream()), "carel_hsic_bwd")
    return gx, gy
