"""Detection losses of upstream coperception/utils/loss.py + CoDetModule.FaFModule.loss_calculator (absent from
/root/reference; README.md:101 names the training scripts that use them).  Build-owned restatement:

  classification: softmax focal loss (alpha = 0.25 on the foreground class, gamma = 2) over every anchor,
  localisation:   smooth-L1 (sigma = 3  ->  beta = 1/9) over the anchors selected by reg_loss_mask,
  both summed and divided by the number of positive anchors (at least 1);  loss = cls + loc.
  normalizer="batch" (oracle/ASSUMPTIONS.md row 49, the second reading; configs.Config.loss_normalizer): divided by the number of maps instead.
"""
import torch
import torch.nn.functional as F

ALPHA, GAMMA, SIGMA = 0.25, 2.0, 3.0


def focal_loss(cls_logits, label_one_hot):
    """cls_logits (N, M, 2) fp32, label_one_hot (N, M, 2) -> per-anchor loss (N, M)."""
    logp = F.log_softmax(cls_logits, dim=-1)
    p_t = (logp.exp() * label_one_hot).sum(-1)
    alpha_t = label_one_hot[..., 1] * ALPHA + label_one_hot[..., 0] * (1.0 - ALPHA)
    return -alpha_t * (1.0 - p_t).pow(GAMMA) * (logp * label_one_hot).sum(-1)


class _DetLossHip(torch.autograd.Function):
    """The whole loss on the hand-written kernels (csrc/det_loss.hip): forward = one pass over the five tensors + a finish launch, backward = one
    pass writing both gradients; as PyTorch ops the same arithmetic is ~45 launches over 31 + 94 MB tensors per 10-map step (0.6 ms of 5.8)."""

    @staticmethod
    def forward(ctx, cls, loc, lab, tgt, mask):
        from .. import ops
        ctx.set_materialize_grads(False)      # the gradients of unused outputs arrive as None, not as zero tensors (a fill launch each)
        out = ops.det_loss_forward(cls, lab, loc, tgt, mask, ALPHA, 1.0 / (SIGMA * SIGMA))
        ctx.save_for_backward(cls, loc, lab, tgt, mask, out)
        n_pos = out[3]
        ctx.mark_non_differentiable(n_pos)
        return out[0], out[1], out[2], n_pos

    @staticmethod
    def backward(ctx, g_loss, g_cls, g_loc, _g_npos=None):
        from .. import ops
        cls, loc, lab, tgt, mask, out = ctx.saved_tensors
        gs = [None if g is None else g.to(torch.float32).contiguous() for g in (g_loss, g_cls, g_loc)]
        dcls, dloc = ops.det_loss_backward(cls, lab, loc, tgt, mask, ALPHA, 1.0 / (SIGMA * SIGMA), out, *gs)
        return dcls, dloc, None, None, None


def _hip_loss_ok(cls, loc, labels, reg_targets, reg_loss_mask):
    from .. import tuning
    if not (cls.is_cuda and tuning.get("TRAIN_HIP") != 0 and tuning.get("TRAIN_LOSS_HIP") != 0):
        return False
    if not (cls.dtype == loc.dtype == labels.dtype == reg_targets.dtype == torch.float32 and reg_loss_mask.dtype in (torch.bool, torch.uint8)):
        return False
    n = cls.numel() // 2
    return (cls.shape[-1] == 2 and labels.numel() == 2 * n and loc.numel() == 6 * n and reg_targets.numel() == 6 * n and reg_loss_mask.numel() == n
            and loc.shape[-1] == 6)


def detection_loss(result, labels, reg_targets, reg_loss_mask, normalizer="positives"):
    """result: {'cls' (N, X*Y*A, 2), 'loc' (N, X, Y, A, 1, 6)};  labels (N, X, Y, A, 2);  reg_targets (N, X, Y, A, 1, 6);
    reg_loss_mask (N, X, Y, A, 1) bool  ->  (loss, cls_loss, loc_loss) scalars.  On the MI355X with the HIP training engine (TRAIN_HIP, and
    TRAIN_LOSS_HIP != 0) and fp32 operands: three launches of csrc/det_loss.hip; otherwise the PyTorch ops below (the specification)."""
    if normalizer not in ("positives", "batch"):
        raise ValueError("normalizer must be 'positives' or 'batch'")
    n = result["cls"].shape[0]
    if _hip_loss_ok(result["cls"], result["loc"], labels, reg_targets, reg_loss_mask):
        loss, cls_loss, loc_loss, n_pos = _DetLossHip.apply(result["cls"].contiguous(), result["loc"].contiguous(), labels.contiguous(),
                                                            reg_targets.contiguous(), reg_loss_mask.contiguous())
        if normalizer == "batch":     # the kernels divide by the positives: rescale by n_pos / n (a device scalar: no synchronisation, capturable)
            k = n_pos / float(n)
            return loss * k, cls_loss * k, loc_loss * k
        return loss, cls_loss, loc_loss
    lab = labels.reshape(n, -1, 2).to(result["cls"].dtype)
    n_pos = lab[..., 1].sum().clamp(min=1.0) if normalizer == "positives" else torch.tensor(float(n), dtype=result["cls"].dtype, device=result["cls"].device)
    cls_loss = focal_loss(result["cls"], lab).sum() / n_pos
    # masked SUM instead of boolean indexing: the same terms, but no data-dependent shape -- no host synchronisation, and the step can be
    # captured in a hipGraph (train/graph_step.py).  Unselected anchors contribute exactly 0 (their terms are finite: targets are 0 there).
    m = reg_loss_mask.reshape(n, -1, 1).to(result["loc"].dtype)
    per = F.smooth_l1_loss(result["loc"].reshape(n, -1, 6), reg_targets.reshape(n, -1, 6).to(result["loc"].dtype), reduction="none",
                           beta=1.0 / (SIGMA * SIGMA))
    loc_loss = (per * m).sum() / n_pos
    return cls_loss + loc_loss, cls_loss, loc_loss
