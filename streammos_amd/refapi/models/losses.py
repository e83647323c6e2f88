"""Training losses of ``AttNet`` (row f2): OHEM cross-entropy (utils/criterion.py:10-28) and the multi-class
Lovasz-softmax surrogate (utils/lovasz_losses.py:147-222, 22-35), restated on plain torch ops."""
import torch
import torch.nn.functional as F


def ohem_cross_entropy(pred, gt, top_ratio=0.2, top_weight=4.0, ignore_index=0):
    """mean CE over all positions + top_weight * mean CE of the hardest top_ratio of them (ignored positions count
    as zero-loss entries, as with ``CrossEntropyLoss(reduce=False, ignore_index=...)``)."""
    per = F.cross_entropy(pred, gt.long(), reduction="none", ignore_index=ignore_index).reshape(-1)
    k = max(int(top_ratio * per.numel()), 1)
    hardest = torch.topk(per, k, largest=True, sorted=False)[0]
    return per.mean() + top_weight * hardest.mean()


def _lovasz_grad(fg_sorted):
    """Gradient of the Lovasz extension of the Jaccard loss w.r.t. sorted errors (lovasz_losses.py:22-35)."""
    total = fg_sorted.sum()
    inter = total - fg_sorted.cumsum(0)
    union = total + (1.0 - fg_sorted).cumsum(0)
    jac = 1.0 - inter / union
    if fg_sorted.numel() > 1:
        jac = torch.cat((jac[:1], jac[1:] - jac[:-1]))
    return jac


def lovasz_softmax(logits, labels, ignore=None):
    """logits [B,C,H,W], labels [B,H,W]; classes='present', per_image=False.  The reference applies the softmax
    itself (lovasz_losses.py:163) and returns 0 when every label is ignored (:159-161)."""
    if ignore is not None and (labels != ignore).sum() == 0:
        return 0
    prob = F.softmax(logits, dim=1)
    c = prob.shape[1]
    prob = prob.permute(0, 2, 3, 1).reshape(-1, c)
    lab = labels.reshape(-1)
    if ignore is not None:
        keep = lab != ignore
        prob, lab = prob[keep], lab[keep]
    if prob.numel() == 0:
        return prob * 0.0
    losses = []
    for cls in range(c):
        fg = (lab == cls).float()
        if fg.sum() == 0:
            continue
        err = (fg - prob[:, cls]).abs()
        err_sorted, perm = torch.sort(err, 0, descending=True)
        losses.append(torch.dot(err_sorted, _lovasz_grad(fg[perm])))
    return sum(losses) / len(losses)
