"""The scoring wrappers' metrics on the device: ROC AUC and the ROC curve as sklearn.metrics computes them
(TwoWL/model/train.py:66-67, SEAL/config/subgraph.py:58-62 call roc_auc_score / roc_curve on the sigmoid scores).
Model-selection arithmetic on [B] score vectors, not part of the aggregation path: plain torch tensor ops."""
import torch


def rank_auc(y, s):
    """ROC AUC with average ranks for tied scores (== sklearn.metrics.roc_auc_score(y, s) for binary y)."""
    s = s.reshape(-1).double()
    y = y.reshape(-1)
    order = torch.argsort(s, stable=True)
    ss = s[order]
    first = torch.ones_like(ss, dtype=torch.bool)
    first[1:] = ss[1:] != ss[:-1]
    grp = torch.cumsum(first, 0) - 1
    cnt = torch.bincount(grp)
    start = torch.cumsum(cnt, 0) - cnt
    avg = (start.double() + (cnt.double() - 1) / 2 + 1)[grp]
    ranks = torch.empty_like(avg)
    ranks[order] = avg
    y = y.bool()
    npos, nneg = int(y.sum()), int((~y).sum())
    return float((ranks[y].sum() - npos * (npos + 1) / 2) / (npos * nneg))


def roc_curve(y, s):
    """(fpr, tpr, thresholds) == sklearn.metrics.roc_curve(y, s) with its defaults (pos_label 1, drop_intermediate):
    scores sorted descending (stable), one point per distinct score, collinear points dropped, (0, 0) prepended with
    threshold inf."""
    s = s.reshape(-1).double()
    y = y.reshape(-1).double()
    order = torch.argsort(s, descending=True, stable=True)
    ss, yy = s[order], y[order]
    distinct = torch.nonzero(ss[1:] != ss[:-1]).reshape(-1)
    idx = torch.cat([distinct, torch.tensor([ss.numel() - 1], device=s.device)])
    tps = torch.cumsum(yy, 0)[idx]
    fps = 1.0 + idx.double() - tps
    thr = ss[idx]
    if tps.numel() > 2:                                         # drop_intermediate: keep the corners only
        d2 = (torch.diff(fps, 2) != 0) | (torch.diff(tps, 2) != 0)
        keep = torch.cat([torch.tensor([True], device=s.device), d2, torch.tensor([True], device=s.device)])
        tps, fps, thr = tps[keep], fps[keep], thr[keep]
    zero = torch.zeros(1, dtype=torch.float64, device=s.device)
    tps, fps = torch.cat([zero, tps]), torch.cat([zero, fps])
    thr = torch.cat([torch.full((1,), float("inf"), dtype=torch.float64, device=s.device), thr])
    return fps / fps[-1], tps / tps[-1], thr
