"""Plain-PyTorch fp32 reference of the TwoWL scoring forward (autograd-capable), reading the
parameters of an ``lpgnn_amd.twowl.LocalWLNet`` instance.  TEST INFRASTRUCTURE: it exists so that a
model can be *trained* (the product is forward-only) and the HIP forward can then be checked for
kernel-substitution AUC parity at a meaningful AUC.  Definitions as in oracle/aggregation.py
(PyG GCNConv / GraphNorm [3P], TwoWL/model/model.py:68-84)."""
import torch
import torch.nn.functional as F


def gcn_conv(x, edge_index, weight, bias):
    n = x.shape[0]
    h = x @ weight.t()
    keep = edge_index[0] != edge_index[1]
    loops = torch.arange(n, device=x.device)
    src = torch.cat([edge_index[0][keep], loops])
    dst = torch.cat([edge_index[1][keep], loops])
    deg = torch.zeros(n, device=x.device).index_add_(0, dst, torch.ones_like(dst, dtype=torch.float))
    dinv = deg.pow(-0.5)
    dinv[torch.isinf(dinv)] = 0
    w = dinv[src] * dinv[dst]
    out = torch.zeros_like(h).index_add_(0, dst, h[src] * w[:, None])
    return out + bias


def graph_norm(x, norm):
    mean = x.mean(0, keepdim=True)
    o = x - mean * norm.mean_scale
    return norm.weight * o / (o.pow(2).mean(0, keepdim=True) + norm.eps).sqrt() + norm.bias


def _seq(seq, x, edge, training):
    conv, norm, drop, act = seq.modlist[0], seq.modlist[1], seq.modlist[2], seq.modlist[3]
    x = graph_norm(gcn_conv(x, edge, conv.lin.weight, conv.bias), norm)
    x = F.dropout(x, drop.p, training)
    return F.relu(x) if isinstance(act, torch.nn.ReLU) else x


def forward(mod, x_deg, edge1, pos, idx, ei2, training=False):
    flip = lambda k: k + 1 - 2 * (k & 1)
    edge2 = torch.stack([flip(ei2[0]), ei2[1]])
    edge2_r = torch.stack([ei2[0], flip(ei2[1])])
    x = mod.emb[0].weight[x_deg]
    x = F.dropout(graph_norm(x, mod.emb[1]), mod.emb[2].p, training)
    for seq in mod.conv1s:
        x = _seq(seq, x, edge1, training)
    x = x[pos[:, 0]] * x[pos[:, 1]]
    for a, b in zip(mod.conv2s, mod.conv2s_r):
        x = _seq(a, x, edge2, training) + _seq(b, x, edge2_r, training)
    x = x[idx]
    return (x[0::2] * x[1::2]) @ mod.pred.weight.t() + mod.pred.bias


def train(mod, bg, steps=300, lr=5e-3, seed=0):
    """the reference's single-minibatch 'epochs' (TwoWL/model/train.py:11-47): sample positives and
    negatives, block the sampled positives out of the graph (sample_block), BCE-with-logits, Adam."""
    from lpgnn_amd import twowl
    g = torch.Generator(device="cpu").manual_seed(seed)
    x0, _, ei, _, pos1, _, ei2 = bg.split(0)
    n_pos, n_tuples = ei.shape[1] // 2, pos1.shape[0]
    n_neg = (n_tuples - ei.shape[1]) // 2
    bs = bg.ys[1].shape[0] // 2
    opt = torch.optim.Adam(mod.parameters(), lr=lr)
    mod.train()
    for _ in range(steps):
        i1 = torch.randperm(n_pos, generator=g)[: bs // 2].to(ei.device)
        i2 = torch.randperm(n_neg, generator=g)[: bs // 2].to(ei.device)
        y = torch.cat([torch.ones(i1.numel()), torch.zeros(i2.numel())]).to(ei.device).unsqueeze(-1)
        idx1 = twowl.double(i1, for_index=True)
        idx2 = twowl.double(i2, for_index=True) + ei.shape[1]
        ei_new, x_new, ei2_new = twowl.sample_block(idx1, bg.num_nodes, ei, ei2, n_tuples=n_tuples)
        opt.zero_grad()
        logits = forward(mod, x_new, ei_new, pos1, torch.cat([idx1, idx2]), ei2_new, training=True)
        F.binary_cross_entropy_with_logits(logits, y).backward()
        opt.step()
    mod.eval()
    return mod
