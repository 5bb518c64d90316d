#!/usr/bin/env python3
"""fb-pages-food through the TwoWL inference pipeline on the device, end to end:
edge list -> seeded split (bit-compatible with the reference's random_split_edges) -> 2-WL tuple
graphs + degree features (device kernels) -> LocalWLNet scoring forward -> AUC / time per forward.

Weights: random init (the reference ships no TwoWL checkpoint), or --state-dict path to a
state_dict saved from the reference's LocalWLNet (keys are identical).

    python examples/twowl_fb_pages_food.py [--edges tests/golden/fb-pages-food.edges] [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpgnn_amd import data, twowl  # noqa: E402


def auc(y, s):
    order = np.argsort(s, kind="mergesort")
    ranks = np.empty(len(s))
    ranks[order] = np.arange(1, len(s) + 1)
    y = y.astype(bool)
    return (ranks[y].sum() - y.sum() * (y.sum() + 1) / 2) / (y.sum() * (~y).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edges", default=os.path.join(ROOT, "tests", "golden", "fb-pages-food.edges"))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--state-dict", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(args.seed)
    t0 = time.time()
    bg = data.load_dataset(args.edges, device=dev)
    bg.preprocess()
    bg.setPosDegreeFeature()
    torch.cuda.synchronize()
    print(f"dataset: {bg.num_nodes} nodes, splits E1={[e.shape[1] for e in bg.edge_indexs]}, "
          f"E2={[e.shape[1] for e in bg.ei2s]}, built in {1e3 * (time.time() - t0):.1f} ms")
    mod = twowl.LocalWLNet(bg.max_x).to(dev).eval()
    if args.state_dict:
        mod.load_state_dict(torch.load(args.state_dict, map_location=dev, weights_only=True))
    for split, name in ((1, "val"), (2, "test")):
        x, _, ei, _, pos1, y, ei2 = bg.split(split)
        cap = twowl.CapturedForward(mod, x, ei, pos1, y.shape[0], ei2)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(100):
            logits = cap()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 100
        mask = torch.zeros(y.shape[0], dtype=torch.bool)
        mask[0::2] = True                                   # TwoWL/model/train.py:62-66: every second label
        score = torch.sigmoid(logits).reshape(-1).cpu().numpy()
        print(f"{name}: {score.size} scored edges, AUC {auc(y.cpu().numpy().reshape(-1)[mask.numpy()], score):.4f}, "
              f"{1e6 * dt:.0f} us per forward")


if __name__ == "__main__":
    main()
