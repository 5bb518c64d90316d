#!/usr/bin/env python3
"""How many destination rows of the bench graph (RMAT |V| = 5 M, |E| = 100 M) receive no edge at all?  These are the
rows of a rank's partial aggregate that a sparse exchange would not have to send."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
for world, rank in ((1, 0), (8, 3)):
    from lpgnn_amd import dist as ldist
    lo, hi = ldist.source_range(5_000_000, world, rank)
    ei = bench.rmat_edges(5_000_000, 100_000_000, 5 + rank, dev, src_range=None if world == 1 else (lo, hi))
    deg = torch.bincount(ei[1], minlength=5_000_000)
    print(f"world {world} rank {rank}: rows without an in-edge: {int((deg == 0).sum())} of 5000000 "
          f"({float((deg == 0).float().mean()):.3f}); rows with <= 2: {float((deg <= 2).float().mean()):.3f}")
