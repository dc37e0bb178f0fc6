"""Could the screened arg-min SKIP whole (row block, column tile) products?  (round-3 experiment)

screen_kernel multiplies every 512-row block of src descriptors with every 64-column tile of ref descriptors.  A tile can be
skipped for a block iff a cheap lower bound shows that none of its columns can be the arg-min of any of the block's rows:
      D(a, b) = |a - b|^2 >= (|a - c_t| - r_t)_+^2      (c_t, r_t: centroid and radius of the tile's 64 descriptors)
against an upper bound T_j of the row's minimum (iteration >= 1: the exact distance to the PREVIOUS iteration's match).
Descriptors are continuous functions of position, so tiles of spatially neighbouring ref points (Morton order) should be
compact, and src rows ordered by the position of their previous match should agree on which tiles matter.

This tool measures, on the engine's own descriptors, the fraction of (row block, tile) products that survive such pruning.
Measurement aid (torch on the GPU); nothing here is on the product path.

    python tools/prune_stats.py [POINTS [PAIRS [FEAT_LEN SHAPE PARTIAL]]]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
feat_len = int(sys.argv[3]) if len(sys.argv) > 3 else 3
shape = sys.argv[4] if len(sys.argv) > 4 else "3dmatch"
partial = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
RB, CT = 512, 64


def morton(xyz):
    """30-bit Morton code of points normalised to their bounding box (torch, on device)."""
    lo, hi = xyz.min(0)[0], xyz.max(0)[0]
    q = ((xyz - lo) / (hi - lo + 1e-9) * 1023).long().clamp(0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


cfg = NetConfig(feat_len=feat_len)
eng = Engine(cfg, 0, max_points=N, max_pairs=P)
eng.load_state_dict(generate_state_dict(cfg, 0))
b = make_batch(N, list(range(10_000, 10_000 + P)), feat_len, shape, partial)
src, ref = torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda()
o = eng.register(src, ref, 5, want_desc=True)
print(f"points {N} pairs {P} shape {shape} partial {partial}: blocks of {RB} rows x tiles of {CT} columns")
# "need+morton": rows grouped by how many tiles they need themselves (8 classes), then by match position: rows that need
# everything (poor previous match, no counterpart) no longer drag the blocks of well-matched rows up to the full list
# "tileT": the upper bound T also takes the exact minimum over the columns of the tile with the NEAREST CENTROID (an actual
# distance of the row, hence valid) - available in iteration 0 too, where rows are ordered by that tile instead of a previous match
# "kmeansT": like "tileT", but the ref columns are ordered by k-means clusters of the DESCRIPTORS (one cluster per tile's worth of
# columns, 10 Lloyd iterations, columns sorted by cluster) instead of the Morton order of their points
def kmeans_order(r, nt, iters=10):
    g = torch.Generator(device="cpu").manual_seed(0)
    cen = r[torch.randperm(r.shape[0], generator=g)[:nt].to(r.device)].clone()
    for _ in range(iters):
        a_ = torch.cdist(r, cen).argmin(1)
        cen = torch.zeros_like(cen).index_add_(0, a_, r) / torch.bincount(a_, minlength=nt).clamp(min=1)[:, None]
    a_ = torch.cdist(r, cen).argmin(1)
    return torch.argsort(a_ * (1 << 20) + torch.arange(r.shape[0], device=r.device))


for order in ("as given", "morton", "need+morton", "T1+morton", "T2+morton", "tileT", "kmeansT"):
    for it in range(0 if order in ("tileT", "kmeansT") else 1, 5):
        kept = total = 0
        radii = []
        for p in range(P):
            r = o["desc_ref"][p]
            a = o["desc_src"][it, p]
            prev = o["idx"][max(it - 1, 0), p].long()
            if order != "as given":
                perm = kmeans_order(r, (N + CT - 1) // CT) if order == "kmeansT" else torch.argsort(morton(ref[p, :, :3]))
                inv = torch.empty_like(perm); inv[perm] = torch.arange(N, device=perm.device)
                r = r[perm]
                prev_pos = inv[prev]                      # position of the previous match in the sorted ref order
                key = prev_pos
                if order == "need+morton":
                    nt_ = (N + CT - 1) // CT
                    rp_ = torch.cat([r, r[-1:].expand(nt_ * CT - N, -1)], 0).view(nt_, CT, 64)
                    c_ = rp_.mean(1)
                    rad_ = (rp_ - c_[:, None, :]).norm(dim=2).max(1)[0]
                    T_ = ((a - r[prev_pos]) ** 2).sum(1) + 1e-5
                    own = (((torch.cdist(a, c_) - rad_[None, :]).clamp(min=0) ** 2) <= T_[:, None]).float().mean(1)
                    key = (own * 7.999).long() * (1 << 20) + prev_pos
                if order in ("T1+morton", "T2+morton"):   # classes of the upper bound itself: 1 / 2 per octave (free: no extra bound pass)
                    T_ = ((a - r[prev_pos]) ** 2).sum(1) + 1e-5
                    cls = T_.view(torch.int32).long() >> (23 if order == "T1+morton" else 22)
                    key = cls * (1 << 20) + prev_pos
                if order in ("tileT", "kmeansT") and it == 0:          # no previous match: rows by the tile with the nearest centroid
                    nt_ = (N + CT - 1) // CT
                    rp_ = torch.cat([r, r[-1:].expand(nt_ * CT - N, -1)], 0).view(nt_, CT, 64)
                    key = torch.cdist(a, rp_.mean(1)).argmin(1) * CT
                rows = torch.argsort(key)                 # src rows ordered by where their previous match sits
                a, prev_pos = a[rows], prev_pos[rows]
            else:
                prev_pos = prev
            T = ((a - r[prev_pos]) ** 2).sum(1) + 1e-5   # upper bound of the row minimum (+ safety margin)
            if order in ("tileT", "kmeansT"):
                nt_ = (N + CT - 1) // CT
                rp_ = torch.cat([r, r[-1:].expand(nt_ * CT - N, -1)], 0).view(nt_, CT, 64)
                tstar = torch.cdist(a, rp_.mean(1)).argmin(1)                     # nearest centroid (rows already in their order)
                Tt = torch.empty_like(T)
                for s0 in range(0, a.shape[0], 4096):
                    blk = rp_[tstar[s0:s0 + 4096]]                                  # [rows, 64, 64]
                    Tt[s0:s0 + 4096] = ((a[s0:s0 + 4096, None, :] - blk) ** 2).sum(2).min(1)[0] + 1e-5
                T = Tt if it == 0 else torch.minimum(T, Tt)
            nt = (N + CT - 1) // CT
            pad = nt * CT - N
            rp = torch.cat([r, r[-1:].expand(pad, -1)], 0).view(nt, CT, 64)
            c = rp.mean(1)
            rad = (rp - c[:, None, :]).norm(dim=2).max(1)[0]
            radii.append(rad.cpu().numpy())
            dist = torch.cdist(a, c)                      # [J, nt]
            lb = (dist - rad[None, :]).clamp(min=0) ** 2
            need = lb <= T[:, None]                       # row j must look at tile t
            nb = (N + RB - 1) // RB
            padr = nb * RB - N
            needp = torch.cat([need, torch.zeros(padr, nt, dtype=torch.bool, device=need.device)], 0).view(nb, RB, nt)
            blk = needp.any(1)                            # [nb, nt]: some row of the block needs the tile
            kept += int(blk.sum()); total += nb * nt
        rr = np.concatenate(radii)
        print(f"  order {order:11s} iter {it}: products kept {kept}/{total} = {kept / total:.3f}; rows' own need {float(need.float().mean()):.3f}; "
              f"tile radius median {np.median(rr):.3f} max {rr.max():.3f}")
eng.close()
