"""Plain-torch statement of the three device stages of ao_amd/ptv2/gva.py (pos_stats, logits,
aggregate).  Test infrastructure: lets the host-side re-association be checked on CPU against the
oracle, and gives the HIP kernels a stage-level reference on GPU."""
import torch


def _pos(coord, idx):
    mask = (idx >= 0).to(coord.dtype)
    safe = idx.clamp(min=0).long()
    pos = (coord[safe] - coord.unsqueeze(1)) * mask.unsqueeze(-1)
    return pos, mask, safe


class TorchImpl:
    @staticmethod
    def pos_stats(coord, idx):
        pos, _, _ = _pos(coord, idx)
        p = pos.double().reshape(-1, 3)
        return p.sum(0), p.t() @ p

    @staticmethod
    def logits(kW, qW, a, b, M, cW, coord, idx):
        pos, mask, safe = _pos(coord, idx)
        P = torch.relu(pos @ a.t() + b)
        W1 = kW[safe] * mask.unsqueeze(-1) - qW.unsqueeze(1) + P @ M + cW
        W1d = W1.double()
        return W1, W1d.sum((0, 1)), (W1d * W1d).sum((0, 1))

    @staticmethod
    def aggregate(W1, sc, sh, Ww2, bw2, v, a, b, coord, idx):
        pos, mask, safe = _pos(coord, idx)
        n, k = idx.shape
        g = sc.shape[0]
        c = v.shape[1]
        y = torch.relu(W1 * sc + sh)
        z = y @ Ww2.t() + bw2
        w = torch.softmax(z, dim=1) * mask.unsqueeze(-1)
        vm = v[safe] * mask.unsqueeze(-1)
        out_v = (vm.view(n, k, g, c // g) * w.unsqueeze(-1)).sum(1).reshape(n, c)
        P = torch.relu(pos @ a.t() + b)
        A = torch.einsum("nsg,nsc->ngc", w, P)
        return out_v, A, w.sum(1)
