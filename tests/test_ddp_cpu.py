"""CPU, world_size 2, gloo: the data-parallel step logic of ao_amd/ptv2/parallel.py (what bench.py runs for
N > 1).  The HIP model cannot run here, so the module under DDP is the CPU oracle of the same network
(oracle/ptv2_ref.RefModule, tiny config); what is checked is the sharding by scene, the gradient averaging
against a single-process computation over both scenes, replica consistency after optimizer steps, and the
max-over-ranks / sum-over-ranks timing reduction."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from tests.conftest import ROOT

TINY = dict(in_channels=6, num_classes=5, patch_embed_depth=1, patch_embed_channels=16, patch_embed_groups=2,
            patch_embed_neighbours=8, enc_depths=(1,), enc_channels=(32,), enc_groups=(4,), enc_neighbours=(8,),
            dec_depths=(1,), dec_channels=(16,), dec_groups=(2,), dec_neighbours=(8,), grid_sizes=(0.12,),
            attn_qkv_bias=True, pe_multiplier=False, pe_bias=True, attn_drop_rate=0.0, drop_path_rate=0.0,
            enable_checkpoint=False, unpool_backend="interp")


def _batch(seeds):
    from ao_amd import synth

    b = synth.scene_batch(seeds, point_max=600, num_classes=5)
    return {k: torch.from_numpy(v) for k, v in b.items()}


def _loss(model, data):
    return F.cross_entropy(model(data), data["segment"], ignore_index=-1)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from ao_amd.ptv2 import parallel
    from oracle import ptv2_ref

    dist.init_process_group("gloo", rank=rank, world_size=world)
    device = torch.device("cpu")
    r, lr, w = parallel.rank_world()
    assert (r, w) == (rank, world)
    seeds = parallel.scene_seeds(rank, 1)
    data = _batch(seeds)
    model = ptv2_ref.RefModule(TINY, seed=3).train()
    net = parallel.wrap_ddp(model, device)
    opt = torch.optim.AdamW(model.parameters(), lr=0.01, weight_decay=0.05)

    # one backward without a step: DDP-averaged gradients
    opt.zero_grad(set_to_none=True)
    _loss(net, data).backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    # the same through the flat all-reduce on an unwrapped copy of the model
    import copy

    twin = copy.deepcopy(model)
    sync = parallel.FlatGradSync(twin)
    twin.zero_grad(set_to_none=True)
    _loss(twin, data).backward()
    sync.sync()
    flat_grads = {k: p.grad.clone() for k, p in twin.named_parameters()}

    # bench.py's default exchange, step by step: FlatAdamW.flatten_grads -> FlatGradSync.reduce_flat (SUM) ->
    # FlatAdamW.step(flat_grad, grad_scale = 1 / world); the HIP update kernel is replaced by its torch statement
    # (tests/_cpu_adamw.py) -- against DDP + torch.optim.AdamW on a third copy
    from tests._cpu_adamw import TorchStatementAdamW

    flat_model, ddp_model = copy.deepcopy(model), copy.deepcopy(model)
    fsync = parallel.FlatGradSync(flat_model)
    fopt = TorchStatementAdamW(flat_model.parameters(), lr=0.006, weight_decay=0.05)
    dnet = parallel.wrap_ddp(ddp_model, device)
    dopt = torch.optim.AdamW(ddp_model.parameters(), lr=0.006, weight_decay=0.05)
    scales = []
    for _ in range(3):
        loss = _loss(flat_model, data)
        fopt.zero_grad(set_to_none=True)
        loss.backward()
        flat = fopt.flatten_grads()
        scale = fsync.reduce_flat(flat)
        scales.append(scale)
        dloss = _loss(dnet, data)
        dopt.zero_grad(set_to_none=True)
        dloss.backward()
        if _ == 0:  # the averaged gradient itself: summed flat buffer x grad_scale == DDP's mean (Adam hides a scale)
            mean0 = [(v * scale).view_as(p).clone() for v, p in zip(fopt._slots(flat), fopt._params)]
            ddp0 = [p.grad.clone() for p in ddp_model.parameters()]
        fopt.step(flat_grad=flat, grad_scale=scale)
        dopt.step()
    flat_exchange = dict(scales=scales, mean0=mean0, ddp0=ddp0, flat={k: p.detach().clone() for k, p in flat_model.named_parameters()},
                         ddp={k: p.detach().clone() for k, p in ddp_model.named_parameters()})

    def step():
        loss = _loss(net, data)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    elapsed, pts, loss = parallel.timed_steps(step, 2, device, data["coord"].shape[0])
    torch.save(dict(seeds=seeds, grads=grads, flat_grads=flat_grads, flat_exchange=flat_exchange, params={k: p.detach().clone() for k, p in model.named_parameters()},
                    elapsed=elapsed, pts=pts, n=data["coord"].shape[0], loss=float(loss.detach())),
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _pre_bn_bias(name):
    """Linear biases with an exactly-zero true gradient: every one that feeds a training-mode BatchNorm (directly, or
    as a per-channel constant through the attention sum) or a softmax over the neighbour slots -- i.e. every Linear
    bias of PT-v2m2 except the last head layer's."""
    return name.endswith("bias") and not name.endswith("norm/bias") and not name.startswith("seg_head/3")


def test_two_rank_data_parallel_step(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert r0["seeds"] == [0] and r1["seeds"] == [1]  # disjoint scenes, no overlap
    assert r0["elapsed"] == r1["elapsed"] and r0["elapsed"] > 0  # MAX over ranks
    assert r0["pts"] == r1["pts"] == r0["n"] + r1["n"]  # SUM over ranks
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k  # replicas stay identical
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k

    # flat all-reduce + FlatAdamW(grad_scale) == DDP + torch.optim.AdamW after 3 steps, and identical on both ranks
    fx0, fx1 = r0["flat_exchange"], r1["flat_exchange"]
    assert fx0["scales"] == [0.5, 0.5, 0.5]
    for k in fx0["flat"]:
        assert torch.equal(fx0["flat"][k], fx1["flat"][k]), k
        # Adam turns a zero-mean-noise gradient (biases in front of a training-mode BatchNorm) into +-lr steps
        # three Adam steps move a weight by up to 3 lr = 1.8e-2; thread-order noise in a near-zero gradient is turned
        # into a fraction of a step by Adam's normalisation (single elements by up to one step): RMS / max criteria
        if not _pre_bn_bias(k):
            d = (fx0["flat"][k] - fx0["ddp"][k]).double()
            assert float(d.pow(2).mean().sqrt()) <= 5e-4 and float(d.abs().max()) <= 6e-3, (k, float(d.abs().max()))
    for a, b in zip(fx0["mean0"], fx0["ddp0"]):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-4, atol=2e-6)

    # single-process reference: mean of the two per-scene gradients (BatchNorm statistics are per scene,
    # exactly as in the two-process run with broadcast_buffers=False)
    from oracle import ptv2_ref

    ref = {}
    for seed in (0, 1):
        model = ptv2_ref.RefModule(TINY, seed=3).train()
        _loss(model, _batch([seed])).backward()
        for k, p in model.named_parameters():
            ref[k] = ref.get(k, 0) + p.grad / 2
    for k, g in ref.items():
        np.testing.assert_allclose(r0["grads"][k].numpy(), g.numpy(), rtol=1e-4, atol=2e-5, err_msg=k)  # pre-BN biases: zero gradient + noise
        np.testing.assert_allclose(r0["flat_grads"][k].numpy(), r0["grads"][k].numpy(), rtol=1e-5, atol=2e-6, err_msg=k)
