"""GPU: the reference's multi-GPU wrap -- `create_ddp_model` = DistributedDataParallel(model, broadcast_buffers=False)
(pointcept/engines/defaults.py:20-43, engines/train_sam_pp2s.py:207-213) -- over this package's segmentor, unchanged trainer
code.  The segmentor tells DDP to leave all parameters but one alone (`_ddp_params_and_buffers_to_ignore`,
ao_amd/ptv2/model.parallel_ddp_ignore) and the native backward averages its ONE flat gradient buffer over the ranks itself.

The box has one MI355X: the two ranks of the second test share cuda:0 and exchange over gloo (RCCL refuses two ranks on one
device); the first test runs the RCCL statement itself in a one-rank "nccl" group."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene(seed, n=3000):
    from ao_amd import synth

    b = synth.scene_batch([seed], point_max=n, room=1)
    return {k: torch.from_numpy(v).cuda() for k, v in b.items()}


def _segmentor(seed):
    import ao_amd.ptv2 as ptv2

    torch.manual_seed(seed)
    cfg = dict(ptv2.S3DIS_BACKBONE, drop_path_rate=0.0)
    return ptv2.DefaultSegmentor(backbone=dict(cfg, type="PT-v2m2")).cuda().train()


def _grads(model, data):
    model.zero_grad(set_to_none=True)
    out = model(dict(data))
    out["loss"].backward()
    inner = model.module if hasattr(model, "module") else model
    return float(out["loss"].detach()), [p.grad.detach().clone() for p in inner.parameters()]


def _worker(rank, world, port, backend, q):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        seg = _segmentor(seed=100 + rank)  # different initial weights per rank: DDP's construction must equalise them
        ddp = DistributedDataParallel(seg, device_ids=[0], output_device=0, broadcast_buffers=False, find_unused_parameters=False)
        owned = [n for n, p in seg.named_parameters() if n not in set(seg._ddp_params_and_buffers_to_ignore)]
        state = torch.cat([p.detach().reshape(-1) for p in seg.parameters()]).cpu()
        loss, grads = _grads(ddp, _scene(seed=rank))
        flat = torch.cat([g.reshape(-1) for g in grads]).cpu()
        # (numpy arrays: pickled by value.  A torch tensor travels as a file descriptor the parent fetches from THIS process's
        # resource sharer -- an EOFError in the parent's q.get() when this process has exited first: seen once in four suite runs)
        q.put((rank, owned, state.numpy(), loss, flat.numpy(), bool(seg.backbone.__dict__.get("_ao_ddp_native_sync"))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_ddp_wrap_in_a_one_rank_rccl_group_changes_nothing_but_the_route():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), "nccl", q))
    p.start()
    rank, owned, state, loss, flat, native = q.get(timeout=600)
    state, flat = torch.from_numpy(state), torch.from_numpy(flat)
    p.join(120)
    assert p.exitcode == 0
    assert native and len(owned) == 1, owned  # DDP keeps one (the smallest) parameter, the native all-reduce the other 839
    seg = _segmentor(seed=100)
    assert torch.equal(torch.cat([p.detach().reshape(-1) for p in seg.parameters()]).cpu(), state)
    ref_loss, ref = _grads(seg, _scene(seed=0))
    assert loss == ref_loss
    assert torch.equal(torch.cat([g.reshape(-1) for g in ref]).cpu(), flat)  # an average over one rank: the same bits


def _two_ranks():
    """Both ranks' results, or None when the rendezvous itself failed (a port taken between the probe and the bind, a worker that
    did not come up): the caller tries once more on a new port.  Numerical results are never retried."""
    import queue

    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, "gloo", q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
        got = [(r, o, torch.from_numpy(st), l, torch.from_numpy(fl), nat) for r, o, st, l, fl, nat in got]
    except queue.Empty:
        got = None
    for p in procs:
        p.join(120 if got is not None else 5)
        if p.is_alive():
            p.kill()  # (this process object only)
            p.join(30)
        if got is not None:
            assert p.exitcode == 0
    return got


def test_two_ranks_average_their_gradients_through_the_native_all_reduce():
    got = _two_ranks() or _two_ranks()
    assert got is not None, "two ranks did not rendezvous (twice)"
    (_, owned0, state0, loss0, flat0, nat0), (_, owned1, state1, loss1, flat1, nat1) = got
    assert nat0 and nat1 and len(owned0) == len(owned1) == 1
    assert torch.equal(state0, state1)  # rank 0's initial weights everywhere (840 tensors: 839 by our broadcast, 1 by DDP's)
    assert torch.equal(flat0, flat1)    # both ranks hold the averaged gradient
    # the same average computed in this process: rank 0's weights, one scene after the other
    seg = _segmentor(seed=100)
    assert torch.equal(torch.cat([p.detach().reshape(-1) for p in seg.parameters()]).cpu(), state0)
    l0, g0 = _grads(seg, _scene(seed=0))
    l1, g1 = _grads(seg, _scene(seed=1))
    assert abs(l0 - loss0) < 1e-6 and abs(l1 - loss1) < 1e-6
    mean = (torch.cat([g.reshape(-1) for g in g0]) + torch.cat([g.reshape(-1) for g in g1])).cpu() / 2
    assert float((mean - flat0).abs().max()) <= 1e-6 * max(1.0, float(mean.abs().max()))


def test_ddp_sync_env_hands_every_parameter_back_to_ddp(monkeypatch):
    import torch.distributed as dist

    monkeypatch.setenv("AO_AMD_DDP_SYNC", "ddp")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        seg = _segmentor(seed=3)
        assert seg._ddp_params_and_buffers_to_ignore == [] and not seg.backbone.__dict__.get("_ao_ddp_native_sync")
    finally:
        dist.destroy_process_group()
