"""world_size-2 gloo test of the flat-gradient exchange (the N>1 path, runnable without GPUs)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sdfstudio_amd.distributed import FlatGradients, broadcast_parameters

    torch.manual_seed(rank)  # different init per rank -> broadcast must equalise
    lin = torch.nn.Linear(5, 3)
    extra = torch.nn.Parameter(torch.randn(7))
    mod = torch.nn.ParameterList([lin.weight, lin.bias, extra])
    broadcast_parameters(mod, src=0)
    flat = FlatGradients(list(mod))
    flat.zero()
    x = torch.full((4, 5), float(rank + 1))
    loss = lin(x).sum() + (extra * (rank + 1)).sum()
    loss.backward()  # accumulates in place into the flat views
    assert lin.weight.grad.data_ptr() == flat.flat.data_ptr()
    flat.all_reduce_mean()
    ret[rank] = (lin.weight.detach().clone(), flat.flat.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    w0, g0 = ret[0]
    w1, g1 = ret[1]
    assert torch.equal(w0, w1), "parameters were not broadcast"
    assert torch.equal(g0, g1), "gradients differ after the all-reduce"
    # d/dW of sum(lin(x)) with x = c: 4*c per entry; mean over ranks c=1,2 -> 6 ; extra grad mean -> 1.5
    assert torch.allclose(g0[:15], torch.full((15,), 6.0))
    assert torch.allclose(g0[15:18], torch.full((3,), 4.0))
    assert torch.allclose(g0[18:], torch.full((7,), 1.5))
