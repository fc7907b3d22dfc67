"""bench.py's HIP backend with the RCCL transport swapped for gloo, so that TWO
ranks can share the ONE GPU of a development box (RCCL refuses two ranks on
one device).  TEST INFRASTRUCTURE, selected with
QMC_BENCH_BACKEND=tests._bench_gpu_gloo: everything else is the product path --
the HIP engine, `DistributedDmc`, the split step, export / import of walker
records, the forced rebalances and checks of `bench.py`.  gloo moves host
buffers only (all-reduce excepted), so the all-gather of the walker counts and
the point-to-point transfers are staged through the host here; on a multi-GPU
node RCCL does both device to device."""
import torch
import torch.distributed as dist

from bench import HipBackend


class Backend(HipBackend):
    dist_backend = 'gloo'
    name = 'hip + gloo transport (tests only: two ranks on one GPU)'

    def __init__(self, local_rank):
        super().__init__(0)              # every rank on the one GPU
        _stage_through_host()


def _stage_through_host():
    if getattr(dist, '_qmc_staged', False):
        return
    dist._qmc_staged = True
    real_all_gather = dist.all_gather
    real_batch = dist.batch_isend_irecv

    def all_gather(tensor_list, tensor, *a, **kw):
        if tensor.is_cuda:
            host = [t.cpu() for t in tensor_list]
            real_all_gather(host, tensor.cpu(), *a, **kw)
            for dst, src in zip(tensor_list, host):
                dst.copy_(src)
            return None
        return real_all_gather(tensor_list, tensor, *a, **kw)

    class _Done:
        def wait(self):
            return True

    def batch_isend_irecv(ops):
        torch.cuda.synchronize()          # the packing kernels have finished
        staged, host_ops = [], []
        for op in ops:
            buf = op.tensor.cpu() if op.op is dist.isend else \
                torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
            staged.append((op, buf))
            host_ops.append(dist.P2POp(op.op, buf, op.peer))
        for req in real_batch(host_ops):
            req.wait()
        for op, buf in staged:
            if op.op is dist.irecv:
                op.tensor.copy_(buf)
        torch.cuda.synchronize()
        return [_Done() for _ in ops]

    dist.all_gather = all_gather
    dist.batch_isend_irecv = batch_isend_irecv
