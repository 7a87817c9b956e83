"""Gene sharding over the GPUs of one node.

Genes are independent whenever Kg == 0 and intercept_mode != 'cell' -- the
very condition under which the reference splits genes into sequential batches
(model_wrap.py:241-260).  Here each rank (one process per GPU) owns one
contiguous gene block; no collective is needed inside the optimisation loop.
RCCL (torch.distributed backend "nccl") is used only for
  * the end-of-fit all-gather of per-gene vectors (weights, intercept, sigma,
    loss_gene, ELBO_gain) -- BASELINE's "RCCL weight all-gather", and
  * the all-reduce of the short loss-trace windows that drive the (global)
    convergence decision.
Cell x gene matrices (Psi, Z_std, Psi_95CI) stay sharded (rank 0 can gather them at the end).

Coupled fits (Kg > 0 or intercept_mode='cell') do have a per-step exchange: the per-cell
parameters are replicated and every step all-reduces the ((max(Kg,4)+2) x Nc) per-cell statistics
(`allreduce_inplace`, 1.2 MB at Nc = 50k) between brie_step_begin and brie_step_end.
"""
import numpy as np


def gene_shard(Ng, rank, world, align=4):
    """[g0, g1) of `rank`; boundaries are multiples of `align` (itself a multiple of 4: one Philox draw = one gene
    quad; fitBRIE passes lcm(4, genes per convergence batch) so that no batch straddles two ranks)."""
    align = int(align)
    if align < 4 or align % 4:
        raise ValueError("align=%d must be a positive multiple of 4" % align)
    per = -(-int(Ng) // int(world))
    per = -(-per // align) * align
    g0 = min(rank * per, Ng)
    g1 = min(g0 + per, Ng)
    return g0, g1


class GeneComm(object):
    """torch.distributed plumbing for gene-sharded fits (RCCL on GPU, gloo on CPU)."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device = device
        self._native, self._native_tried, self.native_error = None, False, None
        # how many ranks of this node share one GPU (world-8 dry runs on one device): each reads the same free-HBM figure when
        # the library sizes its placement search, so the engine hands every handle 0.8 / that many of it
        # (brie_placement_configure).  From the launcher's environment and the device count -- NO collective: the
        # constructor must not add one to a path that has never run between two GPUs.
        self.ranks_on_device = 1
        try:
            import os
            import torch
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
            n_dev = torch.cuda.device_count() if device is not None else 0
            if n_dev > 0 and local_world > n_dev:
                self.ranks_on_device = -(-local_world // n_dev)
        except Exception:
            self.ranks_on_device = 1

    def _on_rccl(self):
        return self.backend == "nccl"

    def native_comm(self, device=None):
        """The library's own RCCL communicator (`brie_comm_*` of include/brie_amd.h) over the same ranks, or None.

        Created once, when the process group runs on RCCL (backend "nccl": one rank per GPU); the 128-byte unique id
        travels through torch.distributed's store -- torch is the rendezvous, the data path is librccl called from
        libbrie_amd.so.  With gloo (CPU tests, two ranks sharing a GPU) there is no native communicator: RCCL
        refuses two ranks on one device."""
        if self._native is not None or self._native_tried:
            return self._native
        self._native_tried = True
        if not self._on_rccl():
            return None
        from . import _capi
        # Every rank takes the same steps whatever happens on it.  (1) Each rank checks what it can check ALONE -- librccl
        # binds, its device exists (brie_comm_available) -- and the ranks agree on that BEFORE a unique id exists: a rank
        # that cannot load RCCL must not leave the others inside ncclCommInitRank.  (2) The id travels, each
        # rank builds its communicator, and the ranks AGREE again (an all-reduce over torch.distributed) that all of them
        # succeeded -- one rank falling back to the torch path alone would leave the others waiting in a collective for
        # ever.  What stays collective: a failure INSIDE ncclCommInitRank on one rank while the others are in it.
        dev = device if device is not None else (self.device.index if hasattr(self.device, "index") else self.device)
        dev = int(dev or 0)
        try:
            self.native_error = _capi.Comm.available(dev)
        except Exception as exc:
            self.native_error = repr(exc)
        if float(self.allreduce_min([0.0 if self.native_error else 1.0])[0]) < 1.0:
            self.native_error = self.native_error or "another rank cannot load RCCL / has no such device"
            return None
        ok = 1.0
        try:
            box = [_capi.Comm.unique_id() if self.rank == 0 else None]
        except Exception as exc:             # rank 0 could not make an id: the others still need the broadcast
            box, ok, self.native_error = [None], 0.0, repr(exc)
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        if ok and box[0] is not None:
            try:
                self._native = _capi.Comm(dev, self.rank, self.world, box[0])
            except Exception as exc:
                ok, self.native_error = 0.0, repr(exc)
        else:
            ok = 0.0
        if float(self.allreduce_min([ok])[0]) < 1.0:
            if self._native is not None:
                self._native.close()
            self._native = None
        return self._native

    def _tensor(self, a, dtype=None):
        import torch
        t = torch.as_tensor(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        if self.backend == "nccl":
            t = t.to(self.device if self.device is not None else "cuda")
        return t

    def allreduce_sum(self, a):
        """Sum a small host vector over ranks (fp64 on the wire)."""
        import torch
        t = self._tensor(np.asarray(a, np.float64), torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def allreduce_min(self, a):
        """Minimum of a small host vector over ranks (e.g. the free HBM every rank reports)."""
        import torch
        t = self._tensor(np.asarray(a, np.float64), torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return t.cpu().numpy()

    def _ranges(self, Ng, ranges):
        return [gene_shard(Ng, r, self.world) for r in range(self.world)] if ranges is None else list(ranges)

    def allgather_genes(self, local, Ng, ranges=None, native=None):
        """local (k, n_local) per-gene rows -> (k, Ng) on every rank.  `ranges`: the [g0, g1) of every rank
        (default: gene_shard with quad alignment).

        The end-of-fit gather of a gene-sharded fit (BASELINE's "RCCL weight all-gather"; replaces the `concate` of
        model_wrap.py:260).  native=None: over torch.distributed -- RCCL when the process group's backend is nccl, gloo
        in the CPU tests -- unless BRIE_NATIVE_GATHER=1 asks for the library's own communicator (`brie_comm_allgather`:
        librccl called from libbrie_amd.so) or the group has ONE rank (the only world in which that path has run on the
        build's 1-GPU boxes).  A fit of hours must not be lost at its last step to a collective nobody has ever seen
        complete between two GPUs: the native path is opt-in until it has, and when asked for it is only taken if EVERY
        rank could build its communicator (native_comm agrees across ranks).  native=False forces the torch path, True
        demands the library's."""
        import torch
        local = np.asarray(local, np.float32)
        if local.ndim == 1:
            local = local.reshape(1, -1)
        ranges = self._ranges(Ng, ranges)
        per = max(b - a for a, b in ranges)
        buf = np.zeros((local.shape[0], per), np.float32)
        buf[:, :local.shape[1]] = local
        import os
        want = native is True or (native is None and (self.world == 1 or os.environ.get("BRIE_NATIVE_GATHER") == "1"))
        nat = self.native_comm() if want else None
        if native is True and nat is None:
            raise RuntimeError("no native communicator: the process group runs on %s, not on RCCL" % self.backend)
        self.last_gather_path = "brie_comm_allgather" if nat is not None else "torch.distributed"
        if nat is not None:
            g = nat.allgather(buf).reshape(self.world, local.shape[0], per)
            return np.concatenate([g[r][:, :b - a] for r, (a, b) in enumerate(ranges)], axis=1)
        t = self._tensor(buf)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return np.concatenate([o.cpu().numpy()[:, :b - a] for o, (a, b) in zip(out, ranges)], axis=1)

    def allreduce_inplace(self, t):
        """Sum a float32 device tensor over ranks in place and return once the result is visible to every
        stream (the per-step exchange of a gene-sharded COUPLED fit: per-cell statistics)."""
        import torch
        if self.backend == "nccl":
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            torch.cuda.current_stream(t.device).synchronize()
        else:                               # gloo (CPU tests, or two ranks sharing one GPU): through the host
            host = t.detach().cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
            if t.is_cuda:
                torch.cuda.current_stream(t.device).synchronize()
        return t

    def gather_columns(self, local, Ng, root=0, ranges=None):
        """Column shards (Nc, n_local) of a cell x gene matrix -> the full (Nc, Ng) matrix on `root`
        (None elsewhere).  Used once per output layer at the end of a fit; inside the loop nothing moves."""
        import torch
        local = np.ascontiguousarray(local, np.float32)
        ranges = self._ranges(Ng, ranges)
        per = max(b - a for a, b in ranges)
        buf = np.zeros((local.shape[0], per), np.float32)
        buf[:, :local.shape[1]] = local
        t = self._tensor(buf)
        out = [torch.empty_like(t) for _ in range(self.world)] if self.rank == root else None
        self.dist.gather(t, out, dst=root, group=self.group)
        if self.rank != root:
            return None
        return np.concatenate([o.cpu().numpy()[:, :b - a] for o, (a, b) in zip(out, ranges)], axis=1)

    def barrier(self):
        self.dist.barrier(group=self.group)
