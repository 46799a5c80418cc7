"""Column sharding over the GPUs of one node.

Columns are fully independent through the whole path (g-points only meet in the broadband sum, which stays on the device
that owns the column), so rank r of N owns one contiguous column range and runs the complete LW+SW solve on it; the only
communication is ONE all-gather of the packed broadband fluxes per solve (RCCL over xGMI on GPUs: backend "nccl"; gloo
in the CPU tests). The reference has no multi-device code at all."""
import torch
import torch.distributed as dist


def column_range(rank, world, ncol_total):
    """Contiguous, balanced partition: the first (ncol_total % world) ranks get one extra column. Returns [start, stop)."""
    base, extra = divmod(ncol_total, world)
    start = rank*base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_atmosphere(atm, rank, world):
    """Column slice of a host-side synthetic.Atmosphere for this rank (numpy, column index is the LAST array axis)."""
    import numpy as np
    from .synthetic import Atmosphere
    s, e = column_range(rank, world, atm.ncol)
    out = {}
    for k, v in atm.__dict__.items():
        if isinstance(v, np.ndarray):
            if k in ("emis_sfc", "sfc_alb_dir", "sfc_alb_dif"):      # (ncol, nbnd)
                out[k] = np.ascontiguousarray(v[s:e])
            else:
                out[k] = np.ascontiguousarray(v[..., s:e])
        elif isinstance(v, dict):
            # 1-D entries are profiles shared by all columns (aerosol mixing ratios may come that way)
            out[k] = {n: (np.ascontiguousarray(a[..., s:e]) if a.ndim > 1 else a) for n, a in v.items()}
        else:
            out[k] = v
    out["ncol"] = e - s
    return Atmosphere(**out)


def gather_fluxes(local, ncol_total, group=None):
    """All-gather packed fluxes (nflux, nlev, ncol_local) -> (nflux, nlev, ncol_total) on every rank.
    Uneven shards are padded to the largest one so that a single all_gather_into_tensor suffices."""
    world = dist.get_world_size(group)
    nmax = -(-ncol_total // world)
    nflux, nlev, nloc = local.shape
    if nloc < nmax:
        pad = torch.zeros((nflux, nlev, nmax), dtype=local.dtype, device=local.device)
        pad[..., :nloc] = local
        local = pad
    out = torch.empty((world*nflux, nlev, nmax), dtype=local.dtype, device=local.device)   # concatenated along dim 0
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    out = out.view(world, nflux, nlev, nmax)
    parts = []
    for r in range(world):
        s, e = column_range(r, world, ncol_total)
        parts.append(out[r, :, :, :e - s])
    return torch.cat(parts, dim=-1)


class FluxGatherer:
    """The flux gather of a resident solver: buffers allocated once, ONE all_gather_into_tensor per solve. `gather(local)` copies
    the rank's packed fluxes (nflux, nlev, ncol_local) into a send buffer and starts the collective WITHOUT waiting for it, so
    that the exchange of solve i travels over xGMI while solve i+1 computes (two send / receive buffer pairs alternate; the
    previous collective is awaited before the next one starts). `result()` waits for the last gather and returns its
    (nflux, nlev, ncol_total) array; `finish()` only waits. Ragged shards go through the same padded send buffer.
    Backend-agnostic: RCCL ("nccl") on GPUs, gloo in the CPU tests."""

    def __init__(self, ncol_total, local_like, group=None, pipelined=True):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.ncol_total = ncol_total
        self.pipelined = pipelined
        nflux, nlev, nloc = local_like.shape
        self.shape = (nflux, nlev)
        self.nmax = -(-ncol_total // self.world)
        s, e = column_range(self.rank, self.world, ncol_total)
        assert e - s == nloc, f"rank {self.rank} owns {e - s} columns but its flux buffer holds {nloc}"
        self.nloc = nloc
        kw = dict(dtype=local_like.dtype, device=local_like.device)
        nbuf = 2 if pipelined else 1
        self.send = [torch.zeros((nflux, nlev, self.nmax), **kw) for _ in range(nbuf)]
        self.recv = [torch.empty((self.world*nflux, nlev, self.nmax), **kw) for _ in range(nbuf)]
        self.work = None
        self.last = None            # index of the buffer pair of the most recent gather
        self.count = 0

    def gather(self, local):
        i = self.count % len(self.send)
        self.count += 1
        self.finish()                                    # the previous exchange (it had a whole solve to complete)
        self.send[i][..., :self.nloc].copy_(local)       # stream-ordered: the solver may overwrite `local` right away
        if self.pipelined:
            self.work = dist.all_gather_into_tensor(self.recv[i], self.send[i], group=self.group, async_op=True)
        else:
            dist.all_gather_into_tensor(self.recv[i], self.send[i], group=self.group)
        self.last = i
        return self.recv[i]

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None

    def result(self):
        self.finish()
        nflux, nlev = self.shape
        out = self.recv[self.last].view(self.world, nflux, nlev, self.nmax)
        parts = []
        for r in range(self.world):
            s, e = column_range(r, self.world, self.ncol_total)
            parts.append(out[r, :, :, :e - s])
        return torch.cat(parts, dim=-1)
