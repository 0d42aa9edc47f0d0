"""Multi-GPU decomposition of the path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests).

The reference has no parallelism at all (SURVEY.md sections 2, 5); this is new.

* Advection is independent per seed (LCS/trajectory.py:80-126): the seed grid is
  cut into contiguous blocks of latitude ROWS, longitude stays whole so the
  stencil's cyclic ``% xsize`` (LCS/tools.py:225-228) is local.  The wind series
  is replicated.  No communication.
* The flow-map gradient couples rows r-2..r+2 (LCS/tools.py:202-207): ONE
  neighbour exchange of 2 halo rows of (x_dep, y_dep) between advection and the
  sigma kernel -- point-to-point send/recv with the previous / next rank only,
  non-periodic (rank 0 and the last rank have one neighbour; the 2 first / last
  global rows use the one-sided rule and need no halo).  No all-reduce anywhere.
* Ensembles (BASELINE config 5) shard start times: no communication at all.
"""
from __future__ import annotations

HALO = 2  # rows; LCS/tools.py:202-207 (4th-order, +-2 points), SURVEY Q12

__all__ = ["HALO", "row_partition", "halo_rows", "halo_exchange", "halo_exchange_into", "ensemble_partition",
           "sharded_lcs", "ensemble_lcs", "ensemble_advect", "native_comm", "ENSEMBLE_CHUNK",
           "interleaved_chunks", "interleaved_partition", "chunk_window", "interleaved_rows", "chunk_halo_exchange", "CHUNK_ROWS"]


def row_partition(ny_global: int, world: int, rank: int):
    """Rows [lo, hi) owned by ``rank``: contiguous, sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} not in [0,{world})")
    if ny_global < world * HALO:
        raise ValueError(f"{ny_global} rows cannot be split over {world} ranks with a {HALO}-row halo")
    base, rem = divmod(ny_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------------------------
# Interleaved, patch-aligned row chunks: the partition for STRONG scaling (a fixed grid over more ranks).
#
# Contiguous blocks hand every rank one latitude band, and the bands do not cost the same: on BASELINE configs[3]
# (8192^2 x 384) the eight blocks take 11.3 ... 14.2 ms (the flow's jets, the polar rows' long zonal travel), the slowest
# sets the step.  Here the rows are cut into chunks of CHUNK_ROWS rows and dealt to the ranks round robin --
# rank r owns chunks r, r + world, r + 2 world, ... -- so every rank holds every band.  A rank advects all of its chunks in
# ONE lc_advect call over the concatenated rows (advection is per seed, LCS/trajectory.py:80-126; the only thing the kernels
# read of a seed's global row is the pole rule: Engine.advect(global_rows=...)), exchanges the 2 sigma-halo rows of every
# chunk with the previous / next rank -- chunk k's neighbours k - 1 and k + 1 belong to ranks r - 1 and r + 1 (mod world: a
# ring, where contiguous blocks form a line), all of a neighbour's rows in ONE message per direction -- and runs lc_sigma once
# per chunk on its window (stencil reach LCS/tools.py:202-207).
#
# Two shape rules, both measured (profiles/r06/shard_costs_*.jsonl, profiles/r05/shard_costs_balanced.txt):
#  * a chunk is a multiple of 64 rows that starts on a multiple of 64 local rows: the two-seed kernel's waves hold 8 x 16-seed
#    patches, its workgroups 64 rows, and a patch that straddles two chunks has its seeds `world` chunks apart in latitude and
#    leaves its LDS tile at every sample (unaligned 128-row chunks: 14.9-16.0 ms per rank of C4 against 11.9-12.7 aligned);
#  * a rank's rows in all are a multiple of 512 where the grid allows it (8 XCDs x 64-row workgroup rows): 1040 rows -- 1024
#    and the 16 halo rows of four chunks advected redundantly instead of exchanged, this file's first cut -- are 17 workgroup
#    rows on 8 XCDs and took 13.8-15.3 ms per rank of C4 where 1024 rows take 11.9-12.7.  Hence the exchange, and no
#    redundant rows (``redundant_halo=True`` remains as the cross-check: bit-identical, no communication).
# Grids whose chunks do not deal out evenly (a rank would get more than 10 % over the mean) keep the contiguous blocks.
# ---------------------------------------------------------------------------------------------------------------
CHUNK_ROWS = 256   # rows of a chunk: 4 workgroup rows of the two-seed kernel
CHUNK_WINDOW = CHUNK_ROWS   # (the name the first cut used)


def interleaved_chunks(ny_global: int, world: int, chunk: int = CHUNK_ROWS):
    """The chunks ``[(lo, hi), ...]`` of the interleaved partition, in row order; chunk ``k`` belongs to rank ``k % world``.
    Chunk boundaries are the multiples of ``chunk`` (a last chunk thinner than the stencil needs joins the one before).
    With fewer than two chunks per rank, or a deal that would leave some rank more than 10 % over the mean, it is
    :func:`row_partition`: one contiguous block per rank."""
    if world < 1 or chunk % 16 or chunk < 16:
        raise ValueError("world >= 1 and a chunk that is a multiple of 16 rows")
    if ny_global < world * HALO:
        raise ValueError(f"{ny_global} rows cannot be split over {world} ranks with a {HALO}-row halo")
    bounds = list(range(0, ny_global, chunk)) + [ny_global]
    if len(bounds) > 2 and bounds[-1] - bounds[-2] < 2 * HALO:
        del bounds[-2]
    ch = [(bounds[k], bounds[k + 1]) for k in range(len(bounds) - 1)]
    per = [sum(hi - lo for lo, hi in ch[r::world]) for r in range(world)]
    if world == 1 or len(ch) < 2 * world or max(per) > 1.1 * ny_global / world:
        return [row_partition(ny_global, world, r) for r in range(world)]
    return ch


def interleaved_partition(ny_global: int, world: int, rank: int, chunk: int = CHUNK_ROWS):
    """The chunks ``[(lo, hi), ...]`` rank ``rank`` owns (ascending)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} not in [0,{world})")
    return interleaved_chunks(ny_global, world, chunk)[rank::world]


def chunk_window(ny_global: int, lo: int, hi: int):
    """Rows ``[a, b)`` the sigma rows of the chunk ``[lo, hi)`` read: the chunk and its halo rows."""
    n_lo, n_hi = halo_rows(ny_global, lo, hi)
    return lo - n_lo, hi + n_hi


def interleaved_rows(ny_global: int, chunks, with_halo: bool = False):
    """Global row index of every row of ``chunks`` concatenated (``with_halo``: of their windows), as a list of ints."""
    rows = []
    for lo, hi in chunks:
        a, b = chunk_window(ny_global, lo, hi) if with_halo else (lo, hi)
        rows.extend(range(a, b))
    return rows


def chunk_halo_exchange(x_own, y_own, chunks, ny_global: int, rank: int, world: int, group=None):
    """The halo exchange of the interleaved partition.  ``x_own`` / ``y_own``: ``(rows, nx)``, this rank's chunks concatenated.
    Returns ``(x_win, y_win)``: every chunk's WINDOW (halo rows received + chunk) concatenated -- what ``lc_sigma`` reads,
    chunk by chunk.  Ring: chunk k's lower halo is the last two rows of chunk k - 1 (rank ``rank - 1``), its upper halo the
    first two rows of chunk k + 1 (rank ``rank + 1``), both mod ``world``; per neighbour and direction ONE message carrying
    (x, y) of all chunks' rows.  Sends are posted [to previous, to next], receives [from next, from previous]: with two ranks
    both neighbours are the same peer and point-to-point operations between a pair match in order."""
    import torch
    import torch.distributed as dist
    if not (x_own.is_contiguous() and y_own.is_contiguous()) or x_own.shape != y_own.shape or x_own.dim() != 2:
        raise ValueError("chunk_halo_exchange: x_own and y_own must be contiguous (rows, nx) buffers of one shape")
    nx = x_own.shape[1]
    offs, o = [], 0
    for lo, hi in chunks:
        offs.append(o)
        o += hi - lo
    if o != x_own.shape[0]:
        raise ValueError("chunk_halo_exchange: the buffers do not hold the chunks' rows")
    has_lo = [lo > 0 for lo, hi in chunks]                  # a chunk below exists (it is rank - 1's)
    has_hi = [hi < ny_global for lo, hi in chunks]          # a chunk above exists (rank + 1's)
    dev = x_own.device
    idx = lambda rows: torch.as_tensor(rows, dtype=torch.long, device=dev)
    first = idx([offs[i] + j for i in range(len(chunks)) if has_lo[i] for j in range(HALO)])                         # -> previous rank
    last = idx([offs[i] + chunks[i][1] - chunks[i][0] - HALO + j for i in range(len(chunks)) if has_hi[i] for j in range(HALO)])  # -> next rank
    send_prev = torch.stack([x_own.index_select(0, first), y_own.index_select(0, first)])
    send_next = torch.stack([x_own.index_select(0, last), y_own.index_select(0, last)])
    recv_next = torch.empty((2, HALO * sum(has_hi), nx), dtype=x_own.dtype, device=dev)      # the chunks above: their first rows
    recv_prev = torch.empty((2, HALO * sum(has_lo), nx), dtype=x_own.dtype, device=dev)      # the chunks below: their last rows
    prev, nxt = (rank - 1) % world, (rank + 1) % world
    via_host = x_own.is_cuda and dist.get_backend(group) == "gloo"       # gloo cannot move device tensors (rehearsal layout only)
    if via_host:
        send_prev, send_next, rn, rp = send_prev.cpu(), send_next.cpu(), recv_next.cpu(), recv_prev.cpu()
    else:
        rn, rp = recv_next, recv_prev
    ops = []
    if send_prev.numel():
        ops.append(dist.P2POp(dist.isend, send_prev, prev, group))
    if send_next.numel():
        ops.append(dist.P2POp(dist.isend, send_next, nxt, group))
    if rn.numel():
        ops.append(dist.P2POp(dist.irecv, rn, nxt, group))
    if rp.numel():
        ops.append(dist.P2POp(dist.irecv, rp, prev, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if via_host:
        recv_next.copy_(rn)
        recv_prev.copy_(rp)
    return _assemble_windows(x_own, y_own, chunks, ny_global, recv_prev, recv_next)


def _assemble_windows(x_own, y_own, chunks, ny_global, recv_prev, recv_next):
    """The chunks' windows, concatenated: [lower halo rows][chunk][upper halo rows] per chunk (2 x 2 small copies per chunk)."""
    import torch
    total = sum(chunk_window(ny_global, lo, hi)[1] - chunk_window(ny_global, lo, hi)[0] for lo, hi in chunks)
    out = []
    for t, k in ((x_own, 0), (y_own, 1)):
        w = torch.empty((total, t.shape[1]), dtype=t.dtype, device=t.device)
        o = src = ip = inx = 0
        for lo, hi in chunks:
            a, b = chunk_window(ny_global, lo, hi)
            if lo - a:
                w[o:o + lo - a].copy_(recv_prev[k, ip:ip + lo - a])
                ip += lo - a
            w[o + lo - a:o + lo - a + hi - lo].copy_(t[src:src + hi - lo])
            if b - hi:
                w[o + hi - a:o + b - a].copy_(recv_next[k, inx:inx + b - hi])
                inx += b - hi
            o += b - a
            src += hi - lo
        out.append(w)
    return out[0], out[1]


def halo_rows(ny_global: int, lo: int, hi: int):
    """(rows needed below lo, rows needed above hi) for sigma on rows [lo, hi)."""
    return min(HALO, lo), min(HALO, ny_global - hi)


def ensemble_partition(n_members: int, world: int, rank: int):
    """Member indices handled by ``rank`` (contiguous blocks)."""
    base, rem = divmod(n_members, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


def native_comm(engine, rank: int, world: int, group=None):
    """RCCL communicator of the C ABI (``lc_comm_create``) for this engine's GPU, created once and kept on
    the engine.  Rank 0 makes the id; ``torch.distributed`` (any backend) only carries those 128 bytes."""
    import torch.distributed as dist
    comm = getattr(engine, "_lc_comm", None)
    if comm is None:
        box = [engine.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        comm = engine.comm_create(world, rank, box[0])
        engine._lc_comm = comm
    return comm


def halo_exchange_into(x_ext, y_ext, n_lo: int, n_hi: int, rank: int, world: int, group=None, engine=None, comm=None):
    """In-place halo exchange.  With ``engine`` and ``comm`` (see :func:`native_comm`) the exchange is the C
    ABI's ``lc_halo_exchange`` (RCCL send/recv straight from / into the buffers, no staging copies);
    otherwise ``torch.distributed`` point-to-point as described below.

    ``x_ext``, ``y_ext``: ``(n_lo + n + n_hi, nx)`` buffers whose middle ``n``
    rows hold this rank's departure points; the first ``n_lo`` / last ``n_hi`` rows are filled with the
    neighbours' boundary rows.  Per neighbour and direction one message per array, straight from / into the buffers
    (2 rows x nx: 32 KiB at nx=4096 fp32 -- latency-bound on xGMI; no collective)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return
    n = x_ext.shape[0] - n_lo - n_hi
    if n < HALO:
        raise ValueError("local block thinner than the halo")
    if not (x_ext.is_contiguous() and y_ext.is_contiguous()) or x_ext.shape != y_ext.shape or x_ext.dim() != 2:
        # rows are sent from and received into the buffers themselves: a strided view would fail inside the transport, per
        # operation and mid-batch, with the peers left waiting
        raise ValueError("halo_exchange_into: x_ext and y_ext must be contiguous (rows, nx) buffers of one shape")
    if comm is not None:
        engine.halo_exchange(comm, x_ext, y_ext, n_lo, n_hi)
        return
    # gloo cannot move device tensors: stage the (tiny) messages through the host.  Only used when
    # rehearsing the N>1 path without RCCL (several ranks on one GPU); nccl sends device memory.
    via_host = x_ext.is_cuda and dist.get_backend(group) == "gloo"
    if not via_host:
        # Row slices of a contiguous (rows, nx) buffer are contiguous: the boundary rows are sent from, and the halo rows
        # received into, the buffers themselves -- one send and one receive per array and neighbour inside ONE batch (a
        # single ncclGroupStart / End with the nccl backend), no staging copies, no kernel launch at all.  (Until round 5
        # both arrays travelled in one stacked message: two gather kernels before and four copy kernels after the exchange,
        # each a launch latency on a path that is nothing but latency.)  Both sides list x before y: point-to-point
        # operations between a pair of ranks match in order.
        ops = []
        if rank > 0:           # previous rank owns the rows just below ours
            assert n_lo == HALO
            ops += [dist.P2POp(dist.isend, t[n_lo:n_lo + HALO], rank - 1, group) for t in (x_ext, y_ext)]
            ops += [dist.P2POp(dist.irecv, t[:n_lo], rank - 1, group) for t in (x_ext, y_ext)]
        if rank < world - 1:   # next rank owns the rows above ours
            assert n_hi == HALO
            ops += [dist.P2POp(dist.isend, t[n_lo + n - HALO:n_lo + n], rank + 1, group) for t in (x_ext, y_ext)]
            ops += [dist.P2POp(dist.irecv, t[n_lo + n:], rank + 1, group) for t in (x_ext, y_ext)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return

    def msg(rows):
        return torch.stack([x_ext[rows], y_ext[rows]]).contiguous().cpu()

    ops, recv_lo, recv_hi = [], None, None
    if rank > 0:
        assert n_lo == HALO
        send = msg(slice(n_lo, n_lo + HALO))
        recv_lo = torch.empty_like(send)
        ops += [dist.P2POp(dist.isend, send, rank - 1, group), dist.P2POp(dist.irecv, recv_lo, rank - 1, group)]
    if rank < world - 1:
        assert n_hi == HALO
        send2 = msg(slice(n_lo + n - HALO, n_lo + n))
        recv_hi = torch.empty_like(send2)
        ops += [dist.P2POp(dist.isend, send2, rank + 1, group), dist.P2POp(dist.irecv, recv_hi, rank + 1, group)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if recv_lo is not None:
        x_ext[:n_lo].copy_(recv_lo[0])
        y_ext[:n_lo].copy_(recv_lo[1])
    if recv_hi is not None:
        x_ext[n_lo + n:].copy_(recv_hi[0])
        y_ext[n_lo + n:].copy_(recv_hi[1])


def halo_exchange(x, y, rank: int, world: int, ny_global: int, lo: int, hi: int, group=None):
    """Copying form: ``x``, ``y`` are this rank's ``(hi-lo, nx)`` blocks; returns ``(x_ext, y_ext, in_row0)``,
    the blocks extended by the halo rows received and the global index of their first row -- the
    input window ``lc_sigma`` wants.  (`sharded_lcs` uses the in-place form and never copies the block.)"""
    import torch
    n_lo, n_hi = halo_rows(ny_global, lo, hi)
    if world == 1:
        return x, y, lo
    x_ext = torch.empty((n_lo + x.shape[0] + n_hi, x.shape[1]), dtype=x.dtype, device=x.device)
    y_ext = torch.empty_like(x_ext)
    x_ext[n_lo:n_lo + x.shape[0]].copy_(x)
    y_ext[n_lo:n_lo + y.shape[0]].copy_(y)
    halo_exchange_into(x_ext, y_ext, n_lo, n_hi, rank, world, group)
    return x_ext, y_ext, lo - n_lo


def sharded_lcs(engine, field, seed_lat_global, seed_lon, timestep, rank: int, world: int, SETTLS_order=0,
                interp_order=1, cyclic_xboundary=True, t0=0, nsteps=None, fd_fp32_cast=True,
                tensor_layout="reference", group=None, redundant_halo=False, native_halo=False, noncyclic_clamp=None,
                partition="contiguous", window=CHUNK_ROWS):
    """This rank's rows of (sigma, x_dep, y_dep) for a row-sharded seed grid.

    ``partition="interleaved"`` (strong scaling; see the comment above :func:`interleaved_chunks`): the rank owns the row
    chunks ``rank, rank + world, ...``, advects them in ONE call and exchanges every chunk's halo rows with the previous / next
    rank (a ring; ``redundant_halo=True``: advects the halo rows itself instead, no communication -- bit-identical, the
    cross-check); the result's ``"rows"`` is then the list of its chunks ``[(lo, hi), ...]`` and ``sigma`` / ``x_dep`` /
    ``y_dep`` hold those rows concatenated in that order (``"global_rows"``: the global row of every output row).  Values are
    bit-identical to the unsharded run's rows.  The reference's non-cyclic outer-product clamp couples all rows through the
    ranks' flags and keeps the contiguous partition; so do grids whose chunks do not deal out evenly.

    ``redundant_halo=True`` advects the halo rows locally instead of exchanging them
    (0.1 % extra work at 4096 rows/GPU); the results are bit-identical and the
    tests use it to check the exchange.  ``native_halo=True`` exchanges through the C ABI
    (``lc_halo_exchange``, RCCL directly) instead of ``torch.distributed`` point-to-point.

    ``cyclic_xboundary=False`` is the reference's outer-product longitude clamp (LCS/trajectory.py:96-97, Q9), which
    couples all seed rows through the offending columns: the ranks OR their column flags after every sub-step
    (``Engine.set_flag_allreduce``: a MAX all-reduce of ``nx`` flags, over the same transport as the halo), so the
    sharded result equals the unsharded one and the reference's.  Collective: every rank must make the call.
    ``noncyclic_clamp='pointwise'`` asks for the per-point clamp instead (no communication, not the reference's rule).
    """
    import numpy as np
    seed_lat_global = np.asarray(seed_lat_global, dtype=field.dtype)
    seed_lon = np.asarray(seed_lon, dtype=field.dtype)
    nyg = seed_lat_global.size
    if partition not in ("contiguous", "interleaved"):
        raise ValueError("partition: 'contiguous' or 'interleaved'")
    outer_clamp = (not cyclic_xboundary) and noncyclic_clamp in (None, "reference_outer")
    if partition == "interleaved" and not outer_clamp:
        chunks = interleaved_partition(nyg, world, rank, window)
        if len(chunks) > 1:
            return _interleaved_lcs(engine, field, seed_lat_global, seed_lon, timestep, rank, world, SETTLS_order, interp_order,
                                    cyclic_xboundary, t0, nsteps, fd_fp32_cast, tensor_layout, noncyclic_clamp, chunks, group,
                                    redundant_halo)
    lo, hi = row_partition(nyg, world, rank)
    n_lo, n_hi = halo_rows(nyg, lo, hi)
    comm = native_comm(engine, rank, world, group) if (native_halo and world > 1) else None
    outer = (not cyclic_xboundary) and noncyclic_clamp in (None, "reference_outer") and world > 1
    if outer:   # the reference's clamp on a row block: the ranks share their offending-column flags
        engine.set_flag_allreduce(group=group, comm=comm)
    try:
        if redundant_halo:
            a, b = lo - n_lo, hi + n_hi
            x_ext, y_ext = engine.advect(field, seed_lat_global[a:b], seed_lon, timestep, SETTLS_order, interp_order,
                                         cyclic_xboundary, t0, nsteps, row0=a, ny_global=nyg, noncyclic_clamp=noncyclic_clamp)
            in_row0 = a
            x, y = x_ext[n_lo:n_lo + hi - lo], y_ext[n_lo:n_lo + hi - lo]
        else:
            x_ext, y_ext = engine.advect(field, seed_lat_global[lo:hi], seed_lon, timestep, SETTLS_order, interp_order,
                                         cyclic_xboundary, t0, nsteps, row0=lo, ny_global=nyg, halo=(n_lo, n_hi),
                                         noncyclic_clamp=noncyclic_clamp)
            halo_exchange_into(x_ext, y_ext, n_lo, n_hi, rank, world, group, engine=engine, comm=comm)
            in_row0 = lo - n_lo
            x, y = x_ext[n_lo:n_lo + hi - lo], y_ext[n_lo:n_lo + hi - lo]
    finally:
        if outer:
            engine.set_flag_allreduce(enable=False)
    dlat = float(seed_lat_global[1] - seed_lat_global[0])
    dlon = float(seed_lon[1] - seed_lon[0])
    sig = engine.sigma(x_ext, y_ext, seed_lat_global[in_row0:in_row0 + x_ext.shape[0]], dlat, dlon, ny_global=nyg,
                       in_row0=in_row0, out_row0=lo, n_out_rows=hi - lo, fd_fp32_cast=fd_fp32_cast,
                       tensor_layout=tensor_layout)
    return {"sigma": sig, "x_dep": x, "y_dep": y, "rows": (lo, hi)}


def _interleaved_lcs(engine, field, seed_lat_global, seed_lon, timestep, rank, world, SETTLS_order, interp_order,
                     cyclic_xboundary, t0, nsteps, fd_fp32_cast, tensor_layout, noncyclic_clamp, chunks, group, redundant_halo):
    import numpy as np
    import torch
    nyg = seed_lat_global.size
    rows = np.asarray(interleaved_rows(nyg, chunks, with_halo=redundant_halo), dtype=np.int64)
    # one advect over the concatenated chunks (redundant_halo: over their windows, and then nothing is exchanged)
    x_all, y_all = engine.advect(field, seed_lat_global[rows], seed_lon, timestep, SETTLS_order, interp_order, cyclic_xboundary,
                                 t0, nsteps, ny_global=nyg, global_rows=rows, noncyclic_clamp=noncyclic_clamp)
    if redundant_halo:
        x_win, y_win = x_all, y_all
    else:
        x_win, y_win = chunk_halo_exchange(x_all, y_all, chunks, nyg, rank, world, group)
    dlat = float(seed_lat_global[1] - seed_lat_global[0])
    dlon = float(seed_lon[1] - seed_lon[0])
    sig, xs, ys, off = [], [], [], 0
    for lo, hi in chunks:
        a, b = chunk_window(nyg, lo, hi)
        xw, yw = x_win[off:off + b - a], y_win[off:off + b - a]
        sig.append(engine.sigma(xw, yw, seed_lat_global[a:b], dlat, dlon, ny_global=nyg, in_row0=a, out_row0=lo,
                                n_out_rows=hi - lo, fd_fp32_cast=fd_fp32_cast, tensor_layout=tensor_layout))
        xs.append(xw[lo - a:lo - a + hi - lo])
        ys.append(yw[lo - a:lo - a + hi - lo])
        off += b - a
    return {"sigma": torch.cat(sig), "x_dep": torch.cat(xs) if redundant_halo else x_all,
            "y_dep": torch.cat(ys) if redundant_halo else y_all, "rows": chunks,
            "global_rows": [r for lo, hi in chunks for r in range(lo, hi)]}


ENSEMBLE_CHUNK = 16   # time levels per launch of ensemble_advect's level-major order (measured on config 5)


def ensemble_advect(engine, field, seed_lat, seed_lon, timestep, members, nsteps: int, SETTLS_order=0, interp_order=1,
                    cyclic_xboundary=True, level_chunk=None, streams: int = 2):
    """Departure points of ensemble members (member ``e`` = start level ``t0 = e``, ``nsteps`` steps), advected in
    LEVEL-MAJOR order: every member's first ``level_chunk`` levels, then every member's next chunk, ... continuing in
    place.  Consecutive members (``members`` = ``range(a, b)``, what ``ensemble_partition`` hands a rank) go through ONE
    launch per chunk (``lc_advect_batch``: member = blockIdx.y): members ``e`` and ``e+1`` of a chunk read time levels
    ``[e + c, e + c + chunk]`` and ``[e + 1 + c, ...]`` -- all but one in common -- so the wind images stay in the
    Infinity Cache / L2, and a launch is as many times deeper as there are members, so no compute unit idles at the
    end of each member's own launch; in float32 at order 1 two consecutive members share a lane of the two-seed kernel
    (same grid point, one level of travel apart, one staged tile of the wind for both) (config 5 on one MI355X: DESIGN 4).  Any other member list falls back to one launch
    per member and chunk (``lc_advect_from``) on ``streams`` HIP streams.  Results are bit-identical to one launch per
    member (the loop of LCS/trajectory.py:80-126 carries only positions from level to level).  ``level_chunk=0``:
    member-major, one launch per member.  Returns ``[(x, y), ...]`` in the order of ``members``; everything is joined
    into the current stream before it returns."""
    import numpy as np
    import torch
    members = list(members)
    chunk = ENSEMBLE_CHUNK if level_chunk is None else int(level_chunk)
    # cyclic_xboundary=False is the reference's outer-product clamp, which is decided per member over its WHOLE series
    # (lc_advect restarts sub-step by sub-step from the chunk in which a parcel first left the box and cannot continue
    # from given positions): member-major, one call per member
    if chunk <= 0 or chunk >= nsteps or not cyclic_xboundary:
        chunk = max(nsteps, 1)
    dtype = getattr(torch, np.dtype(field.dtype).name)
    ny, nx = len(seed_lat), len(seed_lon)
    slat, slon = engine.to_device(seed_lat, field.dtype), engine.to_device(seed_lon, field.dtype)
    consecutive = len(members) > 1 and members == list(range(members[0], members[0] + len(members)))
    if consecutive and cyclic_xboundary:
        prev = engine.level_chunk
        try:
            engine.set_level_chunk(chunk)
            x, y = engine.advect_batch(field, slat, slon, timestep, len(members), nsteps, SETTLS_order, interp_order, True,
                                       t0=members[0], t0_stride=1)
        finally:
            engine.set_level_chunk(prev)
        return [(x[i], y[i]) for i in range(len(members))]
    pos = [(torch.empty((ny, nx), dtype=dtype, device=engine.device), torch.empty((ny, nx), dtype=dtype, device=engine.device))
           for _ in members]
    cur = torch.cuda.current_stream(engine.device)
    side = [torch.cuda.Stream(engine.device) for _ in range(int(streams))] if int(streams) > 1 and len(members) > 1 else []
    for st in side:
        st.wait_stream(cur)                  # the field, the seeds and the buffers above belong to the current stream
    with engine.concurrent_calls(ny * nx, max(len(side), 1)):
        for c0 in range(0, max(nsteps, 1), chunk):
            n = min(chunk, nsteps - c0)
            for i, e in enumerate(members):  # member i keeps its stream: its chunks stay in order
                kw = dict(t0=e + c0, nsteps=n, start=pos[i] if c0 else None, out=pos[i])
                if side:
                    with torch.cuda.stream(side[i % len(side)]):
                        engine.advect(field, slat, slon, timestep, SETTLS_order, interp_order, cyclic_xboundary, **kw)
                else:
                    engine.advect(field, slat, slon, timestep, SETTLS_order, interp_order, cyclic_xboundary, **kw)
    for st in side:
        cur.wait_stream(st)
    return pos


def ensemble_lcs(engine, field, seed_lat, seed_lon, timestep, n_members: int, nsteps: int, rank: int = 0,
                 world: int = 1, SETTLS_order=0, interp_order=1, cyclic_xboundary=True, fd_fp32_cast=True,
                 tensor_layout="reference", return_dpts=False, streams: int = 2, level_chunk=None):
    """BASELINE config 5: member ``e`` starts at time level ``t0 = e`` and runs ``nsteps`` steps over the
    same seed grid.  Members are sharded over ranks in contiguous blocks; nothing is exchanged (gathering
    the sigma fields is the caller's business).  Returns ``(member_indices, sigma[len(members), ny, nx])``,
    with ``return_dpts`` also the members' departure points ``x_dep, y_dep`` (same shape).

    ``streams``: a rank's members are independent, so they alternate between this many HIP streams and one
    member's last workgroups (the tail of its launch) run beside the next member's first ones (config 5 on one
    MI355X: 421 -> 344 ms with 2, 370 with 3).  Results do not depend on it.  All streams are joined before the
    function returns: the outputs are ordinary tensors of the current stream."""
    import torch
    if n_members - 1 + nsteps > field.nt - 1:
        raise ValueError(f"{n_members} members x {nsteps} steps need {n_members + nsteps} time levels, have {field.nt}")
    mine = ensemble_partition(n_members, world, rank)
    out, xs, ys = [], [], []
    if (ENSEMBLE_CHUNK if level_chunk is None else int(level_chunk)) > 0 and len(mine) > 1:
        # level-major advection of all of this rank's members (ensemble_advect), then sigma member by member
        import numpy as np
        seed_lat = np.asarray(seed_lat, dtype=field.dtype)
        seed_lon = np.asarray(seed_lon, dtype=field.dtype)
        pos = ensemble_advect(engine, field, seed_lat, seed_lon, timestep, mine, nsteps, SETTLS_order, interp_order,
                              cyclic_xboundary, level_chunk, streams)
        dlat, dlon = float(seed_lat[1] - seed_lat[0]), float(seed_lon[1] - seed_lon[0])
        for x, y in pos:
            out.append(engine.sigma(x, y, seed_lat, dlat, dlon, fd_fp32_cast=fd_fp32_cast, tensor_layout=tensor_layout))
        st = lambda a: torch.stack(a) if a else None
        if return_dpts:
            return mine, st(out), st([p[0] for p in pos]), st([p[1] for p in pos])
        return mine, st(out)
    cur = torch.cuda.current_stream(engine.device)
    side = [torch.cuda.Stream(engine.device) for _ in range(int(streams))] if int(streams) > 1 and len(mine) > 1 else []

    def one(e):
        return engine.lcs(field, seed_lat, seed_lon, timestep, SETTLS_order=SETTLS_order, interp_order=interp_order,
                          cyclic_xboundary=cyclic_xboundary, t0=e, nsteps=nsteps, fd_fp32_cast=fd_fp32_cast,
                          tensor_layout=tensor_layout)
    with engine.concurrent_calls(len(seed_lat) * len(seed_lon), max(len(side), 1)):
        for i, e in enumerate(mine):
            if side:
                st = side[i % len(side)]
                st.wait_stream(cur)          # the field and the seeds were produced on the current stream
                with torch.cuda.stream(st):
                    r = one(e)
                for v in r.values():
                    if isinstance(v, torch.Tensor):
                        v.record_stream(cur)  # consumed on the current stream below and by the caller
            else:
                r = one(e)
            out.append(r["sigma"])
            if return_dpts:
                xs.append(r["x_dep"])
                ys.append(r["y_dep"])
    for st in side:
        cur.wait_stream(st)
    st = lambda a: torch.stack(a) if a else None
    return (mine, st(out), st(xs), st(ys)) if return_dpts else (mine, st(out))
