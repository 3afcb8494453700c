"""Host-side pieces of the slab decomposition (one process per GPU, torch.distributed launcher).

The reference is single-process; these helpers exist because the lattice is split
into contiguous slabs of its slowest axis, one per rank:
  slab_range      which planes a rank owns            (same arithmetic as fi_slab_partition in libfi_hip)
  halo_width      ghost planes per side               (same rule as fi_halo_width)
  points_of_slab  data points a rank has to see       (cells touching its slab on every level: fi_slab_point_range)
  init_comm       RCCL bootstrap: rank 0 makes the unique id, torch.distributed broadcasts it
Nothing here touches the GPU except init_comm.
"""
import numpy as np


def slab_range(planes, rank, nranks):
    """Planes [lo, hi) of the slowest axis owned by `rank` (floor(rank*planes/nranks) rule)."""
    if not (nranks >= 1 and 0 <= rank < nranks and planes >= 0):
        raise ValueError("bad slab request")
    return (rank * planes) // nranks, ((rank + 1) * planes) // nranks


def halo_width(weights):
    """Ghost planes per side: reach of the widest enabled model stencil (AtA of a k-th difference reaches
    +-k), at least 1 for the cell blocks / gradient_smoothness."""
    reach = 0
    for k, w in enumerate((weights.model_1, weights.model_2, weights.model_3, weights.model_4), start=1):
        if w > 0:
            reach = k
    return max(reach, 1)


def point_range(lo, hi, levels=0):
    """[zlo, zhi) of the slowest coordinate of the points a rank owning planes [lo, hi) has to see (the rule of
    fi_slab_point_range): a cell of coarse level l spans 2^l fine planes and the rank's coarse replicas are built from
    its own points, so the margin grows with the level count -- two cells of the coarsest level below (cell origin
    floor(pos) one plane under the slab, plus one cell for the nearest-neighbour kernels), one above."""
    cell = float(1 << max(int(levels), 0))
    # (two cells above as well once there are levels: a level halved cell-centred sees a point at
    # position / 2^l - (1 - 2^-l) / 2, up to half a coarse cell further down)
    return lo - 2.0 * cell, hi + (2.0 if levels > 0 else 1.0) * cell


def has_replicated_tail(sizes, nranks, levels, reach=2):
    """plan_levels of the library (fi_levels.hip): does a hierarchy of `levels` coarser levels over `nranks` slabs end in
    REPLICATED levels (whole lattices on every rank, once a level's slabs would be thinner than max(reach, 4) planes)?
    Such levels are assembled from ALL the points: fi_slab_point_range then returns everything.  (The library's test switch
    FI_NO_REPLICATED_TAIL cuts the hierarchy instead: honoured here too.)"""
    import os
    if nranks <= 1 or levels <= 0 or os.environ.get("FI_NO_REPLICATED_TAIL"):
        return False
    n = [int(s) for s in sizes]
    planes = n[-1]
    lo = [(r * planes) // nranks for r in range(nranks)]
    hi = [((r + 1) * planes) // nranks for r in range(nranks)]
    for _ in range(int(levels)):
        n = [(s + 1) // 2 for s in n]
        if min(n) < 8:
            return False
        lo = [(v + 1) // 2 for v in lo]
        hi = [(v + 1) // 2 for v in hi]
        if min(h - l for l, h in zip(lo, hi)) < max(reach, 4):
            return True
    return False


def points_of_slab(positions, ndim, lo, hi, levels=0, sizes=None, nranks=1, reach=2):
    """Boolean mask of the points a rank owning planes [lo, hi) of the slowest axis has to upload, for a context
    with `levels` coarser levels (FI_OPT_LEVELS).  With `sizes` and `nranks` the replicated tail of the hierarchy is
    honoured like fi_slab_point_range does: every point when the deepest levels are whole lattices on every rank.
    (LatticeField.point_range() asks the library itself and is the rule to prefer.)"""
    z = np.asarray(positions, np.float32).reshape(-1, ndim)[:, ndim - 1]
    if sizes is not None and has_replicated_tail(sizes, nranks, levels, reach):
        return np.ones(len(z), bool)
    zlo, zhi = point_range(lo, hi, levels)
    return (z >= zlo) & (z < zhi)


_HOST_SEGMENTS = 0


def init_comm(field, device=None, host_staged=False):
    """Create the communicator of a slab LatticeField.  RCCL: the 128-byte unique id made by rank 0 travels through
    torch.distributed (any backend) together with a status byte, so that a failure on rank 0 raises on EVERY rank
    instead of leaving the others in the broadcast; then every rank calls fi_comm_init.  host_staged (tests on a
    one-GPU machine, where RCCL refuses two ranks on one device): the shared-memory test transport of
    fi_comm_init_host -- rank 0 creates the segment, the others attach after a barrier.
    A failure of fi_comm_init itself on some rank is the caller's to agree on (bench.py does) before the next collective."""
    import ctypes
    import os
    import torch
    import torch.distributed as dist
    from . import _capi
    if field.nranks == 1:
        return
    rank0 = dist.get_rank() == 0
    if host_staged:
        global _HOST_SEGMENTS
        _HOST_SEGMENTS += 1                             # every rank makes the same sequence of communicators
        name = "/fi_host_%s_%d_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid(), _HOST_SEGMENTS)
        t = torch.zeros(1, dtype=torch.int32)
        err = None
        if rank0:
            try:
                try:
                    os.unlink("/dev/shm" + name)       # a segment left behind by a killed run
                except OSError:
                    pass
                field.comm_init_host(name, True)
            except Exception as e:                      # noqa: BLE001 -- told to every rank below
                err = e
                t[0] = 1
        dist.broadcast(t, 0)                            # also the barrier: the segment exists before anybody attaches
        if int(t[0]) != 0:
            raise RuntimeError("rank 0 could not create the host segment: %s" % (err or "see rank 0"))
        if not rank0:
            field.comm_init_host(name, False)
        return
    buf = ctypes.create_string_buffer(128)
    status = 0
    err = None
    if rank0:
        try:
            _capi.check(_capi.lib().fi_comm_unique_id(buf))
        except Exception as e:                          # noqa: BLE001 -- told to every rank below
            status, err = 1, e
    t = torch.frombuffer(bytearray(bytes([status]) + buf.raw), dtype=torch.uint8).clone()
    if device is not None and dist.get_backend() == "nccl":
        t = t.to(device)
    dist.broadcast(t, 0)
    raw = bytes(t.cpu().numpy().tobytes())
    if raw[0] != 0:
        raise RuntimeError("rank 0 could not create the RCCL unique id: %s" % (err or "see rank 0"))
    field.comm_init(raw[1:129])
