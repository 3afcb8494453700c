"""world_size-2/3 gloo tests (CPU) of the slab decomposition the multi-GPU path uses.

What runs here is the decomposition ALGORITHM with real message passing -- not the HIP kernels:
  * every rank takes the rows of the oracle's A^T A that belong to its slab (field_interpolation_amd.dist
    .slab_range, the partition rule libfi_hip uses) and checks that their columns stay within
    dist.halo_width(weights) planes of the slab -- i.e. the ghost-plane count the GPU path allocates is enough;
  * a Jacobi-PCG runs with one halo exchange of the search direction per iteration (send/recv with the two
    neighbouring ranks) and all-reduced dot products -- the communication pattern of fi_comm.hip -- and
    must reproduce the single-process oracle PCG iterate for iterate.
The same decomposition on real HIP kernels is covered on the GPU box by tests/test_gpu_slabs.py.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem(oracle, sizes, kw):
    rng = np.random.default_rng(5)
    D = len(sizes)
    c = (np.array(sizes) - 1) / 2.0
    d = rng.normal(size=(150, D))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos = (c + 0.3 * (min(sizes) - 1) * d + rng.normal(scale=0.3, size=d.shape)).astype(np.float32)
    w = oracle.Weights(**kw)
    f = oracle.sdf_from_points(sizes, w, pos, d.astype(np.float32))
    return f, w


def _worker(rank, world, port, sizes, kw, max_it, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from field_interpolation_amd import dist as fdist
    from oracle import fi_oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    f, w = _problem(fi_oracle, sizes, kw)
    AtA, atb, diag = f.normal_equations()
    AtA = AtA.tocsr()
    plane = int(np.prod(sizes[:-1]))
    lo, hi = fdist.slab_range(sizes[-1], rank, world)
    H = fdist.halo_width(w)
    glo, ghi = max(lo - H, 0), min(hi + H, sizes[-1])
    rows = AtA[lo * plane:hi * plane]
    # the halo-width rule: no coupling beyond H planes
    cols = rows.tocoo().col
    assert cols.min() >= glo * plane and cols.max() < ghi * plane, "halo width %d too small" % H
    A_loc = rows[:, glo * plane:ghi * plane]
    own = slice((lo - glo) * plane, (lo - glo) * plane + (hi - lo) * plane)
    b, dinv = atb[lo * plane:hi * plane], 1.0 / diag[lo * plane:hi * plane]

    def exchange(v):          # v: local vector incl. ghost planes; fill ghosts from the neighbours
        reqs, recv_lo, recv_hi = [], None, None
        if rank > 0:
            n = (lo - glo) * plane
            recv_lo = torch.empty(n, dtype=torch.float64)
            reqs.append(dist.isend(torch.from_numpy(v[own][:n].copy()), rank - 1))
            reqs.append(dist.irecv(recv_lo, rank - 1))
        if rank + 1 < world:
            n = (ghi - hi) * plane
            recv_hi = torch.empty(n, dtype=torch.float64)
            reqs.append(dist.isend(torch.from_numpy(v[own][-n:].copy()), rank + 1))
            reqs.append(dist.irecv(recv_hi, rank + 1))
        for r in reqs:
            r.wait()
        if recv_lo is not None:
            v[:own.start] = recv_lo.numpy()
        if recv_hi is not None:
            v[own.stop:] = recv_hi.numpy()

    def allsum(*vals):
        t = torch.tensor(vals, dtype=torch.float64)
        dist.all_reduce(t)
        return t.tolist()

    x = np.zeros(hi * plane - lo * plane)
    r = b.copy()
    z = dinv * r
    p = np.zeros(A_loc.shape[1])
    p[own] = z
    (rz,) = allsum(float(r @ z))
    for _ in range(max_it):
        exchange(p)
        q = A_loc @ p
        (pq,) = allsum(float(p[own] @ q))
        a = rz / pq
        x += a * p[own]
        r -= a * q
        z = dinv * r
        (rz_new,) = allsum(float(r @ z))
        p[own] = z + (rz_new / rz) * p[own]
        rz = rz_new
    gathered = [None] * world
    dist.all_gather_object(gathered, x)
    if rank == 0:
        np.save(out_path, np.concatenate(gathered))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,sizes,kw", [
    (2, [10, 16], dict()),                                   # model_2: two ghost planes
    (3, [6, 5, 12], dict(model_2=0.0, model_1=0.5)),         # model_1: one ghost plane
    (2, [5, 4, 14], dict(model_4=0.3, model_2=0.2)),         # model_4: four ghost planes
])
def test_slab_pcg_over_gloo_matches_single_process(oracle, tmp_path, world, sizes, kw):
    import torch.multiprocessing as mp
    out = str(tmp_path / "x.npy")
    max_it = 25
    mp.spawn(_worker, args=(world, _free_port(), sizes, kw, max_it, out), nprocs=world, join=True)
    x = np.load(out)
    f, _ = _problem(oracle, sizes, kw)
    x_ref, it, _ = f.solve_pcg(np.zeros(f.num_unknowns), max_it, 1e-300, use_double=True)
    assert it == max_it
    np.testing.assert_allclose(x, x_ref, rtol=0, atol=1e-9 * np.abs(x_ref).max())


def test_partition_rule_matches_library():
    """dist.slab_range / dist.halo_width are the same rules libfi_hip applies (host-only entry points)."""
    import ctypes as C
    from field_interpolation_amd import _capi, api, dist as fdist
    L = _capi.lib()
    for planes in (1, 7, 256, 1000):
        for n in (1, 2, 3, 8):
            if n > planes:
                continue
            covered = 0
            for r in range(n):
                lo, hi = C.c_int(), C.c_int()
                _capi.check(L.fi_slab_partition(planes, r, n, C.byref(lo), C.byref(hi)))
                assert (lo.value, hi.value) == fdist.slab_range(planes, r, n)
                assert lo.value == covered and hi.value > lo.value
                covered = hi.value
            assert covered == planes
    for kw, expect in ((dict(), 2), (dict(model_2=0, model_1=1), 1), (dict(model_2=0), 1), (dict(model_4=0.1), 4),
                       (dict(model_3=1.0, model_2=0.0), 3)):
        w = api.Weights(**kw)
        width = C.c_int()
        cw = w._c()
        _capi.check(L.fi_halo_width(C.byref(cw), C.byref(width)))
        assert width.value == fdist.halo_width(w) == expect


def test_points_of_slab_keeps_every_touching_cell():
    from field_interpolation_amd import dist as fdist
    rng = np.random.default_rng(0)
    pos = rng.uniform(-2, 34, size=(2000, 3)).astype(np.float32)
    for lo, hi in ((0, 8), (8, 20), (20, 32)):
        keep = fdist.points_of_slab(pos, 3, lo, hi)
        cz = np.floor(pos[:, 2])
        must = (cz >= lo - 1) & (cz <= hi - 1)
        assert not (must & ~keep).any()


def test_points_of_slab_covers_the_cells_of_every_coarse_level():
    """A rank's coarser levels are assembled from its own points (positions / 2^l): the margin must cover every cell of
    level l that touches the rank's coarse planes ceil(lo / 2^l) .. ceil(hi / 2^l) - 1 (build_levels' rule), plus one
    cell for the nearest-neighbour kernels -- the round-1 margin (2 fine cells) lost cells of level 2 and deeper."""
    from field_interpolation_amd import dist as fdist
    rng = np.random.default_rng(1)
    planes = 512
    pos = rng.uniform(-3, planes + 2, size=(20000, 3)).astype(np.float32)
    z = pos[:, 2].astype(np.float64)
    for nranks in (2, 3, 8):
        for levels in (0, 1, 2, 3, 5):
            for rank in range(nranks):
                lo, hi = fdist.slab_range(planes, rank, nranks)
                keep = fdist.points_of_slab(pos, 3, lo, hi, levels)
                clo, chi = lo, hi
                for lev in range(levels + 1):
                    if lev > 0:
                        clo, chi = (clo + 1) // 2, (chi + 1) // 2
                    # an even extent is halved cell-centred (coarse point j between fine 2j and 2j+1): level lev sees the
                    # point at z / 2^lev - (1 - 2^-lev) / 2; odd extents (vertex-centred) see it at z / 2^lev
                    for zl in (z / 2.0 ** lev, z / 2.0 ** lev - 0.5 * (1.0 - 2.0 ** -lev)):
                        cz = np.floor(zl)                         # cell origin on level lev
                        rz = np.round(zl)                         # nearest-neighbour row on level lev
                        must = ((cz >= clo - 1) & (cz <= chi - 1)) | ((rz >= clo) & (rz <= chi - 1))
                        assert not (must & ~keep).any(), (nranks, levels, rank, lev)
            # and the margins stay small: the ranks together upload each point a bounded number of times
            total = sum(int(fdist.points_of_slab(pos, 3, *fdist.slab_range(planes, r, nranks), levels).sum())
                        for r in range(nranks))
            assert total <= len(pos) * (1.0 + 4.0 * nranks * 2.0 ** levels / planes) + 1


def test_replicated_tail_rule_of_the_point_filter():
    """dist.points_of_slab mirrors fi_slab_point_range: once the deepest levels of a slab hierarchy are replicated (whole
    lattices on every rank) every rank has to upload every point.  (The library's own answer is compared with this rule
    on the GPU: tests/test_gpu_slabs.py.)"""
    from field_interpolation_amd import dist as fdist
    pos = np.random.default_rng(0).uniform(0, 63, size=(500, 3)).astype(np.float32)
    # 64 planes over 4 ranks: level 1 -> 8 planes per slab, level 2 -> 4, level 3 -> 2 < 4: replicated from level 3 on
    assert not fdist.has_replicated_tail([64, 64, 64], 4, 2)
    assert fdist.has_replicated_tail([64, 64, 64], 4, 3)
    assert not fdist.has_replicated_tail([64, 64, 64], 1, 3)
    assert not fdist.has_replicated_tail([64, 64, 64], 8, 0)
    lo, hi = fdist.slab_range(64, 1, 4)
    part = fdist.points_of_slab(pos, 3, lo, hi, 2, sizes=[64, 64, 64], nranks=4)
    assert 0 < part.sum() < len(pos)
    assert fdist.points_of_slab(pos, 3, lo, hi, 3, sizes=[64, 64, 64], nranks=4).all()
    # levels that do not exist (an axis below 8 points) cannot be replicated
    assert not fdist.has_replicated_tail([16, 16, 16], 2, 5)
