"""P1/P2 parity on the GPU (SURVEY.md 8(d) parity protocol): the HIP operator pieces -- AtA*x, A^T b,
diag(AtA) -- against the explicit float64 normal equations the oracle builds from reference-equal
triplets.  Tolerances: fp64 contexts 1e-12 (relative to the largest entry), fp32 contexts 2e-6."""
import numpy as np
import pytest

from util import build_pair, random_points, rel_inf, sphere_points

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-12, "f32": 2e-6}


@pytest.fixture(scope="module")
def fi():
    import field_interpolation_amd as fi
    from field_interpolation_amd import _capi
    assert _capi.device_count() >= 1, "no HIP device visible"
    return fi


def _check_operator(fo, fg, dtype, seed=0):
    AtA, atb, diag = fo.normal_equations()
    n = fo.num_unknowns
    rng = np.random.default_rng(seed)
    assert rel_inf(fg.Atb(), atb) <= TOL[dtype] if np.abs(atb).max() > 0 else not fg.Atb().any()
    assert rel_inf(fg.diag(), diag) <= TOL[dtype]
    absA = abs(AtA)
    for k in range(2):
        x = rng.normal(size=n) if k == 0 else np.linspace(-50, 50, n) + rng.normal(size=n)
        y = fg.apply_AtA(x)
        y_ref = AtA @ x
        scale = (absA @ np.abs(x)).max()        # |A^T A| |x|: the natural rounding scale of the product
        assert np.abs(y - y_ref).max() <= TOL[dtype] * scale


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("sizes", [[33], [17, 12], [9, 8, 7]])
@pytest.mark.parametrize("vk,gk", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_default_model_all_cell_kernels(oracle, fi, dtype, sizes, vk, gk):
    rng = np.random.default_rng(7 + len(sizes) + 10 * vk + gk)
    pos, nrm, pw, val = random_points(rng, sizes, 120)
    w = fi.Weights(data_pos=0.8, data_gradient=1.25, value_kernel=fi.ValueKernel(vk), gradient_kernel=fi.GradientKernel(gk))
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, pw, val, dtype=dtype)
    fg.assemble()
    st = fg.stats()
    assert st["num_data_rows"] == fo.num_rows - _model_rows(oracle, sizes, w)
    _check_operator(fo, fg, dtype)


def _model_rows(oracle, sizes, w):
    from util import oracle_weights
    f = oracle.LatticeField(sizes)
    f.add_field_constraints(oracle_weights(oracle, w))
    return f.num_rows


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("sizes", [[40], [14, 11], [8, 7, 9]])
@pytest.mark.parametrize("kw", [
    dict(model_0=0.3, model_1=0.7, model_2=0.5, model_3=0.9, model_4=1.1),
    dict(model_2=0.0, model_1=1.0),
    dict(model_2=0.0, model_4=0.6),
    dict(model_2=0.5, gradient_smoothness=0.4),
    dict(model_2=0.0, model_3=0.3, gradient_smoothness=1.5, model_0=0.05),
])
def test_every_model_term(oracle, fi, dtype, sizes, kw):
    rng = np.random.default_rng(3)
    pos, nrm, pw, val = random_points(rng, sizes, 40, margin=0.3)
    w = fi.Weights(**kw)
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, pw, None, dtype=dtype)
    _check_operator(fo, fg, dtype)


def test_tiny_and_degenerate_lattices(oracle, fi):
    """Lattices narrower than the stencils: rows that do not fit are not emitted (cpp:265-292)."""
    for sizes in ([1], [2], [3], [4, 1], [2, 2], [1, 5, 2], [3, 2, 2]):
        rng = np.random.default_rng(len(sizes))
        pos, nrm, pw, val = random_points(rng, sizes, 30, margin=0.6)
        w = fi.Weights(model_0=0.2, model_1=0.3, model_2=0.5, model_3=0.7, model_4=0.9, gradient_smoothness=0.2)
        fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, pw, val, dtype="f64")
        _check_operator(fo, fg, "f64")


def test_no_points_and_all_points_outside(oracle, fi):
    sizes = [10, 9]
    w = fi.Weights()
    fg = fi.LatticeField(sizes, dtype="f64")
    fg.add_field_constraints(w)
    assert not fg.Atb().any()
    pos = np.array([[-5.0, 3.0], [100.0, 2.0], [3.0, -7.5], [np.nan, 1.0]], np.float32)
    nrm = np.ones_like(pos)
    fo, fg = build_pair(oracle, fi, sizes, w, pos[:3], nrm[:3], None, None, dtype="f64")
    fg.assemble()
    assert fg.stats()["num_data_rows"] == 0 and fg.stats()["num_cells"] == 0
    _check_operator(fo, fg, "f64")
    fg.add_points(1.0, 1, 1.0, 1, pos[3:], nrm[3:])     # NaN position: ignored, not a crash
    fg.assemble()
    assert fg.stats()["num_data_rows"] == 0


def test_many_points_per_cell_and_multiple_batches(oracle, fi):
    """Thousands of rows in one cell (segmented accumulation) and rows arriving in several calls."""
    sizes = [6, 5, 4]
    rng = np.random.default_rng(11)
    pos = (np.array([2.0, 1.0, 1.0]) + rng.uniform(0, 1, (5000, 3))).astype(np.float32)
    nrm = rng.normal(size=(5000, 3)).astype(np.float32)
    w = fi.Weights()
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, None, None, dtype="f64")
    pos2, nrm2, pw2, _ = random_points(rng, sizes, 64)
    fo.add_points(0.5, 1, 2.0, 0, pos2, nrm2, pw2)
    fg.add_points(0.5, 1, 2.0, 0, pos2, nrm2, pw2)
    _check_operator(fo, fg, "f64")
    assert fg.stats()["num_cells"] <= 5 * 4 * 3 + 64 * 1


def test_single_constraint_calls_match_reference_returns(oracle, fi):
    sizes = [4, 4]
    fo = oracle.LatticeField(sizes)
    fg = fi.LatticeField(sizes, dtype="f64")
    for pos in ([-1.5, 1.0], [-0.5, 1.0], [3.0, 3.0], [3.5, 3.5], [4.0, 1.0], [1.25, 2.75]):
        assert fo.add_value_constraint(pos, 1.5, 0.9) == fg.add_value_constraint(pos, 1.5, 0.9)
    assert not fg.add_value_constraint([1.0, 1.0], 1.0, 0.0)
    for k in (0, 1):
        for pos in ([2.5, 2.5], [3.0, 2.5], [-0.1, 2.5]):
            assert fo.add_gradient_constraint(pos, [1, -2], 0.7, k) == fg.add_gradient_constraint(pos, [1, -2], 0.7, k)
    for pos in ([3.4, -0.4], [3.5, 0.0], [0.0, -0.5], [0.49, 2.51]):
        assert (fo.add_value_constraint_nearest_neighbor(pos, [1, 1], 2.0, 1.1)
                == fg.add_value_constraint_nearest_neighbor(pos, [1, 1], 2.0, 1.1))
    with pytest.raises(ValueError):
        fg.add_gradient_constraint([1.0, 1.0], [1, 0], 1.0, 7)        # reference: ABORT_F (cpp:238)
    w = fi.Weights()
    from util import oracle_weights
    fo.add_field_constraints(oracle_weights(oracle, w))
    fg.add_field_constraints(w)
    _check_operator(fo, fg, "f64")


def test_error_conventions(fi):
    from field_interpolation_amd._capi import FiError
    f = fi.LatticeField([8, 8])
    with pytest.raises(FiError):       # CHECK_NOTNULL_F(normals), field_interpolation.cpp:361
        f.add_points(1.0, fi.ValueKernel.kNearestNeighbor, 1.0, fi.GradientKernel.kCellEdges,
                     np.zeros((3, 2), np.float32), None, None)
    with pytest.raises(FiError):
        f.add_points(1.0, 1, 1.0, 9, np.zeros((3, 2), np.float32), np.zeros((3, 2), np.float32), None)
    with pytest.raises(FiError):
        fi.LatticeField([4, 4, 4, 4])      # MAX_DIM = 3 (hpp:44)
    with pytest.raises(ValueError):
        fi.sdf_from_points([4, 4], fi.Weights(), None)      # CHECK_NOTNULL_F(positions), cpp:382
    # options: out-of-range values are refused, the context keeps what it had
    for bad in (-1e-5, 1.0, 2.0):
        with pytest.raises(FiError):
            f.set_field_tolerance(bad)
    f.set_field_tolerance(0.0)
    f.set_field_tolerance(1e-5)
    with pytest.raises(FiError):
        f.set_mixed_precision(True)       # an FI_F32 context


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("sizes", [[16, 9, 7], [64, 16, 8], [72, 40, 37], [132, 20, 70], [4, 3, 2], [8, 1, 5],
                                   [13, 9, 11], [67, 10, 9], [129, 9, 12], [131, 17, 6], [5, 4, 3], [255, 3, 3]])
@pytest.mark.parametrize("kw", [dict(), dict(model_2=0.0, model_1=0.8), dict(model_0=0.3, model_1=0.6, model_2=1.7)])
def test_lds_marching_kernel_3d(oracle, fi, dtype, sizes, kw):
    """The LDS-tiled z-marching kernel (fi_stencil.hip): tiles, partial tiles, chunk seams, all six boundary faces,
    and rows whose length is not a multiple of the 16-byte group (the last group of a row is stored point by
    point; its loads run on into the next row under zero masks)."""
    rng = np.random.default_rng(sum(sizes))
    pos, nrm, pw, val = random_points(rng, sizes, 200, margin=0.7)
    fo, fg = build_pair(oracle, fi, sizes, fi.Weights(**kw), pos, nrm, pw, val, dtype=dtype)
    _check_operator(fo, fg, dtype)
    x = rng.normal(size=int(np.prod(sizes)))
    np.testing.assert_array_equal(fg.apply_AtA(x), fg.apply_AtA(x))      # no atomics on any shape


def test_marching_kernel_chunk_sizes(oracle, fi, monkeypatch):
    """Every z-chunk length must give the same operator (seams between workgroups along z)."""
    sizes = [32, 18, 41]
    rng = np.random.default_rng(5)
    pos, nrm, pw, val = random_points(rng, sizes, 100, margin=0.5)
    x = rng.normal(size=int(np.prod(sizes)))
    ref = None
    for zc in ("1", "2", "3", "8", "41", "64"):
        monkeypatch.setenv("FI_ZC", zc)
        fo, fg = build_pair(oracle, fi, sizes, fi.Weights(model_1=0.3), pos, nrm, pw, val, dtype="f64")
        y = fg.apply_AtA(x)
        if ref is None:
            AtA, _, _ = fo.normal_equations()
            ref = AtA @ x
        assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()
    monkeypatch.delenv("FI_ZC")
    monkeypatch.setenv("FI_NO_MARCH", "1")
    fo, fg = build_pair(oracle, fi, sizes, fi.Weights(model_1=0.3), pos, nrm, pw, val, dtype="f64")
    assert np.abs(fg.apply_AtA(x) - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_fused_cell_blocks_dense_and_bitwise_reproducible(oracle, fi, dtype, monkeypatch):
    """Cell blocks applied inside the marching kernel: every cell of a 68 x 35 x 21 lattice occupied
    (more than 256 cells per workgroup layer, cells on tile borders, corners and chunk seams), compared
    with the oracle, with the unfused atomic kernel, and run twice for bitwise equality."""
    sizes = [68, 35, 21]
    rng = np.random.default_rng(1)
    n = 60000
    pos = np.stack([rng.uniform(-1.2, s + 0.2, n) for s in sizes], 1).astype(np.float32)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    x = rng.normal(size=int(np.prod(sizes)))
    monkeypatch.setenv("FI_ZC", "8")
    fo, fg = build_pair(oracle, fi, sizes, fi.Weights(), pos, nrm, None, None, dtype=dtype)
    AtA, _, _ = fo.normal_equations()
    ref = AtA @ x
    scale = (abs(AtA) @ np.abs(x)).max()
    y1 = fg.apply_AtA(x)
    y2 = fg.apply_AtA(x)
    np.testing.assert_array_equal(y1, y2)
    assert np.abs(y1 - ref).max() <= TOL[dtype] * scale
    monkeypatch.setenv("FI_NO_FUSE", "1")
    fo2, fg2 = build_pair(oracle, fi, sizes, fi.Weights(), pos, nrm, None, None, dtype=dtype)
    y3 = fg2.apply_AtA(x)
    assert np.abs(y3 - ref).max() <= TOL[dtype] * scale
    assert np.abs(y3 - y1).max() <= TOL[dtype] * scale


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_surface_data_split_launches_with_merged_plain_runs(oracle, fi, dtype):
    """Surface-type data (oriented points on a small sphere) in a lattice of many chunks: fewer than half of the
    workgroups hold cells, so the apply is two launches -- the fused variant over the data workgroups (8-plane
    chunks) and the plain variant over runs of consecutive empty chunks.  Operator and p.q partials against the
    oracle's explicit AtA; run twice for bitwise equality."""
    from util import sphere_points
    sizes = [70, 33, 90]
    rng = np.random.default_rng(5)
    pos, nrm = sphere_points(rng, [21, 21, 21], 3000, noise=0.2)
    pos = (pos + np.array([30.0, 6.0, 50.0])).astype(np.float32)     # a sphere of radius 6 in one corner region
    x = rng.normal(size=int(np.prod(sizes)))
    fo, fg = build_pair(oracle, fi, sizes, fi.Weights(), pos, nrm, None, None, dtype=dtype)
    AtA, _, _ = fo.normal_equations()
    ref = AtA @ x
    scale = (abs(AtA) @ np.abs(x)).max()
    y1 = fg.apply_AtA(x)
    y2 = fg.apply_AtA(x)
    np.testing.assert_array_equal(y1, y2)
    assert np.abs(y1 - ref).max() <= TOL[dtype] * scale
    # the solver consumes the kernel's p.q partials (a run of chunks writes one sum and zeros): the same recurrence
    # as the oracle's PCG must give the same iterate
    if dtype == "f64":
        guess = rng.normal(size=x.size).astype(np.float32)
        xo, ito, erro = fo.solve_pcg(guess, 25, 1e-30, use_double=True)
        _, itg, errg = fg.solve_cg(guess, 25, 1e-30)
        assert itg == ito == 25
        assert rel_inf(fg.solution_f64(), xo) <= 1e-9 and abs(errg - erro) <= 1e-6 * erro


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_gather_assembly_same_bits_as_colour_launches(fi, dtype, monkeypatch):
    """A^T b and diag(A^T A) by the gather launch (dense cell map, cells visited in colour order) and by the 2^D
    parity-colour scatter launches: the same sums in the same order, bit for bit (3-D and 2-D)."""
    rng = np.random.default_rng(9)
    for sizes, n in (([40, 36, 28], 9000), ([150, 37, 19], 60000), ([90, 70], 2500)):
        pos, nrm, pw, val = random_points(rng, sizes, n)
        got = []
        for switch in (None, "FI_NO_TILE_SUMS", "FI_NO_GATHER"):   # 3-D: LDS tiles of points / gather launch / colour launches
            monkeypatch.delenv("FI_NO_GATHER", raising=False)
            monkeypatch.delenv("FI_NO_TILE_SUMS", raising=False)
            if switch:
                monkeypatch.setenv(switch, "1")
            f = fi.LatticeField(sizes, dtype=dtype)
            f.add_field_constraints(fi.Weights(model_1=0.2))
            f.add_points(1.0, fi.ValueKernel.kLinearInterpolation, 1.0, fi.GradientKernel.kCellEdges, pos, nrm, pw, values=val)
            f.assemble()
            got.append((f.Atb().copy(), f.diag().copy()))
        for other in got[1:]:
            np.testing.assert_array_equal(got[0][0], other[0])
            np.testing.assert_array_equal(got[0][1], other[1])
    monkeypatch.delenv("FI_NO_GATHER", raising=False)
    monkeypatch.delenv("FI_NO_TILE_SUMS", raising=False)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_packed_blocks_and_factor_rows_agree(oracle, fi, dtype, monkeypatch):
    """Contexts of mostly multi-row cells keep cells of >= 3 rows as packed blocks (PACK kernel variants); with
    FI_NO_PACK the same cells go through the factor-row loop.  Both against the oracle's explicit AtA."""
    sizes = [50, 21, 37]
    rng = np.random.default_rng(3)
    n = 30000                                          # ~0.8 points, i.e. ~3 rows, per cell: 1 to 20 rows per occupied cell
    pos = np.stack([rng.uniform(-1.2, s + 0.2, n) for s in sizes], 1).astype(np.float32)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    x = rng.normal(size=int(np.prod(sizes)))
    ys = []
    for no_pack in (False, True):
        if no_pack:
            monkeypatch.setenv("FI_NO_PACK", "1")
        else:
            monkeypatch.delenv("FI_NO_PACK", raising=False)
        fo, fg = build_pair(oracle, fi, sizes, fi.Weights(), pos, nrm, None, None, dtype=dtype)
        if not ys:
            AtA, _, _ = fo.normal_equations()
            ref = AtA @ x
            scale = (abs(AtA) @ np.abs(x)).max()
        y = fg.apply_AtA(x)
        np.testing.assert_array_equal(y, fg.apply_AtA(x))
        assert np.abs(y - ref).max() <= TOL[dtype] * scale
        ys.append(y)
    assert np.abs(ys[0] - ys[1]).max() <= TOL[dtype] * scale


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_two_row_cells_on_tile_and_chunk_seams(oracle, fi, dtype, monkeypatch):
    """Cells that hold exactly two value rows travel as record pairs in the prefetched row stream (two scatter passes);
    on the corner of a tile AND of a chunk such a cell has 8 memberships and stays a block record.  A lattice three
    128-wide tiles across with two points in every cell of the planes around the tile / chunk seams, and single
    points elsewhere (so that the context is row-dominated: the non-PACK variant), against the oracle's AtA."""
    sizes = [258, 17, 26]
    monkeypatch.setenv("FI_ZC", "8")                        # chunk seams at z = 7|8, 15|16, 23|24
    rng = np.random.default_rng(21)
    cells = []
    for cz in (6, 7, 8, 15, 22, 23):                         # cells whose corners straddle the chunk seams
        for cy in (6, 7, 8, 14, 15):                         # ... and the 8-row tile seams
            for cx in list(range(120, 136)) + [126, 127, 128, 255, 256]:
                cells.append((cx, cy, cz))
    cells = np.array(cells, np.float32)
    two = np.repeat(cells, 2, axis=0) + rng.uniform(0.05, 0.95, (2 * len(cells), 3)).astype(np.float32)
    one = np.stack([rng.uniform(0, s - 1, 4000) for s in sizes], 1).astype(np.float32)
    pos = np.concatenate([two, one])
    val = rng.normal(size=len(pos)).astype(np.float32)
    x = rng.normal(size=int(np.prod(sizes)))
    w = fi.Weights(model_2=0.7)
    fo = oracle.LatticeField(sizes)
    fo.add_field_constraints(oracle.Weights(model_2=0.7))
    fo.add_value_constraints(pos, val, w.data_pos)
    fg = fi.LatticeField(sizes, dtype=dtype)
    fg.add_field_constraints(w)
    fg.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    AtA, atb, _ = fo.normal_equations()
    ref = AtA @ x
    scale = (abs(AtA) @ np.abs(x)).max()
    y1 = fg.apply_AtA(x)
    np.testing.assert_array_equal(y1, fg.apply_AtA(x))
    assert np.abs(y1 - ref).max() <= TOL[dtype] * scale
    assert np.abs(fg.Atb() - atb).max() <= TOL[dtype] * np.abs(atb).max()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", ["rows", "packed", "factored"])
def test_blocks_by_eight_lanes_equal_the_one_thread_form(fi, dtype, case, monkeypatch):
    """3-D cell blocks are summed by eight lanes per cell (a column of the 8 x 8 block each; many-row cells of row-dominated
    fp32 contexts get their Cholesky factor by the same recurrence spread over the lanes); FI_BLOCKS_PER_THREAD runs the
    one-thread-per-cell kernel of rounds 1-3.  The same sums in the same order: A^T b, the diagonal and the operator's
    output agree bit for bit -- single / few-row cells (data rows as factor rows), a context of mostly multi-row cells
    (packed blocks), and a row-dominated context with clusters of 9-60 rows in some cells (factor rows from the block)."""
    rng = np.random.default_rng(44)
    sizes = [50, 21, 37]
    if case == "rows":
        pos = np.stack([rng.uniform(-1.2, s + 0.2, 30000) for s in sizes], 1).astype(np.float32)
        nrm = None
    elif case == "packed":
        pos = np.stack([rng.uniform(-1.2, s + 0.2, 30000) for s in sizes], 1).astype(np.float32)
        nrm = rng.normal(size=(len(pos), 3)).astype(np.float32)          # 4 rows per point: >= 3 rows in most cells
    else:
        sparse = np.stack([rng.uniform(0, s - 1, 12000) for s in sizes], 1)
        centres = np.stack([rng.integers(2, s - 3, 40) for s in sizes], 1)
        counts = rng.integers(9, 60, 40)
        clusters = np.concatenate([c + rng.uniform(0.02, 0.98, (n, 3)) for c, n in zip(centres, counts)])
        pos = np.concatenate([sparse, clusters]).astype(np.float32)
        nrm = None
    val = rng.normal(size=len(pos)).astype(np.float32)
    x = rng.normal(size=int(np.prod(sizes)))
    w = fi.Weights()
    got = []
    for old in (False, True):
        if old:
            monkeypatch.setenv("FI_BLOCKS_PER_THREAD", "1")
        else:
            monkeypatch.delenv("FI_BLOCKS_PER_THREAD", raising=False)
        f = fi.LatticeField(sizes, dtype=dtype)
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient if nrm is not None else 0.0, w.gradient_kernel, pos, nrm, None,
                     values=val)
        got.append((f.apply_AtA(x).copy(), f.Atb().copy(), f.diag().copy()))
    monkeypatch.delenv("FI_BLOCKS_PER_THREAD", raising=False)
    for a, b in zip(got[0], got[1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", ["dense", "seams", "surface", "tiles of 16 rows"])
def test_lists_from_cell_ranges_equal_the_sorted_lists(fi, dtype, case, monkeypatch):
    """The fused kernel's per-(workgroup, layer, band) record lists are read off the sorted cells as ranges (no sort,
    round 4); FI_LISTS_BY_SORT builds them the way rounds 1-3 did, by a stable radix sort of (list key, slot) pairs.
    Same lists in the same order: the operator's output must agree bit for bit -- dense data (packed blocks), two-row
    cells on tile / chunk seams (record pairs), surface data (split launches), tiles of 16 rows (5 origin rows per band)."""
    rng = np.random.default_rng(33)
    val = None
    nrm = None
    w = fi.Weights()
    if case == "dense":
        sizes = [68, 35, 21]
        pos = np.stack([rng.uniform(-1.2, s + 0.2, 60000) for s in sizes], 1).astype(np.float32)
        nrm = rng.normal(size=(len(pos), 3)).astype(np.float32)
        monkeypatch.setenv("FI_ZC", "8")
    elif case == "seams":
        sizes = [258, 17, 26]
        monkeypatch.setenv("FI_ZC", "8")
        cells = np.array([(cx, cy, cz) for cz in (6, 7, 8, 15, 22, 23) for cy in (6, 7, 8, 14, 15)
                          for cx in list(range(120, 136)) + [126, 127, 128, 255, 256]], np.float32)
        two = np.repeat(cells, 2, axis=0) + rng.uniform(0.05, 0.95, (2 * len(cells), 3)).astype(np.float32)
        one = np.stack([rng.uniform(-1, s, 4000) for s in sizes], 1).astype(np.float32)
        pos = np.concatenate([two, one])
        val = rng.normal(size=len(pos)).astype(np.float32)
        w = fi.Weights(model_2=0.7)
    elif case == "surface":
        from util import sphere_points
        sizes = [70, 33, 90]
        pos, nrm = sphere_points(rng, [21, 21, 21], 3000, noise=0.2)
        pos = (pos + np.array([30.0, 6.0, 50.0])).astype(np.float32)
    else:
        sizes = [30, 48, 19]              # (narrow in x: the tile shape of 16 rows overhangs the lattice least)
        pos = np.stack([rng.uniform(-1.2, s + 0.2, 30000) for s in sizes], 1).astype(np.float32)
        val = rng.normal(size=len(pos)).astype(np.float32)
    x = rng.normal(size=int(np.prod(sizes)))
    got = []
    for by_sort in (False, True):
        if by_sort:
            monkeypatch.setenv("FI_LISTS_BY_SORT", "1")
        else:
            monkeypatch.delenv("FI_LISTS_BY_SORT", raising=False)
        f = fi.LatticeField(sizes, dtype=dtype)
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient if nrm is not None else 0.0, w.gradient_kernel, pos, nrm, None,
                     values=val)
        got.append((f.apply_AtA(x).copy(), f.stats()["spmv_bytes"]))
    monkeypatch.delenv("FI_LISTS_BY_SORT", raising=False)
    np.testing.assert_array_equal(got[0][0], got[1][0])
    assert got[0][1] == got[1][1]          # (the distinct cells per kind behind the algorithmic bytes)


@pytest.mark.parametrize("sizes,dtype", [([30, 26], "f64"), ([30, 26], "f32"), ([12, 10, 9], "f64"), ([40], "f64")])
def test_border_prior_equals_the_reference_rows(oracle, fi, sizes, dtype):
    """src/sdf_field.cpp:218-246: every border lattice point gets add_equation(Weight{w}, Rhs{d}, {{index, 1.0f}}) with
    d = sqrt(min over the points of the squared fp32 coordinate differences).  The oracle side restates that loop with
    numpy in fp32; the device side is fi_add_border_prior (stream compaction + tiled brute force + nearest-neighbour
    value rows)."""
    rng = np.random.default_rng(len(sizes))
    pos, nrm = sphere_points(rng, sizes, 150) if len(sizes) > 1 else (rng.uniform(5, 30, size=(6, 1)).astype(np.float32), None)
    w = fi.Weights()
    bw = 0.7
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, None, None, dtype=dtype)
    # the reference loop (fp32 arithmetic, lattice order)
    D = len(sizes)
    coords = np.stack(np.meshgrid(*[np.arange(s) for s in sizes], indexing="ij"), axis=-1).reshape(-1, D)
    lin = sum(coords[:, d] * int(np.prod(sizes[:d])) for d in range(D))
    border = np.any((coords == 0) | (coords == np.array(sizes) - 1), axis=1)
    order = np.argsort(lin[border])
    bc, bl = coords[border][order], lin[border][order]
    for c, index in zip(bc, bl):
        dd = pos.astype(np.float32) - c.astype(np.float32)
        s = np.zeros(len(pos), np.float32)
        for d in range(D):
            s = (s + dd[:, d] * dd[:, d]).astype(np.float32)
        fo.add_equation(bw, float(np.sqrt(np.float32(s.min()))), [(int(index), 1.0)])
    fg.add_border_prior(bw)
    fg.assemble()
    AtA, atb, diag = fo.normal_equations()
    tol = 1e-12 if dtype == "f64" else 2e-6
    assert rel_inf(fg.Atb(), atb) <= tol
    assert rel_inf(fg.diag(), diag) <= tol
    x = rng.normal(size=int(np.prod(sizes)))
    assert rel_inf(fg.apply_AtA(x), AtA @ x) <= tol
    assert fg.stats()["num_data_rows"] == fo.num_rows - _model_rows(oracle, sizes, w)
    # a second call measures to the DATA points again (the first call's border rows are no distance source): the same
    # rows once more
    for c, index in zip(bc, bl):
        dd = pos.astype(np.float32) - c.astype(np.float32)
        s = np.zeros(len(pos), np.float32)
        for d in range(D):
            s = (s + dd[:, d] * dd[:, d]).astype(np.float32)
        fo.add_equation(bw, float(np.sqrt(np.float32(s.min()))), [(int(index), 1.0)])
    fg.add_border_prior(bw)
    fg.assemble()
    AtA, atb, diag = fo.normal_equations()
    assert rel_inf(fg.Atb(), atb) <= tol and rel_inf(fg.diag(), diag) <= tol


def test_border_prior_is_refused_on_a_slab(fi):
    """A slab context holds only the points near its slab: the nearest point of the whole cloud is not its to know."""
    from field_interpolation_amd._capi import FiError
    f = fi.LatticeField([12, 10, 16], dtype="f32", rank=1, nranks=2)
    f.add_field_constraints(fi.Weights())
    f.add_points(1.0, fi.ValueKernel.kLinearInterpolation, 0.0, fi.GradientKernel.kCellEdges,
                 np.array([[3.0, 4.0, 9.0]], np.float32), None, None)
    with pytest.raises(FiError) as e:
        f.add_border_prior(0.5)
    assert e.value.code == 5      # FI_ERR_UNSUPPORTED


def test_memory_of_destroyed_contexts_is_reused(fi, monkeypatch):
    """fi_memory_pool: the blocks of a destroyed context serve the next one (same answers, bit for bit -- a pooled block
    is handed out only after the device-wide synchronisation of fi_ctx_destroy, and every buffer is written before it is
    read); FI_NO_POOL frees them instead; trimming empties the pool."""
    import gc
    sizes = [48, 40, 36]
    rng = np.random.default_rng(8)
    pos = np.stack([rng.uniform(0, s - 1, 5000) for s in sizes], axis=1).astype(np.float32)
    val = rng.normal(size=len(pos)).astype(np.float32)
    w = fi.Weights(data_gradient=0.0)

    def solve():
        f = fi.LatticeField(sizes, dtype="f32")
        f.add_field_constraints(w)
        f.set_levels(1)
        f.set_polynomial(4)
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.assemble()
        x, it, rel = f.solve_cg(None, 0, 1e-5)
        x = x.copy()
        del f
        gc.collect()
        return x, it

    assert fi.memory_pool(0) == 0
    x0, it0 = solve()
    held = fi.memory_pool()
    assert held > 0                                    # the first context's blocks are in the pool now
    x1, it1 = solve()                                  # ... and this one lives in them
    assert it1 == it0 and np.array_equal(x0, x1)
    assert fi.memory_pool() <= held * 1.05             # nothing new had to be allocated (sizes repeat)
    assert fi.memory_pool(0) == 0
    monkeypatch.setenv("FI_NO_POOL", "1")
    x2, it2 = solve()
    monkeypatch.delenv("FI_NO_POOL", raising=False)
    assert fi.memory_pool() == 0 and it2 == it0 and np.array_equal(x0, x2)


def test_context_lifecycles_do_not_leak(fi):
    """Contexts of several shapes and solver modes created, solved and destroyed in a loop (tools/soak.py in small): the
    device memory in use levels off -- the pool holds what the largest mix of contexts needed, events and streams go with
    their context (700 cycles on the GPU box: flat at 950 MiB, host RSS flat)."""
    import ctypes
    import gc
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        assert hip.hipDeviceSynchronize() == 0
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    rng = np.random.default_rng(2)
    shapes = [[48, 40, 36], [33, 45, 29], [120, 90], [64, 48, 40]]

    def cycle(i):
        sizes = shapes[i % len(shapes)]
        dtype = "f64" if i % 3 == 0 else "f32"
        f = fi.LatticeField(sizes, dtype=dtype)
        w = fi.Weights(data_gradient=0.0)
        f.add_field_constraints(w)
        if i % 2 == 0:
            f.set_levels(2, 1e-5 if dtype == "f32" else 1e-6)
            if i % 4 == 0:
                f.set_multigrid(True)
                if dtype == "f64":
                    f.set_mixed_precision(True)
        if len(sizes) == 3 and i % 5 == 1:
            f.set_polynomial(4)
        pos = np.stack([rng.uniform(0, s - 1, 2000) for s in sizes], axis=1).astype(np.float32)
        val = rng.normal(size=len(pos)).astype(np.float32)
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.assemble()
        x, it, rel = f.solve_cg(None, 0, 1e-5)
        assert rel <= 1e-5
        del f
        gc.collect()

    for i in range(24):
        cycle(i)
    free_a = free_bytes()
    for i in range(24, 72):
        cycle(i)
    free_b = free_bytes()
    assert free_a - free_b <= 16 * 2 ** 20, (free_a - free_b) / 2 ** 20


def test_contexts_on_two_host_threads(fi):
    """Contexts are independent: two host threads, each creating, solving and destroying its own contexts at the same
    time (ctypes drops the interpreter lock during the calls), get the answers of the same work done one after the other,
    bit for bit.  What they share is behind locks: the pool of device blocks, the remembered bounds of the polynomial."""
    import threading

    def work(seed, out):
        rng = np.random.default_rng(seed)
        res = []
        for k in range(6):
            sizes = [[40, 36, 44], [52, 48], [36, 40, 32]][(seed + k) % 3]
            pos = np.stack([rng.uniform(0, s - 1, 2500) for s in sizes], axis=1).astype(np.float32)
            val = rng.normal(size=len(pos)).astype(np.float32)
            w = fi.Weights(data_gradient=0.0)
            f = fi.LatticeField(sizes, dtype="f32" if k % 2 else "f64")
            f.add_field_constraints(w)
            f.set_levels(1)
            if len(sizes) == 3:
                f.set_polynomial(4)
            f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
            f.assemble()
            x, it, rel = f.solve_cg(None, 0, 1e-5)
            res.append((x.copy(), it))
            del f
        out[seed] = res

    serial, parallel = {}, {}
    for seed in (1, 2):
        work(seed, serial)
    threads = [threading.Thread(target=work, args=(seed, parallel)) for seed in (1, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for seed in (1, 2):
        assert len(parallel[seed]) == len(serial[seed])
        for a, b in zip(serial[seed], parallel[seed]):
            assert a[1] == b[1] and np.array_equal(a[0], b[0])


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("sizes,kw", [([40, 33, 29], dict(model_3=0.7, model_2=0.3)), ([37, 41, 30], dict(model_4=0.4, gradient_smoothness=0.3)),
                                      ([300, 200], dict(model_3=0.5, model_1=0.2)), ([5000], dict(model_2=0.5, model_4=0.1))])
def test_wide_stencils_are_bitwise_reproducible(oracle, fi, monkeypatch, dtype, sizes, kw):
    """model_3 / model_4 / gradient_smoothness (field_interpolation.cpp:282-315) and 1-D lattices run the untiled kernels;
    their data cells are applied colour by colour (k_apply_cells: 2^D launches, cells of one parity share no corner), not by
    the atomic scatter of rounds 1-3 (FI_CELLS_ATOMIC) whose sums depend on the order the hardware retires the additions in.
    The same operator as the oracle's explicit AtA, the same bits on every apply, equal to the atomic form up to rounding;
    the tile operator of fi_tile_pass takes the same path."""
    rng = np.random.default_rng(17)
    n = int(0.3 * np.prod(sizes))
    pos, nrm, pw, val = random_points(rng, sizes, n, margin=0.4)
    w = fi.Weights(**kw)
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, pw, None, dtype=dtype)
    _check_operator(fo, fg, dtype)
    x = rng.normal(size=int(np.prod(sizes)))
    y = fg.apply_AtA(x)
    for _ in range(3):
        np.testing.assert_array_equal(y, fg.apply_AtA(x))
    monkeypatch.setenv("FI_CELLS_ATOMIC", "1")
    ya = fg.apply_AtA(x)
    monkeypatch.delenv("FI_CELLS_ATOMIC")
    assert np.abs(ya - y).max() <= (1e-12 if dtype == "f64" else 2e-5) * np.abs(y).max()
    if len(sizes) > 1:
        g = rng.normal(size=int(np.prod(sizes))).astype(np.float32)
        t1, t2 = fg.tile_pass(g, 8), fg.tile_pass(g, 8)
        np.testing.assert_array_equal(t1, t2)


@pytest.mark.parametrize("sizes,npts,kw", [([128, 20, 70], 400, dict()), ([132, 20, 70], 400, dict(model_2=0.0, model_1=0.8)),
                                           ([256, 30, 40], 3000, dict(model_0=0.3, model_1=0.6, model_2=1.7)),
                                           ([256, 32, 40], 3000, dict())])
def test_strip_kernel_equals_the_oracle(oracle, fi, monkeypatch, sizes, npts, kw):
    """fi_strip.hip (round 6, FI_STRIP=1: not the default -- profiles/r6_ablation.md): the fp64 apply of an undivided 3-D lattice
    as wave-private strips, one wave marching a 128 x 4 strip with DPP x neighbours and its cell wave adding the data rows
    through an LDS ring.  The same operator as the oracle's explicit AtA (value rows, pairs, factor rows and packed blocks,
    partial tiles, cells on the lattice's faces), chunk seams included, and the same bits run after run."""
    monkeypatch.setenv("FI_STRIP", "1")
    monkeypatch.setenv("FI_STRIP_CHECK", "1")
    rng = np.random.default_rng(sum(sizes))
    pos, nrm, pw, val = random_points(rng, sizes, npts, margin=0.7)
    x = rng.normal(size=int(np.prod(sizes)))
    ref = None
    for zc in (None, "7", "16"):
        if zc:
            monkeypatch.setenv("FI_STRIP_ZC", zc)
        fo, fg = build_pair(oracle, fi, sizes, fi.Weights(**kw), pos, nrm, pw, val, dtype="f64")
        y = fg.apply_AtA(x)
        if ref is None:
            AtA, _, _ = fo.normal_equations()
            ref = AtA @ x
        assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()
        np.testing.assert_array_equal(y, fg.apply_AtA(x))
    monkeypatch.delenv("FI_STRIP_ZC")
