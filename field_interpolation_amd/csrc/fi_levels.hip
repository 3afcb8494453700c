// fi_levels.hip -- coarser replicas of an assembled problem (build_levels) and the fp32 replica of a mixed-precision context
// (its own cells, or the lumped operator).  Reference role: the coarse lattice of src/sdf_field.cpp:272-288 and the weight
// rescaling rules of field_interpolation.hpp:67-73.
#include "fi_solver_internal.h"
#include "fi_workers.h"

namespace fi {

// Coarser replicas of the assembled problem, each with the lattice halved (fine point 2i <-> coarse point i):
//   * model weights rescaled so that the smoothness energy stays the same functional of the field: a k-th
//     difference on the coarse lattice is 2^k times the fine one and there are 2^D times fewer rows, hence
//     w_k,coarse^2 = w_k^2 * 2^D / 4^k  (gradient_smoothness like k = 2);
//   * the same data points, positions halved; gradients double in coarse lattice units and their rows get
//     half the weight (value rows keep theirs).
// Levels stop when an axis would drop below 8 points.  Hand-built rows (fi_add_rows_coo) have no geometry to
// coarsen: contexts holding them stay single-level.
// build_stream: the stream the levels are ASSEMBLED on (fi_assemble runs this function on a helper thread beside the
// assembly of the finest level); the levels then go back to the solver stream of `c`.
// Pure arithmetic every rank agrees on: how many of the wanted levels exist (extents >= 8) and from which level on the
// slabs would be thinner than max(halo, 4) planes -- the REPLICATED TAIL: those levels are whole lattices on every rank.
// first_tail = levels + 1 when there is none.
int plan_levels(const fi_ctx* c, int* first_tail)
{
	const int D = c->g.ndim;
	int n[3] = {c->g.gn[0], c->g.gn[1], c->g.gn[2]};
	std::vector<int> lo(c->nranks), hi(c->nranks);
	for (int r = 0; r < c->nranks; ++r) {
		lo[r] = static_cast<int>(static_cast<int64_t>(r) * n[D - 1] / c->nranks);
		hi[r] = static_cast<int>(static_cast<int64_t>(r + 1) * n[D - 1] / c->nranks);
	}
	int levels = 0, tail = 0;
	const bool allow_tail = c->nranks > 1 && !test_switch("FI_NO_REPLICATED_TAIL");
	for (int l = 1; l <= c->levels_wanted; ++l) {
		bool ok = true;
		for (int d = 0; d < D; ++d) {
			n[d] = (n[d] + 1) / 2;
			ok = ok && n[d] >= 8;
		}
		if (!ok) { break; }
		if (c->nranks > 1 && !tail) {
			bool thick = true;
			for (int r = 0; r < c->nranks; ++r) {
				lo[r] = (lo[r] + 1) / 2;
				hi[r] = (hi[r] + 1) / 2;
				thick = thick && (hi[r] - lo[r]) >= (c->reach > 4 ? c->reach : 4);
			}
			if (!thick) {
				if (!allow_tail) { break; }
				tail = l;
			}
		}
		levels = l;
	}
	if (first_tail) { *first_tail = tail ? tail : levels + 1; }
	return levels;
}

bool holds_value_rows_only(const fi_ctx* src)
{
	for (const PointBatch* b : src->batches) {
		if (b->n > 0 && b->has_nrm && b->gw != 0.0f) { return false; }
	}
	return src->generic.ntrip == 0;
}

void build_levels(fi_ctx* c, fi_ctx* src, hipStream_t build_stream)  // src: the context holding the point batches (default: c)
{
	if (c->level != 0) { return; }
	if (!src) { src = c; }
	bool wanted = c->levels_wanted > 0;
	if (wanted && c->generic.ntrip > 0) {
		// generic rows that came from points (gradient kLinearInterpolation) can be re-emitted; hand-built ones cannot
		long from_points = 0;
		for (auto* b : src->batches) {
			if (b->has_nrm && b->gk == FI_GRADIENT_LINEAR_INTERPOLATION && b->gw != 0.0f) { from_points += b->n * c->g.ndim; }
		}
		wanted = from_points == c->generic.nrows;
	}
	if (!wanted) {
		if (c->coarse) {
			fi_ctx_destroy(c->coarse);
			c->coarse = nullptr;
		}
		return;
	}
	fi_ctx* fine = c;
	const int D = c->g.ndim;
	// slab ranges of EVERY rank on the current level (pure arithmetic: all ranks agree on where the levels stop)
	std::vector<int> lo(c->nranks), hi(c->nranks);
	for (int r = 0; r < c->nranks; ++r) {
		lo[r] = static_cast<int>(static_cast<int64_t>(r) * c->g.gn[D - 1] / c->nranks);
		hi[r] = static_cast<int>(static_cast<int64_t>(r + 1) * c->g.gn[D - 1] / c->nranks);
	}
	int first_tail = 0;
	const int nlevels = plan_levels(c, &first_tail);
	std::vector<fi_ctx*> built;
	for (int l = 1; l <= nlevels; ++l) {
		// from first_tail on the levels are whole lattices that every rank assembles -- from ALL the data points, which
		// fi_slab_point_range asks the caller for in that case -- and solves in full (fi_ctx::replicated)
		const bool tail = l >= first_tail;
		int sizes[3] = {1, 1, 1}, cc[3] = {0, 0, 0};
		float shift[3] = {0, 0, 0};
		for (int d = 0; d < D; ++d) {
			sizes[d] = (fine->g.gn[d] + 1) / 2;
			// even extents are halved cell-centred (fi_ctx::cc); along the decomposed axis the transfers then reach two
			// planes beyond the slab, which the ghost planes of model_2 and wider stencils cover
			cc[d] = fine->g.gn[d] % 2 == 0 && (d != D - 1 || c->nranks == 1 || c->reach >= 2 || (tail && fine->replicated)) &&
			        !test_switch("FI_VERTEX_LEVELS");
			shift[d] = 0.5f * (fine->pos_shift[d] - (cc[d] ? 0.5f : 0.0f));
		}
		// coarse plane k sits on fine plane 2k: a rank keeps the coarse planes whose fine plane it owns
		for (int r = 0; r < c->nranks; ++r) {
			lo[r] = (lo[r] + 1) / 2;
			hi[r] = (hi[r] + 1) / 2;
		}
		int min_slab = sizes[D - 1];
		if (!tail) {
			for (int r = 0; r < c->nranks; ++r) { min_slab = hi[r] - lo[r] < min_slab ? hi[r] - lo[r] : min_slab; }
		}
		const int co_nranks = tail ? 1 : c->nranks;
		fi_ctx* co = fine->coarse;
		if (co && (co->g.gn[0] != sizes[0] || co->g.gn[1] != sizes[1] || co->g.gn[2] != sizes[2] || co->dtype != c->dtype ||
		           co->halo != c->halo || co->cc[0] != cc[0] || co->cc[1] != cc[1] || co->cc[2] != cc[2] || co->nranks != co_nranks)) {
			fi_ctx_destroy(co);
			co = nullptr;
		}
		if (co) {  // same shape as last time: keep its HBM, drop its rows
			for (auto* pb : co->pending) { co->pending_pool.push_back(pb); }
			co->pending.clear();
			generic_clear(co);
		} else {
			co = create_ctx(D, sizes, c->dtype, tail ? 0 : c->rank, co_nranks);
			stream_give(co->stream, true);  // (new, nothing on it)
			co->stream      = c->stream;
			co->owns_stream = false;
			co->owns_comm   = false;
			co->level       = l;
			co->finer       = fine;
			co->verify_residual = 0;
			co->replicated  = tail;
			if (!tail) {
				co->slab_fixed  = true;
				co->slab_lo     = lo[c->rank];
				co->slab_hi     = hi[c->rank];
			}
			co->halo        = c->halo;
			co->reach       = c->reach;
			for (int d = 0; d < 3; ++d) {
				co->cc[d]        = cc[d];
				co->pos_shift[d] = shift[d];
			}
			compute_geom(co, D, sizes);
			fine->coarse    = co;
		}
		co->min_slab = min_slab;
		co->comm = tail ? nullptr : c->comm;
		co->mg_mode     = c->mg_mode;  // (before the level is assembled: march_setup sizes its grids by the kernels a V-cycle runs)
		co->mg_smoother = c->mg_smoother;
		co->mg_safe     = c->mg_safe;
		co->mg_terms    = c->mg_terms;
		co->mg_pratio   = c->mg_pratio;
		co->mg_kcycle   = c->mg_kcycle;
		co->mg_cheb_degree = c->mg_cheb_degree;
		co->mg_cheb_ratio  = c->mg_cheb_ratio;
		co->value_rows_only = src->value_rows_only;  // (agreed over the ranks: fi_assemble)
		co->any_trip        = src->any_trip;
		co->stream = build_stream ? build_stream : c->stream;
		co->defer_scaling_exchange = build_stream != nullptr;  // a helper thread never talks to the neighbours
		const float vol = static_cast<float>(1 << D);
		fi_weights w = fine->w;
		w.model_0 = fine->w.model_0 * std::sqrt(vol);
		w.model_1 = fine->w.model_1 * std::sqrt(vol / 4.0f);
		w.model_2 = fine->w.model_2 * std::sqrt(vol / 16.0f);
		w.model_3 = fine->w.model_3 * std::sqrt(vol / 64.0f);
		w.model_4 = fine->w.model_4 * std::sqrt(vol / 256.0f);
		w.gradient_smoothness = fine->w.gradient_smoothness * std::sqrt(vol / 16.0f);
		co->w = w;
		built.push_back(co);
		fine = co;
	}
	if (fine->coarse) {  // deeper levels left over from an earlier, larger request
		fi_ctx_destroy(fine->coarse);
		fine->coarse = nullptr;
	}
	// The levels are problems of their own, each a chain of small launches with host round trips for its list sizes:
	// rows from the point batches, cells, lists, diagonal.  On a helper's stream (fi_assemble) the levels beyond the first
	// get a thread and a stream each (config 3's assembly 3.7 -> 2.2 ms, config 5's 24.3 -> 21.1 ms, the accurate leg of
	// config 4 16.2 -> 15.9 ms per step: there the fp64 finest level is the longest chain).  FI_SERIAL_LEVEL_CHAINS: one
	// after the other (tests: the same bits).
	auto assemble_level = [src](fi_ctx* co) {
		AllocStream alloc_on(co->stream);  // (the level's chain runs on co->stream: its buffers' slack is zeroed there)
		const int   l  = co->level;
		const float ps = 1.0f / static_cast<float>(1 << l), ns = static_cast<float>(1 << l);
		chain_mark("begins:", l);
		for (auto* b : src->batches) {
			const float* nrm = b->has_nrm ? b->nrm.as<float>() : nullptr;
			const float* pw  = b->has_pw ? b->pw.as<float>() : nullptr;
			const float* val = b->has_val ? b->val.as<float>() : nullptr;
			const bool   lin = nrm && b->gk == FI_GRADIENT_LINEAR_INTERPOLATION;
			emit_point_rows(co, b->n, b->pos.as<float>(), nrm, pw, val, b->vw, b->vk, lin ? 0.0f : b->gw * ps,
			                lin ? FI_GRADIENT_CELL_EDGES : b->gk, ps, ns);
			if (lin && b->gw != 0.0f) {
				generic_add_gradient_linear(co, b->n, b->pos.as<float>(), nrm, pw, b->gw * ps, ps, ns);
			}
		}
		chain_mark("rows emitted:", l);
		assemble(co);
		chain_mark("rows assembled:", l);
		generic_assemble(co);
		stencil_prepare(co);
		chain_mark("stencil prepared:", l);
		// the polynomial smoother's scaling (and whether the data pin a small level) with the level's assembly, on its chain's
		// stream, instead of at the head of the first solve (undivided levels: over slabs the ghost planes' diagonal comes later)
		operator_prepare(co, co->dtype == FI_F32 && co->g.ndim == 3 && co->mg_smoother == 1 && co->value_rows_only && !co->any_trip &&
		                         co->march.valid && !co->march.wide && co->nranks == 1 && !test_switch("FI_MG_FULL_SMOOTHER"));
		co->tail_prog_valid = false;
		co->dia_valid = false;
		// the data rows as 3^D diagonals: the small-level engine's view of them, and what the full operator's direct launches
		// on levels of up to 2^19 points read (fi_stencil.hip, k_full_direct3)
		chain_mark("operator prepared:", l);
		if (tail_level_supported(co) || stencil_full_direct_wanted(co)) { tail_build_operator(co); }
		chain_mark("done:", l);
		co->assembled = true;
		co->vectors_ready = co->vectors_ready && co->max_blocks >= apply_num_partials(co);
		co->stats.num_unknowns = co->g.nown;
	};
	const bool chains = build_stream != nullptr && built.size() > 1 && !test_switch("FI_SERIAL_LEVEL_CHAINS");
	if (!chains) {
		for (fi_ctx* co : built) { assemble_level(co); }
	} else {
		struct Go {
			hipEvent_t e = nullptr;
			~Go() { if (e) { (void)hipEventDestroy(e); } }
		} go_holder;
		FI_HIP_TRY(hipEventCreateWithFlags(&go_holder.e, hipEventDisableTiming));
		const hipEvent_t go = go_holder.e;
		FI_HIP_TRY(hipEventRecord(go, build_stream));  // (behind the caller's wait for the point batches)
		std::vector<Worker*> workers;  // (persistent threads: fi_workers.h)
		std::vector<int>         codes(built.size(), FI_OK);
		std::vector<std::string> msgs(built.size());
		// chain_of[i]: the level whose stream and thread level i runs on -- every level its own chain, or (FI_LEVEL_CHAINS=n)
		// n chains, the first level alone on `build_stream` and the deeper ones dealt over the other n - 1.  HIP spreads a
		// process's streams over four hardware queues and two chains on one queue take turns launch by launch (round 6's
		// trace: the 128^3 and 64^3 chains of config 4), but fewer, longer chains lose more to their host round trips than
		// they win: config 4's assembly 1.33 ms with 4 chains, 1.32 with 3, 1.55 with 2, 1.49 with all levels on one.
		int max_chains = static_cast<int>(built.size());
		if (const char* v = tuning_switch("FI_LEVEL_CHAINS")) { max_chains = std::atoi(v); }
		max_chains = max_chains < 2 ? 2 : max_chains;
		std::vector<size_t> chain_of(built.size(), 0);
		for (size_t i = 1; i < built.size(); ++i) {
			chain_of[i] = 1 + (i - 1) % static_cast<size_t>(max_chains - 1);
			fi_ctx* head = built[chain_of[i]];
			if (!head->build_stream) {
				head->build_stream = stream_take();
				FI_HIP_TRY(hipEventCreateWithFlags(&head->ev_build, hipEventDisableTiming));
			}
			if (chain_of[i] == i) { FI_HIP_TRY(hipStreamWaitEvent(head->build_stream, go, 0)); }
			built[i]->stream = head->build_stream;
		}
		auto guarded = [&](size_t i) {
			try {
				FI_HIP_TRY(hipSetDevice(c->device));
				assemble_level(built[i]);
			} catch (const Fail& f) {
				codes[i] = f.code;
				msgs[i]  = fi_last_error();
			} catch (...) {
				codes[i] = FI_ERR_HIP;
				msgs[i]  = "unexpected exception while assembling a coarser level";
			}
		};
		auto run_chain = [&](size_t head) {  // the levels of one chain, finest first
			for (size_t i = head; i < built.size(); ++i) {
				if (chain_of[i] == head) { guarded(i); }
			}
		};
		std::vector<size_t> heads;
		for (size_t i = 1; i < built.size(); ++i) {
			if (chain_of[i] == i) { heads.push_back(i); }
		}
		for (size_t head : heads) {
			Worker* w = nullptr;
			try {
				w = worker_pool().acquire();
				w->run([&run_chain, head]() { run_chain(head); });
			} catch (...) {  // no thread to be had: this chain on the caller's thread, behind the first level
				w = nullptr;
			}
			workers.push_back(w);
		}
		guarded(0);
		for (size_t k = 0; k < heads.size(); ++k) {
			Worker* w = workers[k];
			if (w) {
				w->wait();
				worker_pool().release(w);
			} else {
				run_chain(heads[k]);
			}
		}
		// the caller orders `build_stream` against the solver stream: the other chains end in it
		for (size_t head : heads) {
			(void)hipEventRecord(built[head]->ev_build, built[head]->build_stream);
			(void)hipStreamWaitEvent(build_stream, built[head]->ev_build, 0);
		}
		for (size_t i = 0; i < built.size(); ++i) {
			if (codes[i] != FI_OK) {
				for (size_t head : heads) { (void)hipStreamSynchronize(built[head]->build_stream); }
				set_error("%s", msgs[i].c_str());
				throw Fail{codes[i]};
			}
		}
	}
	for (fi_ctx* l = c->coarse; l; l = l->coarse) { l->stream = c->stream; }  // the caller orders the two streams
	// the tail of the hierarchy the small-level engine runs (fi_tail.h): from the coarsest level up while the levels qualify
	{
		std::vector<fi_ctx*> chain;
		for (fi_ctx* l = c->coarse; l; l = l->coarse) { chain.push_back(l); }
		bool ok = true;
		for (size_t k = chain.size(); k-- > 0;) {
			ok = ok && tail_level_supported(chain[k]) && static_cast<int>(chain.size() - k) <= kTailMaxLevels;
			chain[k]->tail_ok = ok;
			chain[k]->tail_prog_valid = false;
		}
	}
	// smoother bounds of the V-cycle (a global power method over all slabs) are estimated by the next multigrid solve
	for (fi_ctx* l = c; l; l = l->coarse) { l->lambda_max = 0; }
}

// fp32 replica of an FI_F64 context for the mixed-precision solve: same lattice, same slab, same weights, the
// same data points (re-emitted from the batches kept in HBM), with the levels and solver options of `c`.
// In three parts, so that fi_assemble can run the replica's finest level and the replica's coarser levels on two helper
// threads beside the fp64 finest level: twin_prepare (the context; cheap, on the caller's thread), twin_assemble (rows +
// finest level on `stream`), build_levels(c->twin, c, stream) and twin_finish.
// The replica's finest level on the LUMPED operator (fi_ctx::lumped): value rows only, a 3-D lattice the marching kernel
// covers, the V-cycle with the polynomial smoother.  FI_NO_LUMPED_TWIN: the replica assembles its own cells (tests).
bool lumped_twin_wanted(const fi_ctx* c)
{
	if (test_switch("FI_NO_LUMPED_TWIN") || test_switch("FI_MG_FULL_SMOOTHER") || test_switch("FI_NO_MARCH")) { return false; }
	const fi_weights& w = c->w;
	return c->g.ndim == 3 && c->mg_mode == 1 && c->mg_smoother == 1 && c->levels_wanted > 0 && c->value_rows_only &&
	       c->generic.ntrip == 0 && !c->any_trip && c->g.gn[0] >= 4 && !(w.model_3 > 0 || w.model_4 > 0 || w.gradient_smoothness > 0) &&
	       (w.model_1 > 0 || w.model_2 > 0);
}

// the replica's dlump and the start of its `diag` (k_model_diag adds the model diagonal) from the row sums the fp64
// level's assembly formed (non-negative for the interpolation kernels; clamped like every bound of the smoother)
__global__ __launch_bounds__(kThreads) void k_lumped_diag(int64_t n, const float* __restrict__ sums, float* __restrict__ dlump,
                                                           float* __restrict__ diag)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const float v = sums[i];
		const float f = v > 0.0f ? v : 0.0f;
		dlump[i] = f;
		diag[i]  = f;
	}
}

// The lumped replica of an ASSEMBLED fp64 context, on the context's stream.  dlump = the row sums of the data term,
// (A_data 1): every cell's record carries its share (the row sums of its block) and the assembly adds them up over the
// lattice points with A^T b and the diagonal -- no pass of its own (until round 4's last build: an apply of the fp64
// operator to a vector of ones, 0.2 ms of the 256^3 assemble's critical path).
void twin_assemble_lumped(fi_ctx* c)
{
	fi_ctx* t = c->twin;
	t->stream = c->stream;
	for (auto* pb : t->pending) { t->pending_pool.push_back(pb); }
	t->pending.clear();
	generic_clear(t);
	assemble(t);  // no rows: atb and diag zeroed, no cells
	const Geom& g = c->g;
	FI_REQUIRE(c->want_lump && c->lump.p, FI_ERR_STATE, "the lumped replica needs the row sums of the fp64 level's assembly");
	generic_assemble(t);
	stencil_prepare(t);
	if (g.nown == g.nloc && t->nranks == 1 && g.nloc > (1 << 16)) {
		// undivided: dlump, the diagonal, both Jacobi scalings and the smoother's scaling in ONE pass over the row sums (until
		// round 4's last build: two fills, the clamp, the model diagonal and the scaling -- five launches, 275 us at 256^3)
		operator_prepare(t, true, c->lump.as<float>());
		return;
	}
	t->dlump.alloc(sizeof(float) * g.nloc);
	if (g.nown != g.nloc) { FI_HIP_TRY(hipMemsetAsync(t->dlump.p, 0, sizeof(float) * g.nloc, c->stream)); }
	hipLaunchKernelGGL(k_lumped_diag, dim3(stream_blocks(g.nown)), dim3(kThreads), 0, c->stream, g.nown,
	                   c->lump.as<float>() + g.own_first, t->dlump.as<float>() + g.own_first, t->diag.as<float>() + g.own_first);
	FI_HIP_TRY(hipGetLastError());
	operator_prepare(t, t->nranks == 1);  // (the smoother's scaling with the assembly; slabs: after the ghost planes' exchange)
}

fi_ctx* twin_prepare(fi_ctx* c)
{
	if (c->level != 0) { return nullptr; }
	if (!(c->mixed && c->dtype == FI_F64)) {
		if (c->twin) {
			fi_ctx_destroy(c->twin);
			c->twin = nullptr;
		}
		return nullptr;
	}
	const int D = c->g.ndim;
	{
		long from_points = 0;
		for (auto* b : c->batches) {
			if (b->has_nrm && b->gk == FI_GRADIENT_LINEAR_INTERPOLATION && b->gw != 0.0f) { from_points += b->n * D; }
		}
		FI_REQUIRE(from_points == c->generic.nrows, FI_ERR_UNSUPPORTED,
		           "mixed precision needs rows that came from points; fi_add_rows_coo rows cannot be replicated");
	}
	int sizes[3] = {c->g.gn[0], c->g.gn[1], c->g.gn[2]};
	fi_ctx* t = c->twin;
	if (t && t->halo != c->halo) {
		fi_ctx_destroy(t);
		t = nullptr;
	}
	if (t) {
		for (auto* pb : t->pending) { t->pending_pool.push_back(pb); }
		t->pending.clear();
		generic_clear(t);
	} else {
		t = create_ctx(D, sizes, FI_F32, c->rank, c->nranks);
		stream_give(t->stream, true);  // (new, nothing on it)
		t->stream      = c->stream;
		t->owns_stream = false;
		t->owns_comm   = false;
		t->slab_fixed  = true;
		t->slab_lo     = c->slab_lo;
		t->slab_hi     = c->slab_hi;
		t->halo        = c->halo;
		t->reach       = c->reach;
		compute_geom(t, D, sizes);
		c->twin = t;
	}
	t->comm            = c->comm;
	t->stream          = c->stream;
	t->defer_scaling_exchange = false;
	t->w               = c->w;
	t->model_set       = true;
	t->verify_residual = 0;
	t->levels_wanted   = c->levels_wanted;
	t->coarse_tol      = c->coarse_tol;
	t->mg_mode         = c->mg_mode;
	t->mg_smoother     = c->mg_smoother;
	t->mg_safe         = c->mg_safe;
	t->mg_terms        = c->mg_terms;
	t->mg_pratio       = c->mg_pratio;
	t->mg_kcycle       = c->mg_kcycle;
	t->mg_cheb_degree  = c->mg_cheb_degree;
	t->mg_cheb_ratio   = c->mg_cheb_ratio;
	t->min_slab        = c->min_slab;
	t->poly_terms      = c->poly_terms;   // (the coarse-to-fine start on the replica solves its levels with them)
	t->poly_ratio      = c->poly_ratio;
	t->value_rows_only = c->value_rows_only;  // (agreed over the ranks: fi_assemble)
	t->any_trip        = c->any_trip;
	t->lumped          = lumped_twin_wanted(c);
	return t;
}

void twin_assemble(fi_ctx* c, hipStream_t build_stream)
{
	fi_ctx* t = c->twin;
	if (t->lumped) {  // (fi_assemble's helper threads never get here: the lumped form needs the assembled fp64 operator)
		FI_REQUIRE(build_stream == nullptr, FI_ERR_STATE, "the lumped replica is built behind the fp64 level");
		twin_assemble_lumped(c);
		return;
	}
	if (build_stream) {
		t->stream = build_stream;
		t->defer_scaling_exchange = true;  // a helper thread never talks to the neighbours
	}
	AllocStream alloc_on(t->stream);  // (the replica's chain runs on t->stream)
	for (auto* b : c->batches) {
		const float* nrm = b->has_nrm ? b->nrm.as<float>() : nullptr;
		const float* pw  = b->has_pw ? b->pw.as<float>() : nullptr;
		const float* val = b->has_val ? b->val.as<float>() : nullptr;
		const bool   lin = nrm && b->gk == FI_GRADIENT_LINEAR_INTERPOLATION;
		emit_point_rows(t, b->n, b->pos.as<float>(), nrm, pw, val, b->vw, b->vk, lin ? 0.0f : b->gw,
		                lin ? FI_GRADIENT_CELL_EDGES : b->gk, 1.0f, 1.0f);
		if (lin && b->gw != 0.0f) { generic_add_gradient_linear(t, b->n, b->pos.as<float>(), nrm, pw, b->gw, 1.0f, 1.0f); }
	}
	assemble(t);
	generic_assemble(t);
	stencil_prepare(t);
	operator_prepare(t);
}

void twin_finish(fi_ctx* c)
{
	fi_ctx* t = c->twin;
	t->assembled = true;
	t->vectors_ready = t->vectors_ready && t->max_blocks >= apply_num_partials(t);
	t->stats.num_unknowns = t->g.nown;
	t->stats.num_levels = 1;
	for (fi_ctx* l = t->coarse; l; l = l->coarse) { t->stats.num_levels += 1; }
}

void build_twin(fi_ctx* c)  // the three parts one after the other, on the context's stream
{
	if (!twin_prepare(c)) { return; }
	twin_assemble(c, nullptr);
	build_levels(c->twin, c);
	twin_finish(c);
}


}  // namespace fi
