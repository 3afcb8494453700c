// fi_group.hip -- the loop-back group: all slabs of a decomposition in one process on one device (tests of the slab
// geometry, halo widths and ownership rules against the undivided solve on a one-GPU machine).
#include "fi_solver_internal.h"

extern "C" {

// ---- loop-back group: all slabs of a decomposition in one process, on one device ------------------
struct fi_group {
	std::vector<fi_ctx*> members;
	int dtype = FI_F32;
};

int fi_group_create(fi_group** out, int ndim, const int* sizes, int dtype, int nranks)
{
	FI_API_BEGIN
	FI_REQUIRE(out != nullptr, FI_ERR_INVALID, "out is null");
	FI_REQUIRE(nranks >= 2, FI_ERR_INVALID, "a group needs at least two slabs");
	auto* g = new fi_group();
	g->dtype = dtype;
	try {
		for (int r = 0; r < nranks; ++r) { g->members.push_back(fi::create_ctx(ndim, sizes, dtype, r, nranks)); }
		fi_ctx* c0 = g->members[0];
		std::vector<fi::CgScalars*> ptrs;
		for (fi_ctx* c : g->members) {
			ptrs.push_back(c->scal.as<fi::CgScalars>());
			if (c != c0) {  // one stream for the whole group: program order is the synchronisation
				fi::stream_give(c->stream, true);  // (new, nothing on it)
				c->stream      = c0->stream;
				c->owns_stream = false;
			}
		}
		c0->group_scal.alloc(sizeof(fi::CgScalars*) * ptrs.size());
		FI_HIP_TRY(hipMemcpy(c0->group_scal.p, ptrs.data(), sizeof(fi::CgScalars*) * ptrs.size(), hipMemcpyHostToDevice));
	} catch (...) {
		for (fi_ctx* c : g->members) { fi_ctx_destroy(c); }
		delete g;
		throw;
	}
	*out = g;
	FI_API_END
}

int fi_group_destroy(fi_group* g)
{
	if (!g) { return FI_OK; }
	for (size_t i = g->members.size(); i-- > 0;) { fi_ctx_destroy(g->members[i]); }  // member 0 owns the stream
	delete g;
	return FI_OK;
}

int fi_group_size(const fi_group* g) { return g ? static_cast<int>(g->members.size()) : 0; }

fi_ctx* fi_group_rank(fi_group* g, int rank)
{
	if (!g || rank < 0 || rank >= static_cast<int>(g->members.size())) { return nullptr; }
	return g->members[rank];
}

int fi_group_assemble(fi_group* g)
{
	FI_API_BEGIN
	FI_REQUIRE(g != nullptr, FI_ERR_INVALID, "null group");
	{  // the data facts a real decomposition agrees on by an all-reduce (fi_ctx::any_trip)
		bool trip = false, grad = false;
		for (fi_ctx* c : g->members) {
			trip = trip || c->generic.ntrip != 0;
			grad = grad || !fi::holds_value_rows_only(c);
		}
		for (fi_ctx* c : g->members) {
			c->facts_forced = true;
			c->forced_trip  = trip;
			c->forced_grad  = grad;
		}
	}
	for (fi_ctx* c : g->members) {
		const int rc = fi_assemble(c);
		if (rc != FI_OK) { return rc; }
	}
	// coarser levels (and the fp32 replicas of mixed precision with theirs): the loop-back dot-product sum needs
	// the scalar blocks of every member of a level
	// The diagonal's ghost planes, like a process per slab gets them through its transport at assembly time: the scaling
	// on the ghost planes is then the neighbour's, the polynomial's first step forms its operand on load and the deep
	// exchange has the scaling of the whole ghost zone.
	auto ghosts = [&](std::vector<fi_ctx*>& lev) {
		if (lev[0]->nranks <= 1 || lev[0]->g.nown == lev[0]->g.nloc) { return; }
		fi::RankSet R(lev.begin(), lev.end());
		fi::halo_exchange(R, &fi_ctx::diag, lev[0]->min_slab >= lev[0]->halo ? lev[0]->halo : lev[0]->reach);
		for (fi_ctx* c : lev) { fi::operator_rescale_with_ghosts(c); }
	};
	ghosts(g->members);
	auto link_chain = [&](std::vector<fi_ctx*> lev) {
		while (lev[0]) {
			std::vector<fi::CgScalars*> ptrs;
			for (fi_ctx*& c : lev) {
				FI_REQUIRE(c != nullptr, FI_ERR_STATE, "members disagree on the number of levels");
				ptrs.push_back(c->scal.as<fi::CgScalars>());
			}
			ghosts(lev);
			lev[0]->group_scal.alloc(sizeof(fi::CgScalars*) * ptrs.size());
			FI_HIP_TRY(hipMemcpy(lev[0]->group_scal.p, ptrs.data(), sizeof(fi::CgScalars*) * ptrs.size(), hipMemcpyHostToDevice));
			for (fi_ctx*& c : lev) { c = c->coarse; }
		}
	};
	std::vector<fi_ctx*> lev, twins;
	for (fi_ctx* c : g->members) {
		lev.push_back(c->coarse);
		twins.push_back(c->twin);
	}
	link_chain(lev);
	link_chain(twins);
	FI_API_END
}

static void group_ready(fi_group* g)
{
	FI_REQUIRE(g != nullptr, FI_ERR_INVALID, "null group");
	for (fi_ctx* c : g->members) {
		fi::check_assembled(c);
		fi::ensure_vectors(c);
	}
	FI_HIP_TRY(hipSetDevice(g->members[0]->device));
}

int fi_group_solve_cg(fi_group* g, const float* guess, int max_iterations, float tol, float* out, int* iterations,
                      float* rel_residual)
{
	FI_API_BEGIN
	group_ready(g);
	int64_t at = 0;
	fi_ctx* c0 = g->members[0];
	c0->stats.coarse_iterations = 0;
	if (!guess && c0->twin && c0->twin->coarse) {
		fi::twin_cascade_guess(g->members);
	} else if (!guess && c0->coarse) {
		g->dtype == FI_F64 ? fi::cascade_guess<double>(g->members) : fi::cascade_guess<float>(g->members);
	} else {
		for (fi_ctx* c : g->members) {
			if (g->dtype == FI_F64) {
				fi::load_owned<double>(c, c->x, guess ? guess + at : nullptr, FI_HOST);
			} else {
				fi::load_owned<float>(c, c->x, guess ? guess + at : nullptr, FI_HOST);
			}
			at += c->g.nown;
		}
	}
	struct Report {
		fi_ctx* c; int* it; float* rel;
		~Report() { if (it) { *it = c->stats.iterations; } if (rel) { *rel = static_cast<float>(c->stats.rel_residual); } }
	} report{c0, iterations, rel_residual};
	if (c0->mg_mode == 1 && (c0->coarse || (c0->twin && c0->twin->coarse))) {
		g->dtype == FI_F64 ? fi::cg_run_mg<double>(g->members, max_iterations, tol)
		                   : fi::cg_run_mg<float>(g->members, max_iterations, tol);
	} else if (fi::poly_ok(c0)) {
		g->dtype == FI_F64 ? fi::cg_run_poly_or_jacobi<double>(g->members, max_iterations, tol)
		                   : fi::cg_run_poly_or_jacobi<float>(g->members, max_iterations, tol);
	} else {
		g->dtype == FI_F64 ? fi::cg_run<double>(g->members, max_iterations, tol) : fi::cg_run<float>(g->members, max_iterations, tol);
	}
	at = 0;
	for (fi_ctx* c : g->members) {
		if (g->dtype == FI_F64) {
			fi::store_owned<double>(c, c->x, out ? out + at : nullptr, FI_HOST);
		} else {
			fi::store_owned<float>(c, c->x, out ? out + at : nullptr, FI_HOST);
		}
		at += c->g.nown;
	}
	FI_API_END
}

int fi_group_apply_AtA_f64(fi_group* g, const double* x, double* y)
{
	FI_API_BEGIN
	group_ready(g);
	FI_REQUIRE(x && y, FI_ERR_INVALID, "null vector");
	g->dtype == FI_F64 ? fi::apply_f64_run<double>(g->members, x, y) : fi::apply_f64_run<float>(g->members, x, y);
	FI_API_END
}

int fi_group_true_residual(fi_group* g, double* rel)
{
	FI_API_BEGIN
	group_ready(g);
	FI_REQUIRE(rel != nullptr, FI_ERR_INVALID, "null output");
	*rel = g->dtype == FI_F64 ? fi::true_residual_run<double>(g->members) : fi::true_residual_run<float>(g->members);
	FI_API_END
}

int fi_group_tile_pass(fi_group* g, const float* guess, int tile_size, float* out)
{
	FI_API_BEGIN
	group_ready(g);
	FI_REQUIRE(tile_size >= 2 && guess && out, FI_ERR_INVALID, "fi_group_tile_pass: tile_size >= 2, guess and out required");
	FI_REQUIRE(g->members[0]->generic.ntrip == 0, FI_ERR_UNSUPPORTED, "the tile pre-solver over triplet rows needs an undivided lattice");
	int64_t at = 0;
	for (fi_ctx* c : g->members) {
		g->dtype == FI_F64 ? fi::load_owned<double>(c, c->x, guess + at, FI_HOST) : fi::load_owned<float>(c, c->x, guess + at, FI_HOST);
		at += c->g.nown;
	}
	g->dtype == FI_F64 ? fi::tile_pass_run<double>(g->members, tile_size) : fi::tile_pass_run<float>(g->members, tile_size);
	at = 0;
	for (fi_ctx* c : g->members) {
		g->dtype == FI_F64 ? fi::store_owned<double>(c, c->x, out + at, FI_HOST) : fi::store_owned<float>(c, c->x, out + at, FI_HOST);
		at += c->g.nown;
	}
	FI_API_END
}

int fi_group_error_map(fi_group* g, const float* solution, float* out)
{
	FI_API_BEGIN
	group_ready(g);
	FI_REQUIRE(solution && out, FI_ERR_INVALID, "fi_group_error_map needs a solution and an output buffer");
	int64_t at = 0;
	for (fi_ctx* c : g->members) {
		g->dtype == FI_F64 ? fi::load_owned<double>(c, c->x, solution + at, FI_HOST)
		                   : fi::load_owned<float>(c, c->x, solution + at, FI_HOST);
		at += c->g.nown;
	}
	fi::halo_exchange(g->members, &fi_ctx::x);
	at = 0;
	for (fi_ctx* c : g->members) {
		fi::error_map(c, c->x.p, c->q.p);
		g->dtype == FI_F64 ? fi::store_owned<double>(c, c->q, out + at, FI_HOST) : fi::store_owned<float>(c, c->q, out + at, FI_HOST);
		at += c->g.nown;
	}
	FI_API_END
}

int fi_group_get_solution_f64(fi_group* g, double* out)
{
	FI_API_BEGIN
	group_ready(g);
	int64_t at = 0;
	for (fi_ctx* c : g->members) {
		g->dtype == FI_F64 ? fi::get_vec_f64_t<double>(c, c->x, out + at) : fi::get_vec_f64_t<float>(c, c->x, out + at);
		at += c->g.nown;
	}
	FI_API_END
}

}  // extern "C"
