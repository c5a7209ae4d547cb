// symx::NewtonsMethod on libmistark: implementation of shim/include/mistark_symx/NewtonsMethod.h over the C ABI (include/mistark.h).
// Replaces symx/src/solver/NewtonsMethod.cpp (and with it second_order/*, the JIT and BlockedSparseMatrix) in a STARK build.
//
// What crosses the boundary and when:
//   first solve()            every DoF set (GlobalPotential::get_dof_maps: host pointer, size, label), every bound array
//                            (MappedWorkspace::maps: DataMap id / data / n_elements / stride), every potential (name, connectivity,
//                            bindings in mws.make_* order); potentials without a hand-written kernel as SymX op sequences
//   every solve()            sizes and pointers re-read (the caller owns and may resize everything), host arrays uploaded, settings
//                            translated, mistark_newton_solve, DoFs written back to the caller's arrays
//   every Newton callback    the DoFs are brought to the caller's arrays first (STARK's callbacks read them there: contact detection,
//                            validity checks), and after before_energy_evaluation what the callback may have refilled goes back to the
//                            device — and only that (see "What is re-sent" below)
// What is re-sent. The reference reads every table and array through its lambdas at every evaluation; sending everything at every evaluation
// (round 2: mistark_upload(-1) + update_connectivity of all 35 contact tables per callback, tens of MB over PCIe and a layout rebuild per
// line-search trial) is what made the drop-in slow. Now every connectivity table and every bound array carries a 64-bit fingerprint of its
// bytes:
//   solve()                  every table and array is fingerprinted (in parallel when OpenMP is on); what changed since the engine last saw it
//                            is sent — an in-place edit at an unchanged address and size (re-targeted attachments, a swapped prescribed-point
//                            list: ADVICE r02) is caught here
//   before_energy_evaluation tables of contact_* / friction_* potentials and every array up to SMALL_ARRAY doubles are fingerprinted in full and
//                            sent when changed (the callback's own products: contact tables, friction data, stiffness scalars). Larger tables
//                            and arrays get a SAMPLED fingerprint at every evaluation (64 bytes of every 4 KiB + both ends: a callback that
//                            recomputes such an array — re-targeted prescribed positions, a rewritten rest shape — is caught at 1/64 of the
//                            cost of reading it), and are re-sent in full when the sample, the address or the size changed
//   end of solve()           every large table and array is fingerprinted in full once more: bytes that changed INSIDE the Newton loop without
//                            the sample noticing (a sparse in-place edit) mean the engine solved with stale data — reported once on stderr
//                            with the name of the remedy, and sent, so that the next solve starts from the caller's data
//   MISTARK_SHIM_STRICT=1    everything fingerprinted in full at every evaluation (the reference's semantics at the reference's cost);
//   MISTARK_SHIM_FAST=1      the round-3 behaviour: large tables and arrays re-checked by address and size only inside the Newton loop
// The DoF vector itself never travels host -> device inside a solve: the engine owns it there.
// MISTARK_SHIM_DRY=1: registration only on a registration-only context (no GPU; solve() returns Successful without touching the DoFs);
// MISTARK_SHIM_DESCRIBE=<file>: the registration (mistark_describe) is written there after every solve.
#include "mistark_symx/NewtonsMethod.h"

#include <compile/Sequence.h>
#include <fmt/format.h>

#include <algorithm>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "mistark.h"

namespace symx
{
	struct NewtonsMethod::Impl
	{
		mistark_ctx* ctx = nullptr;
		bool dry = false;
		NewtonsMethod* self = nullptr;
		struct Arr
		{
			int id = -1;
			const double* host = nullptr;
			int64_t n = -1;
			int stride = 0;
			bool is_dof = false;
			uint64_t print = 0;     // fingerprint of the bytes the engine holds
			uint64_t sprint = 0;    // sampled fingerprint of the same bytes (large arrays)
			bool have_print = false;
		};
		std::map<std::pair<std::uintptr_t, int>, Arr> arrays;  // (DataMap id, stride) -> engine array
		struct Pot
		{
			int id = -1;
			const int32_t* conn = nullptr;
			int32_t n_elem = -1;
			int stride = 0;
			bool dynamic = false;   // contact_* / friction_*: refilled inside the Newton loop
			uint64_t print = 0, sprint = 0;
			std::string name;
		};
		static constexpr int64_t SMALL_ARRAY = 32768;  // doubles: arrays up to this size are fingerprinted at every evaluation
		bool strict = false;        // MISTARK_SHIM_STRICT=1
		bool fast = false;          // MISTARK_SHIM_FAST=1
		bool resolve_stale = true;  // a solve that ran on stale inputs is redone in strict mode (MISTARK_SHIM_NO_RESOLVE=1: warn only)
		int64_t n_stale_solves = 0; // solves whose end-of-solve check found inputs the engine had not seen
		int64_t n_sample_hits = 0, n_stale = 0;  // large tables / arrays re-sent because their sample changed inside the Newton loop; missed edits found at the end of a solve
		int64_t n_uploads = 0, n_table_updates = 0, bytes_sent = 0;  // statistics (MISTARK_SHIM_STATS=1 prints them at destruction)
		double t_tables = 0.0, t_hash = 0.0, t_upload = 0.0;  // inside sync(): table updates in the engine, fingerprints, array uploads
		double t_callbacks = 0.0, t_sync = 0.0, t_dofs = 0.0, t_solve = 0.0;  // seconds: the caller's callbacks, sync(), DoF transfers, all of solve()
		static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
		template <class F>
		auto timed_cb(F&& f)
		{
			const double t0 = now();
			dofs_to_host();
			const double t1 = now();
			t_dofs += t1 - t0;
			struct Acc
			{
				double& t;
				double t0;
				~Acc() { t += now() - t0; }
			} acc{t_callbacks, t1};
			return f();
		}

		// 64-bit fingerprint of a byte range: four multiply-xorshift lanes over 32-byte blocks, folded; chunks of 1 MiB are hashed
		// independently (OpenMP when the build has it) and combined in order
		// (the reference's build recipes compile this file at low optimisation levels: the hash is forced inline and optimised by itself)
		__attribute__((always_inline)) static inline uint64_t mix(uint64_t h, uint64_t v)
		{
			h ^= v;
			h *= 0x9E3779B97F4A7C15ull;
			return h ^ (h >> 29);
		}
		__attribute__((optimize("O3"))) static uint64_t fingerprint_chunk(const unsigned char* p, size_t n)
		{
			uint64_t h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull, h2 = 0xA4093822299F31D0ull, h3 = 0x082EFA98EC4E6C89ull;
			size_t i = 0;
			for (; i + 32 <= n; i += 32) {
				uint64_t w[4];
				std::memcpy(w, p + i, 32);
				h0 = mix(h0, w[0]);
				h1 = mix(h1, w[1]);
				h2 = mix(h2, w[2]);
				h3 = mix(h3, w[3]);
			}
			uint64_t tail = 0;
			for (; i < n; i++) tail = (tail << 8) | p[i];
			return mix(mix(mix(mix(h0, h1), h2), h3), tail ^ (uint64_t)n);
		}
		static uint64_t fingerprint(const void* data, size_t bytes)
		{
			const unsigned char* p = static_cast<const unsigned char*>(data);
			constexpr size_t CH = 1u << 20;
			const long n_ch = (long)((bytes + CH - 1) / CH);
			if (n_ch <= 1) return fingerprint_chunk(p, bytes);
			std::vector<uint64_t> part((size_t)n_ch);
			const int n_thr = (int)std::min<long>(n_ch, 16);
#pragma omp parallel for schedule(static) num_threads(n_thr)
			for (long c = 0; c < n_ch; c++) part[(size_t)c] = fingerprint_chunk(p + (size_t)c * CH, std::min(CH, bytes - (size_t)c * CH));
			uint64_t h = 0x452821E638D01377ull;
			for (uint64_t v : part) h = mix(h, v);
			return h;
		}
		// sampled fingerprint of a large byte range: 64 bytes of every 4 KiB and the last 64 bytes, with the length
		static constexpr size_t SAMPLE_MIN_BYTES = (size_t)SMALL_ARRAY * 8;
		__attribute__((optimize("O3"))) static uint64_t fingerprint_sampled(const void* data, size_t bytes)
		{
			const unsigned char* p = static_cast<const unsigned char*>(data);
			uint64_t h = 0x3F84D5B5B5470917ull ^ (uint64_t)bytes;
			auto take = [&](size_t at) {
				uint64_t w[8];
				std::memcpy(w, p + at, 64);
				for (int i = 0; i < 8; i++) h = mix(h, w[i]);
			};
			if (bytes < 128) return fingerprint_chunk(p, bytes);
			for (size_t at = 0; at + 64 <= bytes; at += 4096) take(at);
			take(bytes - 64);
			return h;
		}
		std::vector<Pot> pots;
		struct SumSlot
		{
			int id = -1;
			std::vector<double> row0;
		};
		std::map<size_t, SumSlot> sum_slots;  // potential index -> the engine array standing for its summation symbols
		std::vector<std::pair<const double*, int64_t>> dof_sets;

		void check(int rc, const char* what) const
		{
			if (rc < 0) throw std::runtime_error(std::string("mistark shim: ") + what + ": " + (ctx ? mistark_last_error(ctx) : "no context"));
		}
		~Impl()
		{
			if (std::getenv("MISTARK_SHIM_STATS"))
				std::cerr << "mistark shim: " << n_uploads << " array uploads, " << n_table_updates << " table updates (" << n_sample_hits << " found by the sampled check inside the Newton loop, " << n_stale
				          << " in-loop edits it missed), " << bytes_sent / 1e6 << " MB sent to the engine; of " << t_solve
				          << " s in solve(): " << t_callbacks << " s in the caller's callbacks, " << t_sync << " s in sync() (" << t_tables << " table updates, " << t_hash << " array fingerprints, " << t_upload << " array uploads), " << t_dofs
				          << " s bringing DoFs to the caller" << std::endl;
			if (ctx && std::getenv("MISTARK_SHIM_STATS")) {
				int64_t ok = 0, bad = 0;
				if (mistark_get_counter(ctx, "host_ranges_pinned", &ok) == 0 && mistark_get_counter(ctx, "host_ranges_not_pinned", &bad) == 0)
					std::cerr << "mistark shim: " << ok << " host arrays page-locked in place, " << bad << " left pageable" << std::endl;
			}
			if (ctx && std::getenv("MISTARK_SHIM_STATS") && std::getenv("MISTARK_VERIFY_DOF_SKIP")) {
				int64_t n = 0;
				if (mistark_get_counter(ctx, "dof_skips_verified", &n) == 0) std::cerr << "mistark shim: " << n << " skipped DoF transfers verified against a real transfer" << std::endl;
			}
			if (ctx) mistark_destroy(ctx);
		}

		void create()
		{
			const char* dry_env = std::getenv("MISTARK_SHIM_DRY");
			dry = dry_env && dry_env[0] == '1';
			const char* strict_env = std::getenv("MISTARK_SHIM_STRICT");
			if (const char* nr = std::getenv("MISTARK_SHIM_NO_RESOLVE")) resolve_stale = !(nr[0] == '1');
			strict = strict_env && strict_env[0] == '1';
			const char* fast_env = std::getenv("MISTARK_SHIM_FAST");
			fast = fast_env && fast_env[0] == '1' && !strict;
			if (dry) {
				check(mistark_create_dry(&ctx), "mistark_create_dry");
			} else {
				const char* dev = std::getenv("MISTARK_DEVICE");
				const int rc = mistark_create(dev ? std::atoi(dev) : 0, &ctx);
				if (rc != 0) throw std::runtime_error("mistark shim: mistark_create failed (" + std::to_string(rc) + "): no MI355X visible; the hot path has no CPU fallback");
				// MISTARK_SHIM_PIN=1: the caller's DoF and state arrays are page-locked where they are (engine option pin_host_arrays: checked
				// hipHostRegister, released on rebind and with the context). Measured at 1 M tets (round 6, profiles/r06_dropin_pin{0,1}.txt): no gain —
				// 4.1 MB of DoFs reach the caller in 0.18 ms either way (23 GB/s: the link, not the staging copy, is the limit) — so it stays off.
				const char* pin = std::getenv("MISTARK_SHIM_PIN");
				if (pin && pin[0] == '1') check(mistark_set_option(ctx, "pin_host_arrays", 1), "pin_host_arrays");
			}
		}

		// DoF sets, arrays and potentials as the caller holds them right now; what changed goes to the engine. full: fingerprint everything
		// (solve()); otherwise (inside the Newton loop) only what a before_energy_evaluation callback produces (see the file header).
		void sync(GlobalPotential& gp, bool full)
		{
			full = full || strict;
			if (!ctx) create();
			// ---- DoF sets (GlobalPotential::add_dof, GlobalPotential.h:55-61)
			const auto& dof_maps = gp.get_dof_maps();
			for (int i = 0; i < gp.get_n_dof_sets(); i++) {
				double* host = dof_maps[(size_t)i].data();
				const int64_t n = gp.get_n_dofs(i);
				if (i >= (int)dof_sets.size()) {
					check(mistark_add_dof_set(ctx, gp.get_dof_label(i).c_str(), host, n), "mistark_add_dof_set");
					dof_sets.emplace_back(host, n);
				} else if (dof_sets[(size_t)i].first != host || dof_sets[(size_t)i].second != n) {
					check(mistark_resize_dof_set(ctx, i, host, n), "mistark_resize_dof_set");
					dof_sets[(size_t)i] = {host, n};
				}
			}
			// ---- potentials
			const auto& potentials = gp.get_potentials();
			for (size_t pi = 0; pi < potentials.size(); pi++) {
				const Potential& pot = *potentials[pi];
				auto mws = pot.get_mws();
				// A summation loop (MappedWorkspace::add_for_each, MappedWorkspace.h:123-130: SymX's fem integrators): its symbols sit among the
				// workspace's symbols where the vector was made; they get a binding of their own (a shim-owned slot holding the first row) and
				// the engine's interpreter runs the rows (mistark_potential_custom_set_summation). The data is fixed at definition time in the
				// reference too (MappedWorkspace.h:549-550 copies it).
				const bool has_sum = mws->has_summation();
				int sum_first_input = -1, n_inputs_so_far = 0;
				std::vector<mistark_binding> bs;
				auto bind_summation = [&]() {
					auto& slot = sum_slots[pi];
					if (slot.id < 0) {
						slot.row0.assign(mws->summation.data.begin(), mws->summation.data.begin() + mws->summation.stride);
						slot.id = mistark_array(ctx, slot.row0.data(), 1, mws->summation.stride);
						check(slot.id, "mistark_array (summation slot)");
					}
					sum_first_input = n_inputs_so_far;
					bs.push_back(mistark_binding{slot.id, mws->summation.stride, -1});
					n_inputs_so_far += mws->summation.stride;
				};
				for (const auto& m : mws->maps) {
					if (has_sum && sum_first_input < 0 && m.first_symbol_idx > mws->summation.first_symbol_idx) bind_summation();
					n_inputs_so_far += m.stride;
					const auto key = std::make_pair(m.id(), (int)m.stride);
					Arr& a = arrays[key];
					const double* host = m.data();
					const int64_t n = m.connectivity_index < 0 ? 1 : (int64_t)m.n_elements();
					int dof_set = -1;  // a DoF map is recognised by the identity of its container (DataMap::id), as SymX does
					for (int i = 0; i < gp.get_n_dof_sets(); i++)
						if (dof_maps[(size_t)i].id() == key.first) dof_set = i;
					if (a.id < 0) {
						a.id = dof_set >= 0 ? mistark_dof_array(ctx, dof_set, m.stride) : mistark_array(ctx, host, n, m.stride);
						check(a.id, "mistark_array");
						a.have_print = false;
					} else if (dof_set < 0 && (a.host != host || a.n != n)) {
						check(mistark_array_rebind(ctx, a.id, host, n), "mistark_array_rebind");
						a.have_print = false;
					}
					a.host = host;
					a.n = n;
					a.stride = (int)m.stride;
					a.is_dof = dof_set >= 0;
					bs.push_back(mistark_binding{a.id, m.stride, m.connectivity_index});
				}
				if (has_sum && sum_first_input < 0) bind_summation();
				const int32_t n_elem = mws->conn.n_elements();
				const int32_t* conn = n_elem > 0 ? mws->conn.data() : nullptr;
				const std::string& name = pot.get_name();
				if (pi >= pots.size()) {
					Pot P;
					bool known = false;
					for (int k = 0; k < mistark_n_supported_potentials() && !known; k++) known = name == mistark_supported_potential(k);
					if (known && has_sum) throw std::runtime_error("mistark shim: potential '" + name + "' carries a summation loop, which the engine's kernel of that name does not expect");
					if (known) {
						P.id = mistark_potential(ctx, name.c_str(), conn, n_elem, mws->conn.stride, bs.data(), (int)bs.size());
					} else {
						// no hand-written kernel under this name (a user-defined energy): SymX's own straight-line op sequence, interpreted on
						// the device (symx/src/compile/Sequence.h:24-41)
						auto flatten = [](const Scalar& expr, std::vector<int32_t>& rows, std::vector<double>& consts) {
							Sequence seq({expr});
							for (const auto& op : seq.ops) {
								rows.insert(rows.end(), {(int32_t)op.type, op.dst, op.a, op.b, op.cond});
								consts.push_back(op.constant);
							}
							return seq.get_n_inputs();
						};
						std::vector<int32_t> ops, cops;
						std::vector<double> cst, ccst;
						const int n_in = flatten(pot.get_expression(), ops, cst);
						if (pot.has_conditional()) flatten(pot.get_condition(), cops, ccst);
						P.id = mistark_potential_custom(ctx, name.c_str(), conn, n_elem, mws->conn.stride, bs.data(), (int)bs.size(), ops.data(), cst.data(), (int)cst.size(), n_in,
						                                cops.empty() ? nullptr : cops.data(), ccst.empty() ? nullptr : ccst.data(), (int)ccst.size());
					}
					check(P.id, ("potential '" + name + "'").c_str());
					if (has_sum)
						check(mistark_potential_custom_set_summation(ctx, P.id, sum_first_input, mws->summation.stride, mws->summation.n_iterations, mws->summation.data.data()),
						      ("summation of potential '" + name + "'").c_str());
					// tables refilled inside the Newton loop (EnergyFrictionalContact.cpp:117-119) go to the small dynamic matrix part
					if (name.rfind("contact_", 0) == 0 || name.rfind("friction_", 0) == 0) check(mistark_potential_set_dynamic(ctx, P.id, 1), "mistark_potential_set_dynamic");
					P.conn = conn;
					P.n_elem = n_elem;
					P.stride = mws->conn.stride;
					P.dynamic = name.rfind("contact_", 0) == 0 || name.rfind("friction_", 0) == 0;
					P.print = n_elem > 0 ? fingerprint(conn, (size_t)n_elem * (size_t)P.stride * sizeof(int32_t)) : 0;
					P.sprint = n_elem > 0 ? fingerprint_sampled(conn, (size_t)n_elem * (size_t)P.stride * sizeof(int32_t)) : 0;
					P.name = name;
					pots.push_back(P);
				} else {
					// The reference refills tables in place: the same address and size may hold other rows. Tables the Newton loop refills are
					// fingerprinted at every evaluation, the others at every solve(); only a table whose bytes changed is sent (each update
					// makes the engine re-validate its indices and rebuild its incidence lists).
					Pot& P = pots[pi];
					const size_t tbytes = (size_t)std::max(n_elem, 0) * (size_t)P.stride * sizeof(int32_t);
					bool changed = P.conn != conn || P.n_elem != n_elem;
					if (!changed && n_elem > 0 && (full || P.dynamic || tbytes <= SAMPLE_MIN_BYTES)) {
						const uint64_t h = fingerprint(conn, tbytes);
						changed = h != P.print;
						P.print = h;
						if (changed) P.sprint = fingerprint_sampled(conn, tbytes);
					} else if (!changed && n_elem > 0 && !fast) {
						// a large static table inside the Newton loop: the sample decides whether it is read in full
						const uint64_t hs = fingerprint_sampled(conn, tbytes);
						if (hs != P.sprint) {
							P.sprint = hs;
							P.print = fingerprint(conn, tbytes);
							changed = true;
							n_sample_hits++;
						}
					} else if (changed) {
						P.print = n_elem > 0 ? fingerprint(conn, tbytes) : 0;
						P.sprint = n_elem > 0 ? fingerprint_sampled(conn, tbytes) : 0;
					}
					if (changed) {
						const double tt = now();
						check(mistark_potential_update_connectivity(ctx, P.id, conn, n_elem), "mistark_potential_update_connectivity");
						t_tables += now() - tt;
						n_table_updates++;
						bytes_sent += (int64_t)n_elem * P.stride * 4;
						P.conn = conn;
						P.n_elem = n_elem;
					}
				}
			}
			if (dry) return;
			// ---- arrays: what changed since the engine last saw it (DoF views never travel this way: mistark_dofs_from_host_arrays at solve())
			for (auto& kv : arrays) {
				Arr& a = kv.second;
				if (a.id < 0 || a.is_dof || !a.host || a.n <= 0) continue;
				const int64_t doubles = a.n * (int64_t)a.stride;
				const double th = now();
				if (a.have_print && !full && doubles > SMALL_ARRAY) {
					// a large array inside the Newton loop: the sample decides whether it is read (and sent) in full
					if (fast) continue;
					const uint64_t hs = fingerprint_sampled(a.host, (size_t)doubles * sizeof(double));
					t_hash += now() - th;
					if (hs == a.sprint) continue;
					n_sample_hits++;
				}
				const uint64_t h = fingerprint(a.host, (size_t)doubles * sizeof(double));
				if (doubles > SMALL_ARRAY) a.sprint = fingerprint_sampled(a.host, (size_t)doubles * sizeof(double));
				t_hash += now() - th;
				if (a.have_print && h == a.print) continue;
				const double tu = now();
				check(mistark_upload(ctx, a.id), "mistark_upload");
				t_upload += now() - tu;
				if (const char* v = std::getenv("MISTARK_SHIM_STATS"); v && v[0] == '2') std::cerr << "  upload " << doubles * 8 << " B " << (now() - tu) * 1e3 << " ms" << std::endl;
				a.print = h;
				a.have_print = true;
				n_uploads++;
				bytes_sent += doubles * 8;
			}
		}

		// End of a solve: what the sampled checks inside the Newton loop may have missed. A large table or array whose bytes differ from what
		// the engine holds was edited in place during the loop without its sample changing: the engine evaluated stale data from that point on.
		// Said once, with the remedy; the data is sent so that the next solve does not start stale as well.
		int verify_after_solve(GlobalPotential& gp)
		{
			if (strict || fast || dry) return 0;
			const int64_t stale_before = n_stale;
			const double t0 = now();
			std::string first;
			const auto& potentials = gp.get_potentials();
			for (size_t pi = 0; pi < pots.size() && pi < potentials.size(); pi++) {
				Pot& P = pots[pi];
				const size_t tbytes = (size_t)std::max(P.n_elem, 0) * (size_t)P.stride * sizeof(int32_t);
				if (P.dynamic || tbytes <= SAMPLE_MIN_BYTES || !P.conn) continue;
				auto mws = potentials[pi]->get_mws();
				if (mws->conn.n_elements() != P.n_elem || mws->conn.data() != P.conn) continue;  // (resized since: the next solve's full pass sends it)
				if (fingerprint(P.conn, tbytes) != P.print) {
					n_stale++;
					if (first.empty()) first = "connectivity of potential '" + P.name + "'";
					P.n_elem = -1;  // (forces the update at the next sync)
				}
			}
			for (auto& kv : arrays) {
				Arr& a = kv.second;
				const int64_t doubles = a.n * (int64_t)a.stride;
				if (a.id < 0 || a.is_dof || !a.host || doubles <= SMALL_ARRAY || !a.have_print) continue;
				if (fingerprint(a.host, (size_t)doubles * sizeof(double)) != a.print) {
					n_stale++;
					if (first.empty()) first = "an array of " + std::to_string(doubles) + " doubles";
					a.have_print = false;
				}
			}
			t_hash += now() - t0;
			const int missed = (int)(n_stale - stale_before);
			if (missed > 0) {  // (every occurrence is said: a stale solve is never silent)
				n_stale_solves++;
				std::cerr << "mistark shim: WARNING: " << first << (missed > 1 ? " (and " + std::to_string(missed - 1) + " more)" : std::string())
				          << " was modified in place inside the Newton loop without its sampled fingerprint changing; the engine evaluated the previous contents for the "
				          << "rest of that solve (" << n_stale_solves << " such solve(s) so far). "
				          << (resolve_stale ? "The solve is REDONE from its initial DoFs with every table and array read in full at every evaluation, and this solver stays "
				                              "in that mode (MISTARK_SHIM_STRICT=1 selects it from the start; MISTARK_SHIM_NO_RESOLVE=1 keeps the stale result and only warns)."
				                            : "MISTARK_SHIM_NO_RESOLVE=1: the stale result is kept. Set MISTARK_SHIM_STRICT=1 for callbacks that edit a few entries of a large array in place.")
				          << std::endl;
			}
			return missed;
		}

		// ---- C callbacks of mistark_newton_solve -> SolverCallbacks (solver_utils.h:29-117). STARK's callbacks read the DoFs from the
		//      caller's arrays: bring them there first.
		// (inside the Newton loop the callbacks read the DoFs and never write them: several callbacks at one iterate share one transfer)
		void dofs_to_host() { check(mistark_dofs_to_host_arrays_if_changed(ctx), "mistark_dofs_to_host_arrays_if_changed"); }
		static Impl& I(void* u) { return *static_cast<Impl*>(u); }
		static void cb_before_eval(void* u)
		{
			Impl& s = I(u);
			s.timed_cb([&] { s.self->callbacks->run_before_energy_evaluation(); return 0; });
			const double t0 = now();
			s.sync(*s.self->global_potential, /*full=*/false);  // tables and data the callback refilled
			s.t_sync += now() - t0;
		}
		static int cb_initial_valid(void* u) { return I(u).timed_cb([&] { return I(u).self->callbacks->run_is_initial_state_valid() ? 1 : 0; }); }
		static int cb_intermediate_valid(void* u) { return I(u).timed_cb([&] { return I(u).self->callbacks->run_is_intermediate_state_valid() ? 1 : 0; }); }
		static void cb_on_invalid(void* u) { I(u).timed_cb([&] { I(u).self->callbacks->run_on_intermediate_state_invalid(); return 0; }); }
		static void cb_on_armijo(void* u) { I(u).timed_cb([&] { I(u).self->callbacks->run_on_armijo_fail(); return 0; }); }
		static int cb_is_converged(void* u) { return I(u).timed_cb([&] { return I(u).self->callbacks->run_is_converged() ? 1 : 0; }); }
		static int cb_converged_valid(void* u) { return I(u).timed_cb([&] { return I(u).self->callbacks->run_is_converged_state_valid() ? 1 : 0; }); }
		static double cb_max_step(void* u) { return I(u).timed_cb([&] { return I(u).self->callbacks->run_max_allowed_step(); }); }
	};

	NewtonsMethod::NewtonsMethod(spGlobalPotential global_potential, spContext context, spSolverCallbacks callbacks)
		: callbacks(callbacks), impl(std::make_unique<Impl>()), global_potential(global_potential), context(context)
	{
		if (!this->callbacks) this->callbacks = SolverCallbacks::create(context);
		impl->self = this;
	}
	NewtonsMethod::~NewtonsMethod() = default;
	std::shared_ptr<NewtonsMethod> NewtonsMethod::create(spGlobalPotential global_potential, spContext context, spSolverCallbacks callbacks)
	{
		return std::make_shared<NewtonsMethod>(global_potential, context, callbacks);
	}
	mistark_ctx* NewtonsMethod::engine() const { return impl->ctx; }
	long NewtonsMethod::mistark_stale_solves() const { return (long)impl->n_stale_solves; }

	SolverReturn NewtonsMethod::solve()
	{
		auto _t = this->context->logger->time("newton_solve");
		Impl& s = *impl;
		const double t_begin = Impl::now();
		struct SolveTimer
		{
			Impl& s;
			double t0;
			~SolveTimer() { s.t_solve += Impl::now() - t0; }
		} solve_timer{s, t_begin};
		s.sync(*global_potential, /*full=*/true);
		s.t_sync += Impl::now() - t_begin;
		if (const char* path = std::getenv("MISTARK_SHIM_DESCRIBE")) {
			const int64_t n = mistark_describe(s.ctx, nullptr, 0);
			std::string buf((size_t)n, '\0');
			mistark_describe(s.ctx, buf.data(), n);
			std::ofstream(path) << buf.c_str() << std::endl;
		}
		this->stats = SolveStats();
		if (s.dry) return SolverReturn::Successful;
		s.check(mistark_dofs_from_host_arrays(s.ctx), "mistark_dofs_from_host_arrays");

		mistark_newton_settings ns;
		mistark_newton_default_settings(&ns);
		const NewtonSettings& S = this->settings;
		ns.max_iterations = S.max_iterations;
		ns.min_iterations = S.min_iterations;
		ns.residual_tolerance_abs = S.residual_tolerance_abs;
		ns.residual_tolerance_rel = S.residual_tolerance_rel;
		ns.step_tolerance = S.step_tolerance;
		ns.max_iterations_as_success = S.max_iterations_as_success ? 1 : 0;
		ns.step_cap = S.step_cap;
		ns.enable_armijo_backtracking = S.enable_armijo_backtracking ? 1 : 0;
		ns.line_search_armijo_beta = S.line_search_armijo_beta;
		ns.max_backtracking_armijo_iterations = S.max_backtracking_armijo_iterations;
		ns.max_backtracking_invalid_state_iterations = S.max_backtracking_invalid_state_iterations;
		switch (S.projection_mode) {
			case ProjectionToPD::Newton: ns.projection_mode = MISTARK_PROJ_NEWTON; break;
			case ProjectionToPD::ProjectedNewton: ns.projection_mode = MISTARK_PROJ_PROJECTED_NEWTON; break;
			case ProjectionToPD::ProjectOnDemand: ns.projection_mode = MISTARK_PROJ_ON_DEMAND; break;
			case ProjectionToPD::Progressive: ns.projection_mode = MISTARK_PROJ_PROGRESSIVE; break;
		}
		ns.projection_eps = S.projection_eps;
		ns.project_to_pd_use_mirroring = S.project_to_pd_use_mirroring ? 1 : 0;
		ns.project_on_demand_countdown = S.project_on_demand_countdown;
		ns.ppn_tightening_factor = S.ppn_tightening_factor;
		ns.ppn_release_factor = S.ppn_release_factor;
		ns.linear_solver = S.linear_solver == LinearSolver::DirectLLT ? MISTARK_SOLVER_DIRECT_LLT : MISTARK_SOLVER_BDPCG;
		ns.cg_max_iterations = S.cg_max_iterations;
		ns.cg_abs_tolerance = S.cg_abs_tolerance;
		ns.cg_rel_tolerance = S.cg_rel_tolerance;
		ns.cg_stop_on_indefiniteness = S.cg_stop_on_indefiniteness ? 1 : 0;
		ns.bailout_residual = S.bailout_residual;

		mistark_newton_callbacks cb{};
		cb.user = &s;
		cb.before_energy_evaluation = Impl::cb_before_eval;
		cb.is_initial_state_valid = Impl::cb_initial_valid;
		cb.is_intermediate_state_valid = Impl::cb_intermediate_valid;
		cb.on_intermediate_state_invalid = Impl::cb_on_invalid;
		cb.on_armijo_fail = Impl::cb_on_armijo;
		cb.is_converged = Impl::cb_is_converged;
		cb.is_converged_state_valid = Impl::cb_converged_valid;
		cb.max_allowed_step = Impl::cb_max_step;

		// (the DoFs the solve starts from, for the one case it has to be redone: inputs edited in place that the sampled checks missed)
		std::vector<double> initial_dofs;
		if (!s.strict && !s.fast && s.resolve_stale) {
			initial_dofs.resize((size_t)global_potential->get_total_n_dofs());
			global_potential->get_dofs(initial_dofs.data());
		}
		mistark_newton_stats st{};
		int rc = 0;
		for (int attempt = 0;; attempt++) {
			st = mistark_newton_stats{};
			rc = mistark_newton_solve(s.ctx, &ns, &cb, &st);
			s.check(rc, "mistark_newton_solve");
			s.check(mistark_dofs_to_host_arrays(s.ctx), "mistark_dofs_to_host_arrays");  // the reference leaves the solution in the caller's DoF arrays (NewtonsMethod.cpp:608-640)
			const int missed = s.verify_after_solve(*global_potential);
			if (missed == 0 || !s.resolve_stale || initial_dofs.empty() || attempt > 0) break;
			// the result above came from stale inputs: never hand it out. From here on this solver reads everything in full at every
			// evaluation (the reference's semantics), and the solve starts again where it started (as the reference itself restarts a solve
			// from `initial_dofs`, NewtonsMethod.cpp:616-619).
			s.strict = true;
			global_potential->set_dofs(initial_dofs.data());
			s.sync(*global_potential, /*full=*/true);
			s.check(mistark_dofs_from_host_arrays(s.ctx), "mistark_dofs_from_host_arrays");
		}
		this->context->logger->set("mistark_stale_solves", (int)s.n_stale_solves);
		if (const char* path = std::getenv("MISTARK_SHIM_SOLVELOG"))  // one line per solve(): Newton iterations, linear solves, CG iterations
			std::ofstream(path, std::ios::app) << st.newton_iterations << " " << st.n_linear_solves << " " << st.cg_iterations << std::endl;
		this->stats.newton_iterations = st.newton_iterations;
		this->stats.cg_iterations = st.cg_iterations;
		this->stats.ls_cap_iterations = st.ls_cap_iterations;
		this->stats.ls_max_iterations = st.ls_max_iterations;
		this->stats.ls_inv_iterations = st.ls_inv_iterations;
		this->stats.ls_bt_iterations = st.ls_bt_iterations;
		this->stats.n_hessians = (uint64_t)st.n_hessians;
		this->stats.n_projected_hessians = (uint64_t)st.n_projected_hessians;
		this->stats.projected_hessians_ratio = st.projected_hessians_ratio;
		// The series the reference logs: one entry per Newton iteration inside the loop (NewtonsMethod.cpp:198-207: Hessian counts and the CG
		// iterations of the iteration's LAST solve; :488-594: the four line-search series), one per solve after it (:249). STARK's console
		// line and YAML output read them.
		auto& lg = *this->context->logger;
		int32_t n_rec = 0;
		s.check(mistark_newton_iteration_log(s.ctx, nullptr, 0, &n_rec), "mistark_newton_iteration_log");
		std::vector<mistark_newton_iteration> rec((size_t)n_rec);
		if (n_rec > 0) s.check(mistark_newton_iteration_log(s.ctx, rec.data(), n_rec, &n_rec), "mistark_newton_iteration_log");
		for (const mistark_newton_iteration& r : rec) {
			if (r.logged) {
				lg.add_and_append("n_hessians", (double)r.n_hessians);
				lg.add_and_append("n_projected_hessians", (double)r.n_projected_hessians);
				lg.add_and_append("projected_hessians_ratio", r.n_hessians > 0 ? (double)r.n_projected_hessians / (double)r.n_hessians : 0.0);
				lg.add_and_append("cg_iterations", r.cg_iterations_last);
			}
			if (r.line_search) {
				lg.add_and_append("ls_cap", r.ls_cap);
				lg.add_and_append("ls_max", r.ls_max);
				lg.add_and_append("ls_inv", r.ls_inv);
				lg.add_and_append("ls_bt", r.ls_bt);
			}
		}
		lg.add_and_append("newton_iterations", st.newton_iterations);
		// stage times of the engine under the reference's timer names (NewtonsMethod.cpp:268,274,390; SecondOrderCompiledGlobal)
		lg.add("mistark_linear_system_solve_s", st.t_linear_solve);
		lg.add("mistark_assembly_s", st.t_assembly);
		lg.add("mistark_project_to_PD_s", st.t_project);
		lg.add("mistark_evaluate_P_grad_hess_s", st.t_eval_pgh);
		lg.add("mistark_evaluate_P_s", st.t_eval_p);
		lg.add("mistark_linear_solves", st.n_linear_solves);
		// SolverReturn and the MISTARK_* result codes share their numbering (solver_utils.h:15-26, mistark.h:158-166)
		return static_cast<SolverReturn>(rc);
	}

	// The summary table of the reference (NewtonsMethod.cpp:643-718): the "Solve" block from the logged series, then the runtime block from
	// the logger's timers (callbacks, newton_solve) followed by the engine's own stage times.
	void NewtonsMethod::print_summary(double total_time) const
	{
		auto* out = this->context->output.get();
		const auto& lg = *this->context->logger;
		const int total_n_newton = (int)lg.get_stats("newton_iterations").total;
		if (total_n_newton == 0) {
			out->print_with_new_line("No Newton iterations were performed. No summary to show.\n");
			return;
		}
		out->print_with_new_line("");
		out->print_with_new_line(fmt::format("  {:<24} {:>10} {:>8} {:>8} {:>8}", "Solve", "Total", "Avg", "Min", "Max"));
		out->print_with_new_line(fmt::format("  {}", std::string(62, '-')));
		const std::vector<std::pair<std::string, std::string>> rows = {{"Newton iterations", "newton_iterations"}, {"CG iterations", "cg_iterations"}, {"Line search cap", "ls_cap"},
		                                                                {"Line search max", "ls_max"},          {"Line search inv", "ls_inv"},      {"Line search bt", "ls_bt"}};
		for (const auto& [label, key] : rows) {
			auto st = lg.get_stats(key);
			out->print_with_new_line(fmt::format("  {:<24} {:>10} {:>8.1f} {:>8} {:>8}", label, (long long)st.total, st.avg, (int)st.min, (int)st.max));
		}
		auto sr = lg.get_stats("projected_hessians_ratio");
		out->print_with_new_line(fmt::format("  {:<24} {:>10} {:>8.1f}% {:>7.1f}% {:>7.1f}%", "Projected hessians", "", 100.0 * sr.avg, 100.0 * sr.min, 100.0 * sr.max));
		if (total_time <= 0.0) {
			total_time = 0.0;
			for (const auto& label : lg.get_timer_labels()) total_time += lg.get_timer_total(label);
		}
		out->print_with_new_line(fmt::format("  {}", std::string(62, '-')));
		out->print_with_new_line("");
		out->print_with_new_line(fmt::format("  {:<40} {:>10}  {:>6}", "Runtime", "Time (s)", "%"));
		out->print_with_new_line(fmt::format("  {}", std::string(60, '-')));
		struct Entry { std::string label; double time; };
		std::vector<Entry> entries;
		for (const auto& label : lg.get_timer_labels()) entries.push_back({label, lg.get_timer_total(label)});
		for (const char* k : {"linear_system_solve", "assembly", "project_to_PD", "evaluate_P_grad_hess", "evaluate_P"})
			entries.push_back({std::string("  mistark: ") + k, lg.get_double(std::string("mistark_") + k + "_s")});
		std::sort(entries.begin(), entries.end(), [](const Entry& a, const Entry& b) { return a.time > b.time; });
		for (const auto& e : entries) {
			if (total_time > 0 && e.time / total_time < 0.001) continue;
			out->print_with_new_line(fmt::format("  {:<40} {:>10.6f}  {:>5.1f}%", e.label, e.time, total_time > 0 ? 100.0 * e.time / total_time : 0.0));
		}
		out->print_with_new_line(fmt::format("  {}", std::string(60, '-')));
		out->print_with_new_line(fmt::format("  {:<40} {:>10.6f}  {:>5.1f}%", "Total", total_time, 100.0));
		out->print_new_line();
	}
}
