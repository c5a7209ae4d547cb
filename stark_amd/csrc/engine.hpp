// engine.hpp — internal declarations of the MI355X engine (not part of the C ABI; see include/mistark.h).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <map>
#include <string>
#include <vector>

#include <memory>

#include "../../include/mistark.h"

#include "potargs.hpp"

namespace mistark {

struct Error : std::runtime_error
{
    using std::runtime_error::runtime_error;
};

#define MS_CHECK(expr)                                                                                                   \
    do {                                                                                                                 \
        hipError_t _e = (expr);                                                                                          \
        if (_e != hipSuccess) {                                                                                          \
            throw ::mistark::Error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
        }                                                                                                                \
    } while (0)

// Registration-only ("dry") contexts (mistark_create_dry): no GPU is touched, device buffers stay empty. Used to check on a machine without
// a GPU what a caller registers (tests/test_shim_cpu.py: the reference's own classes through the SymX shim against the host mirror).
// The flag is a property of the CONTEXT (Context::dry); every C-ABI entry point publishes its context's flag to the calling thread for the
// duration of the call (DryScope), which is where DevBuf::ensure — that knows no context — reads it. A dry context therefore never changes
// what a real context of the same process allocates.
inline bool& dry_mode()
{
    static thread_local bool dry = false;
    return dry;
}
struct DryScope
{
    bool prev;
    explicit DryScope(bool dry) : prev(dry_mode()) { dry_mode() = dry; }
    ~DryScope() { dry_mode() = prev; }
    DryScope(const DryScope&) = delete;
    DryScope& operator=(const DryScope&) = delete;
};
// Growable device buffer
template <class T>
struct DevBuf
{
    T* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap)
    {
        o.p = nullptr;
        o.cap = 0;
    }
    DevBuf& operator=(DevBuf&& o) noexcept
    {
        if (this != &o) {
            if (p) (void)hipFree(p);
            p = o.p;
            cap = o.cap;
            o.p = nullptr;
            o.cap = 0;
        }
        return *this;
    }
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    void ensure(size_t n)
    {
        if (n <= cap) return;
        if (dry_mode()) return;
        if (p) MS_CHECK(hipFree(p));
        p = nullptr;
        // (growth: an eighth, but small buffers double — a reallocation is a hipFree, i.e. a device-wide synchronisation, and the buffers sized by the
        // contact sets grow by a few rows at a time inside the Newton loop)
        size_t want = n + std::max(n / 8, std::min<size_t>(n, (size_t)1 << 20)) + 64;
        MS_CHECK(hipMalloc((void**)&p, want * sizeof(T)));
        // MISTARK_ALLOC_TRACE=1: every (re)allocation on stderr — what still allocates inside a timed region shows up between the caller's markers
        static const bool trace = std::getenv("MISTARK_ALLOC_TRACE") != nullptr;
        if (trace) std::fprintf(stderr, "[alloc] %zu bytes (was %zu)\n", want * sizeof(T), cap * sizeof(T));
        cap = want;
        // MISTARK_POISON=1: fill fresh allocations with a NaN pattern, so that a read of memory nobody wrote shows up in the tests
        // instead of depending on what the allocator hands out (fresh processes get zero pages, long-lived ones do not)
        static const bool poison = std::getenv("MISTARK_POISON") != nullptr;
        if (poison) {
            MS_CHECK(hipMemset(p, 0xFF, want * sizeof(T)));
            MS_CHECK(hipDeviceSynchronize());  // (the memset runs on the null stream, the engine's stream does not wait for that one)
        }
    }
};

constexpr int64_t HOT_SET_ROWS = 1024;  // DoF sets up to this many block rows take the hot path

struct DofSet
{
    std::string label;
    double* host = nullptr;
    int64_t n = 0;       // scalars
    int64_t offset = 0;  // first scalar in the flat DoF vector
};

struct Array
{
    const double* host = nullptr;
    int64_t n_items = 0;
    int stride = 0;
    int dof_set = -1;         // >= 0: view into the DoF vector
    DevBuf<double> own;       // storage when not a view
    double* dev = nullptr;    // resolved at prepare()
    bool need_upload = true;
};

struct CustomProgram;  // custom.hip: an energy given as SymX's op sequence, interpreted on the device
constexpr int KIND_CUSTOM = -2;
struct Potential
{
    std::string name;
    int kind = -1;              // index into the registry of compiled kernels, or KIND_CUSTOM
    std::shared_ptr<CustomProgram> prog;
    int NB = 0;
    int n_elem = 0;
    int conn_stride = 0;
    std::vector<int32_t> conn_host;
    uint64_t conn_version = 0;  // bumped whenever the host connectivity is uploaded again
    DevBuf<int32_t> conn;
    std::vector<mistark_binding> bindings;
    PotArgs args{};
    size_t e_off = 0;   // first element in the element-energy pool
    size_t h_off = 0;   // first double in the element-Hessian pool
    size_t k_off = 0;   // first 3x3 block in the element-Hessian pool (= h_off / 9)
    size_t kp_off = 0;  // first key in the key list of its matrix part
    // "lazy" potentials (closed-form tets): inside the Newton loop their element Hessians are written as FLOAT blocks, upper block triangle
    // only (what the float BSR assembly needs: 360 instead of 1152 bytes per tet); the double blocks of the few elements the PSD
    // projection selects are recomputed on demand (project()). Staged API calls keep the full double pool.
    // Gradient without atomics (closed-form tets: 12 M double atomics were 2/3 of the kernel's time at 1M tets): the kernel writes the four
    // node gradients of an element to a pool and k_grad_gather sums, per block row, the contributions listed in the incidence lists built
    // on the host from the connectivity (fixed order: the gradient of these potentials is bit-reproducible).
    bool grad_gather = false;
    DevBuf<uint32_t> inc_start, inc;  // per block row (+1): first incidence; per incidence: k * n_gpool + pool position
    DevBuf<uint32_t> inc_long;        // block rows with more than GRAD_LONG_ROW incidences
    int n_inc_long = 0;
    DevBuf<double> gpool;
    std::vector<int64_t> inc_sig;     // what the incidence lists were built for
    // sharded runs: the elements this rank evaluates, [elements whose energy counts here | interface elements of other ranks]
    DevBuf<uint32_t> elem_list;
    int n_list = 0, n_eown_list = 0;  // length of the list and of its leading part (elements whose energy counts here)
    int n_eown = 0;                   // elements the energy-only kernels run over (n_elem, or n_eown_list)
    int n_key = 0;                    // elements in the key space / pools of this context (n_elem, or n_list when sharded with a list)
    bool dyn_pool = false;       // its node gradients go to the context's pool of the device-resident tables (Context::dyn_gpool), summed by k_dyn_grad_gather
    bool lazy_capable = false;
    bool ti_projection = false;  // translation-invariant energy: PSD projection on the reduced matrix (k_project_eig_ti)
    size_t hf_off = 0;  // first float in the float pool
    int n_pool_f = 0;   // elements per block pair in the float pool (n_elem rounded up to 64: 16-byte aligned wave stores)
    int part = 0;       // 0: fixed connectivity, 1: connectivity changes inside the Newton loop (contacts)
    bool conn_dirty = true;
    const int32_t* conn_ext = nullptr;  // connectivity written on the device by the contact detector (overrides conn)
};

// One part of the split system matrix in tiled block-CSR form (see kernels.hip, "SpMV").
struct BsrPart
{
    bool dirty = true;              // connectivity changed -> rebuild this part's pattern
    bool have_matrix = false;
    size_t n_keys = 0;
    DevBuf<uint64_t> keys, keys_alt;
    DevBuf<uint32_t> kidx, kidx_alt;
    DevBuf<uint32_t> scan, slot_start;
    DevBuf<uint32_t> slot_of_src;   // per key of this part (potential P: kp_off + (a*NB+b)*n_elem + e) -> BSR slot
    const uint32_t* sorted_src = nullptr;  // key positions in sorted key order (one of kidx / kidx_alt); NO_SRC for structural diagonal keys
    DevBuf<uint32_t> sorted_desc;   // per sorted key: where the gather assembly reads the contribution (make_descriptors)
    DevBuf<uint32_t> sym;           // static part, lazy float pool: per CSR slot SYM_NONE (summed from its own list), SYM_SKIP (written by its transpose's lane) or
                                    // the storage position of the transpose the slot's lane also writes (k_sym_classify / k_assemble_gather)
    bool sym_valid = false;         // ... built for the current descriptors
    int desc_lazy = -1;             // lazy state the descriptors were made for (-1: none)
    int64_t nnzb = 0, ntiles = 0, n_rows = 0;
    DevBuf<uint32_t> colw;          // bit31 = last block of its row, bits 0..30 = block column
    DevBuf<uint32_t> slot_row;      // block row of each slot
    DevBuf<int32_t> tile_first_row; // compact row of the first block of each tile (bit31: continues the previous tile's row)
    DevBuf<int32_t> rowmap;         // compact row -> block row
    DevBuf<int64_t> row_ptr;        // compact rows
    DevBuf<float> vals;             // tiles of 64 blocks: float4 q0[64], float4 q1[64], float s[64]
    DevBuf<uint8_t> slot_dirty;     // per block: a projection round replaced one of its contributions (re-gathered by project(), then cleared)
    DevBuf<uint32_t> long_slots;    // blocks with > LONG_SLOT contributions (summed by k_assemble_long)
    int n_long = 0;
    DevBuf<uint32_t> vlong_slots;   // blocks with > VERY_LONG_SLOT contributions (k_assemble_vlong_part / _fold)
    int n_vlong = 0;
    // contact part only: rows cut into chunks for k_spmv_chunks
    int64_t n_chunks = 0;
    DevBuf<uint32_t> row_chunk0;    // per compact row (+1): first chunk
    DevBuf<int32_t> chunk_row;      // per chunk: compact row
    DevBuf<double> chunk_partial;   // 3 per chunk (long rows only)
    DevBuf<double> yd;              // 3 per compact row: A_dyn x of single-chunk rows
    DevBuf<int32_t> crow_of_row;    // block row -> compact row of this part, -1 if absent
    // static part only: CSR order cut into row-aligned chunks (kernels.hip: build_aligned). `vals` / `scol` hold ntiles tiles in that
    // storage order and store_slot maps a CSR slot to its position there.
    int64_t n_chunks_static = 0;
    int chunk_tiles = 8;            // tiles per chunk of this pattern (chunk_tiles_for)
    DevBuf<uint32_t> store_slot;    // per CSR slot: position (tile * 64 + lane) of the block in vals / scol
    DevBuf<uint32_t> scol;          // per stored position: block column, bit 31 = last block of its row (padding: 0)
    DevBuf<uint64_t> row_pos;       // per block row: position of its first block
    DevBuf<uint32_t> long_rows;     // rows longer than a chunk (stored after the chunks, one wavefront each)
    int n_long_rows = 0;
};

// Sharded problem (world > 1): block rows are partitioned over the ranks (owner map); a rank evaluates every element that
// touches one of its rows (elements on an interface are evaluated by both sides: no gradient / Hessian traffic at all), assembles and
// solves its rows only, and exchanges boundary values of vectors. See shard.hip.
struct Shard
{
    std::vector<int32_t> user_owner;   // explicit partition (mistark_dist_set_row_owner), empty: graph partition
    std::vector<double> coords;        // a position per block row (mistark_dist_set_row_coords): recursive coordinate bisection
    std::vector<int32_t> shared_rows;  // rows any rank may reference from potentials whose connectivity changes (contacts): ghosts everywhere
    std::vector<int32_t> owner;        // per global block row
    std::vector<int64_t> n_own_of, n_send_of;  // per rank
    int64_t n_own = 0, n_ghost = 0, n_loc = 0;
    std::vector<int32_t> grow_h;       // local index -> global block row (owned rows in ascending global order, then ghosts by owner)
    DevBuf<int32_t> lrow, grow;
    // boundary exchange: every rank contributes the values of the rows some other rank holds as ghosts ("send rows", padded to send_stride),
    // the all-gather delivers all of them, a rank picks its ghosts
    int64_t n_send = 0, send_stride = 0;
    DevBuf<int32_t> send_rows;         // local indices of my send rows
    DevBuf<int32_t> ghost_src;         // per ghost: owner rank * send_stride + position in the owner's send rows
    DevBuf<int32_t> send_pos_of_row;   // per own row: its position among my send rows, -1: no other rank holds it as a ghost
    DevBuf<uint32_t> send_mask;        // per send row: the ranks that hold it as a ghost (bit r)
    DevBuf<double> sendbuf, recvbuf;
    // gather of the owned parts of a vector into the global vector on every rank
    int64_t own_stride = 0;            // max n_own over the ranks
    DevBuf<int32_t> grow_all;          // [world][own_stride]: global block row of rank r's local row i (-1: padding)
    DevBuf<double> gath_s, gath_r;
    DevBuf<double> scal_s, scal_r;     // small all-gathers (scalars)
    std::vector<int64_t> sig;          // what the partition and lists were computed for
    int64_t version = 0;               // bumped by mistark_dist_set_row_owner / _add_shared_rows
    int64_t version_lists = 0;         // bumped when the element lists change
    DevBuf<int32_t> err;               // device error flag (a dynamic potential referenced a row that is neither owned nor a ghost)
};

struct PcgCtrl
{
    int done;
    int converged;
    int indef;
    int n_iter;
    double error;
    double bb;
    double rz[2];
    int epoch;  // (pinned copy only) which solve published this: a look-ahead batch of the previous solve may still be on its way
    int pad_;
    double alpha[2];  // sharded PCG with one exchange per iteration (pcg_sharded_fused): the step lengths of the last two iterations
};

struct Context
{
    int device = 0;
    bool dry = false;               // registration only (mistark_create_dry)
    hipStream_t stream = nullptr;
    std::string last_error;

    std::vector<DofSet> dof_sets;
    std::vector<Array> arrays;
    std::vector<Potential> pots;
    bool layout_dirty = true;   // DoF sizes / arrays / potentials changed -> prepare()

    int64_t ndofs = 0, nbr = 0;
    DevBuf<double> u, grad, du, r, z, p, q, tmp_a, tmp_b;
    DevBuf<double> p2;              // second buffer of the search direction (fused direction update: pcg())
    bool no_fuse_dir = true;        // option "fuse_dir" = 1: the direction update formed inside the SpMV (k_spmv_dir; measured slower, kept as a variant)
    size_t n_elem_total = 0, hess_total = 0;
    DevBuf<double> elemE, elemH;
    DevBuf<float> elemHf;           // float pool of the lazy potentials
    size_t hf_total = 0;
    bool lazy_allowed = true;       // option "lazy_hessians": newton_solve may take the lazy path (progressive / no projection)
    bool lazy_eval = false;         // option "lazy_eval": staged mistark_eval calls take it too (tests)
    bool lazy_active = false;       // state of the current element Hessians
    bool no_grad_gather = false;    // option "no_grad_gather": the closed-form tets accumulate their gradient with atomics (cross-check)
    hipStream_t side_stream = nullptr;  // the contact part's pattern build overlaps the element evaluation (eval())
    hipEvent_t side_ev[2] = {nullptr, nullptr};
    bool no_pattern_overlap = false;  // option "no_pattern_overlap"
    hipStream_t aux_stream = nullptr;   // the small potentials of an evaluation run beside the large ones (eval())
    hipEvent_t aux_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool no_eval_overlap = false;     // option "no_eval_overlap"
    bool contact_speculation = false; // option "contact_speculation": the intersection check runs the proximity search of the next evaluation ahead
    bool no_bounded_pattern = false;  // measurement / cross-check: the contact part's pattern with a read-back per stage, like the static part's
    int kernel_dbg = 0;             // option "kernel_dbg": measurement switches inside kernels (PotArgs::dbg)
    DevBuf<uint8_t> is_projected, active_blocks;
    bool have_hessians = false;

    // sparsity pattern / matrix: A = part[0] (fixed connectivity + all diagonal blocks) + part[1] (contacts)
    BsrPart part[2];
    DevBuf<int32_t> diag_slot[2];   // per block row: slot of the diagonal block in each part, -1 if absent
    int spmv_variant = 0;          // micro-benchmark ablation variant
    int spmv_chunk_tiles = 0;      // tiles per SpMV chunk, 0 = by matrix size (chunk_tiles_for); tests force 1..8
    DevBuf<unsigned char> sel_desc;  // descriptor table of k_project_select_multi
    std::vector<unsigned char> sel_desc_host;
    DevBuf<double> vlong_part;
    DevBuf<double> grad_hot;
    DevBuf<int32_t> hot_rows;      // global block row of every hot row
    int n_hot = 0;
    int proj_variant = 0;          // PSD projection, bits: 1 = matrix in LDS (k_project_eig) instead of registers, 2 = no batching of short lists, 4 = IEEE div/sqrt
    int pcg_batch = 0;             // tuning: PCG iterations per launch batch (one batch is always queued ahead of the one the host waits for); 0 = by size
    int spmv_grid_cap = 0;         // tuning: max workgroups of the SpMV kernel (0 = default)
    int hf_layout = 0;             // float pool of the lazy tets: 0 = pair-major Hf[pair][element][9]; 1 = element-major Hf[element][pair][9] (round 5: measured, 2 % slower overall — the tet kernel's strided stores cost more than the gather gains — kept as an option and cross-check)
    int custom_rtc = 1;            // user-defined potentials: kernels emitted from the op sequence and compiled by hipRTC (0: the device interpreter only)
    int custom_timing = 0;         // measurement: HIP events around every launch of a user-defined potential (synchronises), counter "custom_kernel_us"
    double custom_kernel_us = 0.0;
    int64_t n_rtc_builds = 0, n_rtc_launches = 0;
    double t_rtc_builds = 0.0;
    int spmv_nt = -1;              // non-temporal loads of the matrix values in the SpMV: -1 = when the matrix is beyond the Infinity Cache, 0 = never, 1 = always
    bool atomic_assembly = false;  // debug switch: scatter with float atomics instead of the deterministic gather
    bool force_generic = false;    // debug switch: evaluate every potential through the generic hyper-dual path
    bool pcg_holdback = false;     // pcg(): no look-ahead batch while the batch in flight is expected to converge (measured: 1.150 against 1.140 ms per solve, off)
    bool sweep_axis_by_extent = false;  // contact search: sweep / band axes by the extent of the vertices' bounding box (through round 5) instead of their variance
    bool generic_inertia = false;  // ... EnergyLumpedInertia only (its closed form: k_eval_lumped_inertia)
    bool generic_contact = false;  // ... the contact / friction potentials only (their closed forms: contact_closed.hpp)
    int contact_closed_min_lanes = -1;  // closed forms for tables with at least this many (element, DoF pair) lanes; -1: by potential (launch_eval)
    DevBuf<uint8_t> cub_tmp;
    DevBuf<float> dinv;             // 9 floats per block row
    DevBuf<double> dense;           // DirectLLT: the matrix as a dense column-major array (small systems only)
    // DirectLLT beyond that: block-tridiagonal Cholesky of the RCM-ordered matrix (direct.hip)
    DevBuf<double> llt_D, llt_S, llt_y;
    DevBuf<int32_t> llt_perm;
    DevBuf<int> llt_info;
    int llt_mb = 0;
    void* llt_mf = nullptr;  // direct.hip: Multifrontal (nested-dissection fronts of the current pattern)
    uint64_t llt_mf_pattern_version = 0;
    double llt_mf_gb = 0.0;
    bool llt_no_coords = false;  // multifrontal ordering by breadth-first level sets even when positions are known (cross-check)
    int llt_multifrontal = 0;  // 0: when the band costs more than 2 GB, 1: always beyond the dense limit, -1: never
    uint64_t pattern_version = 1, llt_pattern_version = 0;  // bumped by every pattern build
    bool have_matrix = false;
    bool matrix_current = false;    // the assembled matrix reflects the current element Hessians
    bool static_assembled = false;  // eval() has gathered the static part already, on the auxiliary stream (aux_ev[2] marks its end)
    bool no_split_gather = false;   // option "no_split_gather": whole-part assemblies through k_assemble_gather (one lane per block whatever the list length) instead of k_assemble_gather_split
    bool no_sym_gather = false;     // option "no_sym_gather": every block of the static part summed from its own list (k_assemble_gather without the mirror table)
    bool no_eager_assembly = false; // option "no_eager_assembly"
    int pcg_epoch = 0;              // solves started (PcgCtrl::epoch)
    std::vector<int32_t> hot_rows_host;  // what hot_rows holds (prepare uploads it only when it changes)
    DevBuf<uint32_t> proj_list;     // element ids selected for projection (per potential, at e_off)

    // reductions / PCG
    DevBuf<double> partials;        // 4 x MAX_PARTIALS
    DevBuf<PcgCtrl> ctrl;
    DevBuf<int64_t> counters;
    double* h_scratch = nullptr;    // pinned host scratch
    void* h_pin = nullptr;          // pinned staging area of fetch()
    void* h_stage[2] = {nullptr, nullptr};  // pinned staging areas of h2d_staged() (uploads of the caller's pageable arrays)
    // Option "pin_host_arrays" (a drop-in's shim turns it on): the caller's large arrays — DoF sets and bound arrays, whose addresses the engine keeps
    // anyway — are page-locked in place (hipHostRegister, checked) so that transfers to and from them are direct DMA instead of copies through a
    // staging buffer (4 MB of DoFs to the caller before every callback: 0.2 ms pageable). Registered once per (address, size); released when the
    // array is rebound / resized and at destruction; a range that cannot be registered stays pageable (ok = false, not retried).
    struct PinnedRange
    {
        size_t bytes = 0;
        bool ok = false;
    };
    std::map<const void*, PinnedRange> pinned;
    bool pin_host_arrays = false;
    int64_t n_pin_ok = 0, n_pin_failed = 0;
    void* h_small[4] = {nullptr, nullptr, nullptr, nullptr};  // pinned slots of small uploads of host temporaries (kernels.hip: h2d_small)
    hipEvent_t h_small_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned h_small_next = 0;
    hipEvent_t h_stage_ev[2] = {nullptr, nullptr};
    size_t h_pin_bytes = 0;
    uint8_t* pub = nullptr;         // coherent pinned page small read-backs are published into (kernels.hip: publish)
    uint32_t pub_seq = 0;
    double t_eval_issue = 0.0, t_eval_wait = 0.0;  // host seconds of eval(P+g+H): issuing launches / waiting for the read-backs (counters eval_pgh_issue_us, eval_pgh_wait_us)
    // MISTARK_EVAL_EVENTS=1 (measurement): GPU time stamps inside a P+g+H evaluation, microseconds from the start of the tets' kernel, summed over
    // the evaluations (counters evt_tet_us / evt_small_us / evt_gather_us / evt_main_us / evt_pattern_us / evt_n); read at the next evaluation's entry
    hipEvent_t evt[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool evt_armed[6] = {false, false, false, false, false, false};
    double evt_sum[6] = {0, 0, 0, 0, 0, 0};
    int64_t evt_n = 0;
    uint8_t* pub2 = nullptr;        // ... and a second one for a read-back left in flight while another runs (publish_begin2 / publish_end2)
    uint32_t pub2_seq = 0;
    // bumped by everything that can change what a contact detection sees (DoFs, bound arrays, layout): the detector skips a search whose
    // inputs are those of its previous one (the evaluation that opens a Newton iteration repeats the accepted line-search state)
    // a projection round as it travels between its phases (kernels.hip: project_phase_a / _b / _c)
    struct ProjRound
    {
        struct Mark  // what k_proj_mark needs of a potential's list once the eigen-decompositions have run
        {
            int pot;
            const uint32_t* list;
            int nl;
            const double* Hc;
            int n_pool_c;
        };
        std::vector<Mark> marks;
        bool mark_part[2] = {false, false};
    };
    // ... and one started ahead of the solve that may need it (project_speculate)
    struct ProjSpec
    {
        bool active = false;
        int stage = 0;  // 1: phase A queued, counts on their way; 2: phase B queued
        double threshold = 0.0, eps = 0.0;
        int mirroring = 0;
        hipStream_t stream = nullptr;
        hipEvent_t ev_in = nullptr, ev_done = nullptr;
        int64_t* pinned = nullptr;  // 128 counters + flag word, coherent host memory
        uint32_t seq = 0;
        int64_t h[128];
        ProjRound round;
        bool pending = false;  // a request pcg() takes up once its first batches are queued
        double p_eps = 0.0, p_threshold = 0.0;
        int p_mirroring = 0;
    } spec;
    bool proj_speculation = false;  // option "proj_speculation": the next projection round's selection and eigen kernels run beside the solve (measured: no gain)
    int64_t n_proj_speculated = 0, n_proj_adopted = 0;
    uint64_t data_version = 1;
    int64_t n_dof_skips_verified = 0;  // MISTARK_VERIFY_DOF_SKIP=1: skipped transfers checked against a real one
    uint64_t u_version = 1, u_host_version = 0;  // the DoF vector on the device / as last brought to the caller's arrays (mistark_dofs_to_host_arrays_if_changed)
    // Evaluation kernels of the large closed-form potentials launched AHEAD of eval() (eval_prelaunch: while the callback that precedes an
    // evaluation — contact search, a caller's host code — keeps the host and the main stream busy). eval() takes the results if nothing
    // they depend on has changed (same kernel arguments, same pools), launches normally otherwise.
    struct EvalPre
    {
        struct Item
        {
            int pot;
            PotArgs args;
            const void* E;   // where the kernel wrote the element energies (slot 1: Context::elemE_pre, copied into place by eval())
            const void* H;
        };
        bool valid = false;
        int mode = 0;
        bool lazy_active = false;
        hipEvent_t ev_in = nullptr, ev_out = nullptr;
        std::vector<Item> items;
    } pre[2];  // [0]: an energy-only evaluation, [1]: one with gradient (and Hessian) — the latter may be a whole line-search trial ahead
    hipStream_t pre_stream = nullptr;
    DevBuf<double> elemE_pre;  // element energies of slot 1's kernels (the energy-only evaluation of the same state sums elemE meanwhile)
    bool no_eval_prelaunch = false;
    // energy-only evaluations: descriptors of the small potentials that share one launch (kernels.hip: k_eval_p_multi)
    bool no_multi_eval_p = false;
    std::vector<char> multi_p_sent;
    DevBuf<char> multi_p_dev;
    bool late_eager_assembly = false;  // option (measurement): the static part's gather queued behind the gradient gathers and the join, as through round 4
    bool no_multi_eval_pgh = false;  // option: every contact / friction table in a launch of its own (k_eval_pgh) instead of one shared launch (k_eval_pgh_multi)
    std::vector<char> multi_h_sent;
    DevBuf<char> multi_h_dev;
    int64_t n_multi_pgh = 0;         // counter "multi_pgh_launches"
    int64_t n_prelaunch_used = 0, n_prelaunch_dropped = 0;
    // anything that changes what kernels read (arrays, DoFs, tables registered by the caller): contact detection caches and a prelaunched
    // evaluation are void
    // (an array / a table of its own: only a prelaunched kernel that reads it is void — the SymX shim re-sends the contact tables and their
    // data between the callback and the evaluation)
    void touch_array(int id)
    {
        bool read = id < 0 || id >= (int)arrays.size() || arrays[(size_t)id].dof_set >= 0;
        for (const EvalPre& q : pre)
            if (q.valid && !read)
                for (const EvalPre::Item& it : q.items)
                    for (const mistark_binding& b : pots[(size_t)it.pot].bindings) read = read || b.array == id;
        if (read) touch();
        else data_version++;
    }
    void touch_potential(int pot)
    {
        bool mine = false, any = false;
        for (const EvalPre& q : pre) {
            any = any || q.valid;
            if (q.valid)
                for (const EvalPre::Item& it : q.items) mine = mine || it.pot == pot;
        }
        if (mine || !any) touch();
        else data_version++;
    }
    void touch()
    {
        data_version++;
        for (EvalPre& q : pre)
            if (q.valid) {
                (void)hipStreamWaitEvent(stream, q.ev_out, 0);  // (whatever comes next on the main stream must not overtake the kernels still reading)
                q.valid = false;
                n_prelaunch_dropped++;
            }
    }
    bool seg_sort = false;          // option "seg_sort": the contact search's box list and key list sorted by the engine's own kernels (k_seg_sort, k_rank_sort_keys: measured slower)
    bool no_sharded_search = false; // option "no_sharded_search": every rank of a sharded problem sweeps all candidate pairs itself (cross-check)
    bool no_contact_cache = false;  // option "no_contact_cache": every detection request runs the search (cross-check)
    size_t h_scratch_n = 0;

    // SpMV timing
    bool time_spmv = false;
    std::vector<hipEvent_t> ev;
    std::vector<hipEvent_t> pcg_ev;  // batch completion events of the PCG driver
    std::vector<hipEvent_t> stage_ev;  // stage marks of newton_solve (GPU-side stage times without synchronising)
    double spmv_ms_sum = 0.0;
    double spmv_empty_ms_sum = 0.0;  // empty event brackets recorded right behind the sampled launches
    int64_t spmv_n = 0;
    DevBuf<double> grad_aux;  // gradient contributions of the potentials evaluated on the auxiliary stream (eval())
    // Gradient of the potentials whose tables live on the device (contact, friction): their kernels write node gradients to ONE pool
    // (contribution g = offset of the potential + block * n_elem + element); the contributions are sorted by block row (stable: ties in
    // contribution order) whenever the tables change, and one gather adds every row's sum in that order, once. No atomics: the gradient of
    // contact scenes is bit-reproducible from run to run like that of contact-free ones.
    DevBuf<double> dyn_gpool;
    DevBuf<uint32_t> dyn_key, dyn_key_alt, dyn_val, dyn_val_alt;
    DevBuf<unsigned char> dyn_desc;
    DevBuf<uint8_t> dyn_cub_tmp;
    DevBuf<uint32_t> dyn_long;      // [0]: number of long rows, then (first, end) pairs
    DevBuf<double> dyn_long_part;   // chunk sums of the long rows (k_dyn_grad_gather_long)
    const uint32_t* dyn_sorted_key = nullptr;
    const uint32_t* dyn_sorted_val = nullptr;
    int64_t dyn_total = 0;          // contributions of all such potentials
    int dyn_n_desc = 0;
    uint64_t dyn_tables_version = 1, dyn_inc_version = 0;  // bumped by prepare() / the version the sorted lists were built for
    bool no_dyn_pool = false;       // option "no_dyn_pool": atomics instead (cross-check)
    bool atomic_projection = false; // option "atomic_projection": the projection adds float deltas to assembled blocks atomically (arrival order) instead of re-gathering the blocks
    std::vector<mistark_newton_iteration> newton_log;  // per-iteration records of the last newton_solve
    uint64_t* spmv_clk = nullptr;  // pinned: per-workgroup (start, end) of the sampled launches on the device's constant clock
    uint64_t* spmv_clk_sharded = nullptr;  // the same for pcg_sharded (up to 64 samples per solve)
    double spmv_clk_ticks = 0.0;
    int64_t spmv_clk_n = 0;

    // Solver numbering on one GPU: the matrix and the PCG vectors use a permutation of the block rows — Morton order of
    // the positions the caller handed over with mistark_dist_set_row_coords (rows without one, rigid bodies, at the end) — so that the 64 gathers
    // of an SpMV tile fall into few cache lines whatever order the caller numbered its nodes in (the reference's grid generator numbers all
    // hexahedron corners first and the centres behind them: mesh_generators.cpp:301-309). Applied where the keys of the sparsity pattern are
    // formed (PotArgs::lrow, the mechanism of the sharded path), undone where a vector or the matrix leaves the solver (pcg(), get_bsr, spmv).
    bool perm_active = false;
    bool no_row_order = false;      // option "no_row_order": natural numbering (cross-check)
    int row_order_mode = 0;         // option "row_order": 0 = by positions when given, else breadth-first order of the element graph; 1 = breadth-first order even with positions (measurement)
    DevBuf<int32_t> perm, iperm;    // solver row of a block row / block row of a solver row
    std::vector<int32_t> perm_h, iperm_h;
    std::vector<int64_t> perm_sig;  // what the permutation was computed for
    // multi-GPU: elements of every potential are sharded by contiguous ranges; E, gradient and the assembled matrix
    // are summed over the ranks; everything else is replicated
    int rank = 0, world = 1;
    std::unique_ptr<struct Collective> coll;
    bool owns_stream = true;        // false: the stream belongs to the in-process group (LocalCollective)
    Shard sh;
    int64_t mrows() const { return world > 1 ? sh.n_own : nbr; }  // block rows / columns of the matrix this context holds
    int64_t mcols() const { return world > 1 ? sh.n_loc : nbr; }
    DevBuf<double> dist_scalar;
    DevBuf<double> xl;              // sharded PCG: the solution in local numbering
    // sharded PCG through the windows (pcg_sharded_fused): the running tag of its messages, and the option to keep the unfused iteration
    uint32_t fused_tag = 0;
    // the ranks whose CURRENT matrix references a send row of mine as a column (a subset of Shard::send_mask, which lists every rank that
    // holds the row as a ghost — all of them for a contact surface's vertices): the halo of the fused PCG goes to these only
    DevBuf<uint32_t> cg_send_mask;
    DevBuf<double> cg_want_s, cg_want_r;
    uint64_t cg_mask_pattern = 0;
    int64_t cg_mask_lists = -1;
    bool no_halo_subset = false;    // option "no_halo_subset": push to every ghost holder
    bool no_fused_pcg = false;      // option "no_fused_pcg"
    int cg_variant = 0;             // option "cg_variant": 1 = the Chronopoulos-Gear iteration on one GPU, too (pcg_cg; measurement / cross-check)
    int64_t n_fused_solves = 0, n_unfused_solves = 0;  // sharded solves by iteration kind (mistark_dist_info)
    struct FusedReplay  // the last converged fused solve, for fused_pcg_replay (kernels.hip)
    {
        bool valid = false;
        uint32_t base = 0;
        int n = 0;
        uint64_t pattern = 0;
    } fused_replay;
    // sharded projection: delta records of this rank, the common exchange buffer and its sort scratch (kernels.hip: exchange_projection_deltas)
    DevBuf<unsigned char> src_ranges;  // descriptor table of k_make_desc
    struct ContactSystem* contact = nullptr;  // device contact detector (contact.hip), created by mistark_contact_init

    int last_cg_iters = 0;          // iteration count of the previous solve (first-batch predictor)
    // statistics of the last evaluation
    int64_t n_projected_total = 0;

    ~Context();
};

// host-side kernels launchers (kernels.hip)
// device -> host copy of a few scalars through pinned memory + stream synchronisation (a pageable destination would take HIP's
// slow staged path: tens of microseconds of idle GPU per call, dozens of calls per Newton iteration)
void direct_mf_destroy(void* multifrontal);  // direct.hip
void fetch(Context& c, void* dst_host, const void* src_dev, size_t bytes);
// host -> device copy of a caller's (pageable) array on c.stream through two pinned staging areas: HIP's own pageable path pins and unpins
// the range inside every copy (1.5 ms for 4 MB measured); a memcpy into pinned memory and a DMA transfer take a quarter of that. The source
// may be reused when the call returns; the copy is ordered on c.stream like any other.
void h2d_staged(Context& c, void* dst_dev, const void* src_host, size_t bytes);
// option pin_host_arrays: is [host, host + bytes) page-locked (registering it now if it is large enough and not yet known)?
bool host_range_pinned(Context& c, const void* host, size_t bytes);
void host_range_unpin(Context& c, const void* host);
// fills and device-to-device copies as kernels of our own (kernels.hip: the runtime's blit path costs the host 10-25 us per call); a FillQueue
// collects up to FILL_BATCH_MAX regions (4-byte aligned, multiples of 4 bytes) into ONE launch: add(), add(), ..., flush()
constexpr int FILL_BATCH_MAX = 8;
struct FillQueue
{
    hipStream_t stream;
    void* ptr[FILL_BATCH_MAX];
    size_t words[FILL_BATCH_MAX];
    uint32_t value[FILL_BATCH_MAX];
    int n = 0;
    explicit FillQueue(hipStream_t s) : stream(s) {}
    ~FillQueue() { flush(); }
    void add(void* p, int byte_value, size_t bytes);
    void flush();
};
void fill_async(hipStream_t stream, void* p, int byte_value, size_t bytes);
void copy_async(hipStream_t stream, void* dst, const void* src, size_t bytes);
void prepare(Context& c);
void eval_prelaunch(Context& c, int mode, bool lazy);  // kernels.hip: see Context::EvalPre
// shard.hip
void shard_prepare(Context& c);                                        // partition, local numbering, element lists, exchange tables (from prepare())
void shard_halo(Context& c, double* v_local);                           // ghosts of a local vector (3 doubles per local row) from their owners
void shard_halo_global(Context& c, double* v_global);                   // the same for a vector in global numbering (the gradient)
void shard_gather_global(Context& c, const double* v_local, double* v_global);  // owned parts of all ranks -> global vector on every rank
void shard_to_local(Context& c, const double* v_global, double* v_local, bool with_ghosts);
void shard_allgather_scalars(Context& c, const double* mine, int n, double* all_host);  // all_host[r * n + i]; blocking
double shard_sum(Context& c, double mine);
void shard_check(Context& c);
struct ElemTable  // block rows of the elements of one potential: rows[e * nb + k]
{
    const int32_t* rows;
    int64_t n_elem;
    int nb;
};
void rcb_partition_rows(int64_t n_block_rows, int world, const double* xyz, const std::vector<int64_t>& weight, std::vector<int32_t>& owner);
void graph_partition_rows(int64_t n_block_rows, int world, const std::vector<ElemTable>& tables, const uint8_t* hub, std::vector<int32_t>& owner);
void static_graph_order(Context& c, std::vector<int32_t>& rows_in_order);  // shard.hip
void ensure_pattern(Context& c);
void contact_destroy(struct ContactSystem* cs);
int64_t contact_searches(const Context& c, bool repeated);  // counters "contact_searches" / "contact_repeated_searches"
int64_t contact_sharded_searches(const Context& c);  // searches of a sharded problem whose sweep was dealt out to the ranks (contact.hip)
void contact_shared_rows(Context& c, std::vector<int32_t>& rows);  // contact.hip
int register_potential(Context& c, const char* name, const int32_t* conn, int32_t n_elem, int32_t conn_stride, const mistark_binding* bindings, int32_t n_bindings);
void eval(Context& c, int mode, double* E, double* grad_host, double* grad_max_abs = nullptr, bool lazy = false);
void reduce_dot_and_max_abs(Context& c, const double* a, const double* b, int64_t n, double* dot, double* max_abs_a);
bool project_can_speculate(const Context& c);
void project_speculate(Context& c, double eps, int mirroring, double threshold, bool ev_in_recorded = false);
void project_speculate_request(Context& c, double eps, int mirroring, double threshold);
void project_speculate_pending(Context& c);
void project_spec_poll(Context& c);
bool project_spec_adopt(Context& c, double eps, int mirroring, double threshold, int* all_active, int64_t* n_projected_now);
void project_spec_discard(Context& c);
void project(Context& c, double eps, int mirroring, const uint8_t* active_host, bool by_gradient, double threshold, int* all_active,
             int64_t* n_projected_now, int64_t* n_changed_now);
void assemble(Context& c);
void build_preconditioner(Context& c);
double spmv_bench(Context& c, int n);
void rows_from_solver(Context& c, const double* v_solver, double* v_caller);  // Context::perm_active: solver numbering <-> the caller's
void rows_to_solver(Context& c, const double* v_caller, double* v_solver);
void fused_pcg_replay(Context& c, int n_launches, double* s_us, double* v_us);
void spmv_device(Context& c, const double* x, double* y, const double* pdot, double* partials, bool timed);
// rhs_scale: the system solved is A x = rhs_scale * rhs (the Newton loop passes the gradient and -1; single GPU only)
void pcg(Context& c, const double* rhs_dev, double abs_tol, double rel_tol, int max_iter, int stop_on_indef, mistark_pcg_info* info, double rhs_scale = 1.0);
bool direct_llt(Context& c, const double* rhs_dev, double* x_dev);  // direct.hip
// custom.hip
std::shared_ptr<CustomProgram> make_custom_program(const std::string& name, const int32_t* strides, int n_bindings, const int32_t* ops, const double* consts, int n_ops, int n_inputs,
                                                   const int32_t* cond_ops, const double* cond_consts, int n_cond_ops);
void custom_program_set_summation(CustomProgram& G, const std::string& name, int first_input, int stride, int n_iterations, const double* data);
void launch_eval_custom(Context& c, Potential& P, int mode);
void h2d_small(Context& c, void* dst_dev, const void* src_host, size_t bytes);
double reduce_max_abs(Context& c, const double* v, int64_t n);
double reduce_dot(Context& c, const double* a, const double* b, int64_t n);
void vec_axpby(Context& c, double* dst, double a, const double* x, double b, const double* y, int64_t n);
void vec_fill(Context& c, double* dst, double v, int64_t n);
void vec_neg(Context& c, double* dst, const double* x, int64_t n);
int find_kind(const char* name);
int kind_nb(int kind);
int kind_nbind(int kind);
void kind_strides(int kind, int* out);
const char* kind_name(int kind);
int n_kinds();

int newton_solve(Context& c, const mistark_newton_settings& s, const mistark_newton_callbacks* cb, mistark_newton_stats& st);

// custom.hip: what the emitter writes for an op sequence (and, with `compile`, the size of the code object hipRTC makes of it); no context, no GPU
std::string custom_emit_source(const std::string& name, const int32_t* strides, int n_bindings, const int32_t* in_dof, const int32_t* ops, const double* consts, int n_ops, int n_inputs,
                               const int32_t* cond_ops, const double* cond_consts, int n_cond_ops, int NB, bool compile, size_t* code_bytes);
}  // namespace mistark

struct mistark_ctx
{
    mistark::Context c;
};
