// csr5_capi.hip -- the extern "C" boundary of libcsr5hip.so (declared in include/csr5hip.h).
//
// State machine and ownership follow the reference handle (CSR5_cuda/anonymouslib_cuda.h):
//   inputCSR borrows the caller's device CSR arrays and sets _format = CSR          (:61-76)
//   asCSR5   builds tile_ptr / tile_desc / offsets and transposes col/val IN PLACE  (:105-220)
//   spmv     returns UNSUPPORTED_CSR_SPMV (-4) while the format is CSR              (:262-284)
//   asCSR / destroy undo the transpose and drop the CSR5 arrays                     (:78-102, :286-291)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "csr5_internal.h"

using namespace csr5;

namespace {

thread_local std::string g_last_error;

} // namespace

namespace csr5 {
void set_last_error(const std::string &msg) { g_last_error = msg; }
} // namespace csr5

namespace {

int fail_hip(hipError_t e, const char *what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return CSR5HIP_HIP_ERROR;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail_hip(e_, #expr);                                                            \
    } while (0)

struct Buffer {
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        if (ptr) {
            hipError_t e = hipFree(ptr);
            if (e != hipSuccess)
                return e;
            ptr = nullptr;
            cap = 0;
        }
        hipError_t e = hipMalloc(&ptr, bytes);
        if (e == hipSuccess)
            cap = bytes;
        return e;
    }
    void release()
    {
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

struct GraphKey {
    void *y;
    int count;
    int mode;
    bool operator==(const GraphKey &o) const { return y == o.y && count == o.count && mode == o.mode; }
};
struct GraphKeyHash {
    size_t operator()(const GraphKey &k) const
    {
        return std::hash<void *>()(k.y) ^ (size_t)k.count * 1000003u ^ (size_t)k.mode * 7919u;
    }
};

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

} // namespace

struct csr5hip_handle_s {
    int format = -1; // uninitialised until inputCSR, as in the reference
    int value_type = CSR5HIP_F64;
    Geometry g{};
    int sigma_request = 0; // what setSigma stored (resolved value, >= 1)
    int num_offsets = 0;
    hipStream_t stream = nullptr;
    const void *x = nullptr;
    DeviceArrays d{};
    SpmvOptions opt{1, 1, 0, 0, 0, 0}; // fused single-launch SpMV, XCD-contiguous tile ranges
    int nt_request = 1;   // CSR5HIP_OPT_STREAM_NT: 0 off, 1 auto (default), 2 force
    int ldsy_request = 1; // CSR5HIP_OPT_LDS_Y: 0 off, 1 auto (default), 2 force
    int xwin_request = 1; // CSR5HIP_OPT_X_WINDOW: 0 off, 1 auto (default), 2 force
    int xwin_tiles = 0;   // tiles that got a window at conversion
    long long xwin_covered = 0; // non-zeros inside those windows
    long long xwin_lines = 0;   // distinct x lines under the in-window lanes of one sampled gather, summed over those tiles
    // every auxiliary array of the conversion lives in ONE allocation (one hipMalloc, one memset per asCSR5 instead
    // of ten and six: the small-matrix conversion is launch-bound)
    Buffer b_arena;
    void *scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    uint32_t scalar_words[2] = {0, 0}; // landing zone of the two 4-byte reads of the conversion (checkpoint loading)
    uint32_t *host_words = nullptr;    // 32 pinned, device-visible words the last conversion kernels export into
    // deferred carries (Geometry.defer; classification in csr5_format.hip tile_carry_meta): decided at conversion
    int defer_request = 1;       // CSR5HIP_OPT_DEFER_CARRIES: 0 off, 1 auto (default), 2 force
    // narrow column codes of the x-window kernel (csr5_format.hip k_col16): built when that kernel is selected
    int col16_request = 1;       // CSR5HIP_OPT_NARROW_COLUMNS: 0 off, 1 auto (default)
    bool col16_built = false;    // the codes of the current conversion exist
    unsigned col16_wide = 0;     // tiles that span >= 32 768 columns (the codes are used only when there is none)
    Buffer b_col16;              // codes [(p-1) * T / 2 words], then base16 [p]
    // flagged column words of the plain kernel at sigma 4..8 (csr5_format.hip k_col31)
    size_t device_total_bytes = 0; // hipMemGetInfo's total, asked once (build_slabs_impl)
    int col31_request = 1;       // CSR5HIP_OPT_FLAGGED_COLUMNS: 0 off, 1 auto (default), 2 force
    bool col31_built = false;
    Buffer b_col31;              // [(p-1) * T words]
    double wall_clock_khz = 0;         // rate of the device's constant wall clock (phase stamps)
    double t_malloc = 0, t_tile_ptr = 0, t_tile_desc = 0, t_transpose = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::unordered_map<GraphKey, hipGraphExec_t, GraphKeyHash> graphs;
    // column slabs (csr5_slab.hip): the stacked matrix lives in an internal child handle
    int slab_request = 1;  // CSR5HIP_OPT_COLUMN_SLABS: 0 off, 1 auto, 2..64 = S
    int slab_shift = 4;    // CSR5HIP_OPT_SLAB_SHIFT
    int zero_empty = 0;    // CSR5HIP_OPT_ZERO_EMPTY_ROWS
    bool is_child = false; // internal handle of a slab structure: never builds slabs itself
    int slab_mem_mib = 0;  // CSR5HIP_OPT_SLAB_MEMORY_MIB: cap on the structure's device memory (0 = none)
    bool slab_fallback = false; // the structure was wanted but could not be built: plain kernel in use
    unsigned generation = 0;    // bumped whenever captured graphs of this handle go stale (csr5hip_spmv_rotate keys on it)
    int slab_S = 0;        // > 0: spmv() runs child + combine
    int slab_m2 = 0;
    double t_slab = 0;
    csr5hip_handle_s *slab_child = nullptr;
    Buffer b_row_ptr2, b_col2, b_val2, b_val32, b_P, b_rowidx, b_base, b_nonempty;
    Buffer b_slab_tmp; // temporaries of the slab build; kept between conversions only while small (SLAB_TMP_KEEP)
    // LDS hot table of the slab child (k_spmv_range): chosen at conversion, see csr5_slab.hip
    int hot_request = 1;      // CSR5HIP_OPT_SLAB_HOT: 0 off, 1 auto, 2 force
    bool hot_enabled = false; // (child) its columns are hot-encoded as packed codes next to the plain column words (only the values
                              // are transposed): spmv must use the persistent range kernel
    int hot_cover_pct = 0;    // (parent) share of the non-zeros whose column got a table slot
    Buffer b_hot_cols, b_hot_count, b_hot_tile0, b_slab_off, b_lead, b_range_head;
    Buffer b_col_lo, b_col_hi; // packed column codes of a hot child (3 bytes per non-zero)
    // permuted copy of x behind the packed codes (csr5_hot.hip k_x_permute): table images + frequency-ordered cold regions
    Buffer b_cold_base, b_cold_cols, b_xperm;
    int x_snapshot = 0;    // CSR5HIP_OPT_X_SNAPSHOT: 0 = the copy is refreshed by every spmv(), 1 = by setX only
    int narrow_request = 0; // CSR5HIP_OPT_NARROW_VALUES: 1 = keep the hot child's fp64 values as fp32 when every one of them is exact
    bool values_narrowed = false;
    bool xperm_valid = false; // (snapshot mode) the copy holds the current x
    int cold_total = 0;       // entries of the cold region

    // csr5hip_spmv_rotate: one graph over several handles (cold-cache measurement protocol)
    hipGraphExec_t rotate_exec = nullptr;
    std::vector<void *> rotate_key;

    size_t vsize() const { return value_type == CSR5HIP_F64 ? 8 : 4; }
    void drop_graphs()
    {
        generation++;
        for (auto &kv : graphs)
            (void)hipGraphExecDestroy(kv.second);
        graphs.clear();
        if (rotate_exec)
            (void)hipGraphExecDestroy(rotate_exec);
        rotate_exec = nullptr;
        rotate_key.clear();
    }
};

// LDS x-window variant of the fused kernel: forced, or (auto) when the windows found at conversion
// cover at least XWIN_AUTO_COVER_PCT % of ALL non-zeros of the transposed tiles.  Measured on MI355X:
// ~95 % coverage (banded) -> 1.2-1.4x faster from 1.4 k to 28 k tiles; ~50 % coverage (half the
// columns random) -> 0-20 % slower, because every step still needs a divergent global gather.
static int xwin_decision(const csr5hip_handle_s *h)
{
    constexpr int XWIN_AUTO_COVER_PCT = 70;
    constexpr int XWIN_AUTO_MIN_SIGMA = 16;
    constexpr int XWIN_AUTO_MIN_LINES_F64 = 16;
    if ((long long)h->g.n * (long long)h->vsize() >= (1LL << 31))
        return 0; // (the window kernel reads x through a buffer resource with 32-bit byte offsets)
    if (h->xwin_request == 2)
        return 1;
    if (h->xwin_request != 1 || h->g.p <= 1 || h->xwin_tiles <= 0)
        return 0;
    // Staging a 4-KB slice of x costs 8 (fp64) or 16 (fp32) coalesced wave loads, LDS writes and a wave barrier
    // per tile; it replaces sigma gather instructions.  Measured on MI355X (scripts/experiments/xwin_by_rowlen.py
    // and the bench stand-ins), the window pays only when
    //  (a) it covers most non-zeros,
    //  (b) sigma is large enough to amortise the staging (sigma = 6: 8-11 % SLOWER on scircuit-/webbase-like with
    //      80-95 % near-diagonal entries; sigma = 16: faster), and
    //  (c) for fp64, a gather instruction is spread over many 128-byte lines of x (nd24k-like: 25 lines, +9 %;
    //      columns within +-64 of the diagonal: 8 lines, -2 %); fp32 gains at any spread (nd24k-like 1.45x).
    const bool covered = (long long)h->xwin_covered * 100 >=
                         (long long)(h->g.p - 1) * h->g.tile_elems * XWIN_AUTO_COVER_PCT;
    const bool spread = h->value_type == CSR5HIP_F32 ||
                        h->xwin_lines >= (long long)XWIN_AUTO_MIN_LINES_F64 * h->xwin_tiles;
    return covered && h->g.sigma >= XWIN_AUTO_MIN_SIGMA && spread;
}

// y segments through LDS (coalesced flush): pays when a tile holds many rows.  Measured on MI355X
// (scripts/experiments/ldsy_by_rowlen.py): 5-15 % faster at 2-8 non-zeros per row, 1-3 % at 16, even at 20-32,
// 1-3 % slower beyond (one or two segments per tile: the compaction is pure overhead).
static int ldsy_decision(const csr5hip_handle_s *h)
{
    if (h->ldsy_request == 2)
        return 1;
    if (h->ldsy_request != 1 || h->g.m <= 0)
        return 0;
    return (long long)h->g.nnz <= 20LL * h->g.m;
}

// Non-temporal stream loads: measured on MI355X +5 % on R-MAT 22 (0.8 GB of streams) and -18..25 % on matrices
// that fit the 256-MiB Infinity Cache (they are re-streamed from HBM by every SpMV instead of staying cached).
static int nt_decision(const csr5hip_handle_s *h)
{
    if (h->nt_request == 2)
        return 1;
    if (h->nt_request != 1)
        return 0;
    const long long stream_bytes = (long long)h->g.nnz * (4 + (long long)h->vsize());
    return stream_bytes > 256LL * 1024 * 1024;
}

extern "C" {

const char *csr5hip_last_error(void) { return g_last_error.c_str(); }
const char *csr5hip_version(void) { return "csr5hip 0.1 (gfx950, omega=64)"; }

int csr5hip_create(csr5hip_handle *out, int m, int n, int value_type)
{
    if (!out || m < 0 || n < 0)
        return CSR5HIP_INVALID_ARGUMENT;
    if (value_type != CSR5HIP_F64 && value_type != CSR5HIP_F32)
        return CSR5HIP_UNSUPPORTED_VALUE_TYPE;
    csr5hip_handle h = new csr5hip_handle_s();
    h->g.m = m;
    h->g.n = n;
    h->value_type = value_type;
    *out = h;
    return CSR5HIP_SUCCESS;
}

static void release_slabs(csr5hip_handle h);

int csr5hip_free(csr5hip_handle h)
{
    if (!h)
        return CSR5HIP_INVALID_ARGUMENT;
    h->drop_graphs();
    release_slabs(h);
    h->b_arena.release();
    h->b_col16.release();
    h->b_col31.release();
    if (h->host_words)
        (void)hipHostFree(h->host_words);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
    return CSR5HIP_SUCCESS;
}

int csr5hip_set_stream(csr5hip_handle h, void *hip_stream)
{
    if (!h)
        return CSR5HIP_INVALID_ARGUMENT;
    h->stream = (hipStream_t)hip_stream;
    h->xperm_valid = false; // (snapshot mode: the copy was taken on the old stream; the new one takes its own, in order)
    h->drop_graphs();
    return CSR5HIP_SUCCESS;
}

int csr5hip_warmup(csr5hip_handle h)
{
    if (!h)
        return CSR5HIP_INVALID_ARGUMENT;
    HIP_TRY(launch_warmup(h->stream));
    return CSR5HIP_SUCCESS;
}

int csr5hip_input_csr(csr5hip_handle h, int nnz, int32_t *d_row_ptr, int32_t *d_col_idx, void *d_val)
{
    if (!h || nnz < 0)
        return CSR5HIP_INVALID_ARGUMENT;
    h->format = CSR5HIP_FORMAT_CSR;
    h->g.nnz = nnz;
    h->d.row_ptr = d_row_ptr;
    h->d.col = d_col_idx;
    h->d.val = d_val;
    h->drop_graphs();
    return CSR5HIP_SUCCESS;
}

int csr5hip_set_x(csr5hip_handle h, const void *d_x)
{
    if (!h)
        return CSR5HIP_INVALID_ARGUMENT;
    h->x = d_x;
    h->xperm_valid = false; // (snapshot mode: the next spmv() takes a new copy)
    h->drop_graphs();
    return CSR5HIP_SUCCESS;
}

// gfx950 tables in the shape of the reference's (r, s, t, u) rule (anonymouslib_cuda.h:297-313, one table per
// architecture and precision there too): k = nnz/m; sigma = r if k <= r; k if k <= s; s if k <= t; else u.
// fp64: (6, 16, 256, 16); fp32: (8, 16, 256, 16) -- from sweeps of all sigma over mean row lengths 2..512, random and
// near-diagonal columns (scripts/experiments/sigma_table.py; profiles/r01_sigma_table.txt, re-run on the round-4 kernels in
// profiles/r04_sigma_table.txt): the sigma surface is flat on gfx950 and the rules stay within a few per cent of the measured
// best everywhere.  r = 6 instead of the reference's 4 costs random-column fp64 matrices < 1 % and gains 4-8 % where the
// columns are local; fp32 rows are half as wide, so a tile of the same byte size holds twice the elements: short rows
// (k <= 8) with local columns ran 5-8 % faster at sigma = 8 than at 6, at no cost with random columns.  Beyond 256
// non-zeros per row sigma = 32 would gain 3-7 % on random columns but loses 10 % on the nd24k-like stand-in (x-window +
// jitter): u = 16 for fp64.  fp32: 24 won by 3 % for a while in round 5 (a 6-KB tile like fp64's sigma = 16); since the narrow
// codes carry the row-start flags the sigma = 24 kernel needs 136 VGPRs (3 wavefronts per SIMD) and sigma = 16 (78 VGPRs, 6 per
// SIMD) wins: nd24k-like 29.1 / 37.0 us warm / cold against 31.7 / 38.9: u = 16 again.
int csr5hip_auto_sigma(int m, int nnz, int value_type)
{
    const int r = value_type == CSR5HIP_F32 ? 8 : 6, s = 16, t = 256, u = 16;
    const int k = m > 0 ? nnz / m : 0;
    if (k <= r) return r;
    if (k <= s) return k;
    if (k <= t) return s;
    return u;
}

int csr5hip_set_sigma(csr5hip_handle h, int sigma)
{
    if (!h)
        return CSR5HIP_INVALID_ARGUMENT;
    if (sigma == CSR5HIP_AUTO_TUNED_SIGMA)
        sigma = csr5hip_auto_sigma(h->g.m, h->g.nnz, h->value_type);
    if (sigma < CSR5HIP_MIN_SIGMA || sigma > CSR5HIP_MAX_SIGMA)
        return CSR5HIP_INVALID_ARGUMENT;
    h->sigma_request = sigma;
    return CSR5HIP_SUCCESS;
}

static int build_slabs(csr5hip_handle h);
static int build_slabs_impl(csr5hip_handle h);
static int prepare_col16(csr5hip_handle h);
static int prepare_col31(csr5hip_handle h);
// kernel-side tables of the plain (non-slab) path that are built on demand: the narrow column codes of the x-window kernel,
// else the flagged column words of the plain kernel
static int prepare_plain(csr5hip_handle h)
{
    const int rc = prepare_col16(h);
    return rc != CSR5HIP_SUCCESS ? rc : prepare_col31(h);
}

int csr5hip_set_option(csr5hip_handle h, int option, int value)
{
    if (!h)
        return CSR5HIP_INVALID_ARGUMENT;
    switch (option) {
    case CSR5HIP_OPT_SPMV_MODE:
        if (value != 0 && value != 1)
            return CSR5HIP_INVALID_ARGUMENT;
        h->opt.mode = value;
        if (h->format == CSR5HIP_FORMAT_CSR5 && h->slab_S <= 0) {
            const int rc = prepare_plain(h);
            if (rc != CSR5HIP_SUCCESS)
                return rc;
        }
        if (h->slab_S > 0 && h->slab_child->hot_enabled && value != 1) {
            // the hot table exists for the fused kernel only and its column words are encoded: rebuild without it
            h->drop_graphs();
            return build_slabs(h);
        }
        break;
    case CSR5HIP_OPT_XCD_REMAP:
        h->opt.xcd_remap = value ? 1 : 0;
        break;
    case CSR5HIP_OPT_LDS_Y:
        if (value < 0 || value > 2)
            return CSR5HIP_INVALID_ARGUMENT;
        h->ldsy_request = value;
        if (h->format == CSR5HIP_FORMAT_CSR5)
            h->opt.lds_y = ldsy_decision(h);
        break;
    case CSR5HIP_OPT_STREAM_NT:
        if (value < 0 || value > 2)
            return CSR5HIP_INVALID_ARGUMENT;
        h->nt_request = value;
        if (h->format == CSR5HIP_FORMAT_CSR5) {
            h->opt.stream_nt = nt_decision(h);
            if (h->is_child && h->hot_enabled) // another instantiation of the range kernel: its LDS limit
                HIP_TRY(prepare_spmv_hot(h->g, h->d, h->value_type, h->opt));
        }
        break;
    case CSR5HIP_OPT_X_WINDOW:
        if (value < 0 || value > 2)
            return CSR5HIP_INVALID_ARGUMENT;
        h->xwin_request = value;
        if (h->format == CSR5HIP_FORMAT_CSR5) {
            h->opt.x_window = xwin_decision(h);
            const int rc = h->slab_S <= 0 ? prepare_plain(h) : CSR5HIP_SUCCESS;
            if (rc != CSR5HIP_SUCCESS)
                return rc;
        }
        break;
    case CSR5HIP_OPT_COLUMN_SLABS:
        if (value < 0 || value > 64 || (value & (value - 1)))
            return CSR5HIP_INVALID_ARGUMENT;
        h->slab_request = value;
        if (h->format == CSR5HIP_FORMAT_CSR5) {
            h->drop_graphs();
            return build_slabs(h);
        }
        break;
    case CSR5HIP_OPT_SLAB_SHIFT:
        if (value < 0 || value > 24)
            return CSR5HIP_INVALID_ARGUMENT;
        h->slab_shift = value;
        if (h->format == CSR5HIP_FORMAT_CSR5) {
            h->drop_graphs();
            return build_slabs(h);
        }
        break;
    case CSR5HIP_OPT_SLAB_HOT:
        if (value < 0 || value > 2)
            return CSR5HIP_INVALID_ARGUMENT;
        h->hot_request = value;
        if (h->format == CSR5HIP_FORMAT_CSR5) {
            h->drop_graphs();
            return build_slabs(h);
        }
        break;
    case CSR5HIP_OPT_ZERO_EMPTY_ROWS:
        h->zero_empty = value ? 1 : 0;
        break;
    case CSR5HIP_OPT_X_SNAPSHOT:
        if (value != 0 && value != 1)
            return CSR5HIP_INVALID_ARGUMENT;
        h->x_snapshot = value;
        h->xperm_valid = false;
        break;
    case CSR5HIP_OPT_NARROW_VALUES:
        if (value != 0 && value != 1)
            return CSR5HIP_INVALID_ARGUMENT;
        if (h->narrow_request != value) {
            h->narrow_request = value;
            if (h->format == CSR5HIP_FORMAT_CSR5) {
                h->drop_graphs();
                return build_slabs(h);
            }
        }
        break;
    case CSR5HIP_OPT_DEFER_CARRIES:
        if (value < 0 || value > 2)
            return CSR5HIP_INVALID_ARGUMENT;
        if (h->format == CSR5HIP_FORMAT_CSR5 && value != h->defer_request) {
            set_last_error("CSR5HIP_OPT_DEFER_CARRIES takes effect at asCSR5(): set it while the matrix is in CSR form");
            return CSR5HIP_INVALID_ARGUMENT;
        }
        h->defer_request = value;
        break;
    case CSR5HIP_OPT_NARROW_COLUMNS:
        if (value != 0 && value != 1)
            return CSR5HIP_INVALID_ARGUMENT;
        h->col16_request = value;
        if (h->format == CSR5HIP_FORMAT_CSR5 && h->slab_S <= 0) {
            const int rc = prepare_col16(h);
            if (rc != CSR5HIP_SUCCESS)
                return rc;
        }
        break;
    case CSR5HIP_OPT_FLAGGED_COLUMNS:
        if (value < 0 || value > 2)
            return CSR5HIP_INVALID_ARGUMENT;
        h->col31_request = value;
        if (h->format == CSR5HIP_FORMAT_CSR5 && h->slab_S <= 0) {
            const int rc = prepare_col31(h);
            if (rc != CSR5HIP_SUCCESS)
                return rc;
        }
        break;
    case CSR5HIP_OPT_SLAB_MEMORY_MIB:
        if (value < 0)
            return CSR5HIP_INVALID_ARGUMENT;
        h->slab_mem_mib = value;
        if (h->format == CSR5HIP_FORMAT_CSR5) {
            h->drop_graphs();
            return build_slabs(h);
        }
        break;
    default:
        return CSR5HIP_INVALID_ARGUMENT;
    }
    if (h->slab_S > 0 && (option == CSR5HIP_OPT_SPMV_MODE || option == CSR5HIP_OPT_LDS_Y || option == CSR5HIP_OPT_STREAM_NT))
        csr5hip_set_option(h->slab_child, option, value);
    h->drop_graphs();
    return CSR5HIP_SUCCESS;
}

// ---- pieces of the conversion shared by csr5hip_as_csr5 and csr5hip_load -------------------------
// geometry from sigma (anonymouslib_cuda.h:121-137)
static int derive_geometry(csr5hip_handle h, int sigma)
{
    Geometry &g = h->g;
    g.sigma = sigma;
    g.bit_y = bit_y_of(g.sigma);
    g.bit_all = g.bit_y + BIT_SS;
    if (g.bit_all > 31) // the first flag must sit in the first packet (anonymouslib_cuda.h:130)
        return CSR5HIP_UNSUPPORTED_CSR5_OMEGA;
    g.num_packet = num_packet_of(g.sigma);
    g.tile_elems = OMEGA * g.sigma;
    g.p = (int)(((long long)g.nnz + g.tile_elems - 1) / g.tile_elems);
    g.tail_start = g.m;
    g.defer = 0;
    h->num_offsets = 0;
    h->xwin_tiles = 0;
    h->xwin_covered = 0;
    h->xwin_lines = 0;
    h->opt.long_runs = 0;
    h->col16_built = false; // (the column words are about to be permuted again)
    h->col31_built = false;
    h->opt.col31 = 0;
    h->d.col31 = nullptr;
    h->opt.col16 = 0;
    h->d.col16 = nullptr;
    h->d.base16 = nullptr;
    h->t_malloc = h->t_tile_ptr = h->t_tile_desc = h->t_transpose = 0;
    h->drop_graphs();
    return CSR5HIP_SUCCESS;
}

// the auxiliary arrays of the handle (anonymouslib_cuda.h:142-151) plus the carry state: one arena, its zero-initialised
// part cleared by ONE memset.  `offset` is sized by its upper bound (one slot per row start plus one per tile), so the
// conversion does not have to stop for the host to read num_offsets before it can continue (format_cuda.h:331-343
// does; that read now rides with the final one).
static int reserve_aux(csr5hip_handle h)
{
    const Geometry &g = h->g;
    hipStream_t s = h->stream;
    const size_t p1 = (size_t)g.p + 1;
    // (a hot slab child keeps no descriptor array: see build_format_arrays)
    const size_t desc_words = h->is_child && h->hot_enabled ? OMEGA : (size_t)(g.p > 0 ? g.p : 1) * OMEGA * g.num_packet;
    const size_t offset_cap = h->is_child ? p1 : (size_t)g.m + p1; // (a slab child has no empty rows)
    h->scan_tmp_bytes = offset_scan_tmp_bytes((int)p1);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    // zero-initialised part first
    const size_t o_desc = take(desc_words * 4), o_offp = take(p1 * 4), o_cal = take(p1 * h->vsize()),
                 o_acc = take(p1 * h->vsize()), o_cnt = take(p1 * 4), o_counters = take(COUNTER_WORDS * 4), o_tp = take(p1 * 4), o_offset = take(offset_cap * 4);
    const size_t zero_bytes = off;
    const size_t o_meta = take(p1 * 16), o_hdr = take(p1 * 32), o_scan = take(h->scan_tmp_bytes);
    HIP_TRY(h->b_arena.reserve(off));
    char *base = (char *)h->b_arena.ptr;
    h->d.tile_desc = (uint32_t *)(base + o_desc);
    h->d.offset_ptr = (int32_t *)(base + o_offp);
    h->d.calibrator = base + o_cal;
    h->d.carry_acc = base + o_acc;
    h->d.carry_cnt = (uint32_t *)(base + o_cnt);
    h->d.counters = (uint32_t *)(base + o_counters);
    h->d.offset = (int32_t *)(base + o_offset);
    h->d.tile_ptr = (uint32_t *)(base + o_tp);
    h->d.carry_meta = (uint32_t *)(base + o_meta);
    h->d.tile_hdr = (uint32_t *)(base + o_hdr);
    h->scan_tmp = base + o_scan;
    HIP_TRY(hipMemsetAsync(base, 0, zero_bytes, s));
    return CSR5HIP_SUCCESS; // stream-ordered: the conversion kernels follow on the same stream
}

// the two words the host needs from the conversion -- tail start and number of offsets, as in the reference
// (anonymouslib_cuda.h:165-167, format_cuda.h:331-343) -- copied asynchronously; valid after the next synchronisation
static hipError_t read_format_scalars(csr5hip_handle h, bool sync)
{
    const Geometry &g = h->g;
    h->scalar_words[0] = h->scalar_words[1] = 0;
    if (g.p <= 0)
        return hipSuccess;
    hipError_t e = hipMemcpyAsync(&h->scalar_words[0], h->d.tile_ptr + (g.p - 1), 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(&h->scalar_words[1], h->d.offset_ptr + g.p, 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && sync)
        e = hipStreamSynchronize(h->stream);
    return e;
}
static void finish_format_scalars(csr5hip_handle h)
{
    if (h->g.p <= 0)
        return;
    h->g.tail_start = (int)(h->scalar_words[0] & ROW_MASK);
    h->num_offsets = (int)h->scalar_words[1];
}

// ---- narrow column codes (csr5_format.hip k_col16, csr5_spmv.hip C16) ---------------------------------------------------
// Only the x-window kernel reads them (a matrix whose tiles are covered by 4-KB windows of x has local columns), so they are
// built when that kernel is selected: one pass over the tile-ordered column words + one synchronisation for the count of wide
// tiles; 2 bytes per non-zero of device memory.  Stays valid until the next conversion.
static int prepare_col16(csr5hip_handle h)
{
    h->opt.col16 = 0;
    h->d.col16 = nullptr;
    h->d.base16 = nullptr;
    const Geometry &g = h->g;
    const bool windowed = h->opt.x_window != 0;
    if (h->col16_request == 0 || h->is_child || g.p <= 1 || h->opt.mode != 1 || !windowed || !col16_sigma(g.sigma))
        return CSR5HIP_SUCCESS;
    const size_t code_words = (size_t)(g.p - 1) * (g.tile_elems / 2);
    if (!h->col16_built) {
        // The codes are an optional accelerator (2 bytes per non-zero of device memory): when they cannot be built the handle
        // stays on the 32-bit column words -- csr5hip_last_error() says so -- instead of failing asCSR5() / setOption().
        uint32_t *wide = h->d.counters + 5;
        hipError_t e = h->b_col16.reserve((code_words + (size_t)g.p + 1) * 4);
        if (e == hipSuccess)
            e = hipMemsetAsync(wide, 0, 4, h->stream);
        if (e == hipSuccess)
            e = launch_col16(g, h->d, (uint32_t *)h->b_col16.ptr, (int32_t *)h->b_col16.ptr + code_words, wide, h->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(&h->col16_wide, wide, 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) {
            (void)hipGetLastError(); // clear the sticky allocation / launch error
            h->b_col16.release();
            set_last_error(std::string("narrow column codes not built, 32-bit column words in use: ") + hipGetErrorString(e));
            h->drop_graphs();
            return CSR5HIP_SUCCESS;
        }
        h->col16_built = true;
    }
    if (h->col16_wide == 0) {
        h->d.col16 = (const uint32_t *)h->b_col16.ptr;
        h->d.base16 = (const int32_t *)h->b_col16.ptr + code_words;
        h->opt.col16 = 1;
    }
    h->drop_graphs();
    return CSR5HIP_SUCCESS;
}

// ---- flagged column words (csr5_format.hip k_col31, csr5_spmv.hip C31) ----------------------------------------------------
// The plain fused kernel at sigma 4..8 (short rows: the auto rule's r = 6 / 8) reads its column words from a kernel-side copy
// that carries the element's row-start flag in bit 31: no descriptor load (256 B of a sigma = 6 tile's 4.9 KB and one of its
// load instructions), y_offset recomputed from the flags by a wave prefix sum as in the narrow-codes kernel; same gathers, same
// arithmetic, bit-identical results.  An optional accelerator (+4 bytes per non-zero of device memory): when it cannot be built
// the handle stays on column_index + tile_desc.  Not for the x-window kernel (it has the narrow codes) nor for a slab child.
// Auto rule (same-call A/B, scripts/experiments/round6/flagged_ab.py, profiles/r06_probes.txt section 4): the descriptor load rides
// in the tile's first round trip, so dropping it saves BYTES, not latency, and the flag extraction + prefix sum run after the data
// has arrived -- a matrix that streams from HBM gains (2 M rows x 27 per row, sigma 8: 168.9 -> 165.2 us, -2.2 %), the latency-bound
// stand-ins lose (scircuit-like +0.7 % cold, webbase-like without slabs +1.1 %, their local variants +3..5 %): on only when the
// streams exceed the Infinity Cache (the non-temporal rule's size).
static int prepare_col31(csr5hip_handle h)
{
    h->opt.col31 = 0;
    h->d.col31 = nullptr;
    const Geometry &g = h->g;
    const bool pays = (long long)g.nnz * (4 + (long long)h->vsize()) > 256LL * 1024 * 1024;
    if (h->col31_request == 0 || (h->col31_request == 1 && !pays) || h->is_child || g.p <= 1 || h->opt.mode != 1 || h->opt.x_window ||
        !col31_sigma(g.sigma)) {
        h->drop_graphs();
        return CSR5HIP_SUCCESS;
    }
    if (!h->col31_built) {
        hipError_t e = h->b_col31.reserve((size_t)(g.p - 1) * g.tile_elems * 4);
        if (e == hipSuccess)
            e = launch_col31(g, h->d, (uint32_t *)h->b_col31.ptr, h->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) {
            (void)hipGetLastError(); // clear the sticky allocation / launch error
            h->b_col31.release();
            set_last_error(std::string("flagged column words not built, column_index + tile_desc in use: ") + hipGetErrorString(e));
            h->drop_graphs();
            return CSR5HIP_SUCCESS;
        }
        h->col31_built = true;
    }
    h->d.col31 = (const uint32_t *)h->b_col31.ptr;
    h->opt.col31 = 1;
    h->drop_graphs();
    return CSR5HIP_SUCCESS;
}

// what the fused kernel needs on top of the reference's format arrays: carry meta, x windows, tile headers
static int derive_kernel_tables(csr5hip_handle h)
{
    // Deferred carries (CSR5HIP_OPT_DEFER_CARRIES), decided BEFORE the carry meta is written: forced, or (auto) a fused-mode matrix
    // of many tiles.  There the cut rows cost the one-tile kernel twice: a returning atomic at the end of a tile for every
    // hand-shake, and -- for the short-spill ownership that avoids most hand-shakes -- two scattered loads and a gather of the
    // NEXT tile's first 64 elements in every tile (sigma cache lines each: + 2/3 of a sigma = 24 fp32 tile's own lines).  With
    // the flag set no tile finishes its neighbour's spill, every meeting of partials is parked with plain stores and
    // k_calibrate (one thread per tile, ~3 us) adds them: nd24k-like fp32 cold 50.5 -> 37.5 us.  Small matrices keep the
    // in-launch protocol: a second launch costs them more than it saves (rule: csr5_internal.h DEFER_AUTO_*).
    {
        const Geometry &g0 = h->g;
        const bool long_rows = (long long)g0.nnz >= (long long)DEFER_AUTO_LONG_ROW * (g0.m > 0 ? g0.m : 1);
        const bool pays = long_rows ? g0.p - 1 >= DEFER_AUTO_MIN_TILES_LONG
                                    : (long long)(g0.p - 1) * g0.sigma >= DEFER_AUTO_MIN_TILE_SIGMA;
        h->g.defer = !h->is_child && h->opt.mode == 1 && g0.p > 1 && (h->defer_request == 2 || (h->defer_request == 1 && pays)) ? 1 : 0;
    }
    const Geometry &g = h->g;
    hipStream_t s = h->stream;
    if (!h->host_words) {
        HIP_TRY(hipHostMalloc((void **)&h->host_words, 128, hipHostMallocDefault));
        memset(h->host_words, 0, 128);
        int dev = 0, khz = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0)
            h->wall_clock_khz = khz;
        else
            h->wall_clock_khz = 100000.0; // gfx9: 100 MHz
    }
    // carry_meta + x-windows + fused-kernel headers in one launch, then the export of the host's words
    HIP_TRY(launch_tile_tables(g, h->d, (int)h->vsize(), h->host_words, h->is_child && h->hot_enabled, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint32_t *w = h->host_words;
    h->scalar_words[0] = w[0];
    h->scalar_words[1] = w[1];
    finish_format_scalars(h);
    h->xwin_tiles = (int)w[2];
    h->xwin_covered = (long long)w[3];
    h->opt.long_runs = w[4] != 0;
    h->xwin_lines = (long long)w[5];
    if (g.defer)
        h->opt.long_runs = 1; // (the parties of every cut row park: k_calibrate is part of each SpMV)
    // phase times from the kernels' wall-clock stamps (k_row_scan, k_tile_desc, k_transpose, k_tile_tables); a phase
    // whose kernel did not run (single-tile matrices) has no stamp and takes the next one's
    unsigned long long st[4];
    memcpy(st, w + 8, sizeof(st));
    for (int i = 2; i >= 0; i--)
        if (!st[i])
            st[i] = st[i + 1];
    const double per_tick_ms = 1.0 / h->wall_clock_khz;
    h->t_tile_ptr += (double)(st[1] - st[0]) * per_tick_ms;
    h->t_tile_desc += (double)(st[2] - st[1]) * per_tick_ms;
    h->t_transpose += (double)(st[3] - st[2]) * per_tick_ms;
    return CSR5HIP_SUCCESS;
}

// steps 1-2 of the conversion: everything that is derived from row_ptr alone (tile_ptr, tile_desc, offset_ptr,
// offset).  Stream-ordered, no host round trip (see reserve_aux).
static int build_format_arrays(csr5hip_handle h)
{
    Geometry &g = h->g;
    hipStream_t s = h->stream;
    // step 1: tile_ptr (+ empty-row marks) -- flag scatter shares the row pass
    HIP_TRY(launch_row_scan(g, h->d, s));
    // A hot slab child needs nothing else: its bit flags ride in the packed column codes, the range kernel recomputes
    // y_offset from them, and it has no empty rows (no offsets).  (k_tile_desc on the 40 M-row child of R-MAT 24: 0.4 ms.)
    if (h->is_child && h->hot_enabled)
        return CSR5HIP_SUCCESS;

    // step 2: tile_desc, offset_ptr scan, empty-row offsets (the kernel leaves unflagged tiles at once)
    HIP_TRY(launch_tile_desc(g, h->d, s));
    HIP_TRY(launch_offset_scan(g, h->d, h->scan_tmp, h->scan_tmp_bytes, s));
    HIP_TRY(launch_desc_offset(g, h->d, s));
    return CSR5HIP_SUCCESS;
}

static void resolve_variants(csr5hip_handle h)
{
    h->opt.x_window = xwin_decision(h);
    h->opt.lds_y = ldsy_decision(h);
    h->opt.stream_nt = nt_decision(h);
    h->opt.hot = h->hot_enabled ? 1 : 0;
}

int csr5hip_as_csr5(csr5hip_handle h)
{
    if (!h)
        return CSR5HIP_INVALID_ARGUMENT;
    if (h->format == CSR5HIP_FORMAT_CSR5)
        return CSR5HIP_SUCCESS;
    if (h->format != CSR5HIP_FORMAT_CSR)
        return CSR5HIP_UNKOWN_FORMAT;
    if (h->sigma_request < CSR5HIP_MIN_SIGMA)
        h->sigma_request = csr5hip_auto_sigma(h->g.m, h->g.nnz, h->value_type);

    int rc = derive_geometry(h, h->sigma_request);
    if (rc != CSR5HIP_SUCCESS)
        return rc;
    Geometry &g = h->g;
    hipStream_t s = h->stream;

    // ONE host round trip in all, at the end (the reference synchronises after every phase and reads two words in
    // between, anonymouslib_cuda.h:161-208).  The four phase times the reference prints are taken from wall-clock stamps of
    // the stream instead of host timers around synchronisations.
    double t0 = now_ms();
    rc = reserve_aux(h);
    if (rc != CSR5HIP_SUCCESS)
        return rc;
    h->t_malloc += now_ms() - t0;

    if (g.p > 0) {
        rc = build_format_arrays(h);
        if (rc != CSR5HIP_SUCCESS)
            return rc;
        // step 3: in-place tile transpose of column_index and value, then the kernel-side tables
        if (h->is_child && h->hot_enabled)
            HIP_TRY(launch_transpose_values(g, h->d, h->value_type, s)); // (the column codes are read in CSR order)
        else
            HIP_TRY(launch_transpose(g, h->d, h->value_type, true, s));
        // From here on the caller's arrays are in tile order while the handle still says CSR: a failure must
        // put them back (a retry would transpose them a second time).
        auto finish = [&]() -> int {
            return derive_kernel_tables(h); // ends with the one synchronisation; phase times from the kernels' stamps
        };
        rc = finish();
        if (rc != CSR5HIP_SUCCESS) {
            const std::string why = g_last_error;
            if (h->is_child && h->hot_enabled)
                h->format = -1; // only the VALUES of every tile were transposed (private arrays of a slab structure, which the
                                // parent releases on failure): nothing to restore, the handle is unusable until inputCSR
            else if (launch_transpose(g, h->d, h->value_type, false, s) != hipSuccess ||
                     hipStreamSynchronize(s) != hipSuccess)
                h->format = -1; // the arrays could not be restored: the handle is unusable until inputCSR
            g_last_error = why;
            return rc;
        }
    } else {
        HIP_TRY(hipStreamSynchronize(s));
    }
    resolve_variants(h);
    h->format = CSR5HIP_FORMAT_CSR5;
    rc = build_slabs(h); // (ends with prepare_plain when the plain path serves spmv())
    if (rc != CSR5HIP_SUCCESS) {
        // only when the structure was requested explicitly: a failed asCSR5 leaves CSR, as in the reference
        // (anonymouslib_cuda.h:105-220 returns before _format changes)
        const std::string why = g_last_error;
        if (csr5hip_as_csr(h) != CSR5HIP_SUCCESS)
            h->format = -1; // the arrays could not be restored: unusable until inputCSR
        g_last_error = why;
    }
    return rc;
}

// ---- column slabs (csr5_slab.hip) ------------------------------------------------------------------
static void release_slabs(csr5hip_handle h)
{
    if (h->slab_child) {
        csr5hip_free(h->slab_child);
        h->slab_child = nullptr;
    }
    h->values_narrowed = false;
    for (Buffer *b : {&h->b_row_ptr2, &h->b_col2, &h->b_val2, &h->b_val32, &h->b_P, &h->b_rowidx, &h->b_base, &h->b_nonempty, &h->b_col_lo, &h->b_col_hi, &h->b_hot_cols,
                      &h->b_hot_count, &h->b_hot_tile0, &h->b_slab_off, &h->b_lead, &h->b_range_head, &h->b_slab_tmp, &h->b_cold_base, &h->b_cold_cols,
                      &h->b_xperm})
        b->release();
    h->slab_S = 0;
    h->slab_m2 = 0;
    h->hot_cover_pct = 0;
}
// asCSR: the slab structure goes out of use but keeps its memory and its child handle, like the arena (capacity only
// grows until csr5hip_free) -- a reconversion then allocates nothing.  Ten hipMalloc/hipFree pairs were two thirds of
// the 1.9-ms conversion of a 3 M-nnz matrix.
static void deactivate_slabs(csr5hip_handle h)
{
    if (h->slab_child) {
        h->slab_child->drop_graphs();
        h->slab_child->format = -1;
        h->slab_child->hot_enabled = false;
    }
    h->slab_S = 0;
    h->slab_m2 = 0;
    h->hot_cover_pct = 0;
}

// Number of slabs spmv() should use (0 = none).  Auto rule (measured on MI355X, scripts/experiments/slab_*):
// the structure pays when x is larger than one XCD's 4-MB L2 AND the columns are scattered (the per-tile x windows
// found at conversion cover less than half of the non-zeros); a banded matrix keeps its x lines in L2 anyway and
// would only pay for the partial sums.
static int slab_count_for(const csr5hip_handle_s *h)
{
    if (h->is_child || h->slab_request == 0 || h->g.p < 2 || h->g.nnz <= 0 || h->g.m <= 0)
        return 0;
    if (h->slab_request >= 2)
        return h->slab_request;
    const long long xbytes = (long long)h->g.n * (long long)h->vsize();
    if (xbytes < 4LL * 1024 * 1024 || h->g.p < 2 * 256)
        return 0;
    const long long covered_pct = h->xwin_covered * 100 / ((long long)(h->g.p - 1) * h->g.tile_elems);
    if (covered_pct >= 50)
        return 0;
    // one slab per XCD while a slab's share of x stays within a few L2 sizes; two per XCD beyond (measured with the hot
    // table on R-MAT 20 / 22 / 24, x = 8 / 34 / 134 MB: 8 slabs 73 / 329 / 1564 us, 16 slabs 85 / 357 / 1505 us, 32 slabs
    // 119 / 441 / 1674 us: every further round costs a table refill and a workgroup barrier)
    // (beyond R-MAT 24 -- scale 25 / 26, x = 268 / 537 MB -- four slabs per XCD win again: 3 011 vs 3 113 us, 6 691 vs 6 970 us)
    int S = xbytes < 64LL * 1024 * 1024 ? 8 : (xbytes < 256LL * 1024 * 1024 ? 16 : 32);
    // ... as long as every wavefront of the persistent kernel still gets a dozen tiles per slab (256 ranges per slab): below
    // that the pipeline restart and the table refill of every slab cost more than the coverage buys.  One of eight row
    // blocks of R-MAT 24 (25-37 M non-zeros against the full 134-MB x; 12-18 tiles per wavefront and slab at 16 slabs) is
    // faster with 16 slabs once its tables are full -- 165 / 180 / 177 us for blocks 0 / 3 / 7 against 176 / 186 / 169 with
    // 8 (scripts/experiments/shard_alone.py, profiles/r04_shards.txt).
    const long long child_tiles = (long long)h->g.nnz / ((long long)OMEGA * hot_child_sigma((int)h->vsize()));
    while (S > NUM_XCD && child_tiles / ((long long)S * HOT_RANGES_PER_SLAB) < 12)
        S /= 2;
    return S;
}

// Slab count when the hot table is NOT used (auto mode): without the table nothing ties the count to the 8 XCDs, and
// fewer slabs mean fewer segments -- fewer partial sums to store and to combine -- as long as a slab's share of x stays
// around half an XCD's L2 (2 MiB).  webbase-like (x = 8 MB, no popular columns): 4 slabs 24.4-24.9 us, 8 slabs
// 25.4-26.1 us in the same GPU call, for every sigma.
static int slab_count_without_table(const csr5hip_handle_s *h)
{
    const long long xbytes = (long long)h->g.n * (long long)h->vsize();
    int S = 2;
    while (S < 16 && (long long)S * (2LL << 20) < xbytes)
        S *= 2;
    return S;
}

// The column-slab structure is an optional accelerator: when it cannot be built (allocation failure, memory cap) the
// handle stays a valid CSR5 matrix on the plain tile kernel.  Returns SUCCESS in that case -- csr5hip_info.slab_fallback
// and csr5hip_last_error() say what happened -- unless the caller REQUESTED the structure (slab_request >= 2): then the
// error code is returned (the matrix is still usable on the plain path; csr5hip_as_csr5 additionally rolls back to CSR).
static int build_slabs(csr5hip_handle h)
{
    h->slab_fallback = false;
    const int rc = build_slabs_impl(h);
    if (rc == CSR5HIP_SUCCESS) // (the plain path serves spmv() when no structure is active: its narrow column codes, if of use)
        return h->slab_S > 0 || h->is_child ? rc : prepare_plain(h);
    const std::string why = g_last_error;
    (void)hipGetLastError(); // clear a sticky allocation error
    release_slabs(h);
    h->slab_fallback = true;
    h->drop_graphs();
    g_last_error = "column slabs not built, plain tile kernel in use: " + why;
    if (h->slab_request >= 2)
        return rc;
    return prepare_plain(h);
}

static int build_slabs_impl(csr5hip_handle h)
{
    deactivate_slabs(h);
    h->t_slab = 0;
    int S = slab_count_for(h);
    if (!S) {
        release_slabs(h); // not wanted (any more): give the memory back
        return CSR5HIP_SUCCESS;
    }
    const double t0 = now_ms();
    const Geometry &g = h->g;
    hipStream_t s = h->stream;
    const bool auto_count = h->slab_request == 1;
    int S_plain = auto_count ? slab_count_without_table(h) : S; // the count if the table is not used

    // LDS hot table for the persistent kernel (fused mode, S a multiple of the 8 XCDs)?
    const int hot_sigma = hot_child_sigma((int)h->vsize());
    const int hot_T = OMEGA * hot_sigma;
    const int hot_p = (int)(((long long)g.nnz + hot_T - 1) / hot_T);
    // (a slab-local column id -- and with it a cold rank -- must fit the 22 bits a packed column code has for it)
    int bits_s = 0;
    while ((1 << bits_s) < S)
        bits_s++;
    bool hot = h->hot_request != 0 && S % NUM_XCD == 0 && h->opt.mode == 1 && hot_sigma >= 4 && hot_p >= 2 &&
               (long long)g.n * (long long)h->vsize() <= 0x7FFFFFFFLL &&
               slab_local_columns(g.n, bits_s, h->slab_shift) <= ((size_t)1 << 22);
    int hot_capacity = 0;
    if (hot) {
        int dev = 0, lds_max = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
        // the per-wavefront y-compaction regions sit behind the table (k_spmv_range)
        int lds = lds_max - HOT_WAVES * HOT_WAVE_LDS;
        lds = lds > HOT_LDS_BYTES ? HOT_LDS_BYTES : lds;
        hot_capacity = lds / (int)h->vsize();
        // a device (or runtime) that offers little LDS gets no table, forced or not: slot 0 is reserved and the
        // threshold search needs room above it
        if (hot_capacity < 1024) {
            hot = false;
            hot_capacity = 0;
        }
    }
    if (!hot && auto_count)
        S = S_plain; // table ruled out beforehand
    const int S_alloc = S > S_plain ? S : S_plain;

    // all temporaries of the build in one allocation.  Up to 1/64 of the device memory (4.5 GB of 288) they stay with
    // the handle between conversions: giving 2 GB back and asking for it again cost 120-170 ms per reconversion of
    // R-MAT 24 until the runtime's pool had settled, ten times the 16 ms the conversion itself takes.
    // (the device's total memory, asked once per handle: hipMemGetInfo walks the allocator -- a visible share of the 0.5-ms
    //  conversion of a 3 M-nnz matrix when it ran in every build)
    if (!h->device_total_bytes) {
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        h->device_total_bytes = total_b ? total_b : 1;
    }
    const size_t total_b = h->device_total_bytes;
    const size_t SLAB_TMP_KEEP = std::max((size_t)64 << 20, total_b / 64);
    size_t scan_bytes = 0, sel_bytes = 0;
    HIP_TRY(slab_scan_tmp_bytes((size_t)S_alloc * g.p, &scan_bytes));
    HIP_TRY(slab_select_tmp_bytes(g.nnz, &sel_bytes));
    const size_t nb = (size_t)(g.n > 0 ? g.n : 1) * 4, hb = (size_t)S_alloc * slab_hot_buckets() * 4;
    int bits_alloc = 0;
    while ((1 << bits_alloc) < S_alloc)
        bits_alloc++;
    const size_t hotmap_bytes = slab_hotmap_bytes(g.n, S_alloc, bits_alloc, h->slab_shift);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    const size_t o_hist = take((size_t)S_alloc * g.p * 4), o_scan = take(scan_bytes), o_key = take((size_t)g.nnz * 4),
                 o_count = take(16), o_sel = take(sel_bytes), o_cnt = take(nb), o_hotmap = take(hotmap_bytes), o_chist = take(hb),
                 o_thr = take((size_t)S_alloc * 8);
    // ranking of the cold columns behind the packed codes (slab_hot_pack): counts / ranks, sort keys in and out, sources
    size_t cold_words = 0, cold_sort_bytes = 0;
    if (hot) { // (S keeps this value while the table stays in use: bits_s = log2 S)
        cold_words = slab_cold_words(g.n, S, bits_s, h->slab_shift);
        HIP_TRY(slab_cold_sort_tmp_bytes(cold_words, bits_s, &cold_sort_bytes));
    }
    const size_t o_ref = take(cold_words), o_rank = take(cold_words * 4), o_keys = take(cold_words * 4), o_keys2 = take(cold_words * 4),
                 o_src = take(cold_words * 4), o_sort = take(cold_sort_bytes);
    if (h->slab_mem_mib > 0) {
        // Upper bound of what the structure holds: the second copy of column_index / value (+ 3-byte column codes with a
        // hot table), the build temporaries, one partial sum / row byte / child row pointer per (row, slab) segment --
        // at most min(nnz, m * S) of them --, the combine's tables, the child's own conversion arena (descriptor word per
        // 64 sigma elements, tile_ptr, carry state: ~300 B per child tile) and, with a hot table, the table columns, the
        // wavefront-range words and the permuted copy of x with its column lists.
        const unsigned long long seg_max = std::min<unsigned long long>((unsigned long long)g.nnz, (unsigned long long)g.m * S_alloc);
        const unsigned long long child_tiles = (unsigned long long)g.nnz / ((unsigned long long)OMEGA * (hot ? hot_sigma : g.sigma)) + 2;
        unsigned long long need = (unsigned long long)g.nnz * (4 + h->vsize() + (hot ? 3 : 0)) + off + seg_max * (h->vsize() + 5) +
                                  slab_base_words(g.m, S_alloc) * 4ull + (unsigned long long)g.m / 8 + child_tiles * 300ull;
        if (hot && h->narrow_request && h->value_type == CSR5HIP_F64)
            need += (unsigned long long)g.nnz * 4ull; // the fp32 copy of the child's values
        if (hot)
            need += (unsigned long long)S * hot_capacity * (4 + h->vsize()) + (unsigned long long)S * HOT_RANGES_PER_SLAB * (h->vsize() + 4) +
                    std::min<unsigned long long>(cold_words, (unsigned long long)g.nnz) * (4 + h->vsize());
        if (need > (unsigned long long)h->slab_mem_mib << 20)
            return fail_hip(hipErrorOutOfMemory, "column slabs: CSR5HIP_OPT_SLAB_MEMORY_MIB");
    }
    HIP_TRY(h->b_slab_tmp.reserve(off));
    struct TmpGuard { // very big temporaries (4 B per non-zero) do not outlive the build
        Buffer &b;
        size_t keep;
        ~TmpGuard()
        {
            if (b.cap > keep)
                b.release();
        }
    } tmp_guard{h->b_slab_tmp, SLAB_TMP_KEEP};
    char *tb = (char *)h->b_slab_tmp.ptr;
    struct {
        void *hist, *scan_tmp, *key, *count, *sel_tmp;
    } t{tb + o_hist, tb + o_scan, tb + o_key, tb + o_count, tb + o_sel};
    struct {
        void *cnt, *hotmap, *chist, *thr, *covered;
    } ht{tb + o_cnt, tb + o_hotmap, tb + o_chist, tb + o_thr, tb + o_count + 8};

    // Hot columns FIRST: the selection needs only the column indices (the parent's array, any order), and its verdict
    // decides the slab count -- without a table fewer slabs are better, and the partition below then runs once.
    int stride = 1;
    if (hot) {
        const int bits_hot = bits_s;
        // Column use counts come from a sample of the non-zeros (one 64-element chunk in `stride`): ~4 M samples are
        // plenty to rank columns, and a full count serialises on the very columns it is looking for.
        stride = (int)(g.nnz / (4LL * 1024 * 1024));
        stride = stride < 1 ? 1 : (stride > 64 ? 64 : stride); // (R-MAT 24: 1/64 ranks as well as 1/32 -- also the cold regions, round 4: 1/16 and 1/8 leave the L2 misses at 30.8 M and cost 0.7 / 1.8 ms of conversion --, 1/128 costs 1.5 % of the SpMV)
        // A slot is staged by each of the 32 workgroups of the slab's XCD in every SpMV (a coalesced copy out of the
        // permuted x: about what two cold gathers cost), so a column must be used a few times per SpMV to earn one -- and
        // the sample must have seen it twice to say so.  16 uses: with 48 (round 3) a row block of R-MAT 24 -- its columns
        // are used an eighth as often -- left a quarter of its table empty (coverage 60 -> 72 %, 189 -> 180 us on the slowest
        // of the eight blocks, profiles/r04_shards.txt); the whole matrix fills its tables either way.
        int min_count = 16 / stride;
        min_count = min_count < 2 ? 2 : min_count;
        HIP_TRY(h->b_hot_cols.reserve((size_t)S * hot_capacity * 4));
        HIP_TRY(h->b_hot_count.reserve((size_t)S * 4));
        HIP_TRY(h->b_hot_tile0.reserve(((size_t)2 * S + 1) * 4)); // tile0[S + 1], then the slab order of the XCDs [S]
        HIP_TRY(h->b_slab_off.reserve(((size_t)S + 1) * 4));
        HIP_TRY(h->b_lead.reserve((size_t)S * HOT_RANGES_PER_SLAB * h->vsize())); // one leading partial per wavefront range
        HIP_TRY(h->b_range_head.reserve(((size_t)S * HOT_RANGES_PER_SLAB + 1) * 4));
        HIP_TRY(hipMemsetAsync(ht.cnt, 0, nb, s));
        HIP_TRY(hipMemsetAsync(ht.hotmap, 0, slab_hotmap_bytes(g.n, S, bits_hot, h->slab_shift), s));
        HIP_TRY(hipMemsetAsync(ht.chist, 0, (size_t)S * slab_hot_buckets() * 4, s));
        HIP_TRY(hipMemsetAsync(ht.covered, 0, 8, s));
        HIP_TRY(hipMemsetAsync(h->b_hot_cols.ptr, 0, (size_t)S * hot_capacity * 4, s));
        HIP_TRY(slab_hot_select(g.n, g.nnz, S, bits_hot, h->slab_shift, hot_capacity, min_count, stride,
                                (const int32_t *)h->d.col, (uint32_t *)ht.cnt, ht.hotmap, (uint32_t *)ht.chist,
                                (uint32_t *)ht.thr, (int32_t *)h->b_hot_cols.ptr, (int32_t *)h->b_hot_count.ptr,
                                (unsigned long long *)ht.covered, s));
        unsigned long long covered = 0;
        HIP_TRY(hipMemcpyAsync(&covered, ht.covered, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        h->hot_cover_pct = (int)(covered * (unsigned long long)stride * 100 / (unsigned long long)g.nnz); // estimate
        h->hot_cover_pct = h->hot_cover_pct > 100 ? 100 : h->hot_cover_pct;
        // Worth it when a good part of the gathers leaves the vector memory path AND the matrix is large enough to pay for the
        // persistent kernel's fixed costs (a table refill and a workgroup barrier per slab, the range seams, P + combine: a
        // 3 M-nnz matrix takes 44-49 us on this path whatever its columns).  Measured break-evens, cold (round 6,
        // scripts/experiments/round6/locality.py, profiles/r06_locality.md): R-MAT 19 (8.4 M nnz, 90 % covered) wins 13 %,
        // webbase-like x 8 with power-law columns (25 M, 54 %) wins 9 %, the same x 4 (12 M, 50 %) loses 11 %, x 2 loses 33 %, x 1
        // loses 60 %: the table pays from about 5 M non-zeros covered BEYOND the 25 % floor.
        const long long beyond = (long long)g.nnz / 100 * (h->hot_cover_pct - HOT_AUTO_MIN_COVER_PCT);
        hot = h->hot_request == 2 || (h->hot_cover_pct >= HOT_AUTO_MIN_COVER_PCT && beyond >= HOT_AUTO_MIN_COVERED_BEYOND);
        if (!hot && auto_count) {
            if (h->hot_cover_pct >= HOT_AUTO_MIN_COVER_PCT) {
                // Skewed columns on a matrix too small for the table: the popular part of x stays in every XCD's L2 by itself and
                // the plain kernel beats slabs without a table at every size measured (28.0 / 51.3 / 102 us against 32.2 / 56.2 /
                // 110 on webbase-like x 1 / 2 / 4 with power-law columns): no structure at all.
                // (nothing is active -- deactivate_slabs above -- and the selection's buffers stay with the handle like every other
                //  buffer of the structure: freeing and re-allocating them was 0.3 of this path's 0.49 ms per reconversion of a
                //  3 M-nnz matrix; csr5hip_info.slab_hot_cover_pct keeps the estimate that explains the choice)
                h->t_slab = now_ms() - t0; // (the column sample that decided it: part of the conversion's time)
                return CSR5HIP_SUCCESS;
            }
            S = S_plain; // the table was the reason for that slab count
        }
    }
    int bits = 0;
    while ((1 << bits) < S)
        bits++;

    HIP_TRY(h->b_col2.reserve((size_t)g.nnz * 4));
    HIP_TRY(h->b_val2.reserve((size_t)g.nnz * h->vsize()));
    HIP_TRY(slab_partition(g, h->d, h->value_type, S, bits, h->slab_shift, (uint32_t *)t.hist, t.scan_tmp, scan_bytes,
                           (int32_t *)h->b_col2.ptr, h->b_val2.ptr, (uint32_t *)t.key, s));
    HIP_TRY(slab_count_segments(g.nnz, (const uint32_t *)t.key, t.sel_tmp, (unsigned int *)t.count, s));
    unsigned int m2 = 0;
    HIP_TRY(hipMemcpyAsync(&m2, t.count, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(h->b_row_ptr2.reserve(((size_t)m2 + 1) * 4));
    HIP_TRY(h->b_rowidx.reserve((size_t)m2 + 1));
    HIP_TRY(slab_segments(g.nnz, (const uint32_t *)t.key, t.sel_tmp, (int32_t *)h->b_row_ptr2.ptr, (unsigned char *)h->b_rowidx.ptr, s));
    HIP_TRY(h->b_base.reserve(slab_base_words(g.m, S) * 4));
    HIP_TRY(h->b_nonempty.reserve(((size_t)g.m / 32 + 16) * 4));
    HIP_TRY(slab_tables(g.m, (int)m2, g.nnz, S, g.p, h->d.row_ptr, (int32_t *)h->b_row_ptr2.ptr, (const uint32_t *)t.key,
                        (const uint32_t *)t.hist, (uint32_t *)h->b_base.ptr, (uint32_t *)h->b_nonempty.ptr, s));
    HIP_TRY(h->b_P.reserve(((size_t)m2 + 1) * h->vsize()));
    HIP_TRY(hipMemsetAsync(h->b_P.ptr, 0, ((size_t)m2 + 1) * h->vsize(), s));

    if (hot) {
        HIP_TRY(slab_hot_finish(S, g.p, hot_p, hot_T, g.nnz, hot_capacity, (const uint32_t *)t.hist,
                                (int32_t *)h->b_hot_count.ptr, (int32_t *)h->b_hot_tile0.ptr, (int32_t *)h->b_slab_off.ptr, s));
        std::vector<int32_t> first_tile((size_t)S + 1);
        HIP_TRY(hipMemcpyAsync(first_tile.data(), h->b_hot_tile0.ptr, ((size_t)S + 1) * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        // Which slabs an XCD walks.  Slabs differ in size (hub columns: the 16 slabs of R-MAT 24 span 0.86 .. 1.16 of the
        // mean, and two consecutive ones still 1.12) and the kernel ends with its slowest XCD, so the slabs are dealt
        // longest first onto the XCD with the least work that still has a round free.
        const int rounds = S / NUM_XCD;
        std::vector<int> by_size((size_t)S), order((size_t)S, 0), used(NUM_XCD, 0);
        std::vector<long long> load(NUM_XCD, 0);
        for (int k = 0; k < S; k++)
            by_size[k] = k;
        std::stable_sort(by_size.begin(), by_size.end(), [&](int a, int b) {
            return first_tile[a + 1] - first_tile[a] > first_tile[b + 1] - first_tile[b];
        });
        for (int k : by_size) {
            int best = -1;
            for (int x = 0; x < NUM_XCD; x++)
                if (used[x] < rounds && (best < 0 || load[x] < load[best]))
                    best = x;
            order[(size_t)best * rounds + used[best]++] = k;
            load[best] += first_tile[k + 1] - first_tile[k];
        }
        HIP_TRY(hipMemcpyAsync((int32_t *)h->b_hot_tile0.ptr + S + 1, order.data(), (size_t)S * 4, hipMemcpyHostToDevice, s));
        // 3-byte column codes next to the plain words (the child's sigma is a multiple of four: whole dwords per lane)
        int32_t cold_total = 0;
        {
            HIP_TRY(h->b_col_lo.reserve((size_t)g.nnz * 2 + 64));
            HIP_TRY(h->b_col_hi.reserve((size_t)g.nnz + 64));
            // the cold region holds at most one entry per column of every slab, and no more than there are non-zeros
            const size_t cold_cap = std::min(cold_words, (size_t)g.nnz);
            HIP_TRY(h->b_cold_base.reserve(((size_t)S + 1) * 4));
            HIP_TRY(h->b_cold_cols.reserve((cold_cap + 1) * 4));
            HIP_TRY(h->b_xperm.reserve(((size_t)S * hot_capacity + cold_cap + (size_t)hot_T + 1) * h->vsize())); // (+ the x entries of the child's CSR tail)
            HIP_TRY(hipMemsetAsync(tb + o_ref, 0, cold_words, s));
            HIP_TRY(slab_hot_pack(g.n, g.nnz, hot_T, hot_p, S, bits, h->slab_shift, (const int32_t *)h->b_slab_off.ptr, ht.hotmap,
                                  (const uint32_t *)ht.cnt, (const uint32_t *)t.key, (int32_t *)h->b_col2.ptr, (uint16_t *)h->b_col_lo.ptr,
                                  (uint8_t *)h->b_col_hi.ptr, (uint8_t *)(tb + o_ref), (uint32_t *)(tb + o_rank),
                                  (uint32_t *)(tb + o_keys), (uint32_t *)(tb + o_keys2),
                                  (uint32_t *)(tb + o_src), tb + o_sort, cold_sort_bytes, (int32_t *)h->b_cold_base.ptr,
                                  (int32_t *)h->b_cold_cols.ptr, s));
            HIP_TRY(hipMemcpyAsync(&cold_total, (int32_t *)h->b_cold_base.ptr + S, 4, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipStreamSynchronize(s));
        h->cold_total = cold_total;
    }
    HIP_TRY(hipStreamSynchronize(s)); // (`order` and the temporaries are in use until here)

    // the stacked matrix: an ordinary CSR matrix with m2 rows, converted and multiplied by the ordinary kernels
    csr5hip_handle c = h->slab_child ? h->slab_child : new csr5hip_handle_s(); // (kept across asCSR/asCSR5 cycles)
    h->slab_child = c;
    c->is_child = true;
    c->g.m = (int)m2;
    c->g.n = g.n;
    c->value_type = h->value_type;
    c->stream = s;
    c->opt.mode = h->opt.mode;
    c->opt.xcd_remap = 1;
    c->xwin_request = 0;
    c->slab_request = 0;
    c->ldsy_request = h->ldsy_request;
    c->nt_request = h->nt_request;
    c->hot_enabled = hot;
    c->d.col_lo = hot ? (const uint16_t *)h->b_col_lo.ptr : nullptr;
    c->d.col_hi = hot ? (const uint8_t *)h->b_col_hi.ptr : nullptr;
    c->d.slab_off = (const int32_t *)h->b_slab_off.ptr;
    c->d.slab_shift = h->slab_shift;
    c->d.slab_bits = bits;
    c->d.xperm = hot ? h->b_xperm.ptr : nullptr;
    c->d.cold_base = hot ? (const int32_t *)h->b_cold_base.ptr : nullptr;
    c->d.cold_cols = hot ? (const int32_t *)h->b_cold_cols.ptr : nullptr;
    c->d.cold_total = hot ? h->cold_total : 0;
    h->xperm_valid = false;
    c->d.hot_cols = (const int32_t *)h->b_hot_cols.ptr;
    c->d.hot_count = (const int32_t *)h->b_hot_count.ptr;
    c->d.hot_tile0 = (const int32_t *)h->b_hot_tile0.ptr;
    c->d.hot_slabs = S;
    c->d.hot_capacity = hot_capacity;
    c->d.range_lead = h->b_lead.ptr;
    c->d.range_head = (uint32_t *)h->b_range_head.ptr;
    int rc = csr5hip_input_csr(c, g.nnz, (int32_t *)h->b_row_ptr2.ptr, (int32_t *)h->b_col2.ptr, h->b_val2.ptr);
    c->sigma_request = hot ? hot_sigma : g.sigma;
    if (rc == CSR5HIP_SUCCESS)
        rc = csr5hip_as_csr5(c);
    if (rc == CSR5HIP_SUCCESS && hot) {
        // which row every wavefront range of the persistent kernel starts in (k_range_finish adds the seams)
        hipError_t e = launch_range_heads(c->g, c->d, s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);
        if (e != hipSuccess)
            rc = fail_hip(e, "k_range_heads");
    }
    h->values_narrowed = false;
    c->d.val32 = nullptr;
    if (rc == CSR5HIP_SUCCESS && hot && h->narrow_request && h->value_type == CSR5HIP_F64 && g.nnz > 0) {
        // CSR5HIP_OPT_NARROW_VALUES: when every value is exactly representable in fp32 the range kernel streams them as fp32
        // (converted back in registers: the same products, bit for bit); the fp64 array stays for the CSR tail and asCSR
        unsigned *flag = c->d.counters + 5;
        unsigned inexact = 1;
        hipError_t e = launch_fp32_exact((const double *)c->d.val, (size_t)g.nnz, flag, s);
        if (e == hipSuccess)
            e = hipMemcpyAsync(&inexact, flag, sizeof(inexact), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);
        if (e == hipSuccess && !inexact) {
            e = h->b_val32.reserve((size_t)g.nnz * sizeof(float));
            if (e == hipSuccess)
                e = launch_narrow((const double *)c->d.val, (size_t)g.nnz, (float *)h->b_val32.ptr, c->g.tile_elems, c->g.p - 1, s);
            if (e == hipSuccess)
                e = hipStreamSynchronize(s);
            if (e == hipSuccess) {
                c->d.val32 = (const float *)h->b_val32.ptr;
                h->values_narrowed = true;
            }
        }
        if (e != hipSuccess)
            rc = fail_hip(e, "narrow values");
    }
    if (rc != CSR5HIP_SUCCESS) {
        release_slabs(h);
        return rc;
    }
    if (c->hot_enabled) { // the range kernel this child launches needs the CU's whole LDS: raised here, once, not per SpMV
        const hipError_t e = prepare_spmv_hot(c->g, c->d, c->value_type, c->opt);
        if (e != hipSuccess) {
            rc = fail_hip(e, "LDS limit of the range kernel");
            release_slabs(h);
            return rc;
        }
    }
    h->slab_S = S;
    h->slab_m2 = (int)m2;
    h->t_slab = now_ms() - t0;
    return CSR5HIP_SUCCESS;
}

static bool stream_is_capturing(hipStream_t s)
{
    if (s == nullptr)
        return false;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) {
        (void)hipGetLastError(); // (the query itself failed: treat the stream as not capturing)
        return false;
    }
    return cap != hipStreamCaptureStatusNone;
}

// CSR5HIP_OPT_X_SNAPSHOT: the permuted copy of x is taken once per setX -- here, on stream s, in front of the first SpMV
// (or graph capture) that needs it
static hipError_t ensure_x_snapshot(csr5hip_handle h, hipStream_t s)
{
    if (!h->x_snapshot || h->xperm_valid || h->slab_S <= 0 || !h->slab_child->hot_enabled)
        return hipSuccess;
    // a caller capturing its own graph gets the copy recorded with every spmv() (enqueue_spmv)
    if (stream_is_capturing(s))
        return hipSuccess;
    hipError_t e = launch_x_permute(h->slab_child->g, h->slab_child->d, h->value_type, h->x, s);
    if (e == hipSuccess)
        h->xperm_valid = true;
    return e;
}

// one SpMV on stream s: the tile kernel on the matrix itself, or -- with column slabs -- on the stacked matrix
// followed by the combine kernel.  own_graph: s is the private capture stream of spmv_repeat / spmv_rotate, whose graphs
// set_x drops; otherwise s is the caller's stream, which may itself be capturing.
static hipError_t enqueue_spmv(csr5hip_handle h, void *d_y, hipStream_t s, bool own_graph)
{
    if (h->slab_S > 0) {
        csr5hip_handle c = h->slab_child;
        hipError_t e = hipSuccess;
        // CSR5HIP_OPT_X_SNAPSHOT: the copy taken at the last setX may be reused -- except inside a graph the CALLER is
        // capturing: that graph outlives this setX (the handle cannot drop it), so it must carry the copy itself; a replay
        // after the caller rewrote x and called setX again would otherwise read the old copy
        const bool reuse = h->x_snapshot && h->xperm_valid && (own_graph || !stream_is_capturing(s));
        if (c->hot_enabled && !reuse) {
            // the packed codes index the permuted copy of x: taken by every spmv() (x is read live, as the reference reads
            // it), or -- CSR5HIP_OPT_X_SNAPSHOT -- once per setX
            e = launch_x_permute(c->g, c->d, c->value_type, h->x, s);
            if (e != hipSuccess)
                return e;
        }
        e = launch_spmv(c->g, c->d, c->value_type, h->x, h->b_P.ptr, c->opt, s);
        if (e != hipSuccess)
            return e;
        return launch_slab_combine(h->g.m, h->g.tail_start, h->zero_empty, h->slab_S, h->value_type,
                                   (const uint32_t *)h->b_base.ptr, (const unsigned char *)h->b_rowidx.ptr,
                                   (const uint32_t *)h->b_nonempty.ptr, h->b_P.ptr, h->slab_m2, d_y, s);
    }
    if (h->zero_empty && h->g.m > 0) {
        // every row that owns a non-zero is overwritten by the kernel; this defines the others
        hipError_t e = hipMemsetAsync(d_y, 0, (size_t)h->g.m * h->vsize(), s);
        if (e != hipSuccess)
            return e;
    }
    return launch_spmv(h->g, h->d, h->value_type, h->x, d_y, h->opt, s);
}

// ---- serialise / deserialise (checkpoint of the converted matrix) ----------
// File = header + row_ptr + column_index and value IN TILE ORDER + the four CSR5 arrays of the reference
// (anonymouslib_cuda.h:27-52).  Loading skips the conversion: only the kernel-side tables are re-derived.
namespace {
struct Csr5FileHeader {
    char magic[8];      // "CSR5HIP1"
    int32_t value_type, omega, sigma, m, n, nnz, p, num_packet, num_offsets, tail_start;
    int32_t reserved[4];
};
bool write_dev(FILE *f, const void *dptr, size_t bytes, std::vector<char> &stage)
{
    if (!bytes)
        return true;
    stage.resize(bytes);
    if (hipMemcpy(stage.data(), dptr, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return false;
    return fwrite(stage.data(), 1, bytes, f) == bytes;
}
bool read_dev(FILE *f, void *dptr, size_t bytes, std::vector<char> &stage)
{
    if (!bytes)
        return true;
    stage.resize(bytes);
    if (fread(stage.data(), 1, bytes, f) != bytes)
        return false;
    return hipMemcpy(dptr, stage.data(), bytes, hipMemcpyHostToDevice) == hipSuccess;
}
} // namespace

int csr5hip_save(csr5hip_handle h, const char *path)
{
    if (!h || !path)
        return CSR5HIP_INVALID_ARGUMENT;
    if (h->format != CSR5HIP_FORMAT_CSR5) {
        g_last_error = "csr5hip_save: the handle is not in CSR5 format";
        return CSR5HIP_UNKOWN_FORMAT;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    FILE *f = fopen(path, "wb");
    if (!f) {
        g_last_error = std::string("csr5hip_save: cannot open ") + path;
        return CSR5HIP_INVALID_ARGUMENT;
    }
    const Geometry &g = h->g;
    Csr5FileHeader hd;
    memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, "CSR5HIP1", 8);
    hd.value_type = h->value_type, hd.omega = OMEGA, hd.sigma = g.sigma, hd.m = g.m, hd.n = g.n, hd.nnz = g.nnz;
    hd.p = g.p, hd.num_packet = g.num_packet, hd.num_offsets = h->num_offsets, hd.tail_start = g.tail_start;
    std::vector<char> stage;
    bool ok = fwrite(&hd, sizeof hd, 1, f) == 1;
    ok = ok && write_dev(f, h->d.row_ptr, ((size_t)g.m + 1) * 4, stage);
    ok = ok && write_dev(f, h->d.col, (size_t)g.nnz * 4, stage);
    ok = ok && write_dev(f, h->d.val, (size_t)g.nnz * h->vsize(), stage);
    ok = ok && write_dev(f, h->d.tile_ptr, ((size_t)g.p + 1) * 4, stage);
    ok = ok && write_dev(f, h->d.tile_desc, (size_t)g.p * OMEGA * g.num_packet * 4, stage);
    ok = ok && write_dev(f, h->d.offset_ptr, ((size_t)g.p + 1) * 4, stage);
    ok = ok && write_dev(f, h->d.offset, (size_t)h->num_offsets * 4, stage);
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        g_last_error = std::string("csr5hip_save: write failed: ") + path;
        return CSR5HIP_HIP_ERROR;
    }
    return CSR5HIP_SUCCESS;
}

int csr5hip_load(const char *path, csr5hip_handle *out, csr5hip_csr *arrays)
{
    if (!path || !out || !arrays)
        return CSR5HIP_INVALID_ARGUMENT;
    *out = nullptr;
    memset(arrays, 0, sizeof *arrays);
    FILE *f = fopen(path, "rb");
    if (!f) {
        g_last_error = std::string("csr5hip_load: cannot open ") + path;
        return CSR5HIP_INVALID_ARGUMENT;
    }
    Csr5FileHeader hd;
    bool ok = fread(&hd, sizeof hd, 1, f) == 1 && memcmp(hd.magic, "CSR5HIP1", 8) == 0;
    ok = ok && hd.omega == OMEGA && (hd.value_type == CSR5HIP_F64 || hd.value_type == CSR5HIP_F32);
    ok = ok && hd.m >= 0 && hd.n >= 0 && hd.nnz >= 0 && hd.sigma >= CSR5HIP_MIN_SIGMA && hd.sigma <= CSR5HIP_MAX_SIGMA;
    ok = ok && hd.num_offsets >= 0 && hd.tail_start >= 0 && hd.tail_start <= hd.m;
    ok = ok && hd.p == (int)(((long long)hd.nnz + (long long)OMEGA * hd.sigma - 1) / ((long long)OMEGA * hd.sigma));
    ok = ok && hd.num_packet == num_packet_of(hd.sigma);
    if (!ok) {
        fclose(f);
        g_last_error = std::string("csr5hip_load: not a CSR5 checkpoint of this library: ") + path;
        return CSR5HIP_INVALID_ARGUMENT;
    }
    const size_t vs = hd.value_type == CSR5HIP_F64 ? 8 : 4;
    // the file must be exactly as long as its header implies (a truncated or padded file is rejected before any
    // nnz-sized allocation)
    {
        const long long expect = (long long)sizeof hd + ((long long)hd.m + 1) * 4 + (long long)hd.nnz * (4 + (long long)vs) +
                                 ((long long)hd.p + 1) * 8 + (long long)hd.p * OMEGA * hd.num_packet * 4 +
                                 (long long)hd.num_offsets * 4;
        long long have = -1;
        const long pos = ftell(f);
        if (pos >= 0 && fseek(f, 0, SEEK_END) == 0) {
            have = ftell(f);
            if (fseek(f, pos, SEEK_SET) != 0)
                have = -1;
        }
        if (have != expect) {
            fclose(f);
            g_last_error = std::string("csr5hip_load: file size does not match its header: ") + path;
            return CSR5HIP_INVALID_ARGUMENT;
        }
    }
    csr5hip_handle h = nullptr;
    int rc = csr5hip_create(&h, hd.m, hd.n, hd.value_type);
    auto fail_with = [&](int code, const char *msg) {
        if (msg)
            g_last_error = std::string("csr5hip_load: ") + msg;
        fclose(f);
        if (h)
            csr5hip_free(h);
        csr5hip_csr_release(arrays);
        return code;
    };
    if (rc != CSR5HIP_SUCCESS)
        return fail_with(rc, nullptr);
    arrays->m = hd.m, arrays->n = hd.n, arrays->nnz = hd.nnz, arrays->value_type = hd.value_type;
    if (hipMalloc(&arrays->d_row_ptr, ((size_t)hd.m + 1) * 4) != hipSuccess ||
        hipMalloc(&arrays->d_col_idx, (size_t)(hd.nnz ? hd.nnz : 1) * 4) != hipSuccess ||
        hipMalloc(&arrays->d_val, (size_t)(hd.nnz ? hd.nnz : 1) * vs) != hipSuccess)
        return fail_with(CSR5HIP_HIP_ERROR, "device allocation failed");
    std::vector<char> stage;
    ok = read_dev(f, arrays->d_row_ptr, ((size_t)hd.m + 1) * 4, stage);
    ok = ok && read_dev(f, arrays->d_col_idx, (size_t)hd.nnz * 4, stage);
    ok = ok && read_dev(f, arrays->d_val, (size_t)hd.nnz * vs, stage);
    if (!ok)
        return fail_with(CSR5HIP_INVALID_ARGUMENT, "truncated file");
    csr5hip_input_csr(h, hd.nnz, arrays->d_row_ptr, arrays->d_col_idx, arrays->d_val);
    h->sigma_request = hd.sigma;
    rc = derive_geometry(h, hd.sigma);
    if (rc == CSR5HIP_SUCCESS)
        rc = reserve_aux(h);
    if (rc != CSR5HIP_SUCCESS)
        return fail_with(rc, nullptr);
    // row_ptr and column_index are used as addresses by every kernel: check them on the device before anything
    // reads through them (row_ptr monotone from 0 to nnz, columns inside [0, n))
    {
        uint32_t bad = 0;
        hipError_t e = hipMemsetAsync(h->d.counters, 0, 16, h->stream);
        if (e == hipSuccess)
            e = launch_validate_csr(hd.m, hd.n, hd.nnz, arrays->d_row_ptr, arrays->d_col_idx, h->d.counters, h->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(&bad, h->d.counters, 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess)
            e = hipMemsetAsync(h->d.counters, 0, 16, h->stream);
        if (e != hipSuccess)
            return fail_with(fail_hip(e, "csr5hip_load: validation"), nullptr);
        if (bad)
            return fail_with(CSR5HIP_INVALID_ARGUMENT, bad & 1u ? "row_ptr is not a monotone pointer array ending at nnz"
                                                                 : "a column index lies outside [0, n)");
    }
    // The four format arrays are a pure function of row_ptr and sigma: they are RE-DERIVED here (cheap: no transpose)
    // and the file's copies only have to match -- nothing read from the file is ever used as an index unchecked.
    if (hd.p > 0) {
        rc = build_format_arrays(h);
        if (rc != CSR5HIP_SUCCESS)
            return fail_with(rc, nullptr);
        if (read_format_scalars(h, true) != hipSuccess)
            return fail_with(CSR5HIP_HIP_ERROR, "reading the conversion scalars failed");
        finish_format_scalars(h);
        if (h->g.tail_start != hd.tail_start || h->num_offsets != hd.num_offsets)
            return fail_with(CSR5HIP_INVALID_ARGUMENT, "format arrays do not belong to this row_ptr (tail start / offsets)");
    } else if (hd.num_offsets != 0) {
        return fail_with(CSR5HIP_INVALID_ARGUMENT, "format arrays do not belong to this row_ptr");
    }
    {
        std::vector<char> mine;
        auto same = [&](const void *dptr, size_t bytes) -> bool {
            if (!bytes)
                return true;
            stage.resize(bytes);
            mine.resize(bytes);
            if (fread(stage.data(), 1, bytes, f) != bytes)
                return false;
            if (hipMemcpy(mine.data(), dptr, bytes, hipMemcpyDeviceToHost) != hipSuccess)
                return false;
            return memcmp(stage.data(), mine.data(), bytes) == 0;
        };
        ok = same(h->d.tile_ptr, ((size_t)hd.p + 1) * 4);
        ok = ok && same(h->d.tile_desc, (size_t)hd.p * OMEGA * hd.num_packet * 4);
        ok = ok && same(h->d.offset_ptr, ((size_t)hd.p + 1) * 4);
        ok = ok && same(h->d.offset, (size_t)hd.num_offsets * 4);
        if (!ok)
            return fail_with(CSR5HIP_INVALID_ARGUMENT, "format arrays in the file do not match the ones derived from its row_ptr");
    }
    if (hd.p > 0) {
        rc = derive_kernel_tables(h);
        if (rc != CSR5HIP_SUCCESS)
            return fail_with(rc, nullptr);
    }
    fclose(f);
    resolve_variants(h);
    h->format = CSR5HIP_FORMAT_CSR5;
    rc = build_slabs(h); // (ends with prepare_plain when the plain path serves spmv())
    if (rc != CSR5HIP_SUCCESS) {
        csr5hip_free(h);
        csr5hip_csr_release(arrays);
        return rc;
    }
    *out = h;
    return CSR5HIP_SUCCESS;
}

int csr5hip_as_csr(csr5hip_handle h)
{
    if (!h)
        return CSR5HIP_INVALID_ARGUMENT;
    if (h->format == CSR5HIP_FORMAT_CSR)
        return CSR5HIP_SUCCESS;
    if (h->format != CSR5HIP_FORMAT_CSR5)
        return CSR5HIP_UNKOWN_FORMAT;
    h->drop_graphs();
    deactivate_slabs(h);
    HIP_TRY(launch_transpose(h->g, h->d, h->value_type, false, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    // the aux buffers stay cached in the handle (capacity only grows) until csr5hip_free
    h->format = CSR5HIP_FORMAT_CSR;
    return CSR5HIP_SUCCESS;
}

int csr5hip_destroy(csr5hip_handle h) { return csr5hip_as_csr(h); }

int csr5hip_spmv(csr5hip_handle h, double alpha, void *d_y)
{
    (void)alpha; // accepted, not applied: csr5_spmv_cuda.h:22 ("// * alpha")
    if (!h || !d_y)
        return CSR5HIP_INVALID_ARGUMENT;
    if (h->format == CSR5HIP_FORMAT_CSR)
        return CSR5HIP_UNSUPPORTED_CSR_SPMV;
    if (h->format != CSR5HIP_FORMAT_CSR5)
        return CSR5HIP_UNKOWN_FORMAT;
    if (!h->x)
        return CSR5HIP_INVALID_ARGUMENT;
    HIP_TRY(ensure_x_snapshot(h, h->stream));
    HIP_TRY(enqueue_spmv(h, d_y, h->stream, false));
    return CSR5HIP_SUCCESS;
}

int csr5hip_snapshot_x(csr5hip_handle h)
{
    if (!h)
        return CSR5HIP_INVALID_ARGUMENT;
    if (h->format != CSR5HIP_FORMAT_CSR5 || !h->x)
        return CSR5HIP_SUCCESS; // (nothing to copy yet: the first spmv() after asCSR5 / setX takes it)
    HIP_TRY(ensure_x_snapshot(h, h->stream));
    return CSR5HIP_SUCCESS;
}

int csr5hip_spmv_repeat(csr5hip_handle h, double alpha, void *d_y, int count)
{
    if (!h || !d_y || count < 0)
        return CSR5HIP_INVALID_ARGUMENT;
    if (h->format == CSR5HIP_FORMAT_CSR)
        return CSR5HIP_UNSUPPORTED_CSR_SPMV;
    if (h->format != CSR5HIP_FORMAT_CSR5)
        return CSR5HIP_UNKOWN_FORMAT;
    if (!h->x)
        return CSR5HIP_INVALID_ARGUMENT;
    if (count == 0)
        return CSR5HIP_SUCCESS;
    HIP_TRY(ensure_x_snapshot(h, h->stream));
    GraphKey key{d_y, count, h->opt.mode};
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
        // capture on a private stream so the caller's stream state is untouched
        hipStream_t cs = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        int rc = CSR5HIP_SUCCESS;
        if (e == hipSuccess) {
            for (int i = 0; i < count && e == hipSuccess; i++)
                e = enqueue_spmv(h, d_y, cs, true);
            hipError_t e2 = hipStreamEndCapture(cs, &graph);
            if (e == hipSuccess)
                e = e2;
        }
        hipGraphExec_t exec = nullptr;
        if (e == hipSuccess)
            e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph)
            (void)hipGraphDestroy(graph);
        (void)hipStreamDestroy(cs);
        if (e != hipSuccess)
            rc = fail_hip(e, "hipGraph capture of spmv");
        if (rc != CSR5HIP_SUCCESS)
            return rc;
        if (h->graphs.size() >= 8)
            h->drop_graphs();
        it = h->graphs.emplace(key, exec).first;
    }
    HIP_TRY(hipGraphLaunch(it->second, h->stream));
    (void)alpha;
    return CSR5HIP_SUCCESS;
}

int csr5hip_spmv_rotate(csr5hip_handle *hs, void **d_ys, int k, double alpha, int count)
{
    (void)alpha;
    if (!hs || !d_ys || k < 1 || count < 0)
        return CSR5HIP_INVALID_ARGUMENT;
    for (int i = 0; i < k; i++) {
        if (!hs[i] || !d_ys[i] || !hs[i]->x)
            return CSR5HIP_INVALID_ARGUMENT;
        if (hs[i]->format == CSR5HIP_FORMAT_CSR)
            return CSR5HIP_UNSUPPORTED_CSR_SPMV;
        if (hs[i]->format != CSR5HIP_FORMAT_CSR5)
            return CSR5HIP_UNKOWN_FORMAT;
    }
    if (count == 0)
        return CSR5HIP_SUCCESS;
    csr5hip_handle h0 = hs[0];
    for (int i = 0; i < k; i++) {
        // every handle's snapshot on its OWN stream (a later spmv(hs[i]) there is ordered behind it); the rotating graph,
        // launched on hs[0]'s stream, waits for the foreign ones
        const bool pending = hs[i]->x_snapshot && !hs[i]->xperm_valid;
        HIP_TRY(ensure_x_snapshot(hs[i], hs[i]->stream));
        if (pending && hs[i]->stream != h0->stream) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            hipError_t e = hipEventRecord(ev, hs[i]->stream);
            if (e == hipSuccess)
                e = hipStreamWaitEvent(h0->stream, ev, 0);
            (void)hipEventDestroy(ev);
            HIP_TRY(e);
        }
    }
    std::vector<void *> key;
    key.push_back((void *)(intptr_t)count);
    for (int i = 0; i < k; i++) {
        key.push_back(hs[i]);
        key.push_back(d_ys[i]);
        // the graph bakes in every handle's x, format arrays and slab buffers: any change of handle i (set_x, set_option,
        // asCSR / asCSR5, set_stream) bumps its generation and must invalidate the graph cached on hs[0]
        key.push_back((void *)(uintptr_t)hs[i]->generation);
    }
    if (!h0->rotate_exec || h0->rotate_key != key) {
        if (h0->rotate_exec)
            (void)hipGraphExecDestroy(h0->rotate_exec);
        h0->rotate_exec = nullptr;
        h0->rotate_key.clear();
        hipStream_t cs = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            for (int i = 0; i < count && e == hipSuccess; i++)
                e = enqueue_spmv(hs[i % k], d_ys[i % k], cs, true);
            hipError_t e2 = hipStreamEndCapture(cs, &graph);
            if (e == hipSuccess)
                e = e2;
        }
        hipGraphExec_t exec = nullptr;
        if (e == hipSuccess)
            e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph)
            (void)hipGraphDestroy(graph);
        (void)hipStreamDestroy(cs);
        if (e != hipSuccess)
            return fail_hip(e, "hipGraph capture of rotating spmv");
        h0->rotate_exec = exec;
        h0->rotate_key = key;
    }
    HIP_TRY(hipGraphLaunch(h0->rotate_exec, h0->stream));
    return CSR5HIP_SUCCESS;
}

// Measured sigma selection: what the reference's per-architecture
// (r, s, t, u) tables (anonymouslib_cuda.h:297-313, anonymouslib_opencl.h:341-357) approximate, done on
// the matrix itself.  For every candidate sigma: CSR -> CSR5, a few warm SpMVs, then `reps` SpMVs
// replayed from one hipGraph and timed with HIP events; the fastest sigma stays converted.
int csr5hip_autotune_sigma(csr5hip_handle h, void *d_y, int *best_sigma, double *best_us)
{
    if (!h || !d_y)
        return CSR5HIP_INVALID_ARGUMENT;
    if (h->format != CSR5HIP_FORMAT_CSR && h->format != CSR5HIP_FORMAT_CSR5)
        return CSR5HIP_UNKOWN_FORMAT;
    if (!h->x)
        return CSR5HIP_INVALID_ARGUMENT;
    int rc = csr5hip_as_csr(h);
    if (rc != CSR5HIP_SUCCESS)
        return rc;
    static const int candidates[] = {4, 5, 6, 8, 10, 12, 16, 20, 24, 32};
    int best = 0;
    // the slab child that served the last timed candidate: (slabs, hot table, child sigma); a later candidate that converts
    // to the SAME child -- the child's sigma is capped, so every larger parent sigma may -- would time the same kernel again
    int timed_S = -1, timed_hot = -1, timed_child_sigma = -1;
    double best_ms = 1e300;
    for (int sigma : candidates) {
        if ((long long)OMEGA * sigma > (long long)h->g.nnz && sigma != 4)
            continue; // fewer non-zeros than one tile: nothing to choose
        h->sigma_request = sigma;
        rc = csr5hip_as_csr5(h);
        if (rc != CSR5HIP_SUCCESS)
            return rc;
        // The slab decision itself depends on the parent's sigma (tile count, x-window coverage): only a candidate that
        // was CONVERTED and produced the child already timed is skipped.
        if (h->slab_S > 0 && h->slab_child && h->slab_child->hot_enabled) {
            const int cs = h->slab_child->g.sigma;
            if (h->slab_S == timed_S && timed_hot == 1 && cs == timed_child_sigma) {
                rc = csr5hip_as_csr(h);
                if (rc != CSR5HIP_SUCCESS)
                    return rc;
                continue;
            }
            timed_S = h->slab_S, timed_hot = 1, timed_child_sigma = cs;
        } else {
            timed_S = timed_hot = timed_child_sigma = -1;
        }
        for (int i = 0; i < 3 && rc == CSR5HIP_SUCCESS; i++)
            rc = csr5hip_spmv(h, 1.0, d_y);
        // size the timed batch to ~0.5 ms from one timed probe launch
        double probe_ms = 0;
        if (rc == CSR5HIP_SUCCESS) rc = csr5hip_timer_start(h);
        if (rc == CSR5HIP_SUCCESS) rc = csr5hip_spmv(h, 1.0, d_y);
        if (rc == CSR5HIP_SUCCESS) rc = csr5hip_timer_stop(h, &probe_ms);
        int reps = probe_ms > 0 ? (int)(0.5 / probe_ms) : 50;
        reps = reps < 5 ? 5 : (reps > 200 ? 200 : reps);
        double ms = 0;
        if (rc == CSR5HIP_SUCCESS) rc = csr5hip_spmv_repeat(h, 1.0, d_y, reps); // instantiate + warm
        if (rc == CSR5HIP_SUCCESS) rc = csr5hip_timer_start(h);
        if (rc == CSR5HIP_SUCCESS) rc = csr5hip_spmv_repeat(h, 1.0, d_y, reps);
        if (rc == CSR5HIP_SUCCESS) rc = csr5hip_timer_stop(h, &ms);
        if (rc != CSR5HIP_SUCCESS)
            return rc;
        ms /= reps;
        if (ms < best_ms) {
            best_ms = ms;
            best = sigma;
        }
        rc = csr5hip_as_csr(h);
        if (rc != CSR5HIP_SUCCESS)
            return rc;
    }
    if (!best)
        best = csr5hip_auto_sigma(h->g.m, h->g.nnz, h->value_type);
    h->sigma_request = best;
    rc = csr5hip_as_csr5(h);
    if (best_sigma) *best_sigma = best;
    if (best_us) *best_us = best_ms * 1e3;
    return rc;
}

int csr5hip_get_info(csr5hip_handle h, csr5hip_info *info)
{
    if (!h || !info)
        return CSR5HIP_INVALID_ARGUMENT;
    memset(info, 0, sizeof(*info));
    info->format = h->format;
    info->m = h->g.m;
    info->n = h->g.n;
    info->nnz = h->g.nnz;
    info->value_type = h->value_type;
    info->omega = OMEGA;
    info->sigma = h->format == CSR5HIP_FORMAT_CSR5 ? h->g.sigma : h->sigma_request;
    if (h->format == CSR5HIP_FORMAT_CSR5) {
        info->bit_y_offset = h->g.bit_y;
        info->bit_scansum_offset = BIT_SS;
        info->num_packet = h->g.num_packet;
        info->p = h->g.p;
        info->tail_partition_start = h->g.tail_start;
        info->num_offsets = h->num_offsets;
        info->d_tile_ptr = h->d.tile_ptr;
        info->d_tile_desc = h->d.tile_desc;
        info->d_offset_ptr = h->d.offset_ptr;
        info->d_offset = h->d.offset;
    }
    info->x_window_tiles = h->xwin_tiles;
    info->x_window_lines = h->xwin_tiles > 0 ? (int)(h->xwin_lines / h->xwin_tiles) : 0;
    info->x_window_cover_pct = h->g.p > 1 ? (int)(h->xwin_covered * 100 / ((long long)(h->g.p - 1) * h->g.tile_elems)) : 0;
    info->x_window_active = h->opt.x_window;
    info->t_malloc_ms = h->t_malloc;
    info->t_tile_ptr_ms = h->t_tile_ptr;
    info->t_tile_desc_ms = h->t_tile_desc;
    info->t_transpose_ms = h->t_transpose;
    info->column_slabs = h->slab_S;
    info->slab_shift = h->slab_shift;
    info->slab_segments = h->slab_m2;
    info->slab_sigma = h->slab_S > 0 ? h->slab_child->g.sigma : 0;
    info->slab_tiles = h->slab_S > 0 ? h->slab_child->g.p : 0;
    info->t_slab_ms = h->t_slab;
    info->slab_hot = h->slab_S > 0 && h->slab_child->hot_enabled ? 1 : 0;
    info->slab_hot_cover_pct = h->hot_cover_pct;
    info->slab_fallback = h->slab_fallback ? 1 : 0;
    info->slab_x_permuted = h->slab_S > 0 && h->slab_child->hot_enabled ? 1 : 0;
    info->slab_cold_entries = info->slab_x_permuted ? h->cold_total : 0;
    info->x_snapshot = h->x_snapshot;
    info->slab_values_narrowed = h->slab_S > 0 && h->values_narrowed ? 1 : 0;
    info->carries_deferred = h->format == CSR5HIP_FORMAT_CSR5 && h->slab_S <= 0 && h->g.defer ? 1 : 0;
    info->narrow_columns = h->format == CSR5HIP_FORMAT_CSR5 && h->slab_S <= 0 && h->opt.col16 && h->opt.x_window ? 1 : 0;
    info->flagged_columns = h->format == CSR5HIP_FORMAT_CSR5 && h->slab_S <= 0 && h->opt.col31 && !h->opt.x_window && h->opt.mode == 1 ? 1 : 0;
    long long bytes = (long long)h->b_arena.cap;
    for (const Buffer *b : {&h->b_row_ptr2, &h->b_col2, &h->b_val2, &h->b_val32, &h->b_P, &h->b_rowidx, &h->b_base, &h->b_nonempty,
                            &h->b_hot_cols, &h->b_hot_count, &h->b_hot_tile0, &h->b_slab_off, &h->b_lead, &h->b_range_head, &h->b_slab_tmp,
                            &h->b_col_lo, &h->b_col_hi, &h->b_cold_base, &h->b_cold_cols, &h->b_xperm, &h->b_col16, &h->b_col31})
        bytes += (long long)b->cap;
    if (h->slab_child)
        bytes += (long long)h->slab_child->b_arena.cap;
    info->device_bytes = bytes;
    return CSR5HIP_SUCCESS;
}

// ---- device shims ---------------------------------------------------------------------------
int csr5hip_device_count(int *count)
{
    if (!count)
        return CSR5HIP_INVALID_ARGUMENT;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return fail_hip(e, "hipGetDeviceCount");
    }
    return CSR5HIP_SUCCESS;
}

int csr5hip_set_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    return CSR5HIP_SUCCESS;
}

int csr5hip_device_name(int device, char *buf, size_t buflen, double *clock_mhz)
{
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (buf && buflen) {
        // the marketing name comes from libdrm's amdgpu.ids and is empty where that file is missing:
        // fall back to the architecture string (gfx950:...)
        strncpy(buf, prop.name[0] ? prop.name : prop.gcnArchName, buflen - 1);
        buf[buflen - 1] = 0;
    }
    if (clock_mhz)
        *clock_mhz = prop.clockRate * 1e-3;
    return CSR5HIP_SUCCESS;
}

int csr5hip_malloc(void **dptr, size_t bytes)
{
    if (!dptr)
        return CSR5HIP_INVALID_ARGUMENT;
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 4));
    return CSR5HIP_SUCCESS;
}

int csr5hip_device_free(void *dptr)
{
    HIP_TRY(hipFree(dptr));
    return CSR5HIP_SUCCESS;
}

int csr5hip_memcpy_h2d(void *dst, const void *src, size_t bytes)
{
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return CSR5HIP_SUCCESS;
}

int csr5hip_memcpy_d2h(void *dst, const void *src, size_t bytes)
{
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return CSR5HIP_SUCCESS;
}

int csr5hip_memset(void *dptr, int value, size_t bytes)
{
    HIP_TRY(hipMemset(dptr, value, bytes));
    return CSR5HIP_SUCCESS;
}

int csr5hip_synchronize(void)
{
    HIP_TRY(hipDeviceSynchronize());
    return CSR5HIP_SUCCESS;
}

int csr5hip_timer_start(csr5hip_handle h)
{
    if (!h)
        return CSR5HIP_INVALID_ARGUMENT;
    if (!h->ev0) {
        HIP_TRY(hipEventCreate(&h->ev0));
        HIP_TRY(hipEventCreate(&h->ev1));
    }
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    return CSR5HIP_SUCCESS;
}

int csr5hip_timer_stop(csr5hip_handle h, double *ms)
{
    if (!h || !ms || !h->ev0)
        return CSR5HIP_INVALID_ARGUMENT;
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float f = 0;
    HIP_TRY(hipEventElapsedTime(&f, h->ev0, h->ev1));
    *ms = f;
    return CSR5HIP_SUCCESS;
}

} // extern "C"
