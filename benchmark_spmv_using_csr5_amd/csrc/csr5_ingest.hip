// csr5_ingest.hip -- Matrix Market ingest and COO -> CSR on the device.
//
// What the reference CLI does before the hot path (CSR5_avx2/main.cpp; the CUDA main shares it):
//   banner / size line         mmio.h:254-369 through main.cpp:138-157
//   one fscanf per entry       main.cpp:181-208
//   row histogram + scan       main.cpp:186-233
//   counting scatter           main.cpp:241-275  (file order inside a row, the mirror of a symmetric
//                                                 off-diagonal right after its original)
// Here: an mmap'ed file parsed by T host threads (lines are counted first so that every thread knows
// where its entries land -> file order is preserved), then on the device a stable radix sort of
// (row, emission index) pairs -- the emission sequence is exactly the order in which the scatter loop
// of the reference visits (entry, mirror) -- followed by one gather.  The result is the same CSR,
// entry for entry.  Input that the strict line parser does not recognise (entries spread over several
// lines, trailing tokens, ...) is re-read by a sequential scanner built on fscanf itself, i.e. with
// the reference's own token semantics.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <cctype>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "csr5_internal.h"

namespace csr5 {
void set_last_error(const std::string &msg);
}

namespace {

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------------
// host side: header
// ------------------------------------------------------------------------------------------------
struct Head {
    int m = 0, n = 0;
    long long nz = 0;
    int field = 0;
    int symmetric = 0;
    size_t data_begin = 0;  // offset of the first byte after the size line
};

void lower(char *s)
{
    for (; *s; ++s)
        *s = (char)tolower((unsigned char)*s);
}

// one line of at most 1024 characters, like fgets(line, 1025, f): returns the offset after it
size_t take_line(const char *base, size_t size, size_t at, char *line)
{
    size_t k = 0;
    while (at < size && k < 1024) {
        const char c = base[at++];
        line[k++] = c;
        if (c == '\n')
            break;
    }
    line[k] = 0;
    return at;
}

int parse_head(const char *base, size_t size, Head &h)
{
    char line[1026];
    char banner[1026], mtx[1026], crd[1026], field[1026], symm[1026];
    if (size == 0)
        return CSR5HIP_MTX_BAD_BANNER;
    size_t at = take_line(base, size, 0, line);
    if (sscanf(line, "%s %s %s %s %s", banner, mtx, crd, field, symm) != 5)
        return CSR5HIP_MTX_BAD_BANNER;
    lower(mtx), lower(crd), lower(field), lower(symm);
    if (strncmp(banner, "%%MatrixMarket", 14) != 0 || strcmp(mtx, "matrix") != 0)
        return CSR5HIP_MTX_BAD_BANNER;
    if (strcmp(crd, "coordinate") != 0 && strcmp(crd, "array") != 0)
        return CSR5HIP_MTX_BAD_BANNER;
    bool is_complex = false;
    if (!strcmp(field, "real"))         h.field = CSR5HIP_FIELD_REAL;
    else if (!strcmp(field, "integer")) h.field = CSR5HIP_FIELD_INTEGER;
    else if (!strcmp(field, "pattern")) h.field = CSR5HIP_FIELD_PATTERN;
    else if (!strcmp(field, "complex")) is_complex = true;
    else return CSR5HIP_MTX_BAD_BANNER;
    if (!strcmp(symm, "symmetric") || !strcmp(symm, "hermitian")) h.symmetric = 1;
    else if (strcmp(symm, "general") != 0 && strcmp(symm, "skew-symmetric") != 0)
        return CSR5HIP_MTX_BAD_BANNER;
    if (is_complex)
        return CSR5HIP_MTX_COMPLEX;
    if (strcmp(crd, "coordinate") != 0)
        return CSR5HIP_MTX_BAD_SIZE;  // dense "array" files have no "m n nz" line
    // size line: first line that does not start with '%'
    do {
        if (at >= size)
            return CSR5HIP_MTX_BAD_SIZE;
        at = take_line(base, size, at, line);
    } while (line[0] == '%');
    int m = 0, n = 0, nz = 0;
    if (sscanf(line, "%d %d %d", &m, &n, &nz) != 3) {
        // blank line: the reference keeps scanning tokens (mmio.h:359-366)
        while (at < size && isspace((unsigned char)base[at]))
            at++;
        size_t end = at;
        int lines = 0;
        while (end < size && lines < 1) {
            if (base[end] == '\n')
                lines++;
            end++;
        }
        std::string rest(base + at, base + end);
        if (sscanf(rest.c_str(), "%d %d %d", &m, &n, &nz) != 3)
            return CSR5HIP_MTX_BAD_SIZE;
        at = end;
    }
    if (m < 0 || n < 0 || nz < 0)
        return CSR5HIP_MTX_BAD_SIZE;
    h.m = m, h.n = n, h.nz = nz, h.data_begin = at;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// host side: strict line parser (the fast path)
// ------------------------------------------------------------------------------------------------
inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

inline bool parse_index(const char *&p, const char *e, int &out)
{
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-'))
        neg = *p++ == '-';
    if (p >= e || *p < '0' || *p > '9')
        return false;
    long long v = 0;
    while (p < e && *p >= '0' && *p <= '9') {
        v = v * 10 + (*p++ - '0');
        if (v > (long long)INT_MAX + 1)
            return false;
    }
    if (p < e && !is_blank(*p) && *p != '\n')
        return false;
    v = neg ? -v : v;
    if (v > INT_MAX || v < INT_MIN)
        return false;
    out = (int)v;
    return true;
}

const double POW10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                          1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// correctly rounded like strtod / "%lg": exact shortcut when mantissa < 2^53 and |exp10| <= 22
inline bool parse_real(const char *&p, const char *e, double &out)
{
    const char *tok = p;
    const char *q = p;
    bool neg = false;
    if (q < e && (*q == '+' || *q == '-'))
        neg = *q++ == '-';
    unsigned long long mant = 0;
    int digits = 0, exp10 = 0;
    bool any = false, simple = true;
    while (q < e && *q >= '0' && *q <= '9') {
        any = true;
        if (digits < 19) { mant = mant * 10 + (unsigned)(*q - '0'); if (mant) digits++; }
        else exp10++;
        q++;
    }
    if (q < e && *q == '.') {
        q++;
        while (q < e && *q >= '0' && *q <= '9') {
            any = true;
            if (digits < 19) { mant = mant * 10 + (unsigned)(*q - '0'); if (mant) digits++; exp10--; }
            q++;
        }
    }
    if (!any)
        simple = false;
    if (simple && q < e && (*q == 'e' || *q == 'E')) {
        const char *r = q + 1;
        bool eneg = false;
        if (r < e && (*r == '+' || *r == '-'))
            eneg = *r++ == '-';
        if (r < e && *r >= '0' && *r <= '9') {
            int ex = 0;
            while (r < e && *r >= '0' && *r <= '9') {
                if (ex < 100000) ex = ex * 10 + (*r - '0');
                r++;
            }
            exp10 += eneg ? -ex : ex;
            q = r;
        } else {
            simple = false;
        }
    }
    if (simple && (q >= e || is_blank(*q) || *q == '\n')) {
        if (mant < (1ull << 53) && exp10 >= -22 && exp10 <= 22) {
            double v = (double)mant;
            v = exp10 < 0 ? v / POW10[-exp10] : v * POW10[exp10];
            out = neg ? -v : v;
            p = q;
            return true;
        }
    }
    // general token (many digits, huge exponents, inf, nan, hex floats): libc decides
    const char *end = tok;
    while (end < e && !is_blank(*end) && *end != '\n')
        end++;
    if (end == tok)
        return false;
    char buf[128];
    std::string big;
    const char *z;
    const size_t len = (size_t)(end - tok);
    if (len < sizeof buf) { memcpy(buf, tok, len); buf[len] = 0; z = buf; }
    else { big.assign(tok, end); z = big.c_str(); }
    char *stop = nullptr;
    const double v = strtod(z, &stop);
    if (stop != z + len)
        return false;
    out = v;
    p = end;
    return true;
}

struct Chunk {
    size_t begin, end;  // [begin, end) ends right after a '\n' (or at EOF)
    long long lines;    // non-blank lines
    long long first;    // index of its first entry
};

// returns false when a line is not "<int> <int> [<value>]"
bool parse_chunk(const char *base, const Chunk &c, const Head &h, long long limit, int32_t *row,
                 int32_t *col, double *val, std::atomic<long long> &bad_index)
{
    const char *p = base + c.begin;
    const char *e = base + c.end;
    long long k = c.first;
    while (p < e && k < limit) {
        while (p < e && is_blank(*p))
            p++;
        if (p >= e)
            break;
        if (*p == '\n') { p++; continue; }
        int a, b;
        if (!parse_index(p, e, a))
            return false;
        while (p < e && is_blank(*p))
            p++;
        if (!parse_index(p, e, b))
            return false;
        double v = 1.0;
        if (h.field != CSR5HIP_FIELD_PATTERN) {
            while (p < e && is_blank(*p))
                p++;
            if (h.field == CSR5HIP_FIELD_INTEGER) {
                int iv;
                if (!parse_index(p, e, iv))
                    return false;
                v = iv;
            } else if (!parse_real(p, e, v)) {
                return false;
            }
        }
        while (p < e && is_blank(*p))
            p++;
        if (p < e && *p != '\n')
            return false;
        if (p < e)
            p++;
        if (a < 1 || a > h.m || b < 1 || b > h.n || (h.symmetric && (b > h.m || a > h.n))) {
            long long none = -1;
            bad_index.compare_exchange_strong(none, k);
        }
        row[k] = a - 1;
        col[k] = b - 1;
        val[k] = v;
        k++;
    }
    return true;
}

long long count_lines(const char *base, size_t begin, size_t end)
{
    long long lines = 0;
    const char *p = base + begin;
    const char *e = base + end;
    while (p < e) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        const char *stop = nl ? nl : e;
        bool blank = true;
        for (const char *q = p; q < stop; ++q)
            if (!is_blank(*q)) { blank = false; break; }
        if (!blank)
            lines++;
        p = nl ? nl + 1 : e;
    }
    return lines;
}

// the reference's own scanner, token for token (main.cpp:181-208)
int scan_sequential(const char *base, size_t size, const Head &h, int32_t *row, int32_t *col, double *val,
                    long long &bad_index)
{
    FILE *f = fmemopen((void *)(base + h.data_begin), size - h.data_begin, "r");
    if (!f)
        return CSR5HIP_MTX_BAD_SIZE;
    int rc = 0;
    for (long long i = 0; i < h.nz; i++) {
        int a = 0, b = 0, iv = 0, got, want;
        double fv = 1.0;
        if (h.field == CSR5HIP_FIELD_REAL) { got = fscanf(f, "%d %d %lg\n", &a, &b, &fv); want = 3; }
        else if (h.field == CSR5HIP_FIELD_INTEGER) { got = fscanf(f, "%d %d %d\n", &a, &b, &iv); want = 3; fv = iv; }
        else { got = fscanf(f, "%d %d\n", &a, &b); want = 2; }
        if (got != want) { rc = CSR5HIP_MTX_BAD_SIZE; break; }  // the reference would go on with garbage
        if ((a < 1 || a > h.m || b < 1 || b > h.n || (h.symmetric && (b > h.m || a > h.n))) && bad_index < 0)
            bad_index = i;
        row[i] = a - 1;
        col[i] = b - 1;
        val[i] = fv;
    }
    fclose(f);
    return rc;
}

// Plain pageable memory: measured on the MI355X box for 160 MB of triplets, pinning costs 7 ms and saves 3.3 ms
// of the H2D copy (3.4 ms pinned, 6.6 ms pageable), so it does not pay for arrays that are copied once.
void *host_alloc(size_t bytes) { return malloc(bytes ? bytes : 8); }

// ------------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------------
__global__ void k_check_range(long long nz, const int *__restrict__ row, const int *__restrict__ col, int m,
                              int n, int need_square, unsigned *bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nz)
        return;
    const int r = row[i], c = col[i];
    const bool ok = r >= 0 && r < m && c >= 0 && c < n && (!need_square || (c < m && r < n));
    if (!ok)
        atomicAdd(bad, 1u);
}

// emissions per entry: 1, or 2 for a symmetric off-diagonal (main.cpp:213-220)
__global__ void k_emit_count(long long nz, const int *__restrict__ row, const int *__restrict__ col,
                             unsigned *__restrict__ cnt)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nz)
        cnt[i] = row[i] != col[i] ? 2u : 1u;
}

// emission sequence of the scatter loop (main.cpp:243-262): entry i, then its mirror
__global__ void k_emit(long long nz, const int *__restrict__ row, const int *__restrict__ col,
                       const unsigned *__restrict__ at, unsigned *__restrict__ key,
                       unsigned *__restrict__ src)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nz)
        return;
    const int r = row[i], c = col[i];
    const unsigned o = at[i];
    key[o] = (unsigned)r;
    src[o] = (unsigned)i << 1;
    if (r != c) {
        key[o + 1] = (unsigned)c;
        src[o + 1] = ((unsigned)i << 1) | 1u;
    }
}

// row_ptr from the sorted row keys: position i closes rows (key[i-1], key[i]]
__global__ void k_row_ptr(long long nnz, const unsigned *__restrict__ key, int m, int *__restrict__ row_ptr)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nnz)
        return;
    const long long lo = i == 0 ? -1 : (long long)key[i - 1];
    const long long hi = i == nnz ? (long long)m : (long long)key[i];
    for (long long r = lo + 1; r <= hi; ++r)
        row_ptr[r] = (int)i;
}

// MIRRORED: src = entry << 1 | mirror (symmetric files); otherwise src = entry
template <typename VT, bool MIRRORED>
__global__ void k_gather(long long nnz, const unsigned *__restrict__ src, const int *__restrict__ row,
                         const int *__restrict__ col, const double *__restrict__ val, int *__restrict__ col_out,
                         VT *__restrict__ val_out)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nnz)
        return;
    const unsigned s = src[j];
    const unsigned i = MIRRORED ? s >> 1 : s;
    col_out[j] = (MIRRORED && (s & 1u)) ? row[i] : col[i];
    if (val_out)
        val_out[j] = (VT)val[i];
}

int fail(int code, const std::string &msg)
{
    csr5::set_last_error(msg);
    return code;
}

#define ING_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            cleanup();                                                                             \
            return fail(CSR5HIP_HIP_ERROR, std::string(#expr) + ": " + hipGetErrorString(e_));     \
        }                                                                                          \
    } while (0)

inline unsigned blocks_for(long long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

} // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int csr5hip_mtx_read(const char *path, int threads, csr5hip_mtx *out)
{
    if (!path || !out)
        return fail(CSR5HIP_INVALID_ARGUMENT, "csr5hip_mtx_read: null argument");
    memset(out, 0, sizeof *out);
    const double t0 = now_ms();
    const int fd = open(path, O_RDONLY);
    if (fd < 0)
        return fail(CSR5HIP_MTX_CANNOT_OPEN, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        return fail(CSR5HIP_MTX_CANNOT_OPEN, std::string("not a regular file: ") + path);
    }
    const size_t size = (size_t)st.st_size;
    const char *base = (const char *)"";
    void *map = nullptr;
    if (size) {
        map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0); // no MAP_POPULATE: the 64 counting threads fault the pages in 5x faster than one serial populate
        if (map == MAP_FAILED) {
            close(fd);
            return fail(CSR5HIP_MTX_CANNOT_OPEN, std::string("mmap failed: ") + path);
        }
        madvise(map, size, MADV_WILLNEED);
        base = (const char *)map;
    }
    close(fd);
    auto unmap = [&]() { if (map) munmap(map, size); };

    Head h;
    int rc = parse_head(base, size, h);
    if (rc) {
        unmap();
        return fail(rc, rc == CSR5HIP_MTX_COMPLEX ? "Sorry, data type 'COMPLEX' is not supported. "
                      : rc == CSR5HIP_MTX_BAD_SIZE ? "bad Matrix Market size line"
                                                   : "Could not process Matrix Market banner.");
    }
    if (h.nz > (long long)INT_MAX / 2) {
        unmap();
        return fail(CSR5HIP_INVALID_ARGUMENT, "more entries than a 32-bit CSR can hold");
    }
    const bool trace = getenv("CSR5_INGEST_TRACE") != nullptr;
    const double t_head = now_ms();
    int32_t *row = (int32_t *)host_alloc(sizeof(int32_t) * (size_t)h.nz);
    int32_t *col = (int32_t *)host_alloc(sizeof(int32_t) * (size_t)h.nz);
    double *val = (double *)host_alloc(sizeof(double) * (size_t)h.nz);
    auto drop = [&]() {
        free(row);
        free(col);
        free(val);
    };
    if (!row || !col || !val) {
        drop();
        unmap();
        return fail(CSR5HIP_HIP_ERROR, "host allocation failed");
    }

    const double t_alloc = now_ms();
    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    if (T > 64 && threads <= 0) T = 64;
    const size_t data_bytes = size - h.data_begin;
    const size_t min_chunk = threads > 0 ? 16 : (1u << 20);  // an explicit thread count is honoured
    if ((size_t)T > data_bytes / min_chunk + 1)
        T = (int)(data_bytes / min_chunk + 1);

    // cut at line ends
    std::vector<Chunk> chunks((size_t)T);
    size_t at = h.data_begin;
    for (int t = 0; t < T; ++t) {
        size_t end = t == T - 1 ? size : h.data_begin + data_bytes * (size_t)(t + 1) / (size_t)T;
        if (end < at) end = at;
        if (t != T - 1) {
            const char *nl = (const char *)memchr(base + end, '\n', size - end);
            end = nl ? (size_t)(nl - base) + 1 : size;
        }
        chunks[(size_t)t] = Chunk{at, end, 0, 0};
        at = end;
    }
    // ONE set of threads for both passes: a thread parses the chunk whose lines it has just counted, so the
    // second pass finds its 4-MB share of the file in that core's caches.
    std::atomic<long long> bad_index(-1);
    std::vector<char> ok((size_t)T, 1);
    bool fast = false;
    long long total = 0;
    double t_count = 0;
    struct Gate {
        std::mutex m;
        std::condition_variable cv;
        int expected, waiting = 0, generation = 0;
        void wait()
        {
            std::unique_lock<std::mutex> l(m);
            const int g = generation;
            if (++waiting == expected) {
                waiting = 0;
                generation++;
                cv.notify_all();
            } else {
                cv.wait(l, [&] { return generation != g; });
            }
        }
    } gate;
    gate.expected = T;
    auto work = [&](int t) {
        chunks[(size_t)t].lines = count_lines(base, chunks[(size_t)t].begin, chunks[(size_t)t].end);
        gate.wait();
        if (t == 0) {
            for (auto &c : chunks) { c.first = total; total += c.lines; }
            fast = total >= h.nz;
            t_count = now_ms();
        }
        gate.wait();
        if (fast)
            ok[(size_t)t] = parse_chunk(base, chunks[(size_t)t], h, h.nz, row, col, val, bad_index);
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < T; ++t)
            pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
    }
    for (char k : ok) fast = fast && k;
    long long bad = bad_index.load();
    if (!fast) {
        bad = -1;
        rc = scan_sequential(base, size, h, row, col, val, bad);
        if (rc) {
            drop();
            unmap();
            return fail(rc, "the file ends before the announced number of entries");
        }
    }
    unmap();
    if (bad >= 0) {
        drop();
        return fail(CSR5HIP_INVALID_ARGUMENT, "entry " + std::to_string(bad) + " has an index outside the matrix");
    }
    out->m = h.m, out->n = h.n, out->nz = h.nz;
    out->field = h.field, out->symmetric = h.symmetric;
    out->row = row, out->col = col, out->val = val;
    out->threads = T, out->fast_path = fast ? 1 : 0;
    out->file_bytes = (int64_t)size;
    out->alloc_flags = 0;
    out->t_parse_ms = now_ms() - t0;
    if (trace)
        fprintf(stderr, "[csr5hip ingest] %d threads: open+head %.2f ms, alloc %.2f ms, count lines %.2f ms, parse %.2f ms\n",
                T, t_head - t0, t_alloc - t_head, t_count - t_alloc, now_ms() - t_count);
    return CSR5HIP_SUCCESS;
}

extern "C" int csr5hip_mtx_release(csr5hip_mtx *mtx)
{
    if (!mtx)
        return CSR5HIP_INVALID_ARGUMENT;
    free(mtx->row);
    free(mtx->col);
    free(mtx->val);
    memset(mtx, 0, sizeof *mtx);
    return CSR5HIP_SUCCESS;
}

extern "C" int csr5hip_csr_release(csr5hip_csr *csr)
{
    if (!csr)
        return CSR5HIP_INVALID_ARGUMENT;
    if (csr->d_row_ptr) (void)hipFree(csr->d_row_ptr);
    if (csr->d_col_idx) (void)hipFree(csr->d_col_idx);
    if (csr->d_val) (void)hipFree(csr->d_val);
    memset(csr, 0, sizeof *csr);
    return CSR5HIP_SUCCESS;
}

extern "C" int csr5hip_coo_to_csr(int32_t m, int32_t n, int64_t nz, const int32_t *d_row, const int32_t *d_col,
                                  const double *d_val, int symmetric, int value_type, csr5hip_csr *out)
{
    if (!out || m < 0 || n < 0 || nz < 0 || (nz > 0 && (!d_row || !d_col)))
        return fail(CSR5HIP_INVALID_ARGUMENT, "csr5hip_coo_to_csr: bad argument");
    if (value_type != CSR5HIP_F64 && value_type != CSR5HIP_F32)
        return fail(CSR5HIP_UNSUPPORTED_VALUE_TYPE, "csr5hip_coo_to_csr: value_type");
    if (nz > (int64_t)INT_MAX / 2)
        return fail(CSR5HIP_INVALID_ARGUMENT, "more entries than a 32-bit CSR can hold");
    memset(out, 0, sizeof *out);
    hipStream_t s = nullptr;
    constexpr int BS = 256;

    unsigned *d_cnt = nullptr, *d_key = nullptr, *d_key2 = nullptr, *d_src = nullptr, *d_src2 = nullptr;
    unsigned *d_bad = nullptr;
    void *d_tmp = nullptr;
    int32_t *row_ptr = nullptr, *col_idx = nullptr;
    void *val_out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto scratch = [&]() {
        if (d_cnt) (void)hipFree(d_cnt);
        if (d_key) (void)hipFree(d_key);
        if (d_key2) (void)hipFree(d_key2);
        if (d_src) (void)hipFree(d_src);
        if (d_src2) (void)hipFree(d_src2);
        if (d_bad) (void)hipFree(d_bad);
        if (d_tmp) (void)hipFree(d_tmp);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        d_cnt = d_key = d_key2 = d_src = d_src2 = d_bad = nullptr;
        d_tmp = nullptr;
        e0 = e1 = nullptr;
    };
    auto cleanup = [&]() {
        scratch();
        if (row_ptr) (void)hipFree(row_ptr);
        if (col_idx) (void)hipFree(col_idx);
        if (val_out) (void)hipFree(val_out);
    };

    ING_TRY(hipEventCreate(&e0));
    ING_TRY(hipEventCreate(&e1));
    ING_TRY(hipEventRecord(e0, s));

    // indices must be inside the matrix (and inside the square part when they get mirrored)
    ING_TRY(hipMalloc(&d_bad, sizeof(unsigned)));
    ING_TRY(hipMemsetAsync(d_bad, 0, sizeof(unsigned), s));
    if (nz)
        k_check_range<<<blocks_for(nz, BS), BS, 0, s>>>(nz, d_row, d_col, m, n, symmetric ? 1 : 0, d_bad);
    unsigned bad = 0;
    ING_TRY(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, s));
    ING_TRY(hipStreamSynchronize(s));
    if (bad) {
        cleanup();
        return fail(CSR5HIP_INVALID_ARGUMENT, std::to_string(bad) + " COO entries lie outside the matrix");
    }

    // 1. emission sequence
    long long nnz = nz;
    const unsigned *key_in = (const unsigned *)d_row;
    if (symmetric && nz) {
        ING_TRY(hipMalloc(&d_cnt, sizeof(unsigned) * (size_t)(nz + 1)));
        k_emit_count<<<blocks_for(nz, BS), BS, 0, s>>>(nz, d_row, d_col, d_cnt);
        ING_TRY(hipMemsetAsync(d_cnt + nz, 0, sizeof(unsigned), s));
        size_t tmp_bytes = 0;
        ING_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, d_cnt, d_cnt, 0u, (size_t)nz + 1, rocprim::plus<unsigned>(), s));
        ING_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8));
        ING_TRY(rocprim::exclusive_scan(d_tmp, tmp_bytes, d_cnt, d_cnt, 0u, (size_t)nz + 1, rocprim::plus<unsigned>(), s));
        unsigned total = 0;
        ING_TRY(hipMemcpyAsync(&total, d_cnt + nz, sizeof total, hipMemcpyDeviceToHost, s));
        ING_TRY(hipStreamSynchronize(s));
        (void)hipFree(d_tmp);
        d_tmp = nullptr;
        nnz = total;
        if (nnz > INT_MAX) {
            cleanup();
            return fail(CSR5HIP_INVALID_ARGUMENT, "expanded matrix exceeds a 32-bit CSR");
        }
        ING_TRY(hipMalloc(&d_key, sizeof(unsigned) * (size_t)nnz));
        ING_TRY(hipMalloc(&d_src, sizeof(unsigned) * (size_t)nnz));
        k_emit<<<blocks_for(nz, BS), BS, 0, s>>>(nz, d_row, d_col, d_cnt, d_key, d_src);
        key_in = d_key;
    }

    ING_TRY(hipMalloc(&row_ptr, sizeof(int32_t) * ((size_t)m + 1)));
    ING_TRY(hipMalloc(&col_idx, sizeof(int32_t) * (size_t)(nnz ? nnz : 1)));
    if (d_val)
        ING_TRY(hipMalloc(&val_out, (value_type == CSR5HIP_F64 ? 8 : 4) * (size_t)(nnz ? nnz : 1)));

    if (nnz) {
        // 2. stable sort of (row, emission id): file order survives inside a row
        unsigned bits = 1;
        while (bits < 32 && (1ull << bits) < (unsigned long long)(m > 1 ? m : 2))
            bits++;
        ING_TRY(hipMalloc(&d_key2, sizeof(unsigned) * (size_t)nnz));
        ING_TRY(hipMalloc(&d_src2, sizeof(unsigned) * (size_t)nnz));
        size_t tmp_bytes = 0;
        if (symmetric) {
            ING_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key_in, d_key2, d_src, d_src2, (size_t)nnz, 0, bits, s));
            ING_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8));
            ING_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, key_in, d_key2, d_src, d_src2, (size_t)nnz, 0, bits, s));
        } else {
            // the emission id is the entry number itself: no array, a counting iterator feeds the sort
            rocprim::counting_iterator<unsigned> iota(0u);
            ING_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key_in, d_key2, iota, d_src2, (size_t)nnz, 0, bits, s));
            ING_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8));
            ING_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, key_in, d_key2, iota, d_src2, (size_t)nnz, 0, bits, s));
        }
        // 3. row pointer from the run boundaries of the sorted keys, 4. one gather
        k_row_ptr<<<blocks_for(nnz + 1, BS), BS, 0, s>>>(nnz, d_key2, m, row_ptr);
        if (symmetric) {
            if (value_type == CSR5HIP_F64)
                k_gather<double, true><<<blocks_for(nnz, BS), BS, 0, s>>>(nnz, d_src2, d_row, d_col, d_val, col_idx, (double *)val_out);
            else
                k_gather<float, true><<<blocks_for(nnz, BS), BS, 0, s>>>(nnz, d_src2, d_row, d_col, d_val, col_idx, (float *)val_out);
        } else {
            if (value_type == CSR5HIP_F64)
                k_gather<double, false><<<blocks_for(nnz, BS), BS, 0, s>>>(nnz, d_src2, nullptr, d_col, d_val, col_idx, (double *)val_out);
            else
                k_gather<float, false><<<blocks_for(nnz, BS), BS, 0, s>>>(nnz, d_src2, nullptr, d_col, d_val, col_idx, (float *)val_out);
        }
    } else {
        ING_TRY(hipMemsetAsync(row_ptr, 0, sizeof(int32_t) * ((size_t)m + 1), s));
    }
    ING_TRY(hipGetLastError());
    ING_TRY(hipEventRecord(e1, s));
    ING_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    ING_TRY(hipEventElapsedTime(&ms, e0, e1));
    scratch();
    out->m = m, out->n = n, out->nnz = (int32_t)nnz;
    out->d_row_ptr = row_ptr, out->d_col_idx = col_idx, out->d_val = val_out;
    out->value_type = value_type;
    out->t_build_ms = ms;
    return CSR5HIP_SUCCESS;
}

extern "C" int csr5hip_mtx_load(const char *path, int threads, int value_type, csr5hip_csr *out)
{
    if (!out)
        return fail(CSR5HIP_INVALID_ARGUMENT, "csr5hip_mtx_load: null argument");
    memset(out, 0, sizeof *out);
    csr5hip_mtx mtx;
    int rc = csr5hip_mtx_read(path, threads, &mtx);
    if (rc)
        return rc;
    int32_t *d_row = nullptr, *d_col = nullptr;
    double *d_val = nullptr;
    const size_t nz = (size_t)mtx.nz;
    auto cleanup = [&]() {
        if (d_row) (void)hipFree(d_row);
        if (d_col) (void)hipFree(d_col);
        if (d_val) (void)hipFree(d_val);
        csr5hip_mtx_release(&mtx);
    };
    const double t0 = now_ms();
    ING_TRY(hipMalloc(&d_row, sizeof(int32_t) * (nz ? nz : 1)));
    ING_TRY(hipMalloc(&d_col, sizeof(int32_t) * (nz ? nz : 1)));
    ING_TRY(hipMalloc(&d_val, sizeof(double) * (nz ? nz : 1)));
    if (nz) {
        ING_TRY(hipMemcpy(d_row, mtx.row, sizeof(int32_t) * nz, hipMemcpyHostToDevice));
        ING_TRY(hipMemcpy(d_col, mtx.col, sizeof(int32_t) * nz, hipMemcpyHostToDevice));
        ING_TRY(hipMemcpy(d_val, mtx.val, sizeof(double) * nz, hipMemcpyHostToDevice));
    }
    const double t_h2d = now_ms() - t0;
    rc = csr5hip_coo_to_csr(mtx.m, mtx.n, mtx.nz, d_row, d_col, d_val, mtx.symmetric, value_type, out);
    const double t_parse = mtx.t_parse_ms;
    cleanup();
    if (rc)
        return rc;
    out->t_parse_ms = t_parse;
    out->t_h2d_ms = t_h2d;
    return CSR5HIP_SUCCESS;
}
