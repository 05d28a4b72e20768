/* mtx_oracle.c -- CPU restatement of the reference CLI's Matrix Market ingest and COO->CSR build.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as csr5_oracle.c): only tests/, __graft_entry__.smoke() and
 * bench scripts' cpu_baseline legs may call it; the product (libcsr5hip.so) never does.
 *
 * PINNED: tests/test_oracle_golden.py checks it against the reference's own main.cpp ingest
 * (oracle/_ref/libref_ingest.so, built from the reference sources where they lie) and against the
 * CSR fixtures that build produced (tests/golden/mtx/).
 *
 * Follows, sequentially and with the same libc calls:
 *   banner            CSR5_avx2/mmio.h:254-338  (mm_read_banner)
 *   size line         CSR5_avx2/mmio.h:340-369  (mm_read_mtx_crd_size)
 *   entry loop        CSR5_avx2/main.cpp:181-208 (fscanf per entry, 1-based -> 0-based, row histogram)
 *   symmetric count   CSR5_avx2/main.cpp:213-220
 *   exclusive scan    CSR5_avx2/main.cpp:222-233
 *   scatter           CSR5_avx2/main.cpp:241-275 (file order inside a row; the mirrored entry of a
 *                                                 symmetric off-diagonal is emitted right after it)
 * Exit codes of the CLI: -1 cannot open, -2 banner, -3 complex, -4 size line (main.cpp:135-157).
 */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LINE_MAX_MM 1025
#define TOKEN_MAX_MM 64

typedef struct {
    int m, n, nz_file;
    int is_real, is_integer, is_pattern, is_symmetric;
} mtx_head;

/* returns 0, or the CLI exit code */
static int read_head(FILE *f, mtx_head *h)
{
    char line[LINE_MAX_MM];
    char banner[TOKEN_MAX_MM], mtx[TOKEN_MAX_MM], crd[TOKEN_MAX_MM], field[TOKEN_MAX_MM], symm[TOKEN_MAX_MM];
    memset(h, 0, sizeof *h);
    if (!fgets(line, LINE_MAX_MM, f)) return -2;
    if (sscanf(line, "%s %s %s %s %s", banner, mtx, crd, field, symm) != 5) return -2;
    for (char *p = mtx; *p; p++) *p = (char)tolower(*p);
    for (char *p = crd; *p; p++) *p = (char)tolower(*p);
    for (char *p = field; *p; p++) *p = (char)tolower(*p);
    for (char *p = symm; *p; p++) *p = (char)tolower(*p);
    if (strncmp(banner, "%%MatrixMarket", 14) != 0) return -2;
    if (strcmp(mtx, "matrix") != 0) return -2;
    if (strcmp(crd, "coordinate") != 0 && strcmp(crd, "array") != 0) return -2;
    int is_complex = 0;
    if (strcmp(field, "real") == 0) h->is_real = 1;
    else if (strcmp(field, "complex") == 0) is_complex = 1;
    else if (strcmp(field, "pattern") == 0) h->is_pattern = 1;
    else if (strcmp(field, "integer") == 0) h->is_integer = 1;
    else return -2;
    if (strcmp(symm, "general") == 0) ;
    else if (strcmp(symm, "symmetric") == 0 || strcmp(symm, "hermitian") == 0) h->is_symmetric = 1;
    else if (strcmp(symm, "skew-symmetric") == 0) ; /* accepted, NOT expanded (main.cpp:159) */
    else return -2;
    if (is_complex) return -3;
    /* size line: skip comment lines; a blank line falls through to token scanning */
    do {
        if (!fgets(line, LINE_MAX_MM, f)) return -4;
    } while (line[0] == '%');
    if (sscanf(line, "%d %d %d", &h->m, &h->n, &h->nz_file) != 3) {
        int got;
        do {
            got = fscanf(f, "%d %d %d", &h->m, &h->n, &h->nz_file);
            if (got == EOF) return -4;
        } while (got != 3);
    }
    return 0;
}

/* COO (file order, 0-based) -> CSR exactly as the scatter loops do.  row_ptr has m+1 entries;
 * col_out / val_out must hold csr5o_coo_nnz() entries.  val may be NULL (pattern only). */
int csr5o_coo_nnz(int nz, const int *row, const int *col, int symmetric)
{
    int nnz = nz;
    if (symmetric)
        for (int i = 0; i < nz; i++)
            if (row[i] != col[i]) nnz++;
    return nnz;
}

void csr5o_coo_to_csr(int m, int nz, const int *row, const int *col, const double *val, int symmetric,
                      int *row_ptr, int *col_out, double *val_out)
{
    int *counter = (int *)calloc((size_t)m + 1, sizeof(int));
    for (int i = 0; i < nz; i++) counter[row[i]]++;
    if (symmetric)
        for (int i = 0; i < nz; i++)
            if (row[i] != col[i]) counter[col[i]]++;
    int run = 0;
    for (int r = 0; r <= m; r++) {
        int c = counter[r];
        row_ptr[r] = run;
        run += c;
        counter[r] = 0;
    }
    for (int i = 0; i < nz; i++) {
        int at = row_ptr[row[i]] + counter[row[i]]++;
        col_out[at] = col[i];
        if (val) val_out[at] = val[i];
        if (symmetric && row[i] != col[i]) {
            at = row_ptr[col[i]] + counter[col[i]]++;
            col_out[at] = row[i];
            if (val) val_out[at] = val[i];
        }
    }
    free(counter);
}

static struct {
    mtx_head h;
    int *row, *col;
    double *val;
} g;

static void drop(void)
{
    free(g.row); free(g.col); free(g.val);
    g.row = g.col = 0; g.val = 0;
}

/* dims = {m, n, nnz after symmetric expansion, nz in the file, symmetric, field (0 real 1 integer 2 pattern)} */
int csr5o_mtx_read(const char *path, int *dims)
{
    drop();
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int rc = read_head(f, &g.h);
    if (rc) { fclose(f); return rc; }
    const int nz = g.h.nz_file;
    g.row = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));
    g.col = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));
    g.val = (double *)malloc(sizeof(double) * (size_t)(nz > 0 ? nz : 1));
    for (int i = 0; i < nz; i++) {
        int a = 0, b = 0, iv = 0;
        double fv = 0.0;
        if (g.h.is_real) { if (fscanf(f, "%d %d %lg\n", &a, &b, &fv) < 0) {} }
        else if (g.h.is_integer) { if (fscanf(f, "%d %d %d\n", &a, &b, &iv) < 0) {} fv = iv; }
        else if (g.h.is_pattern) { if (fscanf(f, "%d %d\n", &a, &b) < 0) {} fv = 1.0; }
        g.row[i] = a - 1;
        g.col[i] = b - 1;
        g.val[i] = fv;
    }
    fclose(f);
    dims[0] = g.h.m;
    dims[1] = g.h.n;
    dims[2] = csr5o_coo_nnz(nz, g.row, g.col, g.h.is_symmetric);
    dims[3] = nz;
    dims[4] = g.h.is_symmetric;
    dims[5] = g.h.is_real ? 0 : (g.h.is_integer ? 1 : 2);
    return 0;
}

/* the COO triplets of the last csr5o_mtx_read, file order, 0-based */
void csr5o_mtx_coo(int *row, int *col, double *val)
{
    memcpy(row, g.row, sizeof(int) * (size_t)g.h.nz_file);
    memcpy(col, g.col, sizeof(int) * (size_t)g.h.nz_file);
    memcpy(val, g.val, sizeof(double) * (size_t)g.h.nz_file);
}

/* the CSR the reference CLI would hold at main.cpp:281 */
void csr5o_mtx_csr(int *row_ptr, int *col_idx, double *val)
{
    csr5o_coo_to_csr(g.h.m, g.h.nz_file, g.row, g.col, g.val, g.h.is_symmetric, row_ptr, col_idx, val);
}
