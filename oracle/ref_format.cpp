// ref_format.cpp -- thin driver around the REFERENCE's own format code (TEST INFRASTRUCTURE ONLY).
//
// Compiled only where /root/reference exists (this container), from the reference sources where
// they lie, by oracle/Makefile, into oracle/_ref/libref_fmt_w<OMEGA>.so.  No reference source is
// copied into this repository; this file only #includes the reference headers by path and calls
// their free function templates:
//   generate_partition_pointer            CSR5_avx2/detail/avx2/format_avx2.h:65
//   generate_partition_descriptor         CSR5_avx2/detail/avx2/format_avx2.h:239
//   generate_partition_descriptor_offset  CSR5_avx2/detail/avx2/format_avx2.h:351
//   aosoa_transpose                       CSR5_avx2/detail/avx2/format_avx2.h:434
// The format code is scalar and generic in the ANONYMOUSLIB_CSR5_OMEGA macro, so re-instantiating it
// with OMEGA=64 gives the bit-exact goldens for the wave64 layout.
#include "detail/avx2/common_avx2.h"
#ifdef REF_OMEGA
#undef ANONYMOUSLIB_CSR5_OMEGA
#define ANONYMOUSLIB_CSR5_OMEGA REF_OMEGA
#endif
#include "detail/avx2/utils_avx2.h"
#include "detail/avx2/format_avx2.h"

extern "C" int ref_fmt_omega() { return ANONYMOUSLIB_CSR5_OMEGA; }

// Mirrors the parameter derivation of anonymouslibHandle::asCSR5 (CSR5_avx2/anonymouslib_avx2.h:121-137).
// out = {bit_y_offset, bit_scansum_offset, num_packet, p}
extern "C" int ref_fmt_params(int sigma, int nnz, int *out)
{
    int base = 2, bit_y = 1, bit_ss = 1;
    while (base < ANONYMOUSLIB_CSR5_OMEGA * sigma) { base *= 2; bit_y++; }
    base = 2;
    while (base < ANONYMOUSLIB_CSR5_OMEGA) { base *= 2; bit_ss++; }
    if (bit_y + bit_ss > 31) return -2;
    out[0] = bit_y;
    out[1] = bit_ss;
    out[2] = (int)ceil((double)(bit_y + bit_ss + sigma) / 32.0);
    out[3] = (int)ceil((double)nnz / (double)(ANONYMOUSLIB_CSR5_OMEGA * sigma));
    return 0;
}

// row_ptr must have m+2 readable entries (the reference reads row_ptr[m+1] for the last tile,
// format_avx2.h:48-50).  tile_desc must be zero-filled and offset_ptr zero-filled by the caller, as
// asCSR5 does (anonymouslib_avx2.h:145,152).  Returns num_offsets.
extern "C" int ref_fmt_tile(int sigma, int p, int m, int nnz, int bit_y, int bit_ss, int num_packet,
                            const int *row_ptr, unsigned int *tile_ptr, unsigned int *tile_desc,
                            int *offset_ptr)
{
    generate_partition_pointer<int, unsigned int>(sigma, p, m, nnz, tile_ptr, row_ptr);
    int num_offsets = 0;
    generate_partition_descriptor<int, unsigned int>(sigma, p, m, bit_y, bit_ss, num_packet, row_ptr,
                                                     tile_ptr, tile_desc, offset_ptr, &num_offsets);
    return num_offsets;
}

extern "C" void ref_fmt_offset(int sigma, int p, int bit_y, int bit_ss, int num_packet,
                               const int *row_ptr, const unsigned int *tile_ptr,
                               unsigned int *tile_desc, int *offset_ptr, int *offset)
{
    generate_partition_descriptor_offset<int, unsigned int>(sigma, p, bit_y, bit_ss, num_packet,
                                                            row_ptr, tile_ptr, tile_desc, offset_ptr,
                                                            offset);
}

extern "C" void ref_fmt_transpose_f64(int sigma, int nnz, const unsigned int *tile_ptr, int *col,
                                      double *val, int r2c)
{
    aosoa_transpose<int, unsigned int, double>(sigma, nnz, tile_ptr, col, val, r2c != 0);
}

extern "C" void ref_fmt_transpose_f32(int sigma, int nnz, const unsigned int *tile_ptr, int *col,
                                      float *val, int r2c)
{
    aosoa_transpose<int, unsigned int, float>(sigma, nnz, tile_ptr, col, val, r2c != 0);
}
