// anonymouslib_hip.h -- host-only C++ front end of libcsr5hip.so (compiles with plain g++).
//
// Re-creates the reference's public surface so that a program written against
// CSR5_cuda/anonymouslib_cuda.h builds against this header unchanged:
//   template <iT, uiT, vT> class anonymouslibHandle      anonymouslib_cuda.h:11-53
//     anonymouslibHandle(m, n), warmup, inputCSR, asCSR, asCSR5, setX, spmv, destroy, setSigma
//   struct anonymouslib_timer { start(); stop() -> ms }  detail/cuda/utils_cuda.h:6-23
//   getB<iT, vT>(m, nnz), getFLOP<iT>(nnz)               detail/utils.h:10-20
//   ANONYMOUSLIB_* return codes / format ids / OMEGA / AUTO_TUNED_SIGMA
//                                                        detail/common.h:13-22, common_cuda.h:11-15
// Pointer semantics are the CUDA variant's: everything handed to inputCSR / setX / spmv is a DEVICE
// pointer owned by the caller (main.cu:43-63); asCSR5 permutes col_idx / val in place.
// Only <int, unsigned int, double> and <int, unsigned int, float> exist behind the C ABI; any other
// instantiation reports ANONYMOUSLIB_UNSUPPORTED_VALUE_TYPE from inputCSR.
#ifndef ANONYMOUSLIB_HIP_H
#define ANONYMOUSLIB_HIP_H

#include <sys/time.h>

#include <iostream>
#include <type_traits>

#include "csr5hip.h"

#define ANONYMOUSLIB_SUCCESS                   CSR5HIP_SUCCESS
#define ANONYMOUSLIB_UNKOWN_FORMAT             CSR5HIP_UNKOWN_FORMAT
#define ANONYMOUSLIB_UNSUPPORTED_CSR5_OMEGA    CSR5HIP_UNSUPPORTED_CSR5_OMEGA
#define ANONYMOUSLIB_CSR_TO_CSR5_FAILED        CSR5HIP_CSR_TO_CSR5_FAILED
#define ANONYMOUSLIB_UNSUPPORTED_CSR_SPMV      CSR5HIP_UNSUPPORTED_CSR_SPMV
#define ANONYMOUSLIB_UNSUPPORTED_VALUE_TYPE    CSR5HIP_UNSUPPORTED_VALUE_TYPE

#define ANONYMOUSLIB_FORMAT_CSR   CSR5HIP_FORMAT_CSR
#define ANONYMOUSLIB_FORMAT_CSR5  CSR5HIP_FORMAT_CSR5

#define ANONYMOUSLIB_CSR5_OMEGA        CSR5HIP_OMEGA
#define ANONYMOUSLIB_AUTO_TUNED_SIGMA  CSR5HIP_AUTO_TUNED_SIGMA

template <typename iT, typename vT>
double getB(const iT m, const iT nnz)
{
    // the reference CLI's byte model: x counted once per non-zero (detail/utils.h:10-14)
    return (double)(((double)m + 1 + nnz) * sizeof(iT) + (2.0 * nnz + m) * sizeof(vT));
}

template <typename iT>
double getFLOP(const iT nnz)
{
    return 2.0 * (double)nnz;
}

// Timer in milliseconds.  The CUDA variant's timer is a pair of events on the default stream whose stop() waits for the
// device (CSR5_cuda/detail/cuda/utils_cuda.h:6-23), so a caller may time an asynchronous spmv() loop without a
// synchronisation of its own.  Same guarantee here: start() and stop() drain the device before reading the host clock
// (the AVX2 variant's timer is plain gettimeofday, CSR5_avx2/detail/utils.h -- its spmv() is synchronous).
struct anonymouslib_timer {
    timeval t1, t2;
    void start()
    {
        (void)csr5hip_synchronize();
        gettimeofday(&t1, 0);
    }
    double stop()
    {
        (void)csr5hip_synchronize();
        gettimeofday(&t2, 0);
        return (t2.tv_sec - t1.tv_sec) * 1000.0 + (t2.tv_usec - t1.tv_usec) / 1000.0;
    }
};

template <class ANONYMOUSLIB_IT, class ANONYMOUSLIB_UIT, class ANONYMOUSLIB_VT>
class anonymouslibHandle
{
public:
    anonymouslibHandle(ANONYMOUSLIB_IT m, ANONYMOUSLIB_IT n) : _h(0), _err(ANONYMOUSLIB_SUCCESS), _quiet(false)
    {
        const bool ok_index = std::is_same<ANONYMOUSLIB_IT, int>::value &&
                              std::is_same<ANONYMOUSLIB_UIT, unsigned int>::value;
        int vt = std::is_same<ANONYMOUSLIB_VT, double>::value ? CSR5HIP_F64
               : std::is_same<ANONYMOUSLIB_VT, float>::value  ? CSR5HIP_F32 : -1;
        if (!ok_index || vt < 0)
            _err = ANONYMOUSLIB_UNSUPPORTED_VALUE_TYPE;
        else
            _err = csr5hip_create(&_h, (int)m, (int)n, vt);
    }
    ~anonymouslibHandle() { if (_h) csr5hip_free(_h); }

    int warmup() { return _h ? csr5hip_warmup(_h) : _err; }

    int inputCSR(ANONYMOUSLIB_IT nnz, ANONYMOUSLIB_IT *csr_row_pointer,
                 ANONYMOUSLIB_IT *csr_column_index, ANONYMOUSLIB_VT *csr_value)
    {
        if (!_h) return _err;
        return csr5hip_input_csr(_h, (int)nnz, (int32_t *)csr_row_pointer,
                                 (int32_t *)csr_column_index, (void *)csr_value);
    }

    int asCSR() { return _h ? csr5hip_as_csr(_h) : _err; }

    int asCSR5()
    {
        if (!_h) return _err;
        csr5hip_info before;
        csr5hip_get_info(_h, &before);
        const int err = csr5hip_as_csr5(_h);
        if (before.format == ANONYMOUSLIB_FORMAT_CSR && !_quiet) {
            // the stdout lines asCSR5 prints in the reference (anonymouslib_cuda.h:119,211-214)
            csr5hip_info i;
            csr5hip_get_info(_h, &i);
            std::cout << "omega = " << ANONYMOUSLIB_CSR5_OMEGA << ", sigma = " << i.sigma << ". " << std::endl;
            if (err == ANONYMOUSLIB_SUCCESS) {
                std::cout << "CSR->CSR5 malloc time = " << i.t_malloc_ms << " ms." << std::endl;
                std::cout << "CSR->CSR5 tile_ptr time = " << i.t_tile_ptr_ms << " ms." << std::endl;
                std::cout << "CSR->CSR5 tile_desc time = " << i.t_tile_desc_ms << " ms." << std::endl;
                std::cout << "CSR->CSR5 transpose time = " << i.t_transpose_ms << " ms." << std::endl;
            }
        }
        return err;
    }

    int setX(ANONYMOUSLIB_VT *x) { return _h ? csr5hip_set_x(_h, (const void *)x) : _err; }

    int spmv(const ANONYMOUSLIB_VT alpha, ANONYMOUSLIB_VT *y)
    {
        return _h ? csr5hip_spmv(_h, (double)alpha, (void *)y) : _err;
    }

    int destroy() { return _h ? csr5hip_destroy(_h) : _err; }

    void setSigma(int sigma) { if (_h) csr5hip_set_sigma(_h, sigma); }

    // ---- additions (not in the reference class) ----
    int spmv_repeat(const ANONYMOUSLIB_VT alpha, ANONYMOUSLIB_VT *y, int count)
    {
        return _h ? csr5hip_spmv_repeat(_h, (double)alpha, (void *)y, count) : _err;
    }
    int autotuneSigma(ANONYMOUSLIB_VT *y, int *sigma = 0, double *us = 0)
    {
        return _h ? csr5hip_autotune_sigma(_h, (void *)y, sigma, us) : _err;
    }
    int setOption(int option, int value) { return _h ? csr5hip_set_option(_h, option, value) : _err; }
    void setQuiet(bool q) { _quiet = q; }
    csr5hip_handle native() const { return _h; }

private:
    anonymouslibHandle(const anonymouslibHandle &);
    anonymouslibHandle &operator=(const anonymouslibHandle &);
    csr5hip_handle _h;
    int _err;
    bool _quiet;
};

// One matrix on the G GPUs of a node (not in the reference, which is single-device: main.cu:25-26): same member
// names as anonymouslibHandle over csr5hip_multi_*.  inputCSR / setX take DEVICE pointers on devices[0]; y lives
// sharded on the devices (gatherY copies it to a host vector for checks).
template <class ANONYMOUSLIB_IT, class ANONYMOUSLIB_UIT, class ANONYMOUSLIB_VT>
class anonymouslibMultiHandle
{
public:
    anonymouslibMultiHandle(const int *devices, int ngpus, ANONYMOUSLIB_IT m, ANONYMOUSLIB_IT n) : _h(0)
    {
        const int vt = std::is_same<ANONYMOUSLIB_VT, double>::value ? CSR5HIP_F64
                     : std::is_same<ANONYMOUSLIB_VT, float>::value  ? CSR5HIP_F32 : -1;
        _err = vt < 0 ? ANONYMOUSLIB_UNSUPPORTED_VALUE_TYPE : csr5hip_multi_create(&_h, devices, ngpus, (int)m, (int)n, vt);
        _g = ngpus;
    }
    ~anonymouslibMultiHandle() { if (_h) csr5hip_multi_free(_h); }
    int inputCSR(ANONYMOUSLIB_IT nnz, ANONYMOUSLIB_IT *row_ptr, ANONYMOUSLIB_IT *col_idx, ANONYMOUSLIB_VT *val)
    {
        return _h ? csr5hip_multi_input_csr(_h, (int)nnz, (const int32_t *)row_ptr, (const int32_t *)col_idx, val) : _err;
    }
    void setSigma(int sigma) { if (_h) csr5hip_multi_set_sigma(_h, sigma); }
    int setOption(int option, int value) { return _h ? csr5hip_multi_set_option(_h, option, value) : _err; }
    int asCSR5()
    {
        if (!_h) return _err;
        const int err = csr5hip_multi_as_csr5(_h);
        for (int g = 0; g < _g && err == ANONYMOUSLIB_SUCCESS; g++) {
            csr5hip_shard s;
            csr5hip_info i;
            csr5hip_multi_shard(_h, g, &s);
            csr5hip_get_info(s.handle, &i);
            std::cout << "GPU shard " << g << " on device " << s.device << ": rows [" << s.row_lo << ", " << s.row_hi
                      << "), nnz = " << s.nnz << ", omega = " << ANONYMOUSLIB_CSR5_OMEGA << ", sigma = " << i.sigma
                      << ", tiles = " << i.p << std::endl;
        }
        return err;
    }
    int setX(ANONYMOUSLIB_VT *x) { return _h ? csr5hip_multi_set_x(_h, x) : _err; }
    int spmv(const ANONYMOUSLIB_VT alpha) { return _h ? csr5hip_multi_spmv(_h, (double)alpha) : _err; }
    int spmv_repeat(const ANONYMOUSLIB_VT alpha, int count) { return _h ? csr5hip_multi_spmv_repeat(_h, (double)alpha, count) : _err; }
    int synchronize() { return _h ? csr5hip_multi_synchronize(_h) : _err; }
    int gatherY(ANONYMOUSLIB_VT *host_y) { return _h ? csr5hip_multi_gather_y(_h, host_y) : _err; }
    int destroy() { return _h ? csr5hip_multi_destroy(_h) : _err; }
    csr5hip_multi native() const { return _h; }

private:
    anonymouslibMultiHandle(const anonymouslibMultiHandle &);
    anonymouslibMultiHandle &operator=(const anonymouslibMultiHandle &);
    csr5hip_multi _h;
    int _err, _g;
};

#endif // ANONYMOUSLIB_HIP_H
