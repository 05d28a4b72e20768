// csr5_multi.hip -- one matrix on the G GPUs of a node, behind the C ABI (include/csr5hip.h, csr5hip_multi_*).
//
// The reference is single-device (CSR5_cuda/main.cu:25-26 `cudaSetDevice(0)`); this is the MI355X-native addition
// BASELINE.json names: SpMV rows are independent, so the matrix is cut into G contiguous row blocks balanced by
// COST = non-zeros + weight * rows (default weight 2; weight 0 = plain non-zeros) -- split points =
// upper_bound(row_ptr[r] + weight * r, g * total / G) - 1, the primitive the reference uses for tile_ptr
// (utils_cuda.h:25-53) on the cost prefix -- every block gets its own ordinary handle (own CSR5 conversion, own stream) on its
// device, x is replicated ONCE by a single RCCL broadcast over xGMI at set_x time, y stays sharded on the devices,
// and there is no per-SpMV collective.  One host thread drives all devices (launches are asynchronous).
//
// RCCL is opened lazily (dlopen librccl.so.1): programs that never create a multi handle do not load it, and when
// the library is missing -- or a device id is listed twice (several shards on one GPU: how the 1-GPU test box
// exercises this path) -- x is replicated with device-to-device copies instead.
#include <dlfcn.h>

#include <string>
#include <vector>

#include "csr5_internal.h"

namespace csr5 {
void set_last_error(const std::string &msg);
}
using namespace csr5;

namespace {

// the five RCCL entry points used (rccl/rccl.h: ncclCommInitAll :236, ncclCommDestroy :260, ncclBroadcast :591,
// ncclGroupStart/End :923), resolved at run time
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return lib && CommInitAll && CommDestroy && Broadcast && GroupStart && GroupEnd; }
};

Rccl &rccl()
{
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib)
                break;
        }
        if (r.lib) {
            r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
            r.Broadcast = (decltype(r.Broadcast))dlsym(r.lib, "ncclBroadcast");
            r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
            r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        }
    }
    return r;
}
constexpr int NCCL_UINT8 = 1; // ncclUint8 (rccl.h:460): x travels as bytes

int fail(hipError_t e, const char *what)
{
    set_last_error(std::string(what) + ": " + hipGetErrorString(e));
    return CSR5HIP_HIP_ERROR;
}
#define MHIP(expr)                                                                                 \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_, #expr);                                                                \
    } while (0)
// Every entry point leaves the caller's current device as it found it, on every return path.
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { (void)hipGetDevice(&dev); }
    ~DeviceGuard()
    {
        if (dev >= 0)
            (void)hipSetDevice(dev);
    }
};
#define MRC(expr)                                                                                  \
    do {                                                                                           \
        int rc_ = (expr);                                                                          \
        if (rc_ != CSR5HIP_SUCCESS)                                                                \
            return rc_;                                                                            \
    } while (0)

// number of r in [0, size) with row_ptr[r] + weight * r <= key (the cost prefix is sorted: both terms are monotone)
__device__ __forceinline__ int upper_bound_cost(const int32_t *a, long long weight, long long key, int size)
{
    int lo = 0, hi = size;
    while (lo < hi) {
        const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if ((long long)a[mid] + weight * mid <= key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// cut[g] = first row of block g: the last row whose cost prefix (non-zeros + weight * rows before it) is <= g/G of the
// total (cut[0] = 0, cut[G] = m), monotone.  weight = 0: the row that contains non-zero number g*nnz/G.
__global__ void k_row_cuts(int m, int nnz, int G, int weight, const int32_t *__restrict__ row_ptr,
                           int32_t *__restrict__ cut)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    const long long total = (long long)nnz + (long long)weight * m;
    cut[0] = 0;
    for (int g = 1; g < G; g++) {
        int r = upper_bound_cost(row_ptr, weight, (long long)g * total / G, m + 1) - 1;
        r = r < cut[g - 1] ? cut[g - 1] : (r > m ? m : r);
        cut[g] = r;
    }
    cut[G] = m;
}

__global__ void k_rebase(int count, int32_t base, int32_t *__restrict__ row_ptr)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < count)
        row_ptr[i] -= base;
}

} // namespace

struct csr5hip_multi_s {
    int G = 0, m = 0, n = 0, nnz = 0, value_type = CSR5HIP_F64;
    int row_weight = CSR5HIP_MULTI_DEFAULT_ROW_WEIGHT;
    std::vector<int> dev, cut, shard_nnz;
    std::vector<csr5hip_handle> h;
    std::vector<hipStream_t> stream;
    std::vector<void *> row_ptr, col, val, x, y;
    std::vector<char> x_owned; // x[g] was allocated here (shards on devices[0] borrow the caller's vector instead)
    std::vector<hipEvent_t> ev0, ev1;
    std::vector<void *> comm; // ncclComm_t per UNIQUE device, or empty
    bool distinct = true;     // no device id listed twice
    int broadcast_kind = 0;   // what the last set_x used: 1 = RCCL broadcast, 2 = device-to-device copies
    bool own_replicas = false; // CSR5HIP_MULTI_OPT_OWN_REPLICAS: shards on devices[0] get a replica of x too
    bool live_x_borrowers = false; // set_option(CSR5HIP_OPT_X_SNAPSHOT, 0): the shards that BORROW the caller's x read it live
    size_t vsize() const { return value_type == CSR5HIP_F64 ? 8 : 4; }
};

// does shard g read the caller's own vector (it lives on devices[0] and no replica was asked for)?
static bool borrows_x(const csr5hip_multi_s *mh, int g) { return mh->dev[g] == mh->dev[0] && !mh->own_replicas; }

extern "C" {

int csr5hip_multi_create(csr5hip_multi *out, const int *devices, int G, int m, int n, int value_type)
{
    if (!out || !devices || G < 1 || G > 64 || m < 0 || n < 0)
        return CSR5HIP_INVALID_ARGUMENT;
    if (value_type != CSR5HIP_F64 && value_type != CSR5HIP_F32)
        return CSR5HIP_UNSUPPORTED_VALUE_TYPE;
    int ndev = 0;
    MHIP(hipGetDeviceCount(&ndev));
    for (int g = 0; g < G; g++)
        if (devices[g] < 0 || devices[g] >= ndev) {
            set_last_error("csr5hip_multi_create: device id out of range");
            return CSR5HIP_INVALID_ARGUMENT;
        }
    csr5hip_multi mh = new csr5hip_multi_s();
    mh->G = G, mh->m = m, mh->n = n, mh->value_type = value_type;
    mh->dev.assign(devices, devices + G);
    for (int a = 0; a < G; a++)
        for (int b = a + 1; b < G; b++)
            if (devices[a] == devices[b])
                mh->distinct = false;
    mh->h.assign(G, nullptr);
    mh->stream.assign(G, nullptr);
    mh->row_ptr.assign(G, nullptr), mh->col.assign(G, nullptr), mh->val.assign(G, nullptr);
    mh->x.assign(G, nullptr), mh->y.assign(G, nullptr);
    mh->x_owned.assign(G, 0);
    mh->ev0.assign(G, nullptr), mh->ev1.assign(G, nullptr);
    mh->cut.assign(G + 1, 0), mh->shard_nnz.assign(G, 0);
    DeviceGuard restore_device;
    for (int g = 0; g < G; g++) {
        hipError_t e = hipSetDevice(mh->dev[g]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&mh->stream[g], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreate(&mh->ev0[g]);
        if (e == hipSuccess) e = hipEventCreate(&mh->ev1[g]);
        if (e != hipSuccess) {
            (void)csr5hip_multi_free(mh); // releases the streams and events created so far
            return fail(e, "csr5hip_multi_create: stream / event creation");
        }
    }
    *out = mh;
    return CSR5HIP_SUCCESS;
}

int csr5hip_multi_free(csr5hip_multi mh)
{
    if (!mh)
        return CSR5HIP_INVALID_ARGUMENT;
    DeviceGuard restore_device;
    for (void *c : mh->comm)
        if (c && rccl().ok())
            (void)rccl().CommDestroy(c);
    for (int g = 0; g < mh->G; g++) {
        (void)hipSetDevice(mh->dev[g]);
        if (mh->h[g])
            (void)csr5hip_free(mh->h[g]);
        for (void *p : {mh->row_ptr[g], mh->col[g], mh->val[g], mh->y[g]})
            if (p)
                (void)hipFree(p);
        if (mh->x[g] && mh->x_owned[g])
            (void)hipFree(mh->x[g]);
        if (mh->ev0[g]) (void)hipEventDestroy(mh->ev0[g]);
        if (mh->ev1[g]) (void)hipEventDestroy(mh->ev1[g]);
        if (mh->stream[g]) (void)hipStreamDestroy(mh->stream[g]);
    }
    delete mh;
    return CSR5HIP_SUCCESS;
}

int csr5hip_multi_input_csr(csr5hip_multi mh, int nnz, const int32_t *d_row_ptr, const int32_t *d_col_idx,
                            const void *d_val)
{
    DeviceGuard restore_device;
    if (!mh || nnz < 0 || !d_row_ptr || (nnz > 0 && (!d_col_idx || !d_val)))
        return CSR5HIP_INVALID_ARGUMENT;
    mh->nnz = nnz;
    const int G = mh->G;
    // row cuts on device 0, where the matrix lies
    MHIP(hipSetDevice(mh->dev[0]));
    int32_t *d_cut = nullptr;
    MHIP(hipMalloc(&d_cut, ((size_t)G + 1) * 4));
    hipLaunchKernelGGL(k_row_cuts, dim3(1), dim3(64), 0, mh->stream[0], mh->m, nnz, G, mh->row_weight, d_row_ptr, d_cut);
    std::vector<int32_t> cut(G + 1), ptr_at(G + 1);
    hipError_t e = hipMemcpyAsync(cut.data(), d_cut, ((size_t)G + 1) * 4, hipMemcpyDeviceToHost, mh->stream[0]);
    if (e == hipSuccess)
        e = hipStreamSynchronize(mh->stream[0]);
    for (int g = 0; g <= G && e == hipSuccess; g++)
        e = hipMemcpy(&ptr_at[g], d_row_ptr + cut[g], 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_cut);
    if (e != hipSuccess)
        return fail(e, "csr5hip_multi_input_csr: row cuts");
    const size_t vs = mh->vsize();
    for (int g = 0; g < G; g++) {
        mh->cut[g] = cut[g];
        const int mg = cut[g + 1] - cut[g], nz = ptr_at[g + 1] - ptr_at[g];
        mh->shard_nnz[g] = nz;
        MHIP(hipSetDevice(mh->dev[g]));
        if (mh->h[g]) {
            (void)csr5hip_free(mh->h[g]);
            mh->h[g] = nullptr;
        }
        for (void **p : {&mh->row_ptr[g], &mh->col[g], &mh->val[g], &mh->y[g]})
            if (*p) {
                (void)hipFree(*p);
                *p = nullptr;
            }
        MHIP(hipMalloc(&mh->row_ptr[g], ((size_t)mg + 1) * 4));
        MHIP(hipMalloc(&mh->col[g], (size_t)(nz ? nz : 1) * 4));
        MHIP(hipMalloc(&mh->val[g], (size_t)(nz ? nz : 1) * vs));
        MHIP(hipMalloc(&mh->y[g], (size_t)(mg ? mg : 1) * vs));
        MHIP(hipMemsetAsync(mh->y[g], 0, (size_t)(mg ? mg : 1) * vs, mh->stream[g]));
        // the shard's three arrays: device-to-device copies (over xGMI when the devices differ)
        MHIP(hipMemcpyAsync(mh->row_ptr[g], d_row_ptr + cut[g], ((size_t)mg + 1) * 4, hipMemcpyDeviceToDevice, mh->stream[g]));
        if (nz) {
            MHIP(hipMemcpyAsync(mh->col[g], d_col_idx + ptr_at[g], (size_t)nz * 4, hipMemcpyDeviceToDevice, mh->stream[g]));
            MHIP(hipMemcpyAsync(mh->val[g], (const char *)d_val + (size_t)ptr_at[g] * vs, (size_t)nz * vs,
                                hipMemcpyDeviceToDevice, mh->stream[g]));
        }
        hipLaunchKernelGGL(k_rebase, dim3((mg + 1 + 255) / 256), dim3(256), 0, mh->stream[g], mg + 1, ptr_at[g],
                           (int32_t *)mh->row_ptr[g]);
        MHIP(hipGetLastError());
        MRC(csr5hip_create(&mh->h[g], mg, mh->n, mh->value_type));
        MRC(csr5hip_set_stream(mh->h[g], mh->stream[g]));
        MRC(csr5hip_input_csr(mh->h[g], nz, (int32_t *)mh->row_ptr[g], (int32_t *)mh->col[g], mh->val[g]));
        // the multi handle's x contract: set_x captures x (the replicas are ours; see csr5hip_multi_set_x)
        MRC(csr5hip_set_option(mh->h[g], CSR5HIP_OPT_X_SNAPSHOT, mh->live_x_borrowers && borrows_x(mh, g) ? 0 : 1));
        if (mh->x[g]) // (a matrix replaced under an x that is already distributed)
            MRC(csr5hip_set_x(mh->h[g], mh->x[g]));
    }
    mh->cut[G] = cut[G];
    for (int g = 0; g < G; g++) {
        MHIP(hipSetDevice(mh->dev[g]));
        MHIP(hipStreamSynchronize(mh->stream[g]));
    }
    MHIP(hipSetDevice(mh->dev[0]));
    return CSR5HIP_SUCCESS;
}

int csr5hip_multi_set_sigma(csr5hip_multi mh, int sigma)
{
    DeviceGuard restore_device;
    if (!mh)
        return CSR5HIP_INVALID_ARGUMENT;
    for (int g = 0; g < mh->G; g++)
        if (mh->h[g])
            MRC(csr5hip_set_sigma(mh->h[g], sigma));
    return CSR5HIP_SUCCESS;
}

int csr5hip_multi_set_option(csr5hip_multi mh, int option, int value)
{
    DeviceGuard restore_device;
    if (!mh)
        return CSR5HIP_INVALID_ARGUMENT;
    if (option == CSR5HIP_MULTI_OPT_OWN_REPLICAS) { // takes effect at the next set_x
        mh->own_replicas = value != 0;
        return CSR5HIP_SUCCESS;
    }
    if (option == CSR5HIP_MULTI_OPT_ROW_WEIGHT) { // takes effect at the next input_csr
        if (value < 0 || value > 64)
            return CSR5HIP_INVALID_ARGUMENT;
        mh->row_weight = value;
        return CSR5HIP_SUCCESS;
    }
    if (option == CSR5HIP_OPT_X_SNAPSHOT) {
        if (value != 0 && value != 1)
            return CSR5HIP_INVALID_ARGUMENT;
        mh->live_x_borrowers = value == 0; // a library-owned replica cannot change under the handle: those shards stay at 1
    }
    for (int g = 0; g < mh->G; g++)
        if (mh->h[g]) {
            MHIP(hipSetDevice(mh->dev[g]));
            const int v = option == CSR5HIP_OPT_X_SNAPSHOT && !borrows_x(mh, g) ? 1 : value;
            MRC(csr5hip_set_option(mh->h[g], option, v));
        }
    return CSR5HIP_SUCCESS;
}

int csr5hip_multi_as_csr5(csr5hip_multi mh)
{
    DeviceGuard restore_device;
    if (!mh)
        return CSR5HIP_INVALID_ARGUMENT;
    for (int g = 0; g < mh->G; g++) {
        if (!mh->h[g])
            return CSR5HIP_UNKOWN_FORMAT;
        MHIP(hipSetDevice(mh->dev[g]));
        MRC(csr5hip_as_csr5(mh->h[g]));
        MRC(csr5hip_snapshot_x(mh->h[g])); // (x distributed before the conversion: the copy is taken now, not by the first spmv)
    }
    MHIP(hipSetDevice(mh->dev[0]));
    return CSR5HIP_SUCCESS;
}

int csr5hip_multi_destroy(csr5hip_multi mh)
{
    DeviceGuard restore_device;
    if (!mh)
        return CSR5HIP_INVALID_ARGUMENT;
    for (int g = 0; g < mh->G; g++)
        if (mh->h[g]) {
            MHIP(hipSetDevice(mh->dev[g]));
            MRC(csr5hip_destroy(mh->h[g]));
        }
    MHIP(hipSetDevice(mh->dev[0]));
    return CSR5HIP_SUCCESS;
}

// x lives on devices[0]; every other device receives its copy by ONE broadcast.  Shards on devices[0] read d_x itself.
int csr5hip_multi_set_x(csr5hip_multi mh, const void *d_x)
{
    DeviceGuard restore_device;
    if (!mh || !d_x)
        return CSR5HIP_INVALID_ARGUMENT;
    const size_t bytes = (size_t)mh->n * mh->vsize();
    const int G = mh->G;
    for (int g = 0; g < G; g++) {
        MHIP(hipSetDevice(mh->dev[g]));
        if (mh->dev[g] == mh->dev[0] && !mh->own_replicas) {
            if (mh->x[g] && mh->x_owned[g]) { // (a replica from an earlier set_x with own replicas)
                MHIP(hipFree(mh->x[g]));
                mh->x_owned[g] = 0;
            }
            mh->x[g] = const_cast<void *>(d_x); // borrowed, as setX borrows (anonymouslib_cuda.h:222-260)
        } else {
            int first = g; // one replica per device, shared by the shards that live there
            for (int q = 0; q < g; q++)
                if (mh->dev[q] == mh->dev[g]) {
                    first = q;
                    break;
                }
            if (first != g) {
                mh->x[g] = mh->x[first];
            } else if (!mh->x[g] || !mh->x_owned[g]) {
                MHIP(hipMalloc(&mh->x[g], bytes ? bytes : 4));
                mh->x_owned[g] = 1;
            }
        }
    }
    bool any_remote = false;
    for (int g = 0; g < G; g++)
        any_remote = any_remote || mh->dev[g] != mh->dev[0] || mh->own_replicas;
    mh->broadcast_kind = 0;
    if (any_remote && mh->distinct && rccl().ok()) {
        if (mh->comm.empty()) {
            mh->comm.assign(G, nullptr);
            const int rc = rccl().CommInitAll(mh->comm.data(), G, mh->dev.data());
            if (rc != 0) {
                set_last_error(std::string("ncclCommInitAll: ") + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "?"));
                mh->comm.clear();
            }
        }
        if (!mh->comm.empty()) {
            // the ONE collective of the sharded SpMV: n * sizeof(vT) bytes from devices[0] to all, over xGMI
            int rc = rccl().GroupStart();
            for (int g = 0; g < G && rc == 0; g++) {
                if (hipSetDevice(mh->dev[g]) != hipSuccess) { // (no early return: the group must be closed)
                    rc = -1;
                    break;
                }
                rc = rccl().Broadcast(g == 0 ? d_x : mh->x[g], mh->x[g], bytes, NCCL_UINT8, 0, mh->comm[g], mh->stream[g]);
            }
            const int rc2 = rccl().GroupEnd();
            if (rc == 0 && rc2 == 0)
                mh->broadcast_kind = 1;
            else
                set_last_error("ncclBroadcast failed; falling back to device-to-device copies");
        }
    }
    if (any_remote && mh->broadcast_kind == 0) {
        for (int g = 0; g < G; g++) {
            if (mh->dev[g] == mh->dev[0] && !mh->own_replicas)
                continue;
            bool first = true;
            for (int q = 0; q < g; q++)
                first = first && mh->dev[q] != mh->dev[g];
            if (!first)
                continue;
            MHIP(hipSetDevice(mh->dev[g]));
            MHIP(hipMemcpyAsync(mh->x[g], d_x, bytes, hipMemcpyDeviceToDevice, mh->stream[g]));
        }
        mh->broadcast_kind = 2;
    }
    for (int g = 0; g < G; g++) {
        MHIP(hipSetDevice(mh->dev[g]));
        if (mh->h[g]) {
            MRC(csr5hip_set_x(mh->h[g], mh->x[g]));
            // (own_replicas may have changed what the shard reads: only a shard that borrows the caller's vector may read live)
            MRC(csr5hip_set_option(mh->h[g], CSR5HIP_OPT_X_SNAPSHOT, mh->live_x_borrowers && borrows_x(mh, g) ? 0 : 1));
            // the shard's permuted copy of x, ONCE, right behind the broadcast on the shard's stream
            MRC(csr5hip_snapshot_x(mh->h[g]));
        }
        MHIP(hipStreamSynchronize(mh->stream[g]));
    }
    MHIP(hipSetDevice(mh->dev[0]));
    return CSR5HIP_SUCCESS;
}

int csr5hip_multi_spmv(csr5hip_multi mh, double alpha)
{
    DeviceGuard restore_device;
    if (!mh)
        return CSR5HIP_INVALID_ARGUMENT;
    for (int g = 0; g < mh->G; g++) {
        MHIP(hipSetDevice(mh->dev[g]));
        MRC(csr5hip_spmv(mh->h[g], alpha, mh->y[g]));
    }
    return CSR5HIP_SUCCESS;
}

int csr5hip_multi_spmv_repeat(csr5hip_multi mh, double alpha, int count)
{
    DeviceGuard restore_device;
    if (!mh)
        return CSR5HIP_INVALID_ARGUMENT;
    for (int g = 0; g < mh->G; g++) {
        MHIP(hipSetDevice(mh->dev[g]));
        MRC(csr5hip_spmv_repeat(mh->h[g], alpha, mh->y[g], count));
    }
    return CSR5HIP_SUCCESS;
}

int csr5hip_multi_synchronize(csr5hip_multi mh)
{
    DeviceGuard restore_device;
    if (!mh)
        return CSR5HIP_INVALID_ARGUMENT;
    for (int g = 0; g < mh->G; g++) {
        MHIP(hipSetDevice(mh->dev[g]));
        MHIP(hipStreamSynchronize(mh->stream[g]));
    }
    MHIP(hipSetDevice(mh->dev[0]));
    return CSR5HIP_SUCCESS;
}

int csr5hip_multi_timer_start(csr5hip_multi mh)
{
    DeviceGuard restore_device;
    if (!mh)
        return CSR5HIP_INVALID_ARGUMENT;
    for (int g = 0; g < mh->G; g++) {
        MHIP(hipSetDevice(mh->dev[g]));
        MHIP(hipEventRecord(mh->ev0[g], mh->stream[g]));
    }
    return CSR5HIP_SUCCESS;
}

// elapsed device time between timer_start and now: the MAXIMUM over the shards' streams
int csr5hip_multi_timer_stop(csr5hip_multi mh, double *ms_max)
{
    DeviceGuard restore_device;
    if (!mh || !ms_max)
        return CSR5HIP_INVALID_ARGUMENT;
    for (int g = 0; g < mh->G; g++) {
        MHIP(hipSetDevice(mh->dev[g]));
        MHIP(hipEventRecord(mh->ev1[g], mh->stream[g]));
    }
    double worst = 0;
    for (int g = 0; g < mh->G; g++) {
        MHIP(hipSetDevice(mh->dev[g]));
        MHIP(hipEventSynchronize(mh->ev1[g]));
        float f = 0;
        MHIP(hipEventElapsedTime(&f, mh->ev0[g], mh->ev1[g]));
        worst = f > worst ? f : worst;
    }
    *ms_max = worst;
    MHIP(hipSetDevice(mh->dev[0]));
    return CSR5HIP_SUCCESS;
}

int csr5hip_multi_shard(csr5hip_multi mh, int g, csr5hip_shard *out)
{
    DeviceGuard restore_device;
    if (!mh || !out || g < 0 || g >= mh->G)
        return CSR5HIP_INVALID_ARGUMENT;
    out->device = mh->dev[g];
    out->row_lo = mh->cut[g];
    out->row_hi = mh->cut[g + 1];
    out->nnz = mh->shard_nnz[g];
    out->d_y = mh->y[g];
    out->handle = mh->h[g];
    out->x_broadcast = mh->broadcast_kind;
    return CSR5HIP_SUCCESS;
}

// correctness checks only: collect the y shards into one HOST vector of m values (G device-to-host copies)
int csr5hip_multi_gather_y(csr5hip_multi mh, void *h_y)
{
    DeviceGuard restore_device;
    if (!mh || !h_y)
        return CSR5HIP_INVALID_ARGUMENT;
    const size_t vs = mh->vsize();
    for (int g = 0; g < mh->G; g++) {
        const size_t mg = (size_t)(mh->cut[g + 1] - mh->cut[g]);
        if (!mg)
            continue;
        MHIP(hipSetDevice(mh->dev[g]));
        MHIP(hipStreamSynchronize(mh->stream[g]));
        MHIP(hipMemcpy((char *)h_y + (size_t)mh->cut[g] * vs, mh->y[g], mg * vs, hipMemcpyDeviceToHost));
    }
    MHIP(hipSetDevice(mh->dev[0]));
    return CSR5HIP_SUCCESS;
}

// y shards pre-filled with a value pattern (tests: rows without non-zeros must stay untouched)
int csr5hip_multi_fill_y(csr5hip_multi mh, int byte_value)
{
    DeviceGuard restore_device;
    if (!mh)
        return CSR5HIP_INVALID_ARGUMENT;
    for (int g = 0; g < mh->G; g++) {
        const size_t mg = (size_t)(mh->cut[g + 1] - mh->cut[g]);
        MHIP(hipSetDevice(mh->dev[g]));
        if (mg)
            MHIP(hipMemsetAsync(mh->y[g], byte_value, mg * mh->vsize(), mh->stream[g]));
    }
    return CSR5HIP_SUCCESS;
}

} // extern "C"
